"""Helpers of the trainer loop with the call shapes of reference tianshou/tianshou/trainer/utils.py:10-81.

test_episode  one evaluation pass: fresh buffers, policy in eval mode, optional per-epoch hook, `collect(n_episode)`; the
              collector may be a CollectorSet (FB / NX_0 / NX_k), whose merged result dict is passed through unchanged.
gather_info   the summary the trainer returns: step / episode counters and wall-clock split between collecting and
              updating, as formatted strings ("12.34s", "5678.90 step/s") under the reference's key names."""
import time
from typing import Any, Callable, Dict, Optional


def test_episode(policy, collector, test_fn: Optional[Callable], epoch: int, n_episode: int, logger=None,
                 global_step: Optional[int] = None, reward_metric=None, **teacher) -> Dict[str, Any]:
    for prepare in (collector.reset_env, collector.reset_buffer, policy.eval):
        prepare()
    if test_fn is not None:
        test_fn(epoch, global_step)
    outcome = collector.collect(n_episode=n_episode, **teacher)   # teacher: users= / gumbel= of the parity tests
    if reward_metric is not None:
        outcome["rews"] = reward_metric(outcome["rews"])
    if logger is not None and global_step is not None:
        logger.log_test_data(outcome, global_step)
    return outcome


def _seconds(x: float) -> str:
    return "%.2fs" % x


def _rate(steps: int, seconds: float) -> str:
    return "%.2f step/s" % (steps / max(seconds, 1e-9))


def gather_info(start_time, train_c, test_c, best_reward, best_reward_std) -> Dict[str, Any]:
    wall = time.time() - start_time
    learn = wall - test_c.collect_time            # everything that is not test collection ...
    info = {"test_step": test_c.collect_step, "test_episode": test_c.collect_episode, "test_time": _seconds(test_c.collect_time),
            "test_speed": _rate(test_c.collect_step, test_c.collect_time), "best_reward": best_reward,
            "best_result": "%.2f ± %.2f" % (best_reward, best_reward_std), "duration": _seconds(wall)}
    if train_c is not None:
        learn -= train_c.collect_time             # ... nor training collection is model time
        info.update(train_step=train_c.collect_step, train_episode=train_c.collect_episode)
        info["train_time/collector"] = _seconds(train_c.collect_time)
        info["train_speed"] = _rate(train_c.collect_step, wall - test_c.collect_time)
    info["train_time/model"] = _seconds(learn)
    return info
