"""test_episode / gather_info (reference tianshou/tianshou/trainer/utils.py:10-81)."""
import time
from typing import Any, Callable, Dict, Optional


def test_episode(policy, collector, test_fn: Optional[Callable], epoch: int, n_episode: int, logger=None,
                 global_step: Optional[int] = None, reward_metric=None) -> Dict[str, Any]:
    collector.reset_env()
    collector.reset_buffer()
    policy.eval()
    if test_fn:
        test_fn(epoch, global_step)
    result = collector.collect(n_episode=n_episode)
    if reward_metric:
        result["rews"] = reward_metric(result["rews"])
    if logger and global_step is not None:
        logger.log_test_data(result, global_step)
    return result


def gather_info(start_time, train_c, test_c, best_reward, best_reward_std) -> Dict[str, Any]:
    duration = time.time() - start_time
    model_time = duration - test_c.collect_time
    result = {"test_step": test_c.collect_step, "test_episode": test_c.collect_episode, "test_time": f"{test_c.collect_time:.2f}s",
              "test_speed": f"{test_c.collect_step / max(test_c.collect_time, 1e-9):.2f} step/s", "best_reward": best_reward,
              "best_result": f"{best_reward:.2f} ± {best_reward_std:.2f}", "duration": f"{duration:.2f}s",
              "train_time/model": f"{model_time:.2f}s"}
    if train_c is not None:
        model_time -= train_c.collect_time
        result.update({"train_step": train_c.collect_step, "train_episode": train_c.collect_episode,
                       "train_time/collector": f"{train_c.collect_time:.2f}s", "train_time/model": f"{model_time:.2f}s",
                       "train_speed": f"{train_c.collect_step / max(duration - test_c.collect_time, 1e-9):.2f} step/s"})
    return result
