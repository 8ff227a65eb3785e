"""onpolicy_trainer (reference core/trainer/onpolicy.py:30-252): epochs of { collect -> policy.update } until
step_per_epoch transitions, then the test collectors, callbacks and the model-save hook.  Same signature and result
dict; progress bars / TensorBoard are optional (any object with log_train_data / log_update_data / save_data works)."""
import time
from collections import defaultdict
from typing import Callable, Dict, Optional, Union

import numpy as np

from tianshou.trainer.utils import gather_info, test_episode


class _NullLogger:
    def __getattr__(self, name):
        return lambda *a, **k: None

    def restore_data(self):
        return 0, 0, 0


class _MovAvg:  # tianshou.utils.MovAvg(size=100) over the loss streams (statistics.py:7-65)
    def __init__(self, size=100):
        self.size, self.cache = size, []

    def add(self, x):
        self.cache += list(np.atleast_1d(x))
        self.cache = self.cache[-self.size:]

    def get(self):
        return float(np.mean(self.cache)) if self.cache else 0.0


def onpolicy_trainer(policy, train_collector, test_collector, state_tracker, max_epoch: int, step_per_epoch: int,
                     repeat_per_collect: int, episode_per_test: int, batch_size: int, step_per_collect: Optional[int] = None,
                     episode_per_collect: Optional[int] = None, train_fn: Optional[Callable] = None, test_fn: Optional[Callable] = None,
                     stop_fn: Optional[Callable[[float], bool]] = None, save_fn: Optional[Callable] = None,
                     save_checkpoint_fn: Optional[Callable] = None, resume_from_log: bool = False, reward_metric=None, logger=None,
                     verbose: bool = True, test_in_train: bool = True, save_model_fn=None) -> Dict[str, Union[float, str]]:
    logger = logger or _NullLogger()
    start_epoch, env_step, gradient_step = (logger.restore_data() if resume_from_log else (0, 0, 0))
    last_rew, last_len = 0.0, 0
    stat = defaultdict(_MovAvg)
    start_time = time.time()
    train_collector.reset_stat()
    test_collector.reset_stat()
    test_in_train = test_in_train and getattr(train_collector, "policy", policy) is policy
    # the reference evaluates the untrained policy once before the first epoch and starts `best_*` from it (onpolicy.py:126-129)
    test_result = test_episode(policy, test_collector, test_fn, start_epoch, episode_per_test, logger, None, reward_metric)
    best_epoch, best_reward, best_reward_std = start_epoch, test_result["rew"], test_result["rew_std"]
    for cb in getattr(policy, "callbacks", []):
        cb.on_train_begin()
    for epoch in range(1 + start_epoch, 1 + max_epoch):
        policy.train()
        for cb in getattr(policy, "callbacks", []):
            cb.on_epoch_begin(epoch)
        collected = 0
        while collected < step_per_epoch:
            if train_fn:
                train_fn(epoch, env_step)
            result = train_collector.collect(n_step=step_per_collect, n_episode=episode_per_collect)
            if result["n/ep"] > 0 and reward_metric:
                result["rews"] = reward_metric(result["rews"])
            env_step += int(result["n/st"])
            collected += int(result["n/st"])
            logger.log_train_data(result, env_step)
            last_rew, last_len = result.get("rew", last_rew), result.get("len", last_len)
            if result["n/ep"] > 0 and test_in_train and stop_fn and stop_fn(result["rew"]):
                test_result = test_episode(policy, test_collector, test_fn, epoch, episode_per_test, logger, None)
                if stop_fn(test_result["rew"]):
                    if save_fn:
                        save_fn(policy)
                    logger.save_data(epoch, env_step, gradient_step, save_checkpoint_fn)
                    return gather_info(start_time, train_collector, test_collector, test_result["rew"], test_result["rew_std"])
                policy.train()
            losses = policy.update(0, train_collector.buffer, batch_size=batch_size, repeat=repeat_per_collect)
            gradient_step += max([1] + [len(v) for v in losses.values() if isinstance(v, list)])
            for k in losses:
                stat[k].add(losses[k])
                losses[k] = stat[k].get()
            logger.log_update_data(losses, gradient_step)
            if verbose:
                print(f"Epoch #{epoch}: env_step {env_step} mean_len_traj {result['n/st'] / max(result['n/ep'], 1):.2f} "
                      f"R_traj {last_rew:.2f} loss {losses.get('loss', 0.0):.3f}", flush=True)
        test_result = test_episode(policy, test_collector, test_fn, epoch, episode_per_test, logger, None, reward_metric)
        rew, rew_std = test_result["rew"], test_result["rew_std"]
        if best_epoch < 0 or best_reward < rew:
            best_epoch, best_reward, best_reward_std = epoch, rew, rew_std
            if save_fn:
                save_fn(policy)
        logger.save_data(epoch, env_step, gradient_step, save_checkpoint_fn)
        for cb in getattr(policy, "callbacks", []):
            cb.on_epoch_end(epoch, test_result)
        if save_model_fn:
            save_model_fn(epoch=epoch, policy=policy)
        if verbose:
            print(f"Epoch #{epoch}: test_reward: {rew:.6f} ± {rew_std:.6f}, best_reward: {best_reward:.6f} ± {best_reward_std:.6f} in #{best_epoch}", flush=True)
        if stop_fn and stop_fn(best_reward):
            break
    for cb in getattr(policy, "callbacks", []):
        cb.on_train_end()
    return gather_info(start_time, train_collector, test_collector, best_reward, best_reward_std)
