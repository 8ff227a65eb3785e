"""On-policy training driver behind the call `onpolicy_trainer(policy, train_collector, test_collector, state_tracker, ...)` that
CIRS-RL-kuaishou.py / CIRS-RL-taobao.py make (contract: reference core/trainer/onpolicy.py:30-252 -- argument order, the logger and callback
protocol calls, the result dict of gather_info).

The structure is this repository's own.  A `Session` owns the counters and the protocol objects; training is three nested pieces:

    Session.run()            evaluation of the untouched policy, then `Session.epoch()` per epoch, then the summary
    Session.epoch(k)         `Session.round()` until the epoch's transition budget is spent, then evaluation + book-keeping
    Session.round(k)         one collect -> (optional early-stop probe) -> one policy.update, with the loss streams smoothed

Everything the reference's scripts can observe is kept: which hook fires when and with which arguments, how `env_step` /
`gradient_step` advance, the 100-sample moving average of every loss stream handed to `log_update_data`, `best_*` starting from the
pre-training evaluation, test statistics logged with `global_step=None`.
"""
import collections
import time
from typing import Callable, Dict, Optional, Union

import numpy as np

from tianshou.trainer.utils import gather_info, test_episode


class _Quiet:
    """Stands in when no logger is given: accepts every protocol call, remembers nothing."""

    def restore_data(self):
        return 0, 0, 0

    def __getattr__(self, _name):
        return lambda *args, **kwargs: None


class _Trail:
    """Mean over the most recent `span` samples of one loss stream (what the progress line and `log_update_data` show)."""

    def __init__(self, span: int = 100):
        self._recent = collections.deque(maxlen=span)

    def absorb(self, samples) -> float:
        if hasattr(samples, "detach"):                 # a tensor of per-minibatch values
            samples = samples.detach().cpu().numpy()
        # +-inf never enters the window; a COMPUTED NaN does (statistics.py:30,44-46 bans by membership in [inf, nan, -inf], and nan != nan), so a
        # diverged update shows up as a NaN moving average instead of a healthy-looking one
        self._recent.extend(float(x) for x in np.ravel(samples) if not np.isinf(x))
        return sum(self._recent) / len(self._recent) if self._recent else 0.0


class _Best:
    """The best evaluation seen so far and the epoch it came from."""

    def __init__(self, epoch: int, outcome: dict):
        self.epoch, self.reward, self.spread = epoch, outcome["rew"], outcome["rew_std"]

    def offer(self, epoch: int, outcome: dict) -> bool:
        if self.epoch >= 0 and not self.reward < outcome["rew"]:
            return False
        self.epoch, self.reward, self.spread = epoch, outcome["rew"], outcome["rew_std"]
        return True


class Session:
    def __init__(self, policy, train_collector, test_collector, *, epochs, transitions_per_epoch, update_repeat, update_batch, test_episodes,
                 collect_kwargs, hooks, logger, resume, reward_metric, verbose, probe_while_training):
        self.policy, self.train_c, self.test_c = policy, train_collector, test_collector
        self.epochs, self.budget = epochs, transitions_per_epoch
        self.update_kwargs = dict(batch_size=update_batch, repeat=update_repeat)
        self.test_episodes, self.collect_kwargs = test_episodes, collect_kwargs
        self.hooks, self.metric, self.verbose = hooks, reward_metric, verbose
        self.log = logger if logger is not None else _Quiet()
        self.first_epoch, self.env_step, self.gradient_step = self.log.restore_data() if resume else (0, 0, 0)
        self.probe = bool(probe_while_training) and getattr(train_collector, "policy", policy) is policy and hooks["stop_fn"] is not None
        self.trails: Dict[str, _Trail] = collections.defaultdict(_Trail)
        self.began = time.time()
        self.best: Optional[_Best] = None
        self.shown_reward, self.shown_length = 0.0, 0

    # -- protocol fan-out ----------------------------------------------------------------------------------------
    def _tell(self, event: str, *args):
        for listener in getattr(self.policy, "callbacks", ()):
            getattr(listener, event)(*args)

    def _evaluate(self, epoch: int, with_metric: bool = True) -> dict:
        # (global_step stays None, as in the reference: the test statistics are computed but not written)
        return test_episode(self.policy, self.test_c, self.hooks["test_fn"], epoch, self.test_episodes, self.log, None,
                            self.metric if with_metric else None)

    def _checkpoint(self, epoch: int):
        self.log.save_data(epoch, self.env_step, self.gradient_step, self.hooks["save_checkpoint_fn"])

    def _summary(self, reward, spread) -> Dict[str, Union[float, str]]:
        return gather_info(self.began, self.train_c, self.test_c, reward, spread)

    # -- one collect + one update --------------------------------------------------------------------------------
    def round(self, epoch: int):
        """-> (transitions gathered, outcome of a successful early-stop probe or None)."""
        if self.hooks["train_fn"]:
            self.hooks["train_fn"](epoch, self.env_step)
        got = self.train_c.collect(**self.collect_kwargs)
        finished = got["n/ep"] > 0
        if finished and self.metric:
            got["rews"] = self.metric(got["rews"])
        n_new = int(got["n/st"])
        self.env_step += n_new
        self.log.log_train_data(got, self.env_step)              # (adds "rew" / "len" to `got` when episodes finished)
        self.shown_reward, self.shown_length = got.get("rew", self.shown_reward), got.get("len", self.shown_length)
        if finished and self.probe and self.hooks["stop_fn"](got["rew"]):
            outcome = self._evaluate(epoch, with_metric=False)
            if self.hooks["stop_fn"](outcome["rew"]):
                return n_new, outcome
            self.policy.train()
        raw = self.policy.update(0, self.train_c.buffer, **self.update_kwargs)
        self.gradient_step += max([1] + [len(v) for v in raw.values() if isinstance(v, list)])
        smoothed = {name: self.trails[name].absorb(raw[name]) for name in raw}
        raw.update(smoothed)                                      # the caller's dict carries the running means, like the reference's
        self.log.log_update_data(raw, self.gradient_step)
        if self.verbose:
            print(f"[epoch {epoch}] transitions {self.env_step}  episodes/round {int(got['n/ep'])}  mean length {n_new / max(got['n/ep'], 1):.2f}  "
                  f"return {self.shown_reward:.2f}  " + "  ".join(f"{k} {v:.3f}" for k, v in smoothed.items()), flush=True)
        return n_new, None

    # -- one epoch -----------------------------------------------------------------------------------------------
    def epoch(self, k: int):
        """-> summary dict when an early-stop probe ended training inside the epoch, else None."""
        self.policy.train()
        self._tell("on_epoch_begin", k)
        spent = 0
        while spent < self.budget:
            n_new, stopped = self.round(k)
            spent += n_new
            if stopped is not None:
                if self.hooks["save_fn"]:
                    self.hooks["save_fn"](self.policy)
                self._checkpoint(k)
                return self._summary(stopped["rew"], stopped["rew_std"])
        outcome = self._evaluate(k)
        if self.best.offer(k, outcome) and self.hooks["save_fn"]:
            self.hooks["save_fn"](self.policy)
        self._checkpoint(k)
        self._tell("on_epoch_end", k, outcome)
        if self.hooks["save_model_fn"]:
            self.hooks["save_model_fn"](epoch=k, policy=self.policy)
        if self.verbose:
            print(f"[epoch {k}] test return {outcome['rew']:.6f} +- {outcome['rew_std']:.6f}   best {self.best.reward:.6f} +- {self.best.spread:.6f} "
                  f"(epoch {self.best.epoch})", flush=True)
        return None

    def run(self) -> Dict[str, Union[float, str]]:
        self.train_c.reset_stat()
        self.test_c.reset_stat()
        self.best = _Best(self.first_epoch, self._evaluate(self.first_epoch))       # the untouched policy sets the bar
        self._tell("on_train_begin")
        for k in range(self.first_epoch + 1, self.epochs + 1):
            early = self.epoch(k)
            if early is not None:
                return early
            if self.hooks["stop_fn"] and self.hooks["stop_fn"](self.best.reward):
                break
        self._tell("on_train_end")
        return self._summary(self.best.reward, self.best.spread)


def onpolicy_trainer(policy, train_collector, test_collector, state_tracker, max_epoch: int, step_per_epoch: int,
                     repeat_per_collect: int, episode_per_test: int, batch_size: int, step_per_collect: Optional[int] = None,
                     episode_per_collect: Optional[int] = None, train_fn: Optional[Callable] = None, test_fn: Optional[Callable] = None,
                     stop_fn: Optional[Callable[[float], bool]] = None, save_fn: Optional[Callable] = None,
                     save_checkpoint_fn: Optional[Callable] = None, resume_from_log: bool = False, reward_metric=None, logger=None,
                     verbose: bool = True, test_in_train: bool = True, save_model_fn=None) -> Dict[str, Union[float, str]]:
    """`state_tracker` is part of the call shape only: the collectors and the policy already hold it (preprocess_fn / update)."""
    hooks = dict(train_fn=train_fn, test_fn=test_fn, stop_fn=stop_fn, save_fn=save_fn, save_checkpoint_fn=save_checkpoint_fn,
                 save_model_fn=save_model_fn)
    return Session(policy, train_collector, test_collector, epochs=max_epoch, transitions_per_epoch=step_per_epoch,
                   update_repeat=repeat_per_collect, update_batch=batch_size, test_episodes=episode_per_test,
                   collect_kwargs=dict(n_step=step_per_collect, n_episode=episode_per_collect), hooks=hooks, logger=logger,
                   resume=resume_from_log, reward_metric=reward_metric, verbose=verbose, probe_while_training=test_in_train).run()
