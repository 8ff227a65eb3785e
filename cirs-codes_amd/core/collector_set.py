"""CollectorSet (reference core/collector_set.py:13-77): the three test collectors of the CIRS evaluation protocol --
"FB" (free browsing), "NX_0" (no repeated recommendation) and "NX_k" (no repeats, episodes forced to k turns).
Masking of already-recommended ids and the forced length run inside the fused rollout (cirs_rollout_steps)."""
from typing import Any, Callable, Dict, Optional

from core.collector import Collector
from tianshou.data import VectorReplayBuffer


class CollectorSet:
    def __init__(self, policy, envs_dict, buffer_size, env_num, preprocess_fn: Optional[Callable[..., Any]] = None,
                 exploration_noise: bool = False, force_length=10):
        self.collector_dict = {}
        remove_ids = {"FB": False, "NX_0": True, f"NX_{force_length}": True}
        forced = {"FB": 0, "NX_0": 0, f"NX_{force_length}": force_length}
        for name, envs in envs_dict.items():
            self.collector_dict[name] = Collector(policy, envs, VectorReplayBuffer(buffer_size, env_num), preprocess_fn=preprocess_fn,
                                                  exploration_noise=False, remove_recommended_ids=remove_ids[name],
                                                  force_length=forced[name])
        self.env = envs_dict["FB"]
        self.policy, self.preprocess_fn, self.exploration_noise, self.env_num = policy, preprocess_fn, exploration_noise, env_num
        self.collect_step = self.collect_episode = 0
        self.collect_time = 0.0

    def _each(self, fn, *a, **k):
        for c in self.collector_dict.values():
            getattr(c, fn)(*a, **k)

    def _assign_buffer(self, buffer):
        self._each("_assign_buffer", buffer)

    def reset_stat(self):
        self._each("reset_stat")

    def reset_buffer(self, keep_statistics: bool = False):
        self._each("reset_buffer", keep_statistics)

    def reset_env(self):
        self._each("reset_env")

    def _reset_state(self, id):
        self._each("_reset_state", id)

    def collect(self, n_step=None, n_episode=None, random=False, render=None, no_grad=True, users=None, gumbel=None) -> Dict[str, Any]:
        """users / gumbel: optional {collector name: array} teacher forcing (parity tests), see Collector.collect."""
        all_res = {}
        for name, collector in self.collector_dict.items():
            res = collector.collect(n_step, n_episode, random, render, no_grad, users=None if users is None else users[name],
                                    gumbel=None if gumbel is None else gumbel[name])
            all_res.update(res if name == "FB" else {f"{name}_{k}": v for k, v in res.items()})
        fb = self.collector_dict["FB"]
        self.collect_step, self.collect_episode, self.collect_time = fb.collect_step, fb.collect_episode, fb.collect_time
        return all_res
