"""Batch, VectorReplayBuffer, converters (reference tianshou/tianshou/data/{batch,buffer/*,utils/converter}.py).

Batch is a light dict-of-arrays container with the operations the CIRS path uses.  VectorReplayBuffer keeps the
reference's index layout (buffer b owns rows [b*size, b*size+len_b), manager.py / vecbuf.py:26-30) but is filled in
one shot from the device trajectory collected by cirs_rollout_steps instead of one `add` per step."""
from typing import Any, Dict, Optional

import numpy as np
import torch


def to_numpy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    if isinstance(x, Batch):
        return Batch({k: to_numpy(v) for k, v in x.items()})
    return np.asanyarray(x)


def to_torch_as(x, y: torch.Tensor):
    return torch.as_tensor(to_numpy(x) if not isinstance(x, torch.Tensor) else x).to(device=y.device, dtype=y.dtype)


class Batch:
    def __init__(self, batch_dict: Optional[Dict[str, Any]] = None, **kwargs):
        if batch_dict:
            kwargs = dict(batch_dict, **kwargs)
        for k, v in kwargs.items():
            self.__dict__[k] = Batch(v) if isinstance(v, dict) else v

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def get(self, k, d=None):
        return self.__dict__.get(k, d)

    def pop(self, k, d=None):
        return self.__dict__.pop(k, d)

    def update(self, batch=None, **kwargs):
        if batch is not None:
            kwargs = dict(batch.items() if isinstance(batch, Batch) else batch, **kwargs)
        for k, v in kwargs.items():
            self.__dict__[k] = v

    def __contains__(self, k):
        return k in self.__dict__

    def __getitem__(self, index):
        if isinstance(index, str):
            return self.__dict__[index]
        out = Batch()
        for k, v in self.items():
            if isinstance(v, Batch):
                out.__dict__[k] = v[index] if len(v.keys()) else Batch()
            elif v is None or isinstance(v, dict):
                out.__dict__[k] = v
            else:
                out.__dict__[k] = v[index]
        return out

    def __setitem__(self, k, v):
        self.__dict__[k] = v

    def is_empty(self):
        return len(self.__dict__) == 0

    def __len__(self):
        for v in self.values():
            if isinstance(v, Batch):
                if len(v.keys()):
                    return len(v)
            elif v is not None and hasattr(v, "__len__"):
                return len(v)
        return 0

    def __repr__(self):
        return "Batch(" + ", ".join(f"{k}={type(v).__name__}" for k, v in self.items()) + ")"

    def split(self, size, shuffle=True, merge_last=False):
        """Batch.split (batch.py:721-744)."""
        length = len(self)
        indices = np.random.permutation(length) if shuffle else np.arange(length)
        merge_last = merge_last and length % size > 0
        for idx in range(0, length, size):
            if merge_last and idx + size + size >= length:
                yield self[indices[idx:]]
                break
            yield self[indices[idx:idx + size]]


class VectorReplayBuffer:
    """buffer_num ring buffers of equal size (vecbuf.py:26-30).  `fill_from_trajectory` replaces the per-step
    ReplayBufferManager.add (manager.py:91-142) for a finished collect: env b's transitions land at
    _offset[b] .. _offset[b]+len_b-1, exactly where the reference's sequence of adds would have put them."""

    def __init__(self, total_size: int, buffer_num: int, **kwargs):
        assert buffer_num > 0
        self.buffer_num = buffer_num
        self.size = int(np.ceil(total_size / buffer_num))
        self.maxsize = self.size * buffer_num
        self._offset = np.arange(buffer_num) * self.size
        self._lengths = np.zeros(buffer_num, dtype=int)
        self.last_index = self._offset.copy()
        self._meta = Batch()
        self._traj = None

    # ---- filled from the device ------------------------------------------------------------------------------
    def fill_from_trajectory(self, traj, lens: np.ndarray, device_rows=None):
        """traj: cirs_hip.rollout.Trajectory (time-major, device).  Host copies are made lazily on attribute access."""
        lens = np.asarray(lens, dtype=int)
        assert lens.max(initial=0) <= self.size, "episode longer than the per-env buffer"
        self._traj, self._lengths = traj, lens
        self.last_index = self._offset + np.maximum(lens - 1, 0)
        self._meta = Batch()
        self._rows_env = np.repeat(np.arange(self.buffer_num), lens)
        self._rows_t = np.concatenate([np.arange(l) for l in lens]) if lens.sum() else np.zeros(0, int)
        self._index = self._offset[self._rows_env] + self._rows_t

    def _materialise(self):
        if not self._meta.is_empty() or self._traj is None:
            return
        tr, n = self._traj, self.maxsize
        S = tr.obs.shape[-1]
        e, t = self._rows_env, self._rows_t
        obs = torch.zeros((n, S), dtype=torch.float32, device=tr.obs.device)
        obs_next = torch.zeros_like(obs)
        et, tt = torch.as_tensor(e, device=tr.obs.device), torch.as_tensor(t, device=tr.obs.device)
        it = torch.as_tensor(self._index, device=tr.obs.device)
        obs[it] = tr.obs[tt, et]
        obs_next[it] = tr.obs[tt + 1, et]
        act = np.zeros(n, dtype=np.int64); rew = np.zeros(n); done = np.zeros(n, dtype=bool); ctr = np.zeros(n)
        act[self._index] = tr.act.cpu().numpy()[t, e]
        rew[self._index] = tr.rew.cpu().numpy()[t, e]
        done[self._index] = tr.done.cpu().numpy()[t, e].astype(bool)
        ctr[self._index] = tr.ctr.cpu().numpy()[t, e]
        env_id = np.zeros(n, dtype=int); env_id[self._index] = e
        self._meta = Batch(obs=obs, obs_next=obs_next, act=act, rew=rew, done=done, info=Batch(CTR=ctr, env_id=env_id), policy=Batch())

    def __getattr__(self, key):
        if key.startswith("_"):
            raise AttributeError(key)
        self._materialise()
        try:
            return self._meta.__dict__[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __len__(self):
        return int(self._lengths.sum())

    def __getitem__(self, index):
        self._materialise()
        return self._meta[index]

    def sample_index(self, batch_size):
        assert batch_size == 0, "on-policy use: the whole buffer (sample(0))"
        return self._index.copy() if self._traj is not None else np.array([], int)

    def sample(self, batch_size):
        idx = self.sample_index(batch_size)
        return self[idx], idx

    # ring-buffer neighbours (manager.py:194-232) for a buffer that has not wrapped
    def prev(self, index):
        index = np.asarray(index)
        b = np.minimum(index // self.size, self.buffer_num - 1)
        start = self._offset[b]
        p = np.where(index > start, index - 1, index)
        self._materialise()
        return np.where(self._meta.done[p] & (p != index), index, p)

    def next(self, index):
        index = np.asarray(index)
        self._materialise()
        end = np.isin(index, self.last_index) | self._meta.done[index]
        return np.where(end, index, index + 1)

    def unfinished_index(self):
        self._materialise()
        li = self.last_index[self._lengths > 0]
        return li[~self._meta.done[li]]


ReplayBuffer = VectorReplayBuffer  # only the vector flavour is used by CIRS
