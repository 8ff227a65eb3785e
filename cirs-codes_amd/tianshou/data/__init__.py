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


def _stack_value(v, n):
    """Zero storage with n slots for values shaped like one entry of `v` (Batch -> Batch of storages)."""
    if isinstance(v, Batch):
        return Batch({k: _stack_value(x, n) for k, x in v.items()})
    if isinstance(v, torch.Tensor):
        return torch.zeros((n,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
    v = np.asanyarray(v)
    return np.zeros((n,) + v.shape[1:], dtype=v.dtype)


def _store(dst: "Batch", ptrs, src: "Batch"):
    for k, v in src.items():
        if isinstance(v, Batch):
            if not len(v.keys()):
                dst.__dict__.setdefault(k, Batch())
                continue
            if k not in dst or not isinstance(dst[k], Batch) or not len(dst[k].keys()):
                dst.__dict__[k] = Batch()
            _store(dst[k], ptrs, v)
            continue
        if k not in dst:
            dst.__dict__[k] = _stack_value(v, _store.maxsize)
        if isinstance(dst[k], torch.Tensor):
            dst[k][torch.as_tensor(ptrs, device=dst[k].device)] = torch.as_tensor(v).to(dst[k])
        else:
            dst[k][ptrs] = np.asanyarray(v)


class VectorReplayBuffer:
    """buffer_num ring buffers of equal size behind one index space (reference vecbuf.py:26-30, manager.py:16-232,
    base.py:116-183): sub-buffer b owns rows [b*size, (b+1)*size).  Two ways in:
      * `add(batch, buffer_ids)` — the per-step protocol of Collector (manager.py:91-142): ring write per sub-buffer, episode
        reward / length / start-index accounting, returns (ptr, ep_rew, ep_len, ep_idx);
      * `fill_from_trajectory(traj, lens)` — one shot from the device trajectory of cirs_rollout_steps: env b's transitions
        land at _offset[b] .. _offset[b]+len_b-1, exactly where that sequence of adds would have put them (host copies are
        made lazily on attribute access).
    prev / next / unfinished_index follow manager.py:194-232 and base.py:116-141 (ring arithmetic per sub-buffer, vectorised
    over the query instead of a loop over sub-buffers)."""

    _reserved_keys = ("obs", "act", "rew", "done", "obs_next", "info", "policy")

    def __init__(self, total_size: int, buffer_num: int, **kwargs):
        assert buffer_num > 0
        self.buffer_num = buffer_num
        self.size = int(np.ceil(total_size / buffer_num))
        self.maxsize = self.size * buffer_num
        self._offset = np.arange(buffer_num) * self.size
        self.reset()

    def reset(self, keep_statistics: bool = False):
        n = self.buffer_num
        self._lengths = np.zeros(n, dtype=int)
        self._write = np.zeros(n, dtype=int)          # next write position inside each sub-buffer (base.py: self._index)
        self.last_index = self._offset.copy()
        if not keep_statistics:
            self._ep_rew = np.zeros(n, dtype=float)
            self._ep_len = np.zeros(n, dtype=int)
            self._ep_idx = np.zeros(n, dtype=int)     # sub-buffer-local start of the running episode
        self._meta = Batch()
        self._traj = None

    # ---- per-step protocol -------------------------------------------------------------------------------------
    def add(self, batch: Batch, buffer_ids=None):
        """ReplayBufferManager.add (manager.py:91-142) + ReplayBuffer._add_index (base.py:163-183)."""
        assert self._traj is None, "this buffer was filled from a device trajectory; add() needs a fresh buffer"
        b = Batch({k: batch[k] for k in self._reserved_keys if k in batch})
        assert {"obs", "act", "rew", "done"}.issubset(b.keys())
        ids = np.arange(self.buffer_num) if buffer_ids is None else np.asarray(buffer_ids, dtype=int)
        rew = np.asarray(to_numpy(b.rew), dtype=float).reshape(-1)
        done = np.asarray(to_numpy(b.done)).astype(bool).reshape(-1)
        assert len(ids) == len(rew) == len(done)
        ptrs = np.empty(len(ids), dtype=int); ep_rews = np.zeros(len(ids)); ep_lens = np.zeros(len(ids), dtype=int)
        ep_idxs = np.empty(len(ids), dtype=int)
        for k, bid in enumerate(ids):       # sequential on purpose: the same sub-buffer may appear twice in buffer_ids
            ptr = self._write[bid]
            ptrs[k] = ptr + self._offset[bid]
            self.last_index[bid] = ptrs[k]
            self._lengths[bid] = min(self._lengths[bid] + 1, self.size)
            self._write[bid] = (ptr + 1) % self.size
            self._ep_rew[bid] += rew[k]
            self._ep_len[bid] += 1
            ep_idxs[k] = self._ep_idx[bid] + self._offset[bid]
            if done[k]:
                ep_rews[k], ep_lens[k] = self._ep_rew[bid], self._ep_len[bid]
                self._ep_rew[bid], self._ep_len[bid], self._ep_idx[bid] = 0.0, 0, self._write[bid]
        b.rew, b.done = rew, done
        _store.maxsize = self.maxsize
        _store(self._meta, ptrs, b)
        return ptrs, ep_rews, ep_lens, ep_idxs

    # ---- filled from the device ------------------------------------------------------------------------------
    def fill_from_trajectory(self, traj, lens: np.ndarray, device_rows=None):
        """traj: cirs_hip.rollout.Trajectory (time-major, device).  Host copies are made lazily on attribute access."""
        lens = np.asarray(lens, dtype=int)
        assert lens.max(initial=0) <= self.size, "episode longer than the per-env buffer"
        self.reset()
        self._traj = traj
        self._lengths[:len(lens)] = lens
        self._write = self._lengths % self.size
        self._ep_idx = self._write.copy()
        self.last_index = self._offset + np.maximum(self._lengths - 1, 0)
        nb = len(lens)
        self._rows_env = np.repeat(np.arange(nb), lens)
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
        """manager.py:144-172; on-policy use is sample(0) = every stored row, sub-buffer by sub-buffer, oldest first."""
        if batch_size < 0:
            return np.array([], int)
        parts = []
        for b in range(self.buffer_num):      # base.py:185-205 for batch_size == 0: arange(write, size) ++ arange(write)
            L, w = self._lengths[b], self._write[b]
            loc = np.concatenate([np.arange(w, L), np.arange(w)]) if L == self.size else np.arange(L)
            parts.append(loc + self._offset[b])
        allidx = np.concatenate(parts) if parts else np.array([], int)
        if batch_size == 0:
            return allidx
        return np.random.choice(allidx, batch_size)

    def sample(self, batch_size):
        idx = self.sample_index(batch_size)
        return self[idx], idx

    # ---- ring-buffer neighbours (manager.py:194-232) ----------------------------------------------------------------
    def _locate(self, index):
        scalar = not isinstance(index, (list, np.ndarray))
        index = np.atleast_1d(np.asarray(index, dtype=int)) % self.maxsize
        b = index // self.size
        return scalar, index, self._offset[b], np.maximum(1, self._lengths[b]), self.last_index[b]

    def prev(self, index):
        scalar, index, start, cur_len, last = self._locate(index)
        self._materialise()
        done = self._meta.done if "done" in self._meta else np.zeros(self.maxsize, bool)
        sub = (index - start - 1) % cur_len
        end_flag = done[sub + start] | (sub + start == last)
        out = (sub + end_flag) % cur_len + start
        return out[0] if scalar else out

    def next(self, index):
        scalar, index, start, cur_len, last = self._locate(index)
        self._materialise()
        done = self._meta.done if "done" in self._meta else np.zeros(self.maxsize, bool)
        end_flag = done[index] | (index == last)
        out = (index - start + 1 - end_flag) % cur_len + start
        return out[0] if scalar else out

    def unfinished_index(self):
        """base.py:116-119 per sub-buffer: the newest row unless it closed an episode."""
        self._materialise()
        has = self._lengths > 0
        last = (self._write - 1) % np.maximum(1, self._lengths) + self._offset
        done = self._meta.done if "done" in self._meta else np.zeros(self.maxsize, bool)
        return last[has & ~done[last]]


ReplayBuffer = VectorReplayBuffer  # only the vector flavour is used by CIRS
