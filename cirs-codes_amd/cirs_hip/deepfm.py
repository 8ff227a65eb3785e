"""Device engine of the DeepFM user model (csrc/deepfm.hip): pair scoring, full-catalogue sweep, normed_mat."""
import ctypes as C
from typing import Dict

import numpy as np
import torch

from . import abi

# reference state_dict name -> struct field (Q12: the `linear.*` copy is the one forward uses)
STATE_DICT_MAP = {"embedding_dict.user_id.weight": "emb_user", "embedding_dict.photo_id.weight": "emb_item",
                  "embedding_dict.feat.weight": "emb_feat", "linear.embedding_dict.user_id.weight": "lin_user",
                  "linear.embedding_dict.photo_id.weight": "lin_item", "linear.embedding_dict.feat.weight": "lin_feat",
                  "linear.weight": "lin_dense", "dnn.linears.0.weight": "w1", "dnn.linears.0.bias": "b1",
                  "dnn.linears.1.weight": "w2", "dnn.linears.1.bias": "b2", "last.weight": "last", "out.bias": "out_bias"}


class DeviceDeepFM:
    def __init__(self, tensors: Dict[str, torch.Tensor], device="cuda"):
        """tensors: field name (abi.DEEPFM_FIELDS) -> array/tensor; or a reference state_dict (names mapped)."""
        self.device = torch.device(device)
        t = {}
        for k, v in tensors.items():
            f = STATE_DICT_MAP.get(k, k)
            if f in abi.DEEPFM_FIELDS:
                t[f] = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(self.device, torch.float32).contiguous()
        missing = [f for f in abi.DEEPFM_FIELDS if f not in t]
        assert not missing, f"missing DeepFM tensors: {missing}"
        for f in ("lin_user", "lin_item", "lin_feat", "lin_dense", "last", "out_bias"):
            t[f] = t[f].reshape(-1).contiguous()
        self.t = t
        E = t["emb_user"].shape[1]
        assert t["w1"].shape == (64, 6 * E + 1), t["w1"].shape
        self.cfg = abi.DeepFMCfg(n_user_vocab=t["emb_user"].shape[0], n_item_vocab=t["emb_item"].shape[0],
                                 n_feat_vocab=t["emb_feat"].shape[0], emb_dim=E, hidden=64)
        self.w = abi.DeepFMWeights(**{f: t[f].data_ptr() for f in abi.DEEPFM_FIELDS})
        self._lib = abi.lib()

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self, uid, pid, feats, dur):
        """UserModel_Pairwise.forward on explicit columns (ids index the vocab tables)."""
        dev = self.device
        uid = torch.as_tensor(uid).to(dev, torch.int64).contiguous(); pid = torch.as_tensor(pid).to(dev, torch.int64).contiguous()
        feats = torch.as_tensor(feats).to(dev, torch.int32).contiguous(); dur = torch.as_tensor(dur).to(dev, torch.float32).contiguous()
        n = uid.numel()
        out = torch.empty(n, dtype=torch.float32, device=dev)
        abi.check(self._lib.cirs_deepfm_forward(C.byref(self.cfg), C.byref(self.w), uid.data_ptr(), pid.data_ptr(), feats.data_ptr(),
                                                dur.data_ptr(), n, out.data_ptr(), self._stream()), "cirs_deepfm_forward")
        return out

    def gather_fm(self, X):
        """K1-K2 alone: X [n, 7] float32 rows [user_id, photo_id, feat0..3, duration] (the reference's forward input) ->
        linear logit + FM term [n] (cirs_gather_fm; no DNN)."""
        X = torch.as_tensor(X).to(self.device, torch.float32).contiguous()
        assert X.dim() == 2 and X.shape[1] == 7
        out = torch.empty(X.shape[0], dtype=torch.float32, device=self.device)
        abi.check(self._lib.cirs_gather_fm(C.byref(self.cfg), C.byref(self.w), X.data_ptr(), X.shape[0], out.data_ptr(), self._stream()),
                  "cirs_gather_fm")
        return out

    def sweep(self, user_ids, item_ids, item_feats, item_dur, want_pred=True):
        """All (user, item) pairs -> (pred [nu, ni] fp32 or None, minmax [2])."""
        dev = self.device
        user_ids = torch.as_tensor(user_ids).to(dev, torch.int64).contiguous(); item_ids = torch.as_tensor(item_ids).to(dev, torch.int64).contiguous()
        item_feats = torch.as_tensor(item_feats).to(dev, torch.int32).contiguous(); item_dur = torch.as_tensor(item_dur).to(dev, torch.float32).contiguous()
        nu, ni = user_ids.numel(), item_ids.numel()
        pred = torch.empty((nu, ni), dtype=torch.float32, device=dev) if want_pred else None
        mm = torch.empty(2, dtype=torch.float32, device=dev)
        need = self._lib.cirs_deepfm_sweep_workspace_bytes(C.byref(self.cfg), nu, ni)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        abi.check(self._lib.cirs_deepfm_sweep(C.byref(self.cfg), C.byref(self.w), user_ids.data_ptr(), nu, item_ids.data_ptr(),
                                              item_feats.data_ptr(), item_dur.data_ptr(), ni, abi.ptr(pred), mm.data_ptr(), 1,
                                              ws.data_ptr(), ws.numel(), self._stream()), "cirs_deepfm_sweep")
        return pred, mm

    def normed_reward(self, user_ids, item_ids, item_feats, item_dur):
        """KuaishouEnv.compute_normed_reward: float64 (pred - min) / (max - min) over all users x items."""
        pred, mm = self.sweep(user_ids, item_ids, item_feats, item_dur)
        out = torch.empty(pred.shape, dtype=torch.float64, device=self.device)
        abi.check(self._lib.cirs_normed_reward(pred.data_ptr(), pred.numel(), mm.data_ptr(), out.data_ptr(), self._stream()),
                  "cirs_normed_reward")
        return out


def hash_ids(ids, n_buckets: int, device="cuda") -> torch.Tensor:
    """splitmix64(id) mod n_buckets on the device (cirs_hash_ids): maps open-vocabulary ids into a fixed-size table."""
    ids = torch.as_tensor(ids).to(device, torch.int64).contiguous()
    out = torch.empty_like(ids)
    abi.check(abi.lib().cirs_hash_ids(ids.data_ptr(), ids.numel(), int(n_buckets), out.data_ptr(), torch.cuda.current_stream(ids.device).cuda_stream),
              "cirs_hash_ids")
    return out
