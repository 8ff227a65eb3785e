"""Device training of the pairwise DeepFM user model (csrc/deepfm_train.hip, cirs_deepfm_train_step).

Host-side counterpart of fit_data's inner loop (reference core/user_model.py:150-170): parameters, gradients and the Adam
moments live in ONE flat fp32 device buffer each; named views follow the reference's state_dict."""
import ctypes as C
from typing import Dict

import numpy as np
import torch

from . import abi

# (state_dict name, layout slot) in buffer order -- must match train_layout() in csrc/deepfm_train.hip
def layout(U: int, I: int, F: int, E: int):
    K = 6 * E + 1
    return [("embedding_dict.user_id.weight", (U, E)), ("embedding_dict.photo_id.weight", (I, E)), ("embedding_dict.feat.weight", (F, E)),
            ("linear.embedding_dict.user_id.weight", (U, 1)), ("linear.embedding_dict.photo_id.weight", (I, 1)),
            ("linear.embedding_dict.feat.weight", (F, 1)), ("linear.weight", (1, 1)),
            ("dnn.linears.0.weight", (64, K)), ("dnn.linears.0.bias", (64,)), ("dnn.linears.1.weight", (64, 64)), ("dnn.linears.1.bias", (64,)),
            ("last.weight", (1, 64)), ("out.bias", (1, 1)),
            ("ab_embedding_dict.alpha_u.weight", (U, 1)), ("ab_embedding_dict.beta_i.weight", (I, 1)),
            ("linear_model.embedding_dict.user_id.weight", (U, 1)), ("linear_model.embedding_dict.photo_id.weight", (I, 1)),
            ("linear_model.embedding_dict.feat.weight", (F, 1)), ("linear_model.weight", (1, 1))]


class DeepFMTrainer:
    def __init__(self, state_dict: Dict[str, torch.Tensor], *, use_ab=True, lambda_ab=1.0, l2_embedding=1e-5, l2_linear=1e-5, l2_all=1e-1,
                 lr=1e-3, betas=(0.9, 0.999), eps=1e-8, device="cuda"):
        self.device = torch.device(device)
        sd = {k: torch.as_tensor(v) for k, v in state_dict.items()}
        U, E = sd["embedding_dict.user_id.weight"].shape
        I = sd["embedding_dict.photo_id.weight"].shape[0]
        F = sd["embedding_dict.feat.weight"].shape[0]
        self.cfg = abi.DeepFMCfg(n_user_vocab=U, n_item_vocab=I, n_feat_vocab=F, emb_dim=E, hidden=64)
        self._lib = abi.lib()
        total = self._lib.cirs_deepfm_train_param_count(C.byref(self.cfg))
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.views = {}
        off = 0
        for name, shape in layout(U, I, F, E):
            n = int(np.prod(shape))
            self.views[name] = self.flat[off:off + n].view(shape)
            if name in sd:
                self.views[name].copy_(sd[name].to(self.device, torch.float32).reshape(shape))
            elif name.startswith("ab_embedding_dict"):
                # without alpha/beta the model has no such parameters: zeros carry neither a regulariser term nor a gradient
                self.views[name].fill_(1.0 if use_ab else 0.0)
            off += n
        assert off == total
        self.grads = torch.zeros_like(self.flat)
        self.adam_m = torch.zeros_like(self.flat)
        self.adam_v = torch.zeros_like(self.flat)
        self.step_count = 0
        self.use_ab, self.lambda_ab = bool(use_ab), float(lambda_ab)
        self.l2 = (float(l2_embedding), float(l2_linear), float(l2_all))
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self._ws = None
        self.loss = torch.zeros(5, dtype=torch.float32, device=self.device)

    def state_dict(self):
        return {k: v.clone() for k, v in self.views.items() if self.use_ab or not k.startswith("ab_embedding_dict")}

    def step(self, x: torch.Tensor, y: torch.Tensor, score: torch.Tensor):
        """x [n,14] = positive pair columns [user, photo, feat0..3, duration] then the negative pair's (user_model_pairwise.py:136-137);
        y [n] or [n,1]; score [n] or [n,1] = exposure.  Returns the device loss vector {loss, loss_y, bpr, loss_ab, reg_loss}."""
        dev = self.device
        x = torch.as_tensor(x).to(dev)
        n = x.shape[0]
        ids = x[:, [0, 1, 7, 8]].to(torch.int64)
        cols = [ids[:, 0].contiguous(), ids[:, 1].contiguous(), x[:, 2:6].to(torch.int32).contiguous(), x[:, 6].to(torch.float32).contiguous(),
                ids[:, 2].contiguous(), ids[:, 3].contiguous(), x[:, 9:13].to(torch.int32).contiguous(), x[:, 13].to(torch.float32).contiguous()]
        y = torch.as_tensor(y).to(dev, torch.float32).reshape(-1).contiguous()
        ex = torch.as_tensor(score).to(dev, torch.float32).reshape(-1).contiguous()
        need = self._lib.cirs_deepfm_train_workspace_bytes(C.byref(self.cfg), n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        abi.check(self._lib.cirs_deepfm_train_step(
            C.byref(self.cfg), self.flat.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.step_count,
            *[c.data_ptr() for c in cols], y.data_ptr(), ex.data_ptr(), n, int(self.use_ab), self.lambda_ab, *self.l2, self.lr,
            self.betas[0], self.betas[1], self.eps, self.loss.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
            torch.cuda.current_stream(dev).cuda_stream), "cirs_deepfm_train_step")
        self.step_count += 1
        return self.loss
