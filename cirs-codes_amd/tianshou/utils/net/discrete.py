"""Actor / Critic heads (reference tianshou/tianshou/utils/net/discrete.py:11-114); see common.py for how they are used."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from tianshou.utils.net.common import MLP


class Actor(nn.Module):
    def __init__(self, preprocess_net, action_shape, hidden_sizes=(), softmax_output=True, preprocess_net_output_dim=None, device="cpu"):
        super().__init__()
        self.device = device
        self.preprocess = preprocess_net
        self.output_dim = int(np.prod(action_shape))
        input_dim = getattr(preprocess_net, "output_dim", preprocess_net_output_dim)
        self.last = MLP(input_dim, self.output_dim, hidden_sizes, device=device)
        self.softmax_output = softmax_output

    def forward(self, s, state=None, info={}):
        logits, h = self.preprocess(s, state)
        logits = self.last(logits)
        if self.softmax_output:
            logits = F.softmax(logits, dim=-1)
        return logits, h


class Critic(nn.Module):
    def __init__(self, preprocess_net, hidden_sizes=(), last_size=1, preprocess_net_output_dim=None, device="cpu"):
        super().__init__()
        self.device = device
        self.preprocess = preprocess_net
        self.output_dim = last_size
        input_dim = getattr(preprocess_net, "output_dim", preprocess_net_output_dim)
        self.last = MLP(input_dim, last_size, hidden_sizes, device=device)

    def forward(self, s, **kwargs):
        logits, _ = self.preprocess(s, state=kwargs.get("state", None))
        return self.last(logits)
