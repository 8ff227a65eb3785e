"""MLP / Net parameter containers (reference tianshou/tianshou/utils/net/common.py:25-197).

These nn.Modules hold the parameters under the reference's names (`model.model.0.weight`, ...).  On the MI355X path
their arithmetic runs inside csrc/policy.hip / ppo.hip (PPOPolicy binds the parameters into one flat device
buffer), so `forward` here is only a convenience for small host-side checks."""
from typing import Sequence

import numpy as np
import torch
from torch import nn


class MLP(nn.Module):
    def __init__(self, input_dim, output_dim=0, hidden_sizes: Sequence[int] = (), norm_layer=None, activation=nn.ReLU, device=None):
        super().__init__()
        self.device = device
        sizes = [input_dim] + list(hidden_sizes)
        layers = []
        for i, o in zip(sizes[:-1], sizes[1:]):
            layers += [nn.Linear(i, o)] + ([activation()] if activation is not None else [])
        if output_dim > 0:
            layers += [nn.Linear(sizes[-1], output_dim)]
        self.output_dim = output_dim or sizes[-1]
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        x = torch.as_tensor(x, device=self.model[0].weight.device, dtype=torch.float32)
        return self.model(x.flatten(1))


class Net(nn.Module):
    def __init__(self, state_shape, action_shape=0, hidden_sizes: Sequence[int] = (), norm_layer=None, activation=nn.ReLU,
                 device="cpu", softmax=False, concat=False, num_atoms=1, dueling_param=None):
        super().__init__()
        assert not concat and dueling_param is None and num_atoms == 1, "only the plain trunk is used by CIRS"
        self.device = device
        self.softmax = softmax
        input_dim = int(np.prod(state_shape))
        action_dim = int(np.prod(action_shape))
        self.model = MLP(input_dim, action_dim, hidden_sizes, norm_layer, activation, device)
        self.output_dim = self.model.output_dim

    def forward(self, s, state=None, info={}):
        logits = self.model(s)
        if self.softmax:
            logits = torch.softmax(logits, dim=-1)
        return logits, state
