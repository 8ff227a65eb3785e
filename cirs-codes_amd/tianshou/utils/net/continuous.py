"""ActorProb / Critic for continuous actions (reference tianshou/tianshou/utils/net/continuous.py:68-199): the heads of the
VirtualTaobao stack (CIRS-RL-taobao.py:207-209, BASELINE configs[0]: CPU plumbing, no GPU).  Unlike the discrete heads these run
as plain torch modules on the host -- C1 is host code in the reference too -- under the reference's attribute names
(`preprocess`, `mu`, `sigma_param`, `last`) so state_dicts interchange."""
import numpy as np
import torch
from torch import nn

from tianshou.utils.net.common import MLP

SIGMA_MIN, SIGMA_MAX = -20, 2


class ActorProb(nn.Module):
    """s -> (mu, sigma) of a diagonal Gaussian: mu = max_action * tanh(MLP(trunk(s))) unless `unbounded`; sigma = exp of a free
    [A, 1] parameter broadcast over the batch, or exp(clamp(MLP(trunk(s)), -20, 2)) when `conditioned_sigma`."""

    def __init__(self, preprocess_net, action_shape, hidden_sizes=(), max_action=1.0, device="cpu", unbounded=False,
                 conditioned_sigma=False, preprocess_net_output_dim=None):
        super().__init__()
        self.preprocess, self.device = preprocess_net, device
        self.output_dim = int(np.prod(action_shape))
        width = getattr(preprocess_net, "output_dim", preprocess_net_output_dim)
        self.mu = MLP(width, self.output_dim, hidden_sizes, device=device)
        self._c_sigma = conditioned_sigma
        if conditioned_sigma:
            self.sigma = MLP(width, self.output_dim, hidden_sizes, device=device)
        else:
            self.sigma_param = nn.Parameter(torch.zeros(self.output_dim, 1))
        self._max, self._unbounded = max_action, unbounded

    def forward(self, s, state=None, info={}):
        hidden, _ = self.preprocess(s, state)
        mu = self.mu(hidden)
        if not self._unbounded:
            mu = self._max * torch.tanh(mu)
        if self._c_sigma:
            sigma = torch.clamp(self.sigma(hidden), min=SIGMA_MIN, max=SIGMA_MAX).exp()
        else:
            sigma = (self.sigma_param.view(1, -1) + torch.zeros_like(mu)).exp()
        return (mu, sigma), state


class Critic(nn.Module):
    """(s[, a]) -> V / Q: trunk over the flattened (concatenated) input, then an MLP to one output."""

    def __init__(self, preprocess_net, hidden_sizes=(), device="cpu", preprocess_net_output_dim=None):
        super().__init__()
        self.preprocess, self.device, self.output_dim = preprocess_net, device, 1
        width = getattr(preprocess_net, "output_dim", preprocess_net_output_dim)
        self.last = MLP(width, 1, hidden_sizes, device=device)

    def forward(self, s, a=None, info={}):
        s = torch.as_tensor(s, device=self.device, dtype=torch.float32).flatten(1)
        if a is not None:
            s = torch.cat([s, torch.as_tensor(a, device=self.device, dtype=torch.float32).flatten(1)], dim=1)
        hidden, _ = self.preprocess(s)
        return self.last(hidden)
