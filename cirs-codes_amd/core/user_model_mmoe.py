"""UserModel_MMOE (reference core/user_model_mmoe.py:15-262 over core/layers.py MMOELayer/Linear and DeepCTR-Torch's DNN /
PredictionLayer): the user model of the VirtualTaobao experiments (CIRS-UserModel-taobao.py:100-148) -- BASELINE configs[0], CPU
plumbing.  Inference side only: parameters under the reference's state_dict names and `forward`; plain torch on whatever device
the module lives on (this is not a hot path: one 1 x 118 row per env step).

    y_task = PredictionLayer_task( Linear_task(X)  [+ FM over the sparse embeddings, none for the all-dense Taobao features]
                                   + tower_task( MMoE_task( DNN(X) ) ) )
    MMoE: experts = Linear(H, n_experts * expert_dim) reshaped [B, expert_dim, n_experts]; gate_task = softmax(Linear(H, n_experts,
    no bias)); output = experts @ gate."""
import torch
from torch import nn

from core.inputs import compute_input_dim
from core.user_model import UserModel
from deepctr_torch.inputs import DenseFeat, build_input_features


class _Dense(nn.Module):
    """DeepCTR DNN without batch-norm / dropout (dnn_use_bn=False, dnn_dropout=0): `linears.<i>`, ReLU after every layer."""

    def __init__(self, d_in, hidden):
        super().__init__()
        dims = [d_in] + list(hidden)
        self.linears = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])

    def forward(self, x):
        for lin in self.linears:
            x = torch.relu(lin(x))
        return x


class _MMoE(nn.Module):
    def __init__(self, d_in, num_tasks, num_experts, out_dim):
        super().__init__()
        self.num_experts, self.out_dim = num_experts, out_dim
        self.expert_network = nn.Linear(d_in, num_experts * out_dim, bias=True)
        self.gating_networks = nn.ModuleList([nn.Linear(d_in, num_experts, bias=False) for _ in range(num_tasks)])
        for m in (self.expert_network, *self.gating_networks):
            nn.init.normal_(m.weight)

    def forward(self, x):
        experts = self.expert_network(x).reshape(-1, self.out_dim, self.num_experts)
        return [torch.bmm(experts, gate(x).softmax(1).unsqueeze(-1)).squeeze() for gate in self.gating_networks]


class _DenseLinear(nn.Module):
    """core/layers.py Linear for all-dense feature columns: X[:, dense columns] @ weight."""

    def __init__(self, n_dense):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(n_dense, 1))

    def forward(self, x_dense):
        return x_dense.matmul(self.weight)


class _Bias(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros((1, dim)))


class UserModel_MMOE(UserModel):
    def __init__(self, feature_columns, y_columns, num_tasks, tasks, task_logit_dim, num_experts=4, expert_dim=8,
                 dnn_hidden_units=(128, 128), l2_reg_embedding=1e-5, l2_reg_dnn=1e-2, init_std=0.0001, task_dnn_units=None, seed=2021,
                 dnn_dropout=0, dnn_activation="relu", dnn_use_bn=False, device="cpu", padding_idx=None, ab_columns=None):
        super().__init__()
        assert all(isinstance(f, DenseFeat) for f in feature_columns), "the Taobao user model has dense features only (CIRS-UserModel-taobao.py:100)"
        assert dnn_activation == "relu" and not dnn_use_bn and dnn_dropout == 0 and task_dnn_units is None and ab_columns is None
        assert all(v == "regression" for v in tasks.values()), "regression tasks only"
        torch.manual_seed(seed)
        self.feature_columns, self.y_columns, self.tasks, self.task_logit_dim = feature_columns, y_columns, tasks, task_logit_dim
        self.feature_index = build_input_features(feature_columns)
        self.y_index = build_input_features(y_columns)
        self.device = device
        d_in = compute_input_dim(feature_columns)
        self.linear_model = _DenseLinear(d_in)                     # base-class duplicate, unused by forward (as in the reference)
        self.dnn = _Dense(d_in, dnn_hidden_units)
        for lin in self.dnn.linears:
            nn.init.normal_(lin.weight, mean=0, std=init_std)
        self.mmoe_layer = _MMoE(dnn_hidden_units[-1], num_tasks, num_experts, expert_dim)
        self.tower_network = nn.ModuleList([nn.Linear(expert_dim, dim, bias=False) for dim in task_logit_dim.values()])
        self.out = nn.ModuleList([_Bias(dim) for dim in task_logit_dim.values()])
        self.linear_model_task = nn.ModuleList([_DenseLinear(d_in) if dim == 1 else None for dim in task_logit_dim.values()])
        for m in [self.linear_model] + [m for m in self.linear_model_task if m is not None]:
            nn.init.normal_(m.weight, mean=0, std=init_std)
        self.to(device)

    def forward(self, x):
        x = x.to(torch.float32)
        outs = []
        hidden = self.dnn(x)
        mmoe = self.mmoe_layer(hidden)
        for i, name in enumerate(self.tasks):
            logit = torch.zeros([len(x), self.task_logit_dim[name]], device=x.device)
            if self.linear_model_task[i] is not None:
                logit = logit + self.linear_model_task[i](x)
            logit = logit + self.tower_network[i](mmoe[i])
            outs.append(logit + self.out[i].bias)                  # PredictionLayer("regression"): bias only
        return torch.cat(outs, -1)
