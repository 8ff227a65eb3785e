"""Scratch probe: throughput of cirs_deepfm_train_step at the shipped model's shape (7176 x 10729, E = 16), batch 2048, and the
torch-fp32 restatement of the same step on the host cores for comparison."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from cirs_hip.deepfm_train import DeepFMTrainer, layout

U, I, F, E, n = 7176, 10729, 32, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.RandomState(0)
init = {name: rng.normal(0, 0.05, shape).astype(np.float32) for name, shape in layout(U, I, F, E)}
init["embedding_dict.feat.weight"][0] = 0
steps = 50
N = n * steps
col = lambda v: np.asarray(v, np.float64)[:, None]
feats = lambda: np.where(np.arange(4)[None, :] < rng.randint(1, 5, N)[:, None], rng.randint(1, F, (N, 4)), 0)
u = rng.randint(0, U, N)
x = np.concatenate([col(u), col(rng.randint(0, I, N)), feats(), col(rng.uniform(2, 60, N)), col(u), col(rng.randint(0, I, N)), feats(), col(rng.uniform(2, 60, N))], axis=1)
y = rng.uniform(0, 5, (N, 1)); score = rng.gamma(1.0, 0.5, (N, 1))
tr = DeepFMTrainer(init, use_ab=True, lambda_ab=10.0)
xd, yd, sd = torch.as_tensor(x, dtype=torch.float32).cuda(), torch.as_tensor(y, dtype=torch.float32).cuda(), torch.as_tensor(score, dtype=torch.float32).cuda()
for st in range(5):
    tr.step(xd[st * n:(st + 1) * n], yd[st * n:(st + 1) * n], sd[st * n:(st + 1) * n])
torch.cuda.synchronize()
t0 = time.perf_counter()
for st in range(steps):
    tr.step(xd[st * n:(st + 1) * n], yd[st * n:(st + 1) * n], sd[st * n:(st + 1) * n])
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / steps
out = dict(batch=n, us_per_step=1e6 * t, samples_per_s=n / t)
if "--cpu" in sys.argv:
    import nn_oracle
    torch.set_num_threads(16)
    t0 = time.perf_counter()
    nn_oracle.deepfm_train(init, x[:5 * n], y[:5 * n], score[:5 * n], n, 5, True, 10.0)
    tc = (time.perf_counter() - t0) / 5
    out.update(cpu_us_per_step=1e6 * tc, cpu_samples_per_s=n / tc, cpu_threads=16)
print(json.dumps(out))
