"""Scratch diagnostic: per-minibatch loss columns of the C3-size learner test, device vs restatement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cirs-codes_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import nn_oracle, policycase, rolloutcase
from test_gpu_learn import make_learner, rollout_time_value_logp, upload_traj
from cirs_hip.rollout import Trajectory

I, B, T, bs, ent_coef = 10728, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 30, 1024, 0.0
U = 300
rng = np.random.RandomState(I)
tp = rolloutcase.tracker_param_dict(U, I, T, seed=1)
arrs = policycase.random_weights(rng, I, head_scale=1.5)
pp = {k: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)) for k, v in arrs.items()}
lens = rng.randint(8, T + 1, size=B)
users = rng.randint(0, U, B); acts = rng.randint(0, I, (B, T)); rews = rng.uniform(0, 1, (B, T))
dones = np.zeros((B, T), bool); dones[np.arange(B), lens - 1] = True
with torch.no_grad():
    obs_bts = nn_oracle.tracker_states(tp, users, acts, rews).numpy()
value, logp = rollout_time_value_logp(pp, obs_bts, acts, lens)
n = int(lens.sum())
perms = [rng.permutation(n) for _ in range(2)]
hyper = [0.95, 0.95, 0.2, 0.25, ent_coef, 0.5, 1e-3, bs, 2]
traj = Trajectory(B, T, 20, "cuda")
upload_traj(traj, acts, rews, dones, lens, obs_bts, value, logp)
ln, views = make_learner(pp, I, B, T, hyper)
assert ln.prepare(traj, lens) == n
losses = ln.learn(bs, 2, perms=perms).cpu().numpy()
tp_o = {k: v.clone() for k, v in tp.items()}
pp_o = {k: v.clone() for k, v in pp.items()}
out = nn_oracle.ppo_update(tp_o, pp_o, users, acts, rews, dones, lens, perms, gamma=0.95, lam=0.95, eps_clip=0.2, vf_coef=0.25,
                           ent_coef=ent_coef, max_grad_norm=0.5, lr=1e-3, batch_size=bs, repeat=2)
print("n rows", n, "minibatches", len(out["loss"]))
for k in range(len(out["loss"])):
    print(k, "loss %.6f %.6f" % (losses[k, 0], out["loss"][k]), "clip %.6f %.6f" % (losses[k, 1], out["clip"][k]),
          "vf %.6f %.6f" % (losses[k, 2], out["vf"][k]), "ent %.5f %.5f" % (losses[k, 3], out["ent"][k]))
for k, name in {"w1": "actor.preprocess.model.model.0.weight", "wa": "actor.last.model.0.weight", "wc": "critic.last.model.0.weight"}.items():
    got = views[name].cpu().numpy().reshape(pp_o[k].shape)
    print(k, "max |param diff|", np.abs(got - pp_o[k].numpy()).max())
