"""GPU: KV-cached tracker step kernel vs golden vectors recorded from the reference and vs the torch-fp32 restatement."""
import os

import numpy as np
import pytest
import torch

import nn_oracle
from conftest import close

pytestmark = pytest.mark.gpu


def dev_tracker(params_cpu, U, I, B, T, nhead=4):
    from cirs_hip.tracker import DeviceTracker
    params = {k: v.float().cuda().contiguous() for k, v in params_cpu.items()}
    return DeviceTracker(params, U, I, B, T, nhead=nhead)


def run_device(trk, users, acts, rews, last_turn):
    B, T = acts.shape
    states = np.full((B, T + 1, 20), np.nan, np.float32)
    trk.reset()
    s0 = trk.init(torch.as_tensor(users))
    states[:, 0] = s0.cpu().numpy()
    for t in range(T):
        live = np.where(last_turn > t)[0]
        if len(live) == 0:
            break
        out = trk.step(torch.as_tensor(acts[live, t]), torch.as_tensor(rews[live, t]),
                       env_ids=torch.as_tensor(live.astype(np.int32)).cuda())
        states[live, t + 1] = out.cpu().numpy()
    return states


def test_tracker_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "tracker.npz"))
    U, I, B, T = [int(v) for v in z["dims"][:4]]
    p = nn_oracle.tracker_params(z)
    trk = dev_tracker(p, U, I, B, T)
    got = run_device(trk, z["users"], z["acts"], z["rews"], z["last_turn"])
    want = z["states"]
    m = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), m)
    # s_t <= 1e-5 abs with dropout disabled, every step of the episode
    print(f"[tracker vs reference golden] max |err| = {np.abs(got[m] - want[m]).max():.3e} over {int(m.sum())} state entries (bar 1e-5)")
    close(got[m], want[m], 1e-5, 1e-5, "tracker states vs reference golden")
    # x_hist (the reference's self.data) matches too
    x = nn_oracle.tracker_inputs(p, z["users"], z["acts"], z["rews"]).numpy()
    live = np.where(z["last_turn"] == T)[0]
    close(trk.x_hist.cpu().numpy()[live], x[live], 1e-5, 1e-6, "tracker input slots vs restatement")


@pytest.mark.parametrize("U,I,B,T,nhead", [(1411, 3327, 64, 30, 4), (7176, 10728, 1024, 30, 4), (100, 200, 33, 100, 8), (50, 60, 7, 5, 1)])
def test_tracker_vs_restatement_at_baseline_sizes(U, I, B, T, nhead):
    g = torch.Generator().manual_seed(U + B)
    D, S, H = 32, 20, 128

    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    p = {"embedding_dict.feat_user.weight": rn(U, D, scale=0.5), "embedding_dict.feat_item.weight": rn(I, D, scale=0.5),
         "ffn_user.weight": rn(D, D, scale=0.25), "ffn_user.bias": rn(D, scale=0.1),
         "fnn_gate.weight": rn(D, D + 1, scale=0.25), "fnn_gate.bias": rn(D, scale=0.1),
         "decoder.weight": rn(S, D, scale=0.25), "decoder.bias": rn(S, scale=0.1)}
    from cirs_hip.tracker import positional_encoding
    p["pos_encoder.pe"] = positional_encoding(T + 1, D).unsqueeze(1)
    for l in range(2):
        pre = f"transformer_encoder.layers.{l}."
        p[pre + "self_attn.in_proj_weight"] = rn(3 * D, D, scale=0.25); p[pre + "self_attn.in_proj_bias"] = rn(3 * D, scale=0.1)
        p[pre + "self_attn.out_proj.weight"] = rn(D, D, scale=0.25); p[pre + "self_attn.out_proj.bias"] = rn(D, scale=0.1)
        p[pre + "linear1.weight"] = rn(H, D, scale=0.25); p[pre + "linear1.bias"] = rn(H, scale=0.1)
        p[pre + "linear2.weight"] = rn(D, H, scale=0.12); p[pre + "linear2.bias"] = rn(D, scale=0.1)
        for nm in ("norm1", "norm2"):
            p[pre + nm + ".weight"] = 1 + rn(D, scale=0.2); p[pre + nm + ".bias"] = rn(D, scale=0.1)
    rng = np.random.RandomState(B)
    users = rng.randint(0, U, B); acts = rng.randint(0, I, (B, T)); rews = rng.uniform(0, 1, (B, T))
    want = nn_oracle.tracker_states(p, users, acts, rews, nhead=nhead).numpy()
    trk = dev_tracker(p, U, I, B, T, nhead=nhead)
    got = run_device(trk, users, acts, rews, np.full(B, T))
    print(f"[tracker vs restatement U={U} I={I} B={B} T={T} nhead={nhead}] max |err| = {np.abs(got - want).max():.3e}, max |state| = {np.abs(want).max():.2f}")
    close(got, want, 1e-4, 5e-5, f"tracker states vs restatement (nhead={nhead}, T={T})")
    assert int(trk.len.min()) == T + 1
