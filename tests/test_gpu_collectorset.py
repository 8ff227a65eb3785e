"""GPU: SURVEY 8(f1) -- the mirror's test_episode over CollectorSet(FB, NX_0, NX_k) against the REFERENCE's own
tianshou.trainer.utils.test_episode + core.collector_set.CollectorSet.collect (tests/golden/collectorset.npz, recorded by
oracle/gen_golden.py:gen_collectorset with the sampler noise supplied by the harness).  The device rollout is fed the same users
and the same noise (cirs_rollout_steps_noise) and must return the reference's result dict KEY FOR KEY: n/ep, n/st, rews, lens,
idxs (completion order), rew, len, rew_std, len_std for FB and the NX_0_* / NX_k_* copies -- incl. the masking of already
recommended ids (core/policy/utils.py:7-58) and the forced episode length (core/collector.py:253-258) -- and leave the same
transitions in the three buffers."""
import os

import numpy as np
import pytest
import torch

from test_gpu_plugin_surface import load_example

pytestmark = pytest.mark.gpu


def test_test_episode_result_dict_equals_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "collectorset.npz"))
    U, I, B, T, K = [int(v) for v in z["dims"]]
    N, thr, _ = [int(v) for v in z["env_params"]]
    ex = load_example()
    args = ex.get_args(["--n-users", str(U), "--n-items", str(I), "--training-num", str(B), "--test-num", str(B), "--max_turn", str(T),
                        "--force_length", str(K), "--leave_threshold", str(thr), "--num_leave_compute", str(N), "--tau", "10",
                        "--buffer-size", str(B * T)])
    tab, train_envs, st, policy, coll = ex.build(args, table_seed=int(z["seed_tables"]))
    st.eval()      # the fixture was recorded with the reference tracker in eval mode (dropout off)
    # the reference's weights
    st.load_state_dict({k[4:]: torch.as_tensor(z[k]) for k in z.files if k.startswith("trk_")})
    with torch.no_grad():
        for name, v in policy.views.items():
            v.copy_(torch.as_tensor(z["pol_" + name]).reshape(v.shape))
    cs = ex.build_test_collectors(args, policy, st)
    names = ["FB", "NX_0", f"NX_{K}"]
    assert list(cs.collector_dict) == names
    from tianshou.trainer.utils import test_episode
    res = test_episode(policy, cs, None, 1, B, None, None, users={n: z[f"{n}_users"] for n in names}, gumbel={n: z[f"{n}_gumbel"] for n in names})
    want = {k[4:].replace("__", "/"): z[k] for k in z.files if k.startswith("res_")}
    assert set(res) == set(want), set(res) ^ set(want)
    for k, w in want.items():
        got = np.asarray(res[k])
        assert got.shape == w.shape, k
        if k.endswith(("rews", "rew", "rew_std")):
            np.testing.assert_allclose(got, w, rtol=1e-13, atol=0, err_msg=k)       # float64 sums of mat[u, a] in step order
        else:
            assert np.array_equal(got, w), (k, got, w)
    # the transitions behind the dict: every collector's buffer holds the reference's actions / rewards / dones
    for n in names:
        buf = cs.collector_dict[n].buffer
        lens = z[f"{n}_buf_lens"]
        assert np.array_equal(buf._lengths, lens)
        for b in range(B):
            sl = slice(buf._offset[b], buf._offset[b] + lens[b])
            assert np.array_equal(buf.act[sl], z[f"{n}_acts"][b, :lens[b]]), (n, b)
            np.testing.assert_allclose(buf.rew[sl], z[f"{n}_rews"][b, :lens[b]], rtol=1e-13)
            assert np.array_equal(buf.done[sl], z[f"{n}_dones"][b, :lens[b]])
    nx = z["NX_0_acts"]
    assert all(len(set(r[r >= 0].tolist())) == (r >= 0).sum() for r in nx), "NX_0: no id twice (fixture sanity)"
    assert (z[f"res_NX_{K}_lens"] == K).all()
