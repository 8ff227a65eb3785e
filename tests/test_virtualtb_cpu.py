"""CPU (BASELINE configs[0]: VirtualTaobao plumbing, no GPU -- as in the reference): the mirror's VirtualTB env, SimulatedEnv
(VirtualTB branch) and UserModel_MMOE against trajectories recorded from the reference itself (tests/golden/virtualtb.npz, made
by oracle/gen_golden.py:gen_virtualtb with the shipped simulator weights and torch.manual_seed): same generator-call order, so
states / rewards / exit decisions / CTR are reproduced exactly."""
import collections
import os

import numpy as np
import torch

from cirs_hip import gymlite

gym = gymlite.install()


def _run(env, actions):
    s = env.reset()
    states, rews, dones, ctrs = [np.asarray(s, np.float64)], [], [], []
    for a in actions:
        s, r, d, info = env.step(a)
        states.append(np.asarray(s, np.float64)); rews.append(float(r)); dones.append(bool(d)); ctrs.append(float(info["CTR"]))
        if d:
            states.append(np.asarray(env.reset(), np.float64))
    return states, np.array(rews), np.array(dones), np.array(ctrs)


def _check_states(states, want, want_len):
    assert [len(s) for s in states] == want_len.tolist()
    for s, w, n in zip(states, want, want_len):
        np.testing.assert_array_equal(s, w[:n])


def test_virtualtb_env_reproduces_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "virtualtb.npz"))
    N, thr, T = z["env_params"]
    from gym.envs.registration import register
    register(id="VirtualTB-v0", entry_point="environments.VirtualTaobao.virtualTB.envs.virtualTB:VirtualTB",
             kwargs=dict(num_leave_compute=int(N), leave_threshold=float(thr), max_turn=int(T), data_dir=os.path.join(golden_dir, "virtualtb")))
    torch.manual_seed(11)
    env = gym.make("VirtualTB-v0")
    assert env.observation_space.shape == (91,) and env.action_space.shape == (27,)
    states, rews, dones, ctrs = _run(env, z["actions"])
    assert np.array_equal(dones, z["env_dones"]) and dones.sum() >= 5 and (~dones).sum() >= 10      # exit rule + max_turn both exercised
    np.testing.assert_array_equal(rews, z["env_rews"])
    np.testing.assert_array_equal(ctrs, z["env_ctr"])
    _check_states(states, z["env_states"], z["env_state_len"])


def test_mmoe_user_model_and_simulated_env_reproduce_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "virtualtb.npz"))
    from core.user_model_mmoe import UserModel_MMOE
    from deepctr_torch.inputs import DenseFeat
    x_columns = [DenseFeat("user_feat", 91), DenseFeat("feat_item", 27)]        # CIRS-UserModel-taobao.py:100
    y_columns = [DenseFeat("y", 1)]
    tasks = collections.OrderedDict({f.name: "regression" for f in y_columns})
    task_logit_dim = {f.name: f.dimension for f in y_columns}
    model = UserModel_MMOE(x_columns, y_columns, len(tasks), tasks, task_logit_dim, dnn_hidden_units=(128, 128), seed=2022, device="cpu")
    sd = {k[len("mmoe_"):]: torch.as_tensor(z[k]) for k in z.files if k.startswith("mmoe_") and k not in ("mmoe_x", "mmoe_y")}
    assert set(sd) == set(model.state_dict()), set(sd) ^ set(model.state_dict())     # the reference's state_dict names
    model.load_state_dict(sd)
    model.eval()
    with torch.no_grad():
        y = model.forward(torch.as_tensor(z["mmoe_x"])).numpy()
    np.testing.assert_allclose(y, z["mmoe_y"], rtol=1e-6, atol=1e-6)
    from core.inputs import get_dataset_columns
    u, a, f, hu, ha, hf = get_dataset_columns(27, envname="VirtualTB-v0")
    assert (u[0].dimension, a[0].dimension, f[0].dimension, hu, ha, hf) == (88, 27, 1, True, True, True)
    N, thr, T = z["env_params"]
    from gym.envs.registration import register
    register(id="VirtualTB-v0", entry_point="environments.VirtualTaobao.virtualTB.envs.virtualTB:VirtualTB",
             kwargs=dict(num_leave_compute=int(N), leave_threshold=float(thr), max_turn=int(T), data_dir=os.path.join(golden_dir, "virtualtb")))
    for ver in ("v1", "v2"):
        tau, gam = z[f"sim_{ver}_cfg"]
        register(id="SimulatedEnv-v0", entry_point="core.env.simulatedEnv.simulated_env:SimulatedEnv",
                 kwargs=dict(user_model=model, task_name="VirtualTB-v0", version=ver, tau=float(tau), gamma_exposure=float(gam)))
        torch.manual_seed(23)
        sim = gym.make("SimulatedEnv-v0")
        states, rews, dones, ctrs = _run(sim, z["actions"][:24])
        assert np.array_equal(dones, z[f"sim_{ver}_dones"])
        np.testing.assert_allclose(rews, z[f"sim_{ver}_rews"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(ctrs, z[f"sim_{ver}_ctr"], rtol=1e-6, atol=1e-7)
        assert [len(s) for s in states] == z[f"sim_{ver}_state_len"].tolist()
        for s, w, n in zip(states, z[f"sim_{ver}_states"], z[f"sim_{ver}_state_len"]):
            np.testing.assert_allclose(s, w[:n], rtol=1e-6, atol=1e-7)
    assert (z["sim_v2_rews"] < 0).any() or True      # v2 rewards may go negative: clip0 is the identity on scalars
