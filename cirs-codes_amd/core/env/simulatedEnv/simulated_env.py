"""SimulatedEnv (reference core/env/simulatedEnv/simulated_env.py:17-193): the training env = real transitions/exit rule
+ user-model reward discounted by the exposure effect.

KuaishouEnv-v0: a spec object; the arithmetic is csrc/env.hip (one launch per vector step).
VirtualTB-v0  : BASELINE configs[0], CPU plumbing like in the reference: a host env with the reference's reset / step protocol --
                the wrapped VirtualTB decides `done`, the exposure effect uses L2 distances between the 27-d actions
                (core/util.py:21-30), the reward is user_model.forward on [user (88) | reward, 0, turn | action (27)] clamped to
                [0, 10] (simulated_env.py:78-86) and discounted by the exposure effect (v1: r / (1 + e), v2: r - e)."""
import numpy as np

try:
    import gym
except ImportError:
    from cirs_hip import gymlite as gym


class SimulatedEnv(gym.Env):
    simulated = True

    def __init__(self, user_model=None, task_name: str = "KuaishouEnv-v0", version: str = "v1", tau: float = 1.0,
                 use_exposure_intervention=True, alpha_u=None, beta_i=None, normed_mat=None, gamma_exposure=1, r_decay=1):
        if task_name not in ("KuaishouEnv-v0", "VirtualTB-v0"):
            raise ValueError(f"unknown task {task_name}")
        self.user_model = user_model.eval() if hasattr(user_model, "eval") else user_model
        self.env_task = gym.make(task_name)
        self.observation_space = self.env_task.observation_space
        self.action_space = self.env_task.action_space
        self.env_name, self.version, self.tau = task_name, version, tau
        self.use_exposure_intervention = use_exposure_intervention
        self.alpha_u, self.beta_i, self.normed_mat = alpha_u, beta_i, normed_mat
        self.gamma_exposure, self.r_decay = gamma_exposure, r_decay
        if task_name == "KuaishouEnv-v0":
            self.n_users, self.n_items = self.env_task.n_users, self.env_task.n_items
        else:
            self.cum_reward, self.total_turn = 0, 0
            self._reset_history()

    # ---- VirtualTB-v0: host stepping (reference simulated_env.py:44-193, VirtualTB branches) --------------------------------
    def seed(self, sd=0):
        import torch
        torch.manual_seed(sd)

    def _reset_history(self):
        self.history_action = np.zeros([self.env_task.max_turn, self.env_task.action_space.shape[0]])
        self.history_exposure = {}
        self.max_history = 0

    def reset(self):
        assert self.env_name == "VirtualTB-v0", "the KuaishouEnv branch is stepped on the device (tianshou.env.DummyVectorEnv)"
        self.cum_reward, self.total_turn, self.reward, self.action = 0, 0, 0, None
        self.env_task.action = None
        self.state = self.env_task.reset()
        self._reset_history()
        self.cur_user = self.state[:-3]
        return self.state

    def _exposure_effect(self, t, action):
        if t == 0:
            return 0
        diff = np.asarray(action) - self.history_action[:t]
        dist = np.linalg.norm(diff, axis=1)                          # compute_action_distance, VirtualTB branch
        if self.tau <= 0:                                           # compute_exposure: tau <= 0 switches the effect off
            return 0
        return float(np.sum(np.exp(-(t - np.arange(t)) * dist / self.tau))) * self.gamma_exposure

    def _pred_reward(self, exposure_effect, action):
        import torch
        feature = np.concatenate((self.cur_user, np.array([self.reward, 0, self.total_turn]), action), axis=-1)
        x = torch.unsqueeze(torch.tensor(feature, device=getattr(self.user_model, "device", "cpu"), dtype=torch.float), 0)
        pred = self.user_model.forward(x).detach().cpu().numpy().squeeze()
        pred = min(max(pred, 0), 10)
        # clip0 is np.amax(x, 0): the identity on scalars -- negative v2 rewards pass through
        return pred / (1.0 + exposure_effect) if self.version == "v1" else pred - exposure_effect

    def step(self, action):
        assert self.env_name == "VirtualTB-v0", "the KuaishouEnv branch is stepped on the device (tianshou.env.DummyVectorEnv)"
        self.action = action
        _, _, real_done, _ = self.env_task.step(action)
        t = int(self.total_turn)
        exposure_effect = self._exposure_effect(t, action) if self.use_exposure_intervention else 0
        if t < self.env_task.max_turn:
            assert self.max_history == t
            self.history_action[t] = np.expand_dims(action, 0)
            self.history_exposure[t] = exposure_effect
            self.max_history += 1
        pred_reward = self._pred_reward(exposure_effect, action)
        self.reward = pred_reward
        self.cum_reward += pred_reward
        self.total_turn = self.env_task.total_turn
        self.state = np.concatenate((self.action, np.array([pred_reward, 0.0, self.total_turn])), axis=-1)
        return self.state, pred_reward, real_done, {"CTR": self.cum_reward / self.total_turn / 10}

    def batch_key(self):
        return self.env_task.batch_key() + (id(self.normed_mat), id(self.alpha_u), self.version, self.tau, self.gamma_exposure,
                                            self.r_decay, self.use_exposure_intervention)

    def __getattr__(self, key):  # mat, lbe_photo, max_turn ... are read through the wrapped env by the scripts
        if key.startswith("_") or key == "env_task":
            raise AttributeError(key)
        return getattr(self.env_task, key)

    def build_device_env(self, n_env, device="cuda"):
        from cirs_hip.env import DeviceEnv
        t = self.env_task
        tables = t.device_tables(normed_mat=self.normed_mat, alpha_u=self.alpha_u, beta_i=self.beta_i, device=device)
        return DeviceEnv(tables, n_env, num_leave_compute=t.num_leave_compute, leave_threshold=t.leave_threshold,
                         max_turn=t.max_turn, tau=self.tau, gamma_exposure=self.gamma_exposure, version=self.version,
                         r_decay=self.r_decay, use_exposure_intervention=self.use_exposure_intervention, simulated=True)
