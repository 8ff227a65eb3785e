"""SimulatedEnv (reference core/env/simulatedEnv/simulated_env.py:17-193): the training env = real transitions/exit rule
+ user-model reward discounted by the exposure effect.  Spec object; the arithmetic is csrc/env.hip."""
try:
    import gym
except ImportError:
    from cirs_hip import gymlite as gym


class SimulatedEnv(gym.Env):
    simulated = True

    def __init__(self, user_model=None, task_name: str = "KuaishouEnv-v0", version: str = "v1", tau: float = 1.0,
                 use_exposure_intervention=True, alpha_u=None, beta_i=None, normed_mat=None, gamma_exposure=1, r_decay=1):
        if task_name != "KuaishouEnv-v0":
            raise NotImplementedError("VirtualTB-v0 is CPU plumbing in the reference's C1 config and is not on the MI355X path")
        self.user_model = user_model.eval() if hasattr(user_model, "eval") else user_model
        self.env_task = gym.make(task_name)
        self.observation_space = self.env_task.observation_space
        self.action_space = self.env_task.action_space
        self.env_name, self.version, self.tau = task_name, version, tau
        self.use_exposure_intervention = use_exposure_intervention
        self.alpha_u, self.beta_i, self.normed_mat = alpha_u, beta_i, normed_mat
        self.gamma_exposure, self.r_decay = gamma_exposure, r_decay
        self.n_users, self.n_items = self.env_task.n_users, self.env_task.n_items

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
