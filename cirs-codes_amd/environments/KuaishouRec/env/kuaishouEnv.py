"""KuaishouEnv (reference environments/KuaishouRec/env/kuaishouEnv.py:30-231) as a spec object for the batched device env.

Construction keeps the reference's keyword arguments.  One instance describes ONE env of a vector env; the arithmetic
of `step` / `_determine_whether_to_leave` lives in csrc/env.hip and is executed for the whole vector env at once
(tianshou.env.DummyVectorEnv -> cirs_hip.env.DeviceEnv).  Tables are uploaded once per distinct table set."""
import numpy as np

try:
    import gym
    from gym import spaces
except ImportError:  # no gym in the image: the local stand-in provides Env/spaces/register/make
    from cirs_hip import gymlite as gym
    spaces = gym.spaces

_TABLE_CACHE = {}


def _classes(lbe):
    return np.asarray(lbe.classes_ if hasattr(lbe, "classes_") else lbe)


def _to_numpy(x):
    return x.to_numpy() if hasattr(x, "to_numpy") else np.asarray(x)


class KuaishouEnv(gym.Env):
    metadata = {"render.modes": ["human"]}
    simulated = False

    def __init__(self, mat=None, lbe_user=None, lbe_photo=None, list_feat=None, df_photo_env=None, df_dist_small=None,
                 num_leave_compute=5, leave_threshold=1, max_turn=100):
        if mat is None:
            raise NotImplementedError("KuaishouEnv.load_mat needs the KuaiRec CSVs, which the reference does not ship "
                                      "(.gitignore:8-10); pass the tables explicitly (SURVEY §8(f3): loaders are a next row)")
        self.max_turn = max_turn
        self.mat, self.lbe_user, self.lbe_photo = mat, lbe_user, lbe_photo
        self.list_feat, self.df_photo_env, self.df_dist_small = list_feat, df_photo_env, df_dist_small
        # categories per env-encoded item (kuaishouEnv.py:49)
        self.list_feat_small = [self.list_feat[int(x)] for x in _classes(lbe_photo)]
        self.observation_space = spaces.Box(low=0, high=len(self.mat) - 1, shape=(1,), dtype=np.int32)
        self.action_space = spaces.Box(low=0, high=self.mat.shape[1] - 1, shape=(1,), dtype=np.int32)
        self.num_leave_compute = num_leave_compute
        self.leave_threshold = leave_threshold
        self.n_users, self.n_items = self.mat.shape

    # ---- spec protocol used by tianshou.env.DummyVectorEnv -----------------------------------------------------
    def batch_key(self):
        return (id(self.mat), id(self.df_dist_small), self.num_leave_compute, self.leave_threshold, self.max_turn)

    def item_cats(self):
        cats = np.full((self.n_items, 4), -1, dtype=np.int32)
        for i, lst in enumerate(self.list_feat_small):
            assert len(lst) <= 4, "KuaiRec items carry at most 4 categories (feat0..feat3)"
            cats[i, :len(lst)] = lst
        return cats

    def device_tables(self, normed_mat=None, alpha_u=None, beta_i=None, device="cuda"):
        from cirs_hip.env import DeviceEnvTables
        key = (id(self.mat), id(self.df_dist_small), id(normed_mat), id(alpha_u), str(device))
        if key not in _TABLE_CACHE:
            a_env = b_env = None
            if alpha_u is not None:  # alpha_u[lbe_user.inverse_transform(u)], beta_i[...] (simulated_env.py:158-161)
                a_env = np.asarray(alpha_u)[_classes(self.lbe_user), 0].astype(np.float64)
                b_env = np.asarray(beta_i)[_classes(self.lbe_photo), 0].astype(np.float64)
            dist = None if self.df_dist_small is None else _to_numpy(self.df_dist_small)
            _TABLE_CACHE[key] = DeviceEnvTables(self.mat, normed_mat, self.item_cats(), dist=dist, alpha_env=a_env, beta_env=b_env,
                                                device=device, build_dist_on_device=dist is None and False)
        return _TABLE_CACHE[key]

    def build_device_env(self, n_env, device="cuda"):
        from cirs_hip.env import DeviceEnv
        return DeviceEnv(self.device_tables(device=device), n_env, num_leave_compute=self.num_leave_compute,
                         leave_threshold=self.leave_threshold, max_turn=self.max_turn, simulated=False)

    @staticmethod
    def load_mat():
        raise NotImplementedError("KuaiRec data files are not distributed with the reference (SURVEY §0); see make_tables for synthetic ones")

    @staticmethod
    def compute_normed_reward(user_model, lbe_user, lbe_photo, df_photo_env):
        """Full U x I DeepFM sweep + global min-max in float64 (kuaishouEnv.py:113-145) on the device."""
        items = _classes(lbe_photo)
        info = df_photo_env.loc[items]
        feats = info[["feat0", "feat1", "feat2", "feat3"]].to_numpy()
        dur = info["photo_duration"].to_numpy()
        return user_model.device_model().normed_reward(_classes(lbe_user), items, feats, dur).cpu().numpy()
