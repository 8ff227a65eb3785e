"""KuaishouEnv (reference environments/KuaishouRec/env/kuaishouEnv.py:30-231) as a spec object for the batched device env.

Construction keeps the reference's keyword arguments.  One instance describes ONE env of a vector env; the arithmetic
of `step` / `_determine_whether_to_leave` lives in csrc/env.hip and is executed for the whole vector env at once
(tianshou.env.DummyVectorEnv -> cirs_hip.env.DeviceEnv).  Tables are uploaded once per distinct table set."""
import json
import os

import numpy as np

try:
    import gym
    from gym import spaces
except ImportError:  # no gym in the image: the local stand-in provides Env/spaces/register/make
    from cirs_hip import gymlite as gym
    spaces = gym.spaces

_TABLE_CACHE = {}
DATAPATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _classes(lbe):
    return np.asarray(lbe.classes_ if hasattr(lbe, "classes_") else lbe)


def _to_numpy(x):
    return x.to_numpy() if hasattr(x, "to_numpy") else np.asarray(x)


class KuaishouEnv(gym.Env):
    metadata = {"render.modes": ["human"]}
    simulated = False

    def __init__(self, mat=None, lbe_user=None, lbe_photo=None, list_feat=None, df_photo_env=None, df_dist_small=None,
                 num_leave_compute=5, leave_threshold=1, max_turn=100):
        self.max_turn = max_turn
        if mat is not None:
            self.mat, self.lbe_user, self.lbe_photo = mat, lbe_user, lbe_photo
            self.list_feat, self.df_photo_env, self.df_dist_small = list_feat, df_photo_env, df_dist_small
        else:  # kuaishouEnv.py:43-45: read the KuaiRec files under DATAPATH (not shipped with the reference)
            self.mat, self.lbe_user, self.lbe_photo, self.list_feat, self.df_photo_env, self.df_dist_small = self.load_mat()
        # categories per env-encoded item (kuaishouEnv.py:49)
        self.list_feat_small = [self.list_feat[int(x)] for x in _classes(self.lbe_photo)]
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
            # a missing distance table is NOT rebuilt silently: a bare KuaishouEnv (no exposure term) does not read it
            tables = DeviceEnvTables(self.mat, normed_mat, self.item_cats(), dist=dist, alpha_env=a_env, beta_env=b_env,
                                     device=device, build_dist_on_device=False)
            # the key is made of id()s: keep the keyed host objects alive next to the entry, otherwise CPython may hand a
            # collected object's id to a different matrix and a later env would silently hit this entry
            _TABLE_CACHE[key] = (tables, (self.mat, self.df_dist_small, normed_mat, alpha_u))
        return _TABLE_CACHE[key][0]

    def build_device_env(self, n_env, device="cuda"):
        from cirs_hip.env import DeviceEnv
        return DeviceEnv(self.device_tables(device=device), n_env, num_leave_compute=self.num_leave_compute,
                         leave_threshold=self.leave_threshold, max_turn=self.max_turn, simulated=False)

    @staticmethod
    def load_mat(DATAPATH=None):
        """The KuaiRec files -> (mat, lbe_user, lbe_photo, list_feat, df_photo_env, df_dist_small), same objects as reference
        kuaishouEnv.py:61-111: `small_matrix.csv` (user_id, photo_id, watch_ratio clipped at 5) as a dense user x item
        matrix over the sorted unique ids, `item_categories.json` as per-photo category lists and feat0..feat3 columns
        (ids shifted by one, 0 = no category), `photo_mean_duration.json`, and the item-item distance table (cached CSV,
        otherwise built on the device)."""
        import pandas as pd
        from sklearn.preprocessing import LabelEncoder
        from core.util import get_distance_mat
        root = DATAPATH or os.environ.get("CIRS_DATAPATH") or globals()["DATAPATH"]
        log = pd.read_csv(os.path.join(root, "small_matrix.csv"), usecols=["user_id", "photo_id", "watch_ratio"])
        ratio = np.minimum(log["watch_ratio"].to_numpy(dtype=np.float64), 5.0)
        lbe_user, lbe_photo = LabelEncoder().fit(log["user_id"].unique()), LabelEncoder().fit(log["photo_id"].unique())
        rows, cols = lbe_user.transform(log["user_id"]), lbe_photo.transform(log["photo_id"])
        mat = np.zeros((len(lbe_user.classes_), len(lbe_photo.classes_)), dtype=np.float64)
        np.add.at(mat, (rows, cols), ratio)                      # duplicate (user, photo) rows add up, like a COO matrix
        mat[~np.isfinite(mat)] = ratio.mean()
        with open(os.path.join(root, "item_categories.json")) as fh:
            cat_json = json.load(fh)
        list_feat = [cat_json[str(i)]["feature_index"] for i in range(len(cat_json))]
        feat = np.zeros((len(list_feat), 4), dtype=np.int64)     # category id + 1, 0 = empty slot
        for i, cats in enumerate(list_feat):
            feat[i, :len(cats)] = np.asarray(cats, dtype=np.int64) + 1
        with open(os.path.join(root, "photo_mean_duration.json")) as fh:
            duration = {int(k): v for k, v in json.load(fh).items()}
        seen = log["photo_id"].unique()                           # first-seen order, as the reference indexes df_photo_env
        df_photo_env = pd.DataFrame(feat[seen], index=pd.Index(seen, name="photo_id"), columns=["feat0", "feat1", "feat2", "feat3"])
        df_photo_env["photo_duration"] = [duration[int(x)] for x in seen]
        df_dist_small = get_distance_mat(list_feat, lbe_photo.classes_, DATAPATH=root)
        return mat, lbe_user, lbe_photo, list_feat, df_photo_env, df_dist_small

    @staticmethod
    def compute_normed_reward(user_model, lbe_user, lbe_photo, df_photo_env):
        """Full U x I DeepFM sweep + global min-max in float64 (kuaishouEnv.py:113-145) on the device."""
        items = _classes(lbe_photo)
        info = df_photo_env.loc[items]
        feats = info[["feat0", "feat1", "feat2", "feat3"]].to_numpy()
        dur = info["photo_duration"].to_numpy()
        return user_model.device_model().normed_reward(_classes(lbe_user), items, feats, dur).cpu().numpy()
