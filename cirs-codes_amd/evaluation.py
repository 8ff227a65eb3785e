"""Evaluation callbacks of the CIRS protocol on device trajectories (reference evaluation.py).

  get_feat_dominate_dict   evaluation.py:10-77  ("ifeat_feat": share of recommendations carrying a dominating category)
  Callback_Coverage_Count  evaluation.py:286-371 (CV, CV_turn and ifeat_* for the FB / NX_0 / NX_k test collectors)

The reference walks the replay buffers on the host (buffer.prev / buffer.next); here each collector's buffer carries the
device trajectory of its fused rollout and the counts come from cirs_eval_coverage (integer kernel, bit-exact)."""
import numpy as np
import torch

from cirs_hip.evalmetrics import CoverageCounter, dominated_values, item_flags


def _feat_columns(df_item_val):
    return [c for c in df_item_val.columns if str(c).startswith("feat")]  # .filter(regex="^feat", axis=1)


def get_feat_dominate_dict(df_item_val, all_acts_origin, item_feat_domination, top_rate=0.6):
    """Same contract as the reference: df_item_val is indexed by the ORIGINAL item ids, all_acts_origin are original ids."""
    if item_feat_domination is None:  # for yahoo
        return dict()
    out = {}
    pos = df_item_val.index.get_indexer(np.asarray(all_acts_origin))
    assert (pos >= 0).all(), "recommended item missing from df_item_val"
    counter = CoverageCounter(len(df_item_val))
    if "feat" in item_feat_domination:  # kuairec / kuairand: multi-hot categories in the feat* columns
        flags = item_flags(df_item_val[_feat_columns(df_item_val)].to_numpy(), dominated_values(item_feat_domination["feat"], top_rate))
        _, n, fl = counter.count(torch.as_tensor(pos), torch.as_tensor(flags))
        out["ifeat_feat"] = fl / n
    else:  # coat: one column per feature
        for feat_name, sorted_items in item_feat_domination.items():
            flags = item_flags(df_item_val[[feat_name]].to_numpy(), dominated_values(sorted_items, top_rate))
            _, n, fl = counter.count(torch.as_tensor(pos), torch.as_tensor(flags))
            out["ifeat_" + feat_name] = fl / n
    return out


def interactive_evaluation(model, env, dataset_val, is_softmax, epsilon, is_ucb, k, need_transform, num_trajectory, item_feat_domination,
                           remove_recommended, force_length=0, top_rate=0.6, users=None, seed=0):
    """reference evaluation.py:79-151 with the num_trajectory trajectories run in LOCK-STEP on the device: one catalogue
    sweep per trajectory user (the model is static), then per step cirs_select_items -> env step (cirs_rollout_static).
    Same result dict.  With is_ucb the arm counts change after every single recommendation of every trajectory
    (core/user_model.py:303-314, 340-342): that path runs the trajectories one after the other, one device step per
    recommendation (_interactive_evaluation_ucb).  Users can be supplied (the reference draws them with the unseeded `random`)."""
    from cirs_hip.evalmetrics import CoverageCounter
    from cirs_hip.static_policy import StaticRollout
    assert k == 1 and need_transform, "built for KuaishouEnv (need_transform=True, k=1)"
    B = int(num_trajectory)
    df_item_val = dataset_val.df_photo_env
    item_index = df_item_val.index.to_numpy()
    assert np.array_equal(item_index, np.asarray(env.lbe_photo.classes_)), "df_photo_env must be in env item order (lbe_photo.classes_)"
    I = len(item_index)
    dev_env = env.build_device_env(B)
    if users is None:
        users = np.random.randint(0, env.mat.shape[0], B)
    users = np.asarray(users)
    raw_users = np.asarray(env.lbe_user.classes_)[users]
    feats = df_item_val[["feat0", "feat1", "feat2", "feat3"]].to_numpy()
    dur = df_item_val["photo_duration"].to_numpy()
    scores, _ = model.device_model().sweep(raw_users, item_index, feats, dur)   # [B, I] u_value of every trajectory user
    ro = StaticRollout(dev_env)
    T = dev_env.max_turn if force_length <= 0 else min(force_length, dev_env.max_turn)
    if is_ucb:
        return _interactive_evaluation_ucb(model, env, dataset_val, is_softmax, epsilon, B, item_feat_domination, remove_recommended,
                                           force_length, top_rate, users, raw_users, item_index, feats, dur, seed)
    ro.run(torch.as_tensor(users), scores, softmax=is_softmax, epsilon=epsilon, seed=seed, remove_recommended=remove_recommended,
           force_length=force_length, n_steps=T)
    valid = ro.act >= 0
    total_turns = int(valid.sum())
    cumulative_reward = float((ro.rew * valid).sum())
    total_click_loss = float(((ro.value.double() - ro.rew).abs() * valid).sum())
    ctr = cumulative_reward / total_turns
    click_loss = total_click_loss / total_turns
    cc = CoverageCounter(I, device=ro.act.device)
    flags = None
    if item_feat_domination is not None and "feat" in item_feat_domination:
        flags = torch.as_tensor(item_flags(feats, dominated_values(item_feat_domination["feat"], top_rate)))
    hit_item, n_acts, n_fl = cc.count(ro.act, flags)
    eval_result_RL = {"click_loss": click_loss, "CV": f"{hit_item / I:.5f}", "CV_turn": f"{hit_item / n_acts:.5f}", "ctr": ctr,
                      "len_tra": total_turns / B, "R_tra": cumulative_reward / B}
    if flags is not None:
        eval_result_RL["ifeat_feat"] = n_fl / n_acts
    if remove_recommended:
        eval_result_RL = {f"NX_{force_length}_" + key: v for key, v in eval_result_RL.items()}
    interactive_evaluation.last_rollout = ro
    return eval_result_RL


def _interactive_evaluation_ucb(model, env, dataset_val, is_softmax, epsilon, B, item_feat_domination, remove_recommended, force_length,
                                top_rate, users, raw_users, item_index, feats, dur, seed):
    """is_ucb=True (reference evaluation.py:87-120 with core/user_model.py:303-314, 340-342): the bonus sqrt(2 ln n_rec / n_each) is
    added to u_value when nothing is excluded (recommended_ids == [], i.e. always without remove_recommended, only at the first step
    of a trajectory with it), and (n_rec, n_each) change after EVERY recommendation -- sequential by construction: trajectories one
    after the other on a 1-env device env, one cirs_rollout_static step per recommendation, counts kept on the model like the
    reference's.  reward_pred (click_loss) includes the bonus when it was applied, as in the reference (value_rec = u_value[...])."""
    from cirs_hip.evalmetrics import CoverageCounter
    from cirs_hip.static_policy import StaticRollout
    I = len(item_index)
    dev_env = env.build_device_env(1)
    ro = StaticRollout(dev_env)
    T = dev_env.max_turn if force_length <= 0 else min(force_length, dev_env.max_turn)
    if not hasattr(model, "n_rec"):
        model.compile_UCB(I)
    total_turns, cumulative_reward, total_click_loss = 0, 0.0, 0.0
    all_acts = []
    for i in range(B):
        scores, _ = model.device_model().sweep(raw_users[i:i + 1], item_index, feats, dur)   # [1, I]
        ro.begin(torch.as_tensor(users[i:i + 1]), remove_recommended)
        n_acts = 0
        for t in range(T):
            bonus = None
            if not remove_recommended or n_acts == 0:
                bonus = torch.as_tensor(((2 * np.log(model.n_rec) / model.n_each) ** 0.5).astype(np.float32))
            ro.step(t, scores, softmax=is_softmax, epsilon=epsilon, seed=seed, rng_base=i * dev_env.max_turn, force_length=force_length,
                    bonus=bonus)
            a = int(ro.act[t, 0])
            if a < 0:
                break
            model.n_rec += 1
            model.n_each[a] += 1
            n_acts += 1
            all_acts.append(a)
            r = float(ro.rew[t, 0])
            total_turns += 1
            cumulative_reward += r
            total_click_loss += abs(float(ro.value[t, 0]) - r)
            if bool(ro.done[t, 0]):
                break
    ctr = cumulative_reward / total_turns
    click_loss = total_click_loss / total_turns
    acts = torch.as_tensor(np.asarray(all_acts, dtype=np.int64))
    cc = CoverageCounter(I)
    flags = None
    if item_feat_domination is not None and "feat" in item_feat_domination:
        flags = torch.as_tensor(item_flags(feats, dominated_values(item_feat_domination["feat"], top_rate)))
    hit_item, n_acts_all, n_fl = cc.count(acts, flags)
    res = {"click_loss": click_loss, "CV": f"{hit_item / I:.5f}", "CV_turn": f"{hit_item / n_acts_all:.5f}", "ctr": ctr,
           "len_tra": total_turns / B, "R_tra": cumulative_reward / B}
    if flags is not None:
        res["ifeat_feat"] = n_fl / n_acts_all
    if remove_recommended:
        res = {f"NX_{force_length}_" + key: v for key, v in res.items()}
    return res


def test_static_model_in_RL_env(model, env, dataset_val, is_softmax=True, epsilon=0, is_ucb=False, k=1, need_transform=False,
                                num_trajectory=100, item_feat_domination=None, force_length=10, top_rate=0.6, users=None, seed=0):
    """reference evaluation.py:153-176: free browsing, no-overlap, no-overlap with forced length."""
    out = {}
    for remove, fl in ((False, 0), (True, 0), (True, force_length)):
        out.update(interactive_evaluation(model, env, dataset_val, is_softmax, epsilon, is_ucb, k, need_transform, num_trajectory,
                                          item_feat_domination, remove_recommended=remove, force_length=fl, top_rate=top_rate, users=users,
                                          seed=seed))
    return out


test_static_model_in_RL_env.__test__ = False  # not a pytest test


class Callback_Coverage_Count:
    def __init__(self, test_collector_set, df_item_val, need_transform, item_feat_domination, lbe_photo, top_rate):
        self.collector_dict = test_collector_set.collector_dict
        self.num_items = test_collector_set.env.mat[0].shape[1]
        self.df_item_val = df_item_val
        self.need_transform = need_transform
        self.item_feat_domination = item_feat_domination
        self.lbe_photo = lbe_photo
        self.top_rate = top_rate
        self._counter = None
        self._flags = {}
        if item_feat_domination is not None:
            # flags per ENV item id: env id -> original id (lbe_photo.inverse_transform == classes_[id]) -> df_item_val row
            ids = np.arange(self.num_items)
            # LabelEncoder.inverse_transform(ids) == classes_[ids]
            origin = (self.lbe_photo.inverse_transform(ids) if hasattr(self.lbe_photo, "inverse_transform")
                      else np.asarray(self.lbe_photo.classes_)[ids]) if need_transform else ids
            rows = df_item_val.loc[origin]
            if "feat" in item_feat_domination:
                self._flags["ifeat_feat"] = item_flags(rows[_feat_columns(rows)].to_numpy(), dominated_values(item_feat_domination["feat"], top_rate))
            else:
                for feat_name, sorted_items in item_feat_domination.items():
                    self._flags["ifeat_" + feat_name] = item_flags(rows[[feat_name]].to_numpy(), dominated_values(sorted_items, top_rate))

    def on_epoch_begin(self, epoch):
        pass

    def on_train_begin(self):
        pass

    def on_train_end(self):
        pass

    def on_epoch_end(self, epoch, results=None, **kwargs):
        results_all = {}
        for name, collector in self.collector_dict.items():
            ro = getattr(collector.buffer, "_rollout", None)
            assert ro is not None, "collector has not collected yet"
            act = ro.traj.act  # [T, B] int64 on the device, -1 once an env has finished
            if self._counter is None:
                self._counter = CoverageCounter(self.num_items, device=act.device)
                self._flags = {k: torch.as_tensor(v).to(act.device) for k, v in self._flags.items()}
            res = {}
            hit = n = None
            for key, fl in (self._flags or {None: None}).items():
                hit, n, n_fl = self._counter.count(act, fl)
                if key is not None:
                    res[key] = n_fl / n
            res["CV"] = hit / self.num_items
            res["CV_turn"] = hit / n
            results_all.update({name + "_" + k: v for k, v in res.items()} if name != "FB" else res)
        results.update(results_all)
        return results
