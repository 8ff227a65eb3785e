// rollout.hip -- the Collector hot loop as a stream of launches with no host synchronisation (reference:
// core/collector.py:219-317).  Per vector step: policy trunk + MFMA actor head + merge (policy.hip), env step
// (env.hip), tracker decode step (tracker.hip); finished envs propagate as act = -1 so no compaction is needed and
// every env keeps its row (RNG keyed by env id -> results independent of which other envs are still alive).
#include "common.h"

namespace cirs {

__global__ __launch_bounds__(256) void mark_visited_kernel(const int64_t* __restrict__ act, int n, int n_items,
                                                           uint32_t* __restrict__ visited) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long a = act[j];
    if (a < 0) return;
    const int words = (n_items + 31) / 32;
    atomicOr(&visited[(size_t)j * words + (a >> 5)], 1u << (a & 31));
}

// core/collector.py:253-258: with force_length every env's done flag is replaced by (cnt_loop >= force_length)
__global__ __launch_bounds__(256) void force_done_kernel(uint8_t* __restrict__ st_done, uint8_t* __restrict__ done_row,
                                                         const int64_t* __restrict__ act, int n, int force_done) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (act[j] < 0) return;
    st_done[j] = (uint8_t)force_done;
    done_row[j] = (uint8_t)force_done;
}

// rows -> (raw uid, raw pid, feats, duration) of the chosen pair; finished rows (act < 0) score item 0 and are ignored
__global__ __launch_bounds__(256) void online_pairs_kernel(const int32_t* __restrict__ env_user, const int64_t* __restrict__ act, int n,
                                                           cirs_online_reward o) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long a = act[j] < 0 ? 0 : act[j];
    o.uid_buf[j] = o.raw_uid[env_user[j]];
    o.pid_buf[j] = o.raw_pid[a];
#pragma unroll
    for (int q = 0; q < 4; ++q) o.feat_buf[(size_t)j * 4 + q] = o.item_feats[(size_t)a * 4 + q];
    o.dur_buf[j] = o.item_dur[a];
}

}  // namespace cirs

extern "C" int cirs_rollout_steps(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                                  const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w,
                                  cirs_tracker_state* trk_st, const cirs_policy_cfg* pol_cfg,
                                  const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env,
                                  int32_t t_begin, int32_t t_end, uint64_t seed, uint32_t rng_base, uint32_t* visited,
                                  int32_t force_length, void* workspace, int64_t workspace_bytes, void* stream) {
    return cirs_rollout_steps_online(env_cfg, env_tab, env_st, trk_cfg, trk_w, trk_st, pol_cfg, pol_w, traj, n_env, t_begin, t_end,
                                     seed, rng_base, visited, force_length, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int cirs_rollout_steps_online(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                                  const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w,
                                  cirs_tracker_state* trk_st, const cirs_policy_cfg* pol_cfg,
                                  const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env,
                                  int32_t t_begin, int32_t t_end, uint64_t seed, uint32_t rng_base, uint32_t* visited,
                                  int32_t force_length, const cirs_online_reward* online, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    using namespace cirs;
    cirs_env_tables tab_local;
    if (online) {
        CIRS_REQUIRE(env_tab && online->cfg && online->w && online->raw_uid && online->raw_pid && online->item_feats && online->item_dur &&
                     online->pred_minmax && online->uid_buf && online->pid_buf && online->feat_buf && online->dur_buf && online->pred_buf,
                     "online reward: null field");
        tab_local = *env_tab;
        tab_local.pred_online = online->pred_buf;
        tab_local.pred_minmax = online->pred_minmax;
        env_tab = &tab_local;
    }
    CIRS_REQUIRE(env_cfg && env_tab && env_st && trk_cfg && trk_w && trk_st && pol_cfg && pol_w && traj, "null argument");
    CIRS_REQUIRE(traj->obs && traj->act && traj->rew && traj->done && traj->logp && traj->value && traj->ctr, "trajectory pointer null");
    CIRS_REQUIRE(n_env > 0 && t_begin >= 0 && t_end <= env_cfg->max_turn && t_begin <= t_end, "bad step range");
    CIRS_REQUIRE(trk_cfg->n_env == n_env, "tracker n_env mismatch");
    CIRS_REQUIRE(trk_cfg->dim_state == pol_cfg->dim_state, "tracker/policy dim_state mismatch");
    CIRS_REQUIRE(trk_cfg->max_len >= env_cfg->max_turn + 1, "tracker max_len < max_turn + 1");
    const long B = n_env, S = trk_cfg->dim_state;
    hipStream_t s = (hipStream_t)stream;
    for (int t = t_begin; t < t_end; ++t) {
        float* obs_t = traj->obs + (size_t)t * B * S;
        float* obs_n = traj->obs + (size_t)(t + 1) * B * S;
        int64_t* act_t = traj->act + (size_t)t * B;
        double* rew_t = traj->rew + (size_t)t * B;
        uint8_t* done_t = traj->done + (size_t)t * B;
        // policy(obs_t): finished envs (env_st->done) are skipped and get act = -1
        if (int rc = cirs_actor_sample(pol_cfg, pol_w, obs_t, S, n_env, nullptr, seed, rng_base + (uint32_t)t, nullptr,
                                       visited, env_st->done, act_t, traj->logp + (size_t)t * B,
                                       traj->value + (size_t)t * B, workspace, workspace_bytes, stream))
            return rc;
        if (visited) {
            hipLaunchKernelGGL(mark_visited_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, s, act_t, n_env,
                               pol_cfg->n_items, visited);
            CIRS_CHECK_LAUNCH("mark_visited_kernel");
        }
        if (online) {  // score the chosen (user, item) pairs with the DeepFM user model
            hipLaunchKernelGGL(online_pairs_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, s, env_st->user, act_t, n_env, *online);
            CIRS_CHECK_LAUNCH("online_pairs_kernel");
            if (int rc = cirs_deepfm_forward(online->cfg, online->w, online->uid_buf, online->pid_buf, online->feat_buf, online->dur_buf,
                                             n_env, online->pred_buf, stream))
                return rc;
        }
        // env.step: obs_next id == action, so the int64 obs row doubles as scratch we do not keep
        if (int rc = cirs_env_step(env_cfg, env_tab, env_st, act_t, nullptr, n_env, (int64_t*)workspace, rew_t, done_t,
                                   traj->ctr + (size_t)t * B, nullptr, stream))
            return rc;
        if (force_length > 0) {
            hipLaunchKernelGGL(force_done_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, s, env_st->done, done_t, act_t,
                               n_env, (t + 1 >= force_length) ? 1 : 0);
            CIRS_CHECK_LAUNCH("force_done_kernel");
        }
        // preprocess_fn(obs_next, rew): tracker appends one position for every env that acted this step
        if (int rc = cirs_tracker_step(trk_cfg, trk_w, trk_st, act_t, rew_t, nullptr, nullptr, n_env, obs_n, S, stream))
            return rc;
    }
    return CIRS_OK;
}
