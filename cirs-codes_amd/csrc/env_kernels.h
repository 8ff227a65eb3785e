// env_kernels.h -- the per-env step of SimulatedEnv(KuaishouEnv) as a wavefront-level device function, shared by the
// stand-alone env_step_kernel (env.hip) and the fused rollout step kernel (rollout.hip).  See env.hip for the design.
#pragma once
#include "common.h"

namespace cirs {

constexpr int kEnvsPerBlock = 4;

// One wavefront steps env e (row j of the call) with `action`; lane 0 writes the outputs (obs_out / expo_out may be null:
// the next observation of this env is the action id itself).
struct EnvStepResult { double reward; int done; };   // what lane 0 wrote for this env (reward 0 / done 1 for a no-op)

// Action-independent state of env e, loaded ahead of time (fused rollout: while the sampler partials are merged).
// hk[q] = history entry lane + 64 q (entries at or beyond the current turn are never used).
constexpr int kEnvPrefetchIters = 2;   // covers max_turn <= 128; longer histories fall back to loads inside the loop
struct EnvPrefetch { int u, t, done; int32_t hk[kEnvPrefetchIters]; double cum; };
__device__ __forceinline__ EnvPrefetch env_prefetch(const cirs_env_cfg& cfg, const cirs_env_state& st, int e, int lane) {
    EnvPrefetch p;
    const int32_t* hist = st.hist_action + (size_t)e * cfg.max_turn;
#pragma unroll
    for (int q = 0; q < kEnvPrefetchIters; ++q) p.hk[q] = lane + 64 * q < cfg.max_turn ? hist[lane + 64 * q] : 0;
    p.u = st.user[e]; p.t = st.turn[e]; p.done = st.done[e]; p.cum = st.cum_reward[e];
    return p;
}
// env.reset for envs j < n (kuaishouEnv.py:155-171 / simulated_env.py:63-75): thread gtid of gthreads walks (env, history slot) pairs
__device__ __forceinline__ void env_reset_body(const cirs_env_cfg& cfg, const cirs_env_state& st, const int32_t* __restrict__ users,
                                               const int32_t* __restrict__ env_ids, int n, int64_t* __restrict__ obs_out, long gtid, long gthreads) {
    const int T = cfg.max_turn;
    const long total = (long)n * T;
    for (long i = gtid; i < total; i += gthreads) {
        const int j = (int)(i / T), k = (int)(i % T);
        const int e = env_ids ? env_ids[j] : j;
        st.hist_action[(size_t)e * T + k] = 0;
        if (k == 0) {
            st.user[e] = users[j];
            st.turn[e] = 0;
            st.done[e] = 0;
            st.cum_reward[e] = 0.0;
            if (obs_out) obs_out[j] = users[j];
        }
    }
}

__device__ __forceinline__ void env_step_wave(const cirs_env_cfg& cfg, const cirs_env_tables& tab, const cirs_env_state& st, int e, int j,
                                              int64_t action, int lane, int64_t* __restrict__ obs_out, double* __restrict__ rew_out,
                                              uint8_t* __restrict__ done_out, double* __restrict__ ctr_out,
                                              double* __restrict__ expo_out, EnvStepResult* res = nullptr,
                                              const EnvPrefetch* pf = nullptr) {
    if (res) { res->reward = 0.0; res->done = 1; }
    const int T = cfg.max_turn;
    const long I = cfg.n_items;
    int32_t* hist = st.hist_action + (size_t)e * T;

    if ((pf ? pf->done : (int)st.done[e]) || action < 0 || action >= I) {  // finished envs are never stepped by the collector: no-op
        if (lane == 0) {
            if (obs_out) obs_out[j] = action;
            rew_out[j] = 0.0;
            done_out[j] = 1;
            ctr_out[j] = 0.0;
            if (expo_out) expo_out[j] = 0.0;
        }
        return;
    }
    const int u = pf ? pf->u : st.user[e];
    const int t = pf ? pf->t : st.turn[e];
    const uint32_t cats_a = tab.item_cats[action];
    // lane 0's table reads depend only on (user, action): issued now, consumed after the history scan
    double pre_pred = 0.0, pre_a = 1.0, pre_b = 1.0, pre_mat = 0.0;
    if (lane == 0) {
        if (cfg.simulated) {
            if (!tab.pred_online) pre_pred = tab.normed_mat[(size_t)u * I + action];
            if (cfg.use_exposure && cfg.has_ab) { pre_a = tab.alpha_env[u]; pre_b = tab.beta_env[action]; }
        } else {
            pre_mat = tab.mat[(size_t)u * I + action];
        }
    }

    // ---- (a) exit rule + (c) repeat count: integer scans over the history ---------------------------------
    // window = sequence_action[t-N : t] with Python negative-start wrap
    long start = (long)t - cfg.num_leave_compute;
    if (start < 0) {
        start += t;
        if (start < 0) start = 0;
    }
    if (start > t) start = t;
    // counts of each of the action's (<=4) categories inside the window, 16 bits each, packed in a u64
    unsigned long long packed_counts = 0;
    int repeat = 0;
    const bool want_exposure = cfg.simulated && cfg.use_exposure && t > 0 && cfg.tau > 0;
    double expo_part = 0.0;
    for (int k = lane, q = 0; k < t; k += CIRS_WAVE, ++q) {
        const int32_t hk = (pf && q < kEnvPrefetchIters) ? pf->hk[q] : hist[k];
        repeat += (hk == (int32_t)action);
        uint32_t cats_h = 0;
        const bool in_window = k >= start;
        if (in_window || (want_exposure && cfg.dist_mode == 1)) cats_h = tab.item_cats[hk];
        if (in_window) {
#pragma unroll
            for (int ia = 0; ia < CIRS_MAX_CATS_PER_ITEM; ++ia) {
                const uint32_t ca = (cats_a >> (8 * ia)) & 0xFFu;
                if (ca == CIRS_CAT_NONE) continue;
                int c = 0;
#pragma unroll
                for (int ih = 0; ih < CIRS_MAX_CATS_PER_ITEM; ++ih) c += (((cats_h >> (8 * ih)) & 0xFFu) == ca);
                packed_counts += (unsigned long long)c << (16 * ia);
            }
        }
        // ---- (b) exposure effect term, float64 -----------------------------------------------------------
        if (want_exposure) {
            const double d = cfg.dist_mode == 0 ? tab.dist[(size_t)action * I + hk] : jaccard_dist(cats_a, cats_h);
            const double t_diff = (double)(t - k);
            expo_part += exp(-t_diff * d / cfg.tau);
        }
    }
    packed_counts = wave_sum_u64(packed_counts);
    repeat = wave_sum_i32(repeat);
    const double exposure_effect = want_exposure ? wave_sum_f64(expo_part) : 0.0;

    if (lane != 0) return;

    int done = 0;
    if (t > 0) {
#pragma unroll
        for (int ia = 0; ia < CIRS_MAX_CATS_PER_ITEM; ++ia) {
            const uint32_t ca = (cats_a >> (8 * ia)) & 0xFFu;
            const int cnt = (int)((packed_counts >> (16 * ia)) & 0xFFFFull);
            if (ca != CIRS_CAT_NONE && cnt > cfg.leave_threshold) done = 1;
        }
    }
    if (t >= T - 1) done = 1;

    double reward, exposure_gamma = 0.0;
    if (cfg.simulated) {
        if (cfg.use_exposure && t > 0) {
            double e_new = exposure_effect;
            if (cfg.has_ab) e_new = exposure_effect * pre_a * pre_b;
            exposure_gamma = e_new * cfg.gamma_exposure;
        }
        double pred;
        if (tab.pred_online) {  // online DeepFM score of this row's (user, action), min-max normalised in float64
            const double lo = (double)tab.pred_minmax[0], hi2 = (double)tab.pred_minmax[1];
            pred = ((double)tab.pred_online[j] - lo) / (hi2 - lo);
        } else {
            pred = pre_pred;
        }
        reward = cfg.version == 1 ? pred / (1.0 + exposure_gamma) : pred - exposure_gamma;
        // num_actions[action] - 1 == occurrences before this step (this step's own append is guarded by t < T)
        const int num_repeat = (t < T) ? repeat : repeat - 1;
        if (cfg.r_decay != 1.0) reward = reward * pow(cfg.r_decay, (double)num_repeat);   // pow(1, n) == 1 exactly: skipped
    } else {
        reward = pre_mat;
    }
    if (t < T) hist[t] = (int32_t)action;
    const double cum = (pf ? pf->cum : st.cum_reward[e]) + reward;
    st.cum_reward[e] = cum;
    st.turn[e] = t + 1;
    st.done[e] = (uint8_t)done;

    if (obs_out) obs_out[j] = action;
    rew_out[j] = reward;
    done_out[j] = (uint8_t)done;
    if (res) { res->reward = reward; res->done = done; }
    ctr_out[j] = cfg.simulated ? cum / (double)(t + 1) / 10.0 : cum;
    if (expo_out) expo_out[j] = exposure_gamma;
}

}  // namespace cirs
