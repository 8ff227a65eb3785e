// env.hip -- batched SimulatedEnv(KuaishouEnv) step for gfx950.
//
// One 64-lane wavefront per environment, 4 envs per 256-thread workgroup.  Lanes stride over the episode
// history (t <= max_turn <= ~100), so the three per-step scans of the reference
//   (a) exit rule: multiset of categories over the recent-N window  (kuaishouEnv.py:199-218)
//   (b) exposure effect: sum_k exp(-(t-k) * dist[a, hist_k] / tau)   (simulated_env.py:147-168, util.py:34-46)
//   (c) repeat count of the chosen item                              (simulated_env.py:129-132,188)
// are each one gather per lane followed by a wavefront butterfly reduction.  All reward arithmetic is float64 like
// the reference's NumPy scalars; exit decisions are integer and bit-exact.
//
// HBM traffic per env-step (table mode): 4*t B history + 8*t B dist gathers (one 32 B sector each, physically)
// + 4*(N+1) B category words + 16 B (mat, normed_mat) + 16 B (alpha, beta) + ~40 B state/outputs.  The path is
// latency-bound, not bandwidth-bound: the design goal is one launch for all envs and no host sync.
#include "env_kernels.h"

namespace cirs {

__global__ __launch_bounds__(256) void env_step_kernel(cirs_env_cfg cfg, cirs_env_tables tab, cirs_env_state st,
                                                       const int64_t* __restrict__ actions,
                                                       const int32_t* __restrict__ env_ids, int n,
                                                       int64_t* __restrict__ obs_out, double* __restrict__ rew_out,
                                                       uint8_t* __restrict__ done_out, double* __restrict__ ctr_out,
                                                       double* __restrict__ expo_out) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * kEnvsPerBlock + (threadIdx.x >> 6);
    if (j >= n) return;  // wave-uniform
    const int e = env_ids ? env_ids[j] : j;
    env_step_wave(cfg, tab, st, e, j, actions[j], lane, obs_out, rew_out, done_out, ctr_out, expo_out);
}

__global__ __launch_bounds__(256) void env_reset_kernel(cirs_env_cfg cfg, cirs_env_state st,
                                                        const int32_t* __restrict__ users,
                                                        const int32_t* __restrict__ env_ids, int n,
                                                        int64_t* __restrict__ obs_out) {
    env_reset_body(cfg, st, users, env_ids, n, obs_out, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

__global__ __launch_bounds__(256) void dist_jaccard_kernel(const uint32_t* __restrict__ item_cats, int n_items,
                                                           double* __restrict__ dist_out) {
    const long total = (long)n_items * n_items;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / n_items), j = (int)(idx % n_items);
        dist_out[idx] = jaccard_dist(item_cats[i], item_cats[j]);
    }
}

static int validate_cfg(const cirs_env_cfg* cfg) {
    CIRS_REQUIRE(cfg != nullptr, "cfg is null");
    CIRS_REQUIRE(cfg->n_users > 0 && cfg->n_items > 0, "n_users/n_items must be positive");
    CIRS_REQUIRE(cfg->max_turn > 0 && cfg->max_turn <= 16383, "max_turn out of range (1..16383)");
    CIRS_REQUIRE(cfg->num_leave_compute >= 0, "num_leave_compute must be >= 0");
    CIRS_REQUIRE(cfg->version == 1 || cfg->version == 2, "version must be 1 (v1) or 2 (v2)");
    CIRS_REQUIRE(cfg->dist_mode == 0 || cfg->dist_mode == 1, "dist_mode must be 0 (table) or 1 (jaccard)");
    return CIRS_OK;
}

}  // namespace cirs

extern "C" int cirs_env_reset(const cirs_env_cfg* cfg, cirs_env_state* st, const int32_t* users,
                              const int32_t* env_ids, int32_t n, int64_t* obs_out, void* stream) {
    using namespace cirs;
    if (int rc = validate_cfg(cfg)) return rc;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(st && st->user && st->turn && st->done && st->hist_action && st->cum_reward, "env state has null field");
    CIRS_REQUIRE(users != nullptr, "users is null");
    const long total = (long)n * cfg->max_turn;
    const int grid = cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048;
    hipLaunchKernelGGL(env_reset_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *cfg, *st, users, env_ids, n,
                       obs_out);
    CIRS_CHECK_LAUNCH("env_reset_kernel");
    return CIRS_OK;
}

extern "C" int cirs_env_step(const cirs_env_cfg* cfg, const cirs_env_tables* tab, cirs_env_state* st,
                             const int64_t* actions, const int32_t* env_ids, int32_t n, int64_t* obs_out,
                             double* rew_out, uint8_t* done_out, double* ctr_out, double* expo_out, void* stream) {
    using namespace cirs;
    if (int rc = validate_cfg(cfg)) return rc;
    if (n <= 0) return CIRS_OK;  // empty batch: nothing to do (pointers may legitimately be null)
    CIRS_REQUIRE(tab && tab->item_cats, "tables: item_cats is null");
    CIRS_REQUIRE(tab->mat || cfg->simulated, "tables: mat is null");
    CIRS_REQUIRE(!cfg->simulated || tab->normed_mat || (tab->pred_online && tab->pred_minmax), "tables: neither normed_mat nor online scores given");
    CIRS_REQUIRE(cfg->dist_mode == 1 || tab->dist || !(cfg->simulated && cfg->use_exposure),
                 "tables: dist is null in table mode");
    CIRS_REQUIRE(!(cfg->simulated && cfg->has_ab) || (tab->alpha_env && tab->beta_env), "tables: alpha/beta null");
    CIRS_REQUIRE(st && st->user && st->turn && st->done && st->hist_action && st->cum_reward, "env state has null field");
    CIRS_REQUIRE(actions && obs_out && rew_out && done_out && ctr_out, "null action/output pointer");
    if (n <= 0) return CIRS_OK;
    hipLaunchKernelGGL(env_step_kernel, dim3(cdiv(n, kEnvsPerBlock)), dim3(256), 0, (hipStream_t)stream, *cfg, *tab,
                       *st, actions, env_ids, n, obs_out, rew_out, done_out, ctr_out, expo_out);
    CIRS_CHECK_LAUNCH("env_step_kernel");
    return CIRS_OK;
}

extern "C" int cirs_dist_jaccard(const uint32_t* item_cats, int32_t n_items, double* dist_out, void* stream) {
    using namespace cirs;
    CIRS_REQUIRE(item_cats && dist_out && n_items > 0, "bad arguments");
    const long total = (long)n_items * n_items;
    const int grid = cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192;
    hipLaunchKernelGGL(dist_jaccard_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, item_cats, n_items, dist_out);
    CIRS_CHECK_LAUNCH("dist_jaccard_kernel");
    return CIRS_OK;
}
