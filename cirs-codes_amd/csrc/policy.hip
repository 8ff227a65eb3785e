// policy.hip -- shared trunk (20->64->64), critic head and the full-catalogue actor head with a fused sampler.
//
// Rollout-side policy forward for gfx950 (reference: tianshou/utils/net/{common,discrete}.py, core/policy/ppo.py:111-163).
//
//   trunk_kernel       one wavefront per env row: h1 = relu(W1 s + b1), h2 = relu(W2 h1 + b2), value = wc.h2 + bc.
//                      Sequential-k fmaf chains (the order the oracle restates), lane o owns output feature o.
//   actor_head_kernel  logits^T tile = Wa[32 items x 64] * H2^T[64 x 32 envs] on the fp32 matrix cores
//                      (v_mfma_f32_32x32x2_f32: exact f32 fma chain, D = fma(a_k1,b_k1, fma(a_k0,b_k0,C))), accumulator
//                      initialised with the bias.  Operand k-relabelling: MFMA step kk pairs k0 = kk with
//                      k1 = 32+kk so every lane streams 32 CONTIGUOUS floats (one 128-B line) of its row of Wa --
//                      no LDS staging, no strided reads.  The epilogue never writes logits: per (env,item) it adds
//                      Gumbel noise (Philox, or harness-supplied) and keeps a per-lane scalar running arg-max and
//                      online log-sum-exp (a lane owns one env row).  Per-chunk partials {score, idx, m, s} go to
//                      the caller's workspace.
//   actor_merge_kernel merges partials across chunks: action id (ties -> lowest id), logp with Categorical's clamp.
//
// Roofline: 2*64*I flop per env row (1.37 MFLOP at I = 10728) against 4*64*I bytes of Wa re-read per 32-env tile
// from L2/MALL (2.7 MB, cache resident) -> MFMA-bound at fp32 (157 TF peak), HBM traffic ~ Wa once per launch.
#include "policy_kernels.h"
#include "internal.h"

namespace cirs {

// merge partials across chunks: action id (ties -> lowest id), logp with Categorical's clamp (actor_merge_wave)
__global__ __launch_bounds__(256) void actor_merge_kernel(int n, int n_pad, int n_chunks, ActorPartialView pv,
                                                          const float* __restrict__ wa, const float* __restrict__ ba,
                                                          const float* __restrict__ h2,
                                                          const uint8_t* __restrict__ skip,
                                                          int64_t* __restrict__ act_out, float* __restrict__ logp_out) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    if (skip && skip[j]) {
        if (lane == 0) {
            act_out[j] = -1;
            if (logp_out) logp_out[j] = 0.f;
        }
        return;
    }
    actor_merge_wave(j, lane, n_pad, n_chunks, pv, wa, ba, h2, act_out, logp_out);
}

static int validate_policy(const cirs_policy_cfg* cfg, const cirs_policy_weights* w) {
    CIRS_REQUIRE(cfg && w, "policy cfg/weights null");
    CIRS_REQUIRE(cfg->n_items > 0, "n_items must be positive");
    if (cfg->hidden != kH) return fail(CIRS_E_UNSUPPORTED, "this build supports hidden == 64 only");
    CIRS_REQUIRE(cfg->dim_state > 0 && cfg->dim_state <= 64, "dim_state must be in 1..64");
    CIRS_REQUIRE(w->w1 && w->b1 && w->w2 && w->b2 && w->wa && w->ba && w->wc && w->bc, "policy weight pointer null");
    return CIRS_OK;
}

}  // namespace cirs

extern "C" int64_t cirs_policy_workspace_bytes(const cirs_policy_cfg* cfg, int32_t n) {
    using namespace cirs;
    if (!cfg || n <= 0) return 0;
    return (int64_t)(ws_h2_floats(n) + 4 * ws_partial_elems(n, cfg->n_items)) * 4;
}

extern "C" int cirs_actor_sample(const cirs_policy_cfg* cfg, const cirs_policy_weights* w, const float* state,
                                 int64_t state_stride, int32_t n, const float* gumbel, uint64_t seed,
                                 uint32_t rng_step, const int32_t* env_ids, const uint32_t* visited,
                                 const uint8_t* skip, int64_t* act_out, float* logp_out, float* value_out,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace cirs;
    if (int rc = validate_policy(cfg, w)) return rc;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(state && act_out && workspace, "null state/act/workspace");
    CIRS_REQUIRE(state_stride >= cfg->dim_state, "state_stride < dim_state");
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(workspace_bytes >= cirs_policy_workspace_bytes(cfg, n), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* h2 = (float*)workspace;
    const int n_pad = n_pad_of(n);
    const HeadGrid hg = sampler_grid(cfg->n_items, n_pad);
    const int n_chunks = hg.n_chunks;
    ActorPartialView pv = partial_view(workspace, n, cfg->n_items);
    hipLaunchKernelGGL(trunk_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, *cfg, *w, state, (long)state_stride, n, skip, h2,
                       value_out, nullptr);
    CIRS_CHECK_LAUNCH("trunk_kernel");
    const dim3 grid(hg.grid_x, hg.n_row_blocks);
    hipLaunchKernelGGL(actor_head_kernel, grid, dim3(256), 0, s, *cfg, w->wa, w->ba, h2, n, gumbel, seed,
                       rng_step, env_ids, visited, skip, pv, n_pad, hg.tiles_per_chunk);
    CIRS_CHECK_LAUNCH("actor_head_kernel");
    hipLaunchKernelGGL(actor_merge_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, n, n_pad, n_chunks, pv, w->wa, w->ba,
                       h2, skip, act_out, logp_out);
    CIRS_CHECK_LAUNCH("actor_merge_kernel");
    return CIRS_OK;
}
