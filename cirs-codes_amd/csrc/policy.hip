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
    actor_merge_wave(j, lane, n_pad, n_chunks, pv, act_out, logp_out);
}

__global__ __launch_bounds__(256) void actor_shard_tuple_kernel(int n, int n_pad, int n_chunks, ActorPartialView pv,
                                                               const uint8_t* __restrict__ skip, float* __restrict__ out5) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    if (skip && skip[j]) {
        if (lane == 0) {
            out5[j] = -INFINITY; reinterpret_cast<int32_t*>(out5)[(size_t)n + j] = 0x7FFFFFFF;
            out5[(size_t)2 * n + j] = 0.f; out5[(size_t)3 * n + j] = -INFINITY; out5[(size_t)4 * n + j] = 0.f;
        }
        return;
    }
    actor_shard_tuple_wave(j, lane, n_pad, n_chunks, pv, n, out5);
}

// two-level sampler, stages 2 + 3 for n rows (one wavefront per row): final action / log-prob, or (tuple_out) the shard tuple
__global__ __launch_bounds__(256) void actor_pick_kernel(PickArgs a, int n, const int32_t* __restrict__ env_ids,
                                                         const uint8_t* __restrict__ skip, int64_t* __restrict__ act_out,
                                                         float* __restrict__ logp_out, float* __restrict__ tuple_out) {
    __shared__ __attribute__((aligned(16))) float stage[4][kPickStage];
    __shared__ float hs[4][kH];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wv;
    if (j >= n) return;
    Cand r{-INFINITY, 0.f, -INFINITY, 0.f, 0x7FFFFFFF};
    if (!(skip && skip[j])) r = actor_pick_wave(a, j, env_ids ? env_ids[j] : j, lane, hs[wv], stage[wv], nullptr);
    if (lane != 0) return;
    if (tuple_out) {
        tuple_out[j] = r.bs;
        reinterpret_cast<int32_t*>(tuple_out)[(size_t)n + j] = r.bi;
        tuple_out[(size_t)2 * n + j] = r.bi != 0x7FFFFFFF ? r.bz : 0.f;
        tuple_out[(size_t)3 * n + j] = r.m;
        tuple_out[(size_t)4 * n + j] = r.s;
        return;
    }
    act_out[j] = r.bi == 0x7FFFFFFF ? -1 : (int64_t)r.bi;
    if (logp_out) logp_out[j] = cand_logp(r);
}

// cross-rank merge of W shard tuples per env row, in RANK ORDER (fixed): candidate with the highest noisy score (ties -> lowest
// global id), running (max, sum-exp) folded rank by rank; logp of the winner with Categorical's clamp (as actor_merge_wave)
__global__ __launch_bounds__(256) void actor_merge_shards_kernel(const float* __restrict__ tuples, int n_shards, int n,
                                                                 const uint8_t* __restrict__ skip, int64_t* __restrict__ act_out,
                                                                 float* __restrict__ logp_out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (skip && skip[j]) {
        act_out[j] = -1;
        if (logp_out) logp_out[j] = 0.f;
        return;
    }
    float bs = -INFINITY, m = -INFINITY, s = 0.f, zb = 0.f;
    int bi = 0x7FFFFFFF;
    for (int r = 0; r < n_shards; ++r) {
        const float* t = tuples + (size_t)r * 5 * n;
        const float os = t[j];
        const int oi = reinterpret_cast<const int32_t*>(t)[(size_t)n + j];
        if (os > bs || (os == bs && oi < bi)) { bs = os; bi = oi; zb = t[(size_t)2 * n + j]; }
        const float om = t[(size_t)3 * n + j], osum = t[(size_t)4 * n + j];
        const float mn = fmaxf(m, om);
        if (mn > -INFINITY) {
            s = s * __expf(m - mn) + osum * __expf(om - mn);
            m = mn;
        }
    }
    act_out[j] = bi == 0x7FFFFFFF ? -1 : (int64_t)bi;
    if (logp_out) {
        float lp = 0.f;
        if (bi != 0x7FFFFFFF) {
            const float lse = m + __logf(s);
            float p = __expf(zb - lse);
            const float eps = 1.1920928955078125e-7f;
            p = fminf(fmaxf(p, eps), 1.0f - eps);
            lp = __logf(p);
        }
        logp_out[j] = lp;
    }
}

// rows of a row-major fp32 table: out[k, :] = table[idx[k], :]   (16 B per lane; row_floats % 4 == 0) -- the local side of a
// row-sharded embedding lookup
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int row_floats, const int64_t* __restrict__ idx,
                                                          long n, float* __restrict__ out) {
    const int q = row_floats >> 2;
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n * q) return;
    const long r = k / q;
    const int c = (int)(k - r * q);
    const long id = idx[r];     // id < 0: an empty message slot of the sharded lookup -> zeros
    reinterpret_cast<float4*>(out)[k] = id < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4*>(table + (size_t)id * row_floats)[c];
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
    // + slack: the fused rollout may carve one (256-byte aligned) workspace per env group out of this buffer
    // + the packed weight image of the fused rollout's step kernel (internal.h: TrkImg) at its end
    // + the logit store of the fused rollout's sampler while it is small (policy_kernels.h: ws_zstore_floats)
    // + the bf16 planes of the actor head for the chunk-mass kernels (policy_kernels.h: ws_rplanes_bytes), in front of the image
    return (int64_t)(ws_h2_floats(n) + 5 * ws_partial_elems(n, cfg->n_items) + ws_zstore_floats(n, cfg->n_items)) * 4 + 8192 + kTrkImgBytes +
           (int64_t)ws_rplanes_bytes(cfg->n_items) + 512;
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
    if (!gumbel) {   // counter-based sampler: two-level Gumbel-max (chunk masses, then chunk + item draws)
        const int nch = n_chunks_of(cfg->n_items);
        const int cpw = mass_chunks_per_wg(nch, hg.n_row_blocks);
        uint4* planes = ws_rplanes(workspace, workspace_bytes, cfg->n_items);
        if (int rc = build_rplanes(w->wa, cfg->n_items, planes, s)) return rc;
        hipLaunchKernelGGL(actor_mass_kernel, dim3(cdiv(nch, cpw), hg.n_row_blocks), dim3(kMassThreads), 0, s, *cfg, (const uint4*)planes, w->ba, (const float*)h2, n,
                           env_ids, visited, skip, pv.m, n_pad, cpw, 0, 0);
        CIRS_CHECK_LAUNCH("actor_mass_kernel");
        PickArgs pa{pv.m, n_pad, nch, w->wa, w->ba, h2, visited, cfg->n_items, 0, 0, seed, rng_step};
        hipLaunchKernelGGL(actor_pick_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, pa, n, env_ids, skip, act_out, logp_out, (float*)nullptr);
        CIRS_CHECK_LAUNCH("actor_pick_kernel");
        return CIRS_OK;
    }
    const dim3 grid(hg.grid_x, hg.n_row_blocks);
    hipLaunchKernelGGL(actor_head_kernel, grid, dim3(256), 0, s, *cfg, w->wa, w->ba, h2, n, gumbel, seed,
                       rng_step, env_ids, visited, skip, pv, n_pad, hg.tiles_per_chunk);
    CIRS_CHECK_LAUNCH("actor_head_kernel");
    hipLaunchKernelGGL(actor_merge_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, n, n_pad, n_chunks, pv, skip, act_out, logp_out);
    CIRS_CHECK_LAUNCH("actor_merge_kernel");
    return CIRS_OK;
}

extern "C" int cirs_critic_values(const cirs_policy_cfg* cfg, const cirs_policy_weights* w, const float* state, int64_t state_stride, int32_t n,
                                  float* value_out, void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace cirs;
    if (int rc = validate_policy(cfg, w)) return rc;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(state && value_out && workspace, "null state/value/workspace");
    CIRS_REQUIRE(state_stride >= cfg->dim_state, "state_stride < dim_state");
    CIRS_REQUIRE(workspace_bytes >= cirs_policy_workspace_bytes(cfg, n), "workspace too small");
    hipLaunchKernelGGL(trunk_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, *cfg, *w, state, (long)state_stride, n,
                       (const uint8_t*)nullptr, (float*)workspace, value_out, nullptr);
    CIRS_CHECK_LAUNCH("trunk_kernel (values)");
    return CIRS_OK;
}

extern "C" int cirs_actor_shard_partials(const cirs_policy_cfg* cfg_shard, const cirs_policy_weights* w_shard, const float* state,
                                         int64_t state_stride, int32_t n, uint64_t seed, uint32_t rng_step, const int32_t* env_ids,
                                         const uint32_t* visited, const uint8_t* skip, int32_t item_base, int32_t n_items_total,
                                         float* tuples_out, float* value_out, void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace cirs;
    if (int rc = validate_policy(cfg_shard, w_shard)) return rc;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(state && tuples_out && workspace, "null state/tuples/workspace");
    CIRS_REQUIRE(state_stride >= cfg_shard->dim_state, "state_stride < dim_state");
    CIRS_REQUIRE(item_base >= 0 && (item_base % CIRS_SAMPLER_CHUNK) == 0, "item_base must be a non-negative multiple of the sampler chunk (128 items)");
    CIRS_REQUIRE(n_items_total >= item_base + cfg_shard->n_items, "n_items_total < item_base + shard size");
    CIRS_REQUIRE(workspace_bytes >= cirs_policy_workspace_bytes(cfg_shard, n), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* h2 = (float*)workspace;
    const int n_pad = n_pad_of(n);
    const HeadGrid hg = sampler_grid(cfg_shard->n_items, n_pad);
    ActorPartialView pv = partial_view(workspace, n, cfg_shard->n_items);
    hipLaunchKernelGGL(trunk_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, *cfg_shard, *w_shard, state, (long)state_stride, n, skip, h2,
                       value_out, nullptr);
    CIRS_CHECK_LAUNCH("trunk_kernel");
    const int nch = n_chunks_of(cfg_shard->n_items);
    const int cpw = mass_chunks_per_wg(nch, hg.n_row_blocks);
    uint4* planes = ws_rplanes(workspace, workspace_bytes, cfg_shard->n_items);
    if (int rc = build_rplanes(w_shard->wa, cfg_shard->n_items, planes, s)) return rc;
    hipLaunchKernelGGL(actor_mass_kernel, dim3(cdiv(nch, cpw), hg.n_row_blocks), dim3(kMassThreads), 0, s, *cfg_shard, (const uint4*)planes, w_shard->ba,
                       (const float*)h2, n, env_ids, visited, skip, pv.m, n_pad, cpw, item_base, n_items_total);
    CIRS_CHECK_LAUNCH("actor_mass_kernel");
    PickArgs pa{pv.m, n_pad, nch, w_shard->wa, w_shard->ba, h2, visited, cfg_shard->n_items, item_base, n_items_total, seed, rng_step};
    hipLaunchKernelGGL(actor_pick_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, pa, n, env_ids, skip, (int64_t*)nullptr, (float*)nullptr, tuples_out);
    CIRS_CHECK_LAUNCH("actor_pick_kernel");
    return CIRS_OK;
}

extern "C" int cirs_actor_merge_shards(const float* tuples, int32_t n_shards, int32_t n, const uint8_t* skip, int64_t* act_out,
                                       float* logp_out, void* stream) {
    using namespace cirs;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(tuples && act_out && n_shards > 0, "null tuples / act_out or n_shards <= 0");
    hipLaunchKernelGGL(actor_merge_shards_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, tuples, n_shards, n, skip,
                       act_out, logp_out);
    CIRS_CHECK_LAUNCH("actor_merge_shards_kernel");
    return CIRS_OK;
}

extern "C" int cirs_gather_rows(const float* table, int32_t row_floats, const int64_t* idx, int64_t n, float* out, void* stream) {
    using namespace cirs;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(table && idx && out, "null argument");
    CIRS_REQUIRE(row_floats > 0 && (row_floats & 3) == 0, "row_floats must be a positive multiple of 4");
    const long total = n * (row_floats >> 2);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, table, row_floats, idx, (long)n, out);
    CIRS_CHECK_LAUNCH("gather_rows_kernel");
    return CIRS_OK;
}
