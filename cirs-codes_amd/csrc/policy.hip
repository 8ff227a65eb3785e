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
#include "common.h"
#include "rng.h"

namespace cirs {

constexpr int kH = 64;            // hidden width (checked at the ABI)
constexpr int kTileM = 32;        // env rows per MFMA tile
constexpr int kTileN = 32;        // items per MFMA tile
constexpr int kTilesPerChunk = 4; // item tiles handled by one wave before the cross-lane reduction
constexpr int kChunkItems = kTileN * kTilesPerChunk;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ActorPartialView {  // SoA in the workspace, each array [n_chunks][n_pad]
    float* score;
    int32_t* idx;
    float* m;
    float* s;
};

__host__ __device__ inline int n_chunks_of(int n_items) { return (n_items + kChunkItems - 1) / kChunkItems; }
__host__ __device__ inline int n_pad_of(int n) { return ((n + kTileM - 1) / kTileM) * kTileM; }

__host__ inline size_t ws_h2_floats(int n) { return (size_t)n_pad_of(n) * kH; }
__host__ inline size_t ws_partial_elems(int n, int n_items) { return (size_t)n_chunks_of(n_items) * n_pad_of(n); }

__host__ __device__ inline ActorPartialView partial_view(void* ws, int n, int n_items) {
    float* base = (float*)ws + (size_t)n_pad_of(n) * kH;
    const size_t e = (size_t)n_chunks_of(n_items) * n_pad_of(n);
    ActorPartialView v;
    v.score = base;
    v.idx = (int32_t*)(base + e);
    v.m = base + 2 * e;
    v.s = base + 3 * e;
    return v;
}

// ---- trunk: one wave per row ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trunk_kernel(cirs_policy_cfg cfg, cirs_policy_weights w,
                                                    const float* __restrict__ state, long state_stride, int n,
                                                    const uint8_t* __restrict__ skip, float* __restrict__ h2_out,
                                                    float* __restrict__ value_out) {
    __shared__ float lds[4][2][kH];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wv;
    if (j >= n) return;
    float* xs = lds[wv][0];
    float* hs = lds[wv][1];
    const int S = cfg.dim_state;
    if (skip && skip[j]) {
        h2_out[(size_t)j * kH + lane] = 0.f;
        if (lane == 0 && value_out) value_out[j] = 0.f;
        return;
    }
    if (lane < S) xs[lane] = state[(size_t)j * state_stride + lane];
    __builtin_amdgcn_wave_barrier();
    // layer 1: lane o, chain over k = 0..S-1 starting from the bias
    float acc = w.b1[lane];
    const float* w1r = w.w1 + (size_t)lane * S;
    for (int k = 0; k < S; ++k) acc = __builtin_fmaf(w1r[k], xs[k], acc);
    hs[lane] = fmaxf(acc, 0.f);
    __builtin_amdgcn_wave_barrier();
    // layer 2
    acc = w.b2[lane];
    const float4* w2r = reinterpret_cast<const float4*>(w.w2 + (size_t)lane * kH);
#pragma unroll
    for (int k4 = 0; k4 < kH / 4; ++k4) {
        const float4 wv4 = w2r[k4];
        acc = __builtin_fmaf(wv4.x, hs[4 * k4 + 0], acc);
        acc = __builtin_fmaf(wv4.y, hs[4 * k4 + 1], acc);
        acc = __builtin_fmaf(wv4.z, hs[4 * k4 + 2], acc);
        acc = __builtin_fmaf(wv4.w, hs[4 * k4 + 3], acc);
    }
    const float h2 = fmaxf(acc, 0.f);
    h2_out[(size_t)j * kH + lane] = h2;
    __builtin_amdgcn_wave_barrier();
    xs[lane] = h2;  // S <= 64
    __builtin_amdgcn_wave_barrier();
    if (lane == 0 && value_out) {  // critic: sequential chain (bit-reproducible), 64 fma
        float v = w.bc[0];
        for (int k = 0; k < kH; ++k) v = __builtin_fmaf(w.wc[k], xs[k], v);
        value_out[j] = v;
    }
}

// ---- actor head ------------------------------------------------------------------------------------------------
// Transposed tile: ZT[32 items x 32 rows] = Wa_tile[32 x 64] * H2_tile^T[64 x 32].  MFMA A operand = this lane's ITEM
// row of Wa, B operand = this lane's ENV row of H2 (both 32 contiguous floats, k = hi*32 + kk).  In the C/D layout a
// lane then owns ONE env row (col = lane & 31) and 16 items (item(s) = (s&3) + 8*(s>>2) + 4*hi), so the running
// arg-max and the online log-sum-exp are per-lane SCALARS; the only cross-lane step is one exchange between the
// two half-waves at the end of the chunk.  Per accumulator group (s>>2) the 4 items are consecutive -> one Philox
// block and one float4 bias load serve them.
// grid = (n_chunks, ceil(n_pad/32/4)); block = 4 waves = 4 env tiles walking the same item chunk (shared Wa lines).
// kSample: true  -> Gumbel-max sampling + LSE (rollout);  false -> LSE (+ sum exp(z-m) z for the entropy) only.
template <bool kSample>
__global__ __launch_bounds__(256, 2) void actor_head_kernel(cirs_policy_cfg cfg, const float* __restrict__ wa,
                                                            const float* __restrict__ ba,
                                                            const float* __restrict__ h2, int n,
                                                            const float* __restrict__ gumbel, uint64_t seed,
                                                            uint32_t rng_step, const int32_t* __restrict__ env_ids,
                                                            const uint32_t* __restrict__ visited,
                                                            const uint8_t* __restrict__ skip, ActorPartialView pv,
                                                            int n_pad) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int row0 = (blockIdx.y * 4 + wv) * kTileM;
    if (row0 >= n_pad) return;
    const int I = cfg.n_items;
    const int chunk = blockIdx.x;
    const int vis_words = (I + 31) / 32;
    const int jr = row0 + lo;  // this lane's env row
    const bool active = jr < n && !(skip && skip[jr]);
    const size_t po = (size_t)chunk * n_pad + jr;
    if (__ballot(active) == 0ull) {  // every row of this env tile is finished: publish neutral partials
        if (hi == 0) {
            pv.score[po] = -INFINITY; pv.idx[po] = 0x7FFFFFFF; pv.m[po] = -INFINITY; pv.s[po] = 0.f;
            if (!kSample) pv.score[po] = 0.f;
        }
        return;
    }
    const int e = active ? (env_ids ? env_ids[jr] : jr) : 0;

    // B operand: this lane's env row of H2, k = hi*32 + kk
    float hrow[32];
    if (jr < n) {
        const float4* src = reinterpret_cast<const float4*>(h2 + (size_t)jr * kH + hi * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = src[q];
            hrow[4 * q + 0] = v.x; hrow[4 * q + 1] = v.y; hrow[4 * q + 2] = v.z; hrow[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 32; ++q) hrow[q] = 0.f;
    }
    float best_score = -INFINITY, run_m = -INFINITY, run_s = 0.f, run_t = 0.f;
    int best_idx = 0x7FFFFFFF;

    for (int it = 0; it < kTilesPerChunk; ++it) {
        const int tile0 = chunk * kChunkItems + it * kTileN;  // multiple of 32
        if (tile0 >= I) break;
        const int item_a = tile0 + lo;  // A-operand item of this lane
        float wrow[32];
        if (item_a < I) {
            const float4* src = reinterpret_cast<const float4*>(wa + (size_t)item_a * kH + hi * 32);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = src[q];
                wrow[4 * q + 0] = v.x; wrow[4 * q + 1] = v.y; wrow[4 * q + 2] = v.z; wrow[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 32; ++q) wrow[q] = 0.f;
        }
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // bias of the 4 consecutive items of accumulator group g
            const int i0 = tile0 + 8 * g + 4 * hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[4 * g + q] = (i0 + q) < I ? ba[i0 + q] : 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[kk], hrow[kk], acc, 0, 0, 0);

        if (!active) continue;
        const uint32_t vis = (kSample && visited) ? visited[(size_t)e * vis_words + (tile0 >> 5)] : 0u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i0 = tile0 + 8 * g + 4 * hi;
            float g4[4];
            if (kSample && !gumbel) {
                const u32x4 rr = philox4x32_10((uint32_t)i0 >> 2, (uint32_t)e, rng_step, CIRS_RNG_STREAM_ACTOR,
                                               (uint32_t)seed, (uint32_t)(seed >> 32));
                g4[0] = gumbel_from_bits(rr.x); g4[1] = gumbel_from_bits(rr.y);
                g4[2] = gumbel_from_bits(rr.z); g4[3] = gumbel_from_bits(rr.w);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int item = i0 + q;
                if (item >= I) continue;
                if (kSample && ((vis >> (item & 31)) & 1u)) continue;
                const float z = acc[4 * g + q];
                if (kSample) {
                    const float gn = gumbel ? gumbel[(size_t)jr * I + item] : g4[q];
                    const float sc = z + gn;
                    if (sc > best_score) {  // items ascend within a lane: strict > keeps the lowest id on ties
                        best_score = sc; best_idx = item;
                    }
                }
                // online log-sum-exp with ONE exp per element: ex = exp(-|z - m|)
                const float dlt = z - run_m;
                const float ex = __expf(-fabsf(dlt));
                if (dlt > 0.f) {
                    run_s = __builtin_fmaf(run_s, ex, 1.0f);
                    if (!kSample) run_t = __builtin_fmaf(run_t, ex, z);
                    run_m = z;
                } else {
                    run_s += ex;
                    if (!kSample) run_t = __builtin_fmaf(ex, z, run_t);
                }
            }
        }
    }
    // combine the two half-waves (same env row, disjoint items)
    {
        const float os = __shfl_xor(best_score, 32, CIRS_WAVE);
        const int oi = __shfl_xor(best_idx, 32, CIRS_WAVE);
        if (os > best_score || (os == best_score && oi < best_idx)) { best_score = os; best_idx = oi; }
        const float om = __shfl_xor(run_m, 32, CIRS_WAVE), osum = __shfl_xor(run_s, 32, CIRS_WAVE);
        const float ot = __shfl_xor(run_t, 32, CIRS_WAVE);
        const float mn = fmaxf(run_m, om);
        if (mn > -INFINITY) {
            const float fa = __expf(run_m - mn), fb = __expf(om - mn);
            run_s = run_s * fa + osum * fb;
            run_t = run_t * fa + ot * fb;
            run_m = mn;
        }
    }
    if (hi == 0) {
        if (kSample) { pv.score[po] = best_score; pv.idx[po] = best_idx; }
        else { pv.score[po] = run_t; }
        pv.m[po] = run_m; pv.s[po] = run_s;
    }
}

// merge partials across chunks (one wavefront per env row, lanes stride over chunks); recompute the chosen item's
// logit with the SAME k-order as the MFMA chain (bias, then for kk: k = kk, k = 32+kk) so logp is consistent with
// the sampled distribution.
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
    float bs = -INFINITY, m = -INFINITY, s = 0.f;
    int bi = 0x7FFFFFFF;
    for (int c = lane; c < n_chunks; c += CIRS_WAVE) {  // within a lane chunks ascend: strict > keeps the lowest id
        const size_t o = (size_t)c * n_pad + j;
        const float os = pv.score[o];
        const int oi = pv.idx[o];
        if (os > bs) { bs = os; bi = oi; }
        const float om = pv.m[o], osum = pv.s[o];
        const float mn = fmaxf(m, om);
        if (mn > -INFINITY) {
            s = s * __expf(m - mn) + osum * __expf(om - mn);
            m = mn;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float os = __shfl_xor(bs, off, CIRS_WAVE);
        const int oi = __shfl_xor(bi, off, CIRS_WAVE);
        if (os > bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
        const float om = __shfl_xor(m, off, CIRS_WAVE), osum = __shfl_xor(s, off, CIRS_WAVE);
        const float mn = fmaxf(m, om);
        if (mn > -INFINITY) {
            s = s * __expf(m - mn) + osum * __expf(om - mn);
            m = mn;
        }
    }
    if (lane != 0) return;
    act_out[j] = bi == 0x7FFFFFFF ? -1 : (int64_t)bi;
    if (logp_out) {
        float lp = 0.f;
        if (bi != 0x7FFFFFFF) {
            const float* wr = wa + (size_t)bi * kH;
            const float* hr = h2 + (size_t)j * kH;
            float z = ba[bi];
            for (int kk = 0; kk < 32; ++kk) {
                z = __builtin_fmaf(hr[kk], wr[kk], z);
                z = __builtin_fmaf(hr[32 + kk], wr[32 + kk], z);
            }
            const float lse = m + __logf(s);
            float p = __expf(z - lse);  // softmax prob of the chosen item (over unmasked items)
            const float eps = 1.1920928955078125e-7f;
            p = fminf(fmaxf(p, eps), 1.0f - eps);  // torch probs_to_logits clamp
            lp = __logf(p);
        }
        logp_out[j] = lp;
    }
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
    CIRS_REQUIRE(state && act_out && workspace, "null state/act/workspace");
    CIRS_REQUIRE(state_stride >= cfg->dim_state, "state_stride < dim_state");
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(workspace_bytes >= cirs_policy_workspace_bytes(cfg, n), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* h2 = (float*)workspace;
    const int n_pad = n_pad_of(n), n_chunks = n_chunks_of(cfg->n_items);
    ActorPartialView pv = partial_view(workspace, n, cfg->n_items);
    hipLaunchKernelGGL(trunk_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, *cfg, *w, state, (long)state_stride, n, skip, h2,
                       value_out);
    CIRS_CHECK_LAUNCH("trunk_kernel");
    const dim3 grid(n_chunks, cdiv(n_pad / kTileM, 4));
    hipLaunchKernelGGL(actor_head_kernel<true>, grid, dim3(256), 0, s, *cfg, w->wa, w->ba, h2, n, gumbel, seed,
                       rng_step, env_ids, visited, skip, pv, n_pad);
    CIRS_CHECK_LAUNCH("actor_head_kernel");
    hipLaunchKernelGGL(actor_merge_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, n, n_pad, n_chunks, pv, w->wa, w->ba,
                       h2, skip, act_out, logp_out);
    CIRS_CHECK_LAUNCH("actor_merge_kernel");
    return CIRS_OK;
}
