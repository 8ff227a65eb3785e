// policy_kernels.h -- device code shared by policy.hip (rollout) and ppo.hip (learner): trunk MLP and the MFMA
// actor head.  See policy.hip for the design notes.
#pragma once
#include "bf16x6.h"
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
    float* z;   // sampler only: the candidate's own logit (the MFMA accumulator value), so the merge needs no second look at Wa
};

__host__ __device__ inline int n_chunks_of(int n_items) { return (n_items + kChunkItems - 1) / kChunkItems; }
__host__ __device__ inline int n_pad_of(int n) { return ((n + kTileM - 1) / kTileM) * kTileM; }

__host__ inline size_t ws_h2_floats(int n) { return (size_t)n_pad_of(n) * kH; }
__host__ inline size_t ws_partial_elems(int n, int n_items) { return (size_t)n_chunks_of(n_items) * n_pad_of(n); }
// Logit store of the two-level sampler for SMALL env counts (round 5): actor_mass_kernel keeps the 128 logits of every (chunk, env row) it has just
// computed -- [n_chunks][n_pad][128] floats -- and the step kernel's pick reads the two logits a lane needs of the drawn chunk instead of re-reading
// the chunk's 32 KB of head rows and redoing 128 dot products at the head of its dependent chain (~8 k of the step kernel's 58 k cycles).  The
// values are the ones the recompute reproduces bit for bit (the MFMA accumulators).  Only while the store stays small (L2 resident): at C3 it would be
// 44 MB per vector step.
constexpr size_t kZStoreMaxFloats = 1u << 20;     // 4 MB
__host__ inline size_t ws_zstore_floats(int n, int n_items) {
    const size_t f = ws_partial_elems(n, n_items) * kChunkItems;
    return f <= kZStoreMaxFloats ? f : 0;
}

__host__ __device__ inline ActorPartialView partial_view(void* ws, int n, int n_items) {
    float* base = (float*)ws + (size_t)n_pad_of(n) * kH;
    const size_t e = (size_t)n_chunks_of(n_items) * n_pad_of(n);
    ActorPartialView v;
    v.score = base;
    v.idx = (int32_t*)(base + e);
    v.m = base + 2 * e;
    v.s = base + 3 * e;
    v.z = base + 4 * e;   // (the learner's statistics pass sizes its buffer for four arrays and never touches z)
    return v;
}

// ---- trunk: one wave per row ---------------------------------------------------------------------------------
// trunk on one row whose input already sits in LDS (xs[0..S)): h1 = relu(W1 x + b1), h2 = relu(W2 h1 + b2),
// value = wc . h2 + bc.  Sequential-k fmaf chains (the order the oracle restates); lane o owns output feature o.
// LDS copies of the weights (optional; a workgroup that runs several rows stages them once with coalesced loads -- a lane reading
// its own 256-byte row of W2 from memory costs 64 cache lines per load instruction): lw1 [64][ld1] (ld1 odd), lw2 [64][kLdsRow2],
// lwc [64].  Same fma chains either way.
constexpr int kLdsRow2 = kH + 4;
__device__ __forceinline__ void trunk_compute(const cirs_policy_cfg& cfg, const cirs_policy_weights& w, float* xs, float* hs, int lane,
                                              int j, float* __restrict__ h2_out, float* __restrict__ value_out,
                                              float* __restrict__ h1_out, const float* lw1 = nullptr, int ld1 = 0,
                                              const float* lw2 = nullptr, const float* lwc = nullptr) {
    const int S = cfg.dim_state;
    __builtin_amdgcn_wave_barrier();
    // layer 1: lane o, chain over k = 0..S-1 starting from the bias
    float acc = w.b1[lane];
    const float* w1r = lw1 ? lw1 + lane * ld1 : w.w1 + (size_t)lane * S;
    for (int k = 0; k < S; ++k) acc = __builtin_fmaf(w1r[k], xs[k], acc);
    hs[lane] = fmaxf(acc, 0.f);
    if (h1_out) h1_out[(size_t)j * kH + lane] = hs[lane];
    __builtin_amdgcn_wave_barrier();
    // layer 2
    acc = w.b2[lane];
    const float4* w2r = lw2 ? reinterpret_cast<const float4*>(lw2 + lane * kLdsRow2) : reinterpret_cast<const float4*>(w.w2 + (size_t)lane * kH);
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
        const float* wcr = lwc ? lwc : w.wc;
        for (int k = 0; k < kH; ++k) v = __builtin_fmaf(wcr[k], xs[k], v);
        value_out[j] = v;
    }
}

// row_index (nullable): row j reads state row row_index[j] (rows j >= n_valid read nothing and produce zeros);
// obs_copy (nullable, [n, S]): the gathered input rows are kept for the weight-gradient GEMM of the learner.
__device__ __forceinline__ void trunk_rows(const cirs_policy_cfg& cfg, const cirs_policy_weights& w,
                                           const float* __restrict__ state, long state_stride, int n,
                                           const uint8_t* __restrict__ skip, float* __restrict__ h2_out,
                                           float* __restrict__ value_out, float* __restrict__ h1_out,
                                           const int32_t* __restrict__ row_index, int n_valid, float* __restrict__ obs_copy,
                                           float (*lds)[2][kH], const float* lw1 = nullptr, int ld1 = 0, const float* lw2 = nullptr,
                                           const float* lwc = nullptr) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wv;
    if (j >= n) return;
    float* xs = lds[wv][0];
    float* hs = lds[wv][1];
    const int S = cfg.dim_state;
    if (skip && skip[j]) {
        h2_out[(size_t)j * kH + lane] = 0.f;
        if (h1_out) h1_out[(size_t)j * kH + lane] = 0.f;
        if (lane == 0 && value_out) value_out[j] = 0.f;
        return;
    }
    if (row_index) {
        const bool ok = j < n_valid;
        const float x = (ok && lane < S) ? state[(size_t)row_index[j] * state_stride + lane] : 0.f;
        if (lane < S) {
            xs[lane] = x;
            if (obs_copy) obs_copy[(size_t)j * S + lane] = x;
        }
    } else if (lane < S) {
        xs[lane] = state[(size_t)j * state_stride + lane];
    }
    trunk_compute(cfg, w, xs, hs, lane, j, h2_out, value_out, h1_out, lw1, ld1, lw2, lwc);
}

static __global__ __launch_bounds__(256) void trunk_kernel(cirs_policy_cfg cfg, cirs_policy_weights w,
                                                           const float* __restrict__ state, long state_stride, int n,
                                                           const uint8_t* __restrict__ skip, float* __restrict__ h2_out,
                                                           float* __restrict__ value_out,
                                                           float* __restrict__ h1_out,
                                                           const int32_t* __restrict__ row_index = nullptr, int n_valid = 0,
                                                           float* __restrict__ obs_copy = nullptr) {
    __shared__ float lds[4][2][kH];
    trunk_rows(cfg, w, state, state_stride, n, skip, h2_out, value_out, h1_out, row_index, n_valid, obs_copy, lds);
}

// ---- actor head ------------------------------------------------------------------------------------------------
// Transposed tile: ZT[32 items x 32 rows] = Wa_tile[32 x 64] * H2_tile^T[64 x 32].  MFMA A operand = this lane's ITEM
// row of Wa, B operand = this lane's ENV row of H2 (both 32 contiguous floats, k = hi*32 + kk).  In the C/D layout a
// lane then owns ONE env row (col = lane & 31) and 16 items (item(s) = (s&3) + 8*(s>>2) + 4*hi), so the running
// arg-max and the online log-sum-exp are per-lane SCALARS; the only cross-lane step is one exchange between the
// two half-waves at the end of the chunk.  Per accumulator group (s>>2) the 4 items are consecutive -> one Philox
// block and one float4 bias load serve them.
// grid = (n_chunks, ceil(n_pad/32/4)); block = 4 waves = 4 env tiles walking the same item chunk (shared Wa lines).
// (The PPO update's statistics-only variant of this kernel moved to ppo.hip: head_stats_kernel, on the bf16 matrix pipe.)
// Launch geometry of the sampler.  Unlike the MFMA-heavy PPO head kernels this one is VALU-bound (Philox: 20 quarter-rate
// v_mad_u64_u32 per 4 items, two fixed-order logs per item) and wants MANY small co-resident workgroups (60 VGPRs, 17 KB LDS:
// up to 3 per CU here): measured at C3 per launch, 4 tiles per chunk (672 workgroups) 42.9 us, 6 tiles (448) 46.4 us,
// 11 tiles (248, one per CU) 54.7 us.  kSamplerWgsPerCu = 3 resolves to the 4-tile floor at this size.
constexpr int kSamplerWgsPerCu = 3;
struct HeadGrid { int tiles_per_chunk, n_chunks, grid_x, n_row_blocks; };
constexpr int kLdsStride = 68;  // row stride (floats) of a staged 32 x 64 tile: ds_read_b128 of 16 lanes x 16 rows -> 64 banks

static __global__ __launch_bounds__(256, 2) void actor_head_kernel(cirs_policy_cfg cfg, const float* __restrict__ wa,
                                                            const float* __restrict__ ba,
                                                            const float* __restrict__ h2, int n,
                                                            const float* __restrict__ gumbel, uint64_t seed,
                                                            uint32_t rng_step, const int32_t* __restrict__ env_ids,
                                                            const uint32_t* __restrict__ visited,
                                                            const uint8_t* __restrict__ skip, ActorPartialView pv,
                                                            int n_pad, int tiles_per_chunk, int item_base = 0,
                                                            int n_items_total = 0) {
    // Column-sharded head (BASELINE configs[4]): wa / ba / cfg.n_items describe THIS rank's item shard, whose first item has the
    // global id item_base (a multiple of 32); noise counters, the visited bitmap, harness noise and the returned candidate ids use
    // GLOBAL item ids (n_items_total = size of the whole catalogue), so a shard computes exactly what the full kernel computes
    // for its items.  item_base = 0, n_items_total = 0: the whole catalogue on one device.
    // The Wa tile (32 items x 64) is staged in LDS once per workgroup and shared by its four env tiles; double
    // buffered: global loads of tile t+1 are issued before the MFMAs of tile t (one barrier per tile).
    __shared__ __attribute__((aligned(16))) float sW[2][kTileN * kLdsStride];
    __shared__ __attribute__((aligned(16))) float sB[2][kTileN];   // 16-byte aligned: the accumulator init reads 4 consecutive biases as one ds_read_b128
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int row0 = (blockIdx.y * 4 + wv) * kTileM;
    const int I = cfg.n_items;
    const int chunk = blockIdx.x;
    const int I_tot = n_items_total > 0 ? n_items_total : I;
    const int vis_words = (I_tot + 31) / 32;
    const int jr = row0 + lo;  // this lane's env row
    const bool active = row0 < n_pad && jr < n && !(skip && skip[jr]);
    const bool wave_live = __ballot(active) != 0ull;  // some row of this env tile still runs
    const size_t po = (size_t)chunk * n_pad + jr;
    const int e = active ? (env_ids ? env_ids[jr] : jr) : 0;

    // B operand: this lane's env row of H2, k = hi*32 + kk
    float hrow[32];
    if (wave_live) {
        // the wave's 32 x 64 tile of hidden rows is 8 KB of consecutive memory: read coalesced (8 x 1 KB), handed to the lanes through LDS
        __shared__ __attribute__((aligned(16))) float sH[4][kTileM * kLdsStride];
        typedef float mass_v4 __attribute__((ext_vector_type(4)));
        mass_v4* st4 = reinterpret_cast<mass_v4*>(sH[wv]);
        mass_v4 t8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = row0 + 4 * q + (lane >> 4);
            t8[q] = r < n ? *reinterpret_cast<const mass_v4*>(h2 + (size_t)r * kH + (lane & 15) * 4) : mass_v4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) st4[(4 * q + (lane >> 4)) * (kLdsStride / 4) + (lane & 15)] = t8[q];
        __builtin_amdgcn_wave_barrier();
        const mass_v4* src = reinterpret_cast<const mass_v4*>(&sH[wv][lo * kLdsStride + hi * 32]);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const mass_v4 v = src[q];
            hrow[4 * q + 0] = v.x; hrow[4 * q + 1] = v.y; hrow[4 * q + 2] = v.z; hrow[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 32; ++q) hrow[q] = 0.f;
    }
    float best_score = -INFINITY, best_z = 0.f, run_m = -INFINITY, run_s = 0.f;
    int best_idx = 0x7FFFFFFF;

    const int st_item = tid >> 3, st_col = (tid & 7) * 8;  // staging role: 2 float4 of the tile
    const int first_tile = chunk * tiles_per_chunk * kTileN;
    const int n_tiles = max(0, min(tiles_per_chunk, (I - first_tile + kTileN - 1) / kTileN));
    // a workgroup whose four env tiles are all finished only publishes neutral partials
    __shared__ int s_any;
    if (tid == 0) s_any = 0;
    __syncthreads();
    if (wave_live && lane == 0) s_any = 1;
    __syncthreads();
    const bool block_live = s_any != 0;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
    float gb = 0.f;
#define CIRS_ISSUE(TILE0)                                                                                  \
    do {                                                                                                   \
        const int item_ = (TILE0) + st_item;                                                               \
        if (item_ < I) {                                                                                   \
            const float4* src_ = reinterpret_cast<const float4*>(wa + (size_t)item_ * kH + st_col);        \
            g0 = src_[0]; g1 = src_[1];                                                                    \
        } else {                                                                                           \
            g0 = make_float4(0.f, 0.f, 0.f, 0.f); g1 = g0;                                                 \
        }                                                                                                  \
        if (tid < kTileN) gb = ((TILE0) + tid) < I ? ba[(TILE0) + tid] : 0.f;                              \
    } while (0)
#define CIRS_COMMIT(BUF)                                                                                   \
    do {                                                                                                   \
        float4* dst_ = reinterpret_cast<float4*>(&sW[BUF][st_item * kLdsStride + st_col]);                 \
        dst_[0] = g0; dst_[1] = g1;                                                                        \
        if (tid < kTileN) sB[BUF][tid] = gb;                                                               \
    } while (0)
    if (block_live) {
        if (n_tiles > 0) { CIRS_ISSUE(first_tile); CIRS_COMMIT(0); }
        __syncthreads();
        for (int it = 0; it < n_tiles; ++it) {
            const int buf = it & 1;
            const int tile0 = first_tile + it * kTileN;  // multiple of 32
            if (it + 1 < n_tiles) CIRS_ISSUE(tile0 + kTileN);
            if (wave_live) {
                const float* tw = sW[buf];
                float wrow[32];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 t4 = *reinterpret_cast<const float4*>(&tw[lo * kLdsStride + hi * 32 + 4 * q]);
                    wrow[4 * q] = t4.x; wrow[4 * q + 1] = t4.y; wrow[4 * q + 2] = t4.z; wrow[4 * q + 3] = t4.w;
                }
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = sB[buf][(r & 3) + 8 * (r >> 2) + 4 * hi];
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[kk], hrow[kk], acc, 0, 0, 0);

                if (active) {
                    const uint32_t vis = visited ? visited[(size_t)e * vis_words + ((item_base + tile0) >> 5)] : 0u;
                    // log-sum-exp per tile: the lane's maximum first, then ONE rescale of the running sums and one exp per
                    // element (masked / padded elements carry -inf and add exp(-inf) = 0)
                    float zt[16];
                    float tmax = -INFINITY;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int i0 = tile0 + 8 * g + 4 * hi;
                        float g4[4];
                        if (!gumbel) {
                            const u32x4 rr = philox4x32_10((uint32_t)(item_base + i0) >> 2, (uint32_t)e, rng_step, CIRS_RNG_STREAM_ACTOR,
                                                           (uint32_t)seed, (uint32_t)(seed >> 32));
                            g4[0] = gumbel_from_bits(rr.x); g4[1] = gumbel_from_bits(rr.y);
                            g4[2] = gumbel_from_bits(rr.z); g4[3] = gumbel_from_bits(rr.w);
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int item = i0 + q;
                            const bool valid = item < I && !((vis >> (item & 31)) & 1u);
                            const float z = acc[4 * g + q];
                            zt[4 * g + q] = valid ? z : -INFINITY;
                            tmax = fmaxf(tmax, zt[4 * g + q]);
                            if (valid) {
                                const float gn = gumbel ? gumbel[(size_t)jr * I_tot + item_base + item] : g4[q];
                                const float sc = z + gn;
                                if (sc > best_score) {  // items ascend within a lane: strict > keeps the lowest id on ties
                                    best_score = sc; best_idx = item_base + item; best_z = z;
                                }
                            }
                        }
                    }
                    if (tmax > -INFINITY) {
                        const float mn = fmaxf(run_m, tmax);
                        const float keep = __expf(run_m - mn);      // run_m = -inf on the first tile: keep = 0
                        float ss = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float ex = __expf(zt[r] - mn);
                            ss += ex;
                        }
                        run_s = __builtin_fmaf(run_s, keep, ss);
                        run_m = mn;
                    }
                }
            }
            if (it + 1 < n_tiles) CIRS_COMMIT(buf ^ 1);
            __syncthreads();
        }
    }
#undef CIRS_ISSUE
#undef CIRS_COMMIT
    if (row0 >= n_pad) return;
    // combine the two half-waves (same env row, disjoint items)
    {
        const float os = __shfl_xor(best_score, 32, CIRS_WAVE), oz = __shfl_xor(best_z, 32, CIRS_WAVE);
        const int oi = __shfl_xor(best_idx, 32, CIRS_WAVE);
        if (os > best_score || (os == best_score && oi < best_idx)) { best_score = os; best_idx = oi; best_z = oz; }
        const float om = __shfl_xor(run_m, 32, CIRS_WAVE), osum = __shfl_xor(run_s, 32, CIRS_WAVE);
        const float mn = fmaxf(run_m, om);
        if (mn > -INFINITY) {
            const float fa = __expf(run_m - mn), fb = __expf(om - mn);
            run_s = run_s * fa + osum * fb;
            run_m = mn;
        }
    }
    if (hi == 0) {
        pv.score[po] = best_score; pv.idx[po] = best_idx; pv.z[po] = best_z;
        pv.m[po] = run_m; pv.s[po] = run_s;
    }
}

// merge the per-chunk partials of env row j (one wavefront, lanes stride over chunks); recompute the chosen item's
// logit with the SAME k-order as the MFMA chain (bias, then for kk: k = kk, k = 32+kk) so logp is consistent with the
// sampled distribution.  Lane 0 writes act / logp; the action id is returned in every lane.
// ---- two-level sampler, stage 1: log-mass of every 128-item chunk ----------------------------------------------------------
// Same tiling as actor_head_kernel (ZT[32 items x 32 rows] per MFMA tile, a lane owns ONE env row and 16 items of a tile), no noise at all: the four
// tiles of a chunk stay in the accumulators, M_c = max of the lane pair's valid logits, S_c = sum of exp(z - M_c) in register order (tile, r) per
// half-wave, S_c = S_hi0 + S_hi1, L_c = M_c + log S_c.  One float per (chunk, env row) leaves the kernel: lmass[c][row] (-inf: no valid item).
//
// Round 6: the logits are fp32 products from two fp16 pieces per operand on the matrix pipe ("f16x3", bf16x6.h: three MFMAs per 16 k, operands pre-scaled by
// exact powers of two) -- the arithmetic of the learner's head_stats_kernel, accumulator split and all (bias + h*h terms | cross terms, met once), so the
// rollout's log-probabilities and the learner's come from the same numbers -- and the mass uses the hardware's exp2 / log2.  Rounds 2-5 kept this stage on the fp32 MFMA with fma-only exp / log so that the C oracle could restate L_c BIT
// FOR BIT (22.9 us per launch at C3, 39 % of the fp32 MFMA peak; no arithmetic model of v_mfma_f32_32x32x16_bf16 reproduces it exactly:
// tools/probes/mfma_model_probe.hip).  The masses now agree with the oracle's to ~1e-6 and SURVEY 8(c)'s own protocol applies to the draw: "action
// indices identical wherever the top-2 margin > 1e-6; report violations, expected 0" (tests/policycase.py: assert_draws_match; the oracle reports both
// margins of its two-level draw).  Still exact and order-fixed: a logit depends on its (item, env row) alone -- not on the tile, the workgroup, the launch
// geometry or the shard --, the sum keeps its order, and stage 3 (the item inside the drawn chunk) still runs scalar fp32 fma chains.  So every path that
// forms chunk masses (this kernel, the small-count kernel below, item shards) gives the same bits, and ranks / replays stay bit-identical.
//
// Wa arrives as fp16 planes (wa_rplanes_kernel, once per cirs_rollout_steps / cirs_actor_sample call -- the weights do not change inside a call):
// per item tile of 32, [2 planes h | l][32 items][64 cols] fp16 of kMassScWa * Wa = 512 uint4 (items beyond the catalogue: zero rows; the chunk count is
// rounded up to whole chunks).  A tile's 8 KB are staged in LDS once per workgroup (row stride 144 B: conflict-free 16-byte operand reads), double buffered.
constexpr int kMassRowB = 144;                      // LDS row stride (bytes) of one plane of a staged tile: 64 fp16 + 16
constexpr int kMassPlaneB = kTileN * kMassRowB;     // one plane of a tile in LDS
constexpr int kMassTileU4 = 512;                    // uint4 per item tile in global memory
constexpr int kMassThreads = 512;                    // workgroup of actor_mass_kernel: 4 row tiles x 2 tile halves
constexpr float kMassScWa = 256.f, kMassScH2 = 64.f, kMassScZ = kMassScWa * kMassScH2, kMassScZi = 1.0f / kMassScZ;   // exact power-of-two prescales (ppo.hip: kScWa / kScH2)
__host__ __device__ inline size_t ws_rplanes_bytes(int n_items) { return (size_t)n_chunks_of(n_items) * kTilesPerChunk * kMassTileU4 * 16; }

// grid = tiles of whole chunks (n_chunks * 4), 256 threads: thread -> (item = tid / 8, 8 consecutive columns)
static __global__ __launch_bounds__(256) void wa_rplanes_kernel(const float* __restrict__ wa, int I, uint4* __restrict__ planes) {
    const int tid = threadIdx.x, tile = blockIdx.x;
    const int item = tile * kTileN + (tid >> 3), col = 8 * (tid & 7);
    typedef float rp_v4 __attribute__((ext_vector_type(4)));
    rp_v4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (item < I) {
        const rp_v4* src = reinterpret_cast<const rp_v4*>(wa + (size_t)item * kH + col);
        a = src[0]; b = src[1];
    }
    a *= kMassScWa; b *= kMassScWa;
    const Planes2 pl = split8h(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
    uint4* out = planes + (size_t)tile * kMassTileU4;
    out[tid] = __builtin_bit_cast(uint4, pl.h); out[256 + tid] = __builtin_bit_cast(uint4, pl.l);
}

// B operand of the logits product: this lane's env row of H2 (col = lane & 31), element j of k-step s = column 16 s + 8 hi + j, split into fp16 pieces of
// kMassScH2 * H2.  tile: the wave's 32 x 64 fp32 tile in LDS (row stride kLdsStride floats).
__device__ __forceinline__ void mass_split_hidden(const float* tile, int lo, int hi, Planes2 (&hz)[4]) {
    typedef float ms_v4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const ms_v4* src = reinterpret_cast<const ms_v4*>(&tile[lo * kLdsStride + 16 * s4 + 8 * hi]);
        const ms_v4 a = src[0] * kMassScH2, b = src[1] * kMassScH2;
        hz[s4] = split8h(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
    }
}
// one 32 x 32 logits tile: A = the staged planes of the item tile (lane = item lo, 8 consecutive columns per k-step), B = hz, accumulator = bias + h*h terms |
// cross terms (two chains), met once -- head_stats_kernel's sequence (ppo.hip)
// (sb: the tile's biases PRE-SCALED by kMassScZ -- the accumulators hold kMassScZ z until the last line)
__device__ __forceinline__ f32x16 mass_logits_tile(const unsigned char* tw, const float* sb, const Planes2 (&hz)[4], int lo, int hi) {
    Planes2 za[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const unsigned char* ap = tw + lo * kMassRowB + (16 * s4 + 8 * hi) * 2;
        za[s4].h = *reinterpret_cast<const f16x8*>(ap);
        za[s4].l = *reinterpret_cast<const f16x8*>(ap + kMassPlaneB);
    }
    f32x16 acc, accs, acct;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = sb[acc_row(r, hi)]; accs[r] = 0.f; acct[r] = 0.f; }
    mfma_f16x3_split2(za[0], hz[0], za[1], hz[1], acc, accs, acct);
    mfma_f16x3_split2(za[2], hz[2], za[3], hz[3], acc, accs, acct);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = (acc[r] + (accs[r] + acct[r])) * kMassScZi;
    return acc;
}
// e^(z - M) as the hardware's exp2 of one fma (nml = -M log2 e); masked elements carry z = -inf -> 0.  ONE definition for every mass kernel: same bits.
__device__ __forceinline__ float mass_exp(float z, float nml) { return __builtin_amdgcn_exp2f(__builtin_fmaf(z, 1.4426950408889634f, nml)); }
__device__ __forceinline__ float mass_log(float s) { return __builtin_amdgcn_logf(s) * 0.6931471805599453f; }

#ifdef CIRS_MASS_PROF
// stage timestamps of workgroup (0, 0) / wave 0 (probe builds only: tools/probes/mass_prof.py)
static __device__ unsigned long long g_mass_prof[32];
#define CIRS_MSTAMP(K) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_mass_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
#define CIRS_MSTAMP_IF(COND, K) do { if ((COND) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_mass_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CIRS_MSTAMP(K) do { } while (0)
#define CIRS_MSTAMP_IF(COND, K) do { } while (0)
#endif
// grid = (ceil(n_chunks / chunks_per_wg), row blocks); 8 waves = 4 env row tiles x 2 halves of a chunk's four item tiles (wave = (row tile wv & 3, half wv >> 2):
// tiles 2 half, 2 half + 1).  Round 6, from stage stamps (tools/probes/mass_prof.py):
//  * staging by CHUNK: the four plane tiles of the NEXT chunk (32 KB) are requested at the top of a chunk and committed to the other LDS buffer at its end -- one
//    data barrier per chunk (tile-by-tile double buffering exposed a global-load latency per tile: a tile's MFMAs + operand reads are ~0.6 k cycles, a load under
//    load 1.5-2 k; the kernel ran at 2.4 k cycles per tile);
//  * two waves per SIMD: with one wave per SIMD the kernel was instruction-issue bound (~6 cycles per instruction: 3.1 k cycles for a chunk's 48 MFMAs + operand
//    reads, 2.1 k for its maximum + 64 exponentials).  The two waves of a row tile split the chunk's tiles; what has an order -- the half-wave sums
//    S_hi = ((0 + e(t0, r0)) + ... + e(t3, r15)) -- is handed from the first to the second through LDS, the maximum is order-free: the same bits as one wave
//    walking all four tiles (and as actor_mass_small_kernel below).
static __global__ __launch_bounds__(512, 1) void actor_mass_kernel(cirs_policy_cfg cfg, const uint4* __restrict__ planes,
                                                                   const float* __restrict__ ba, const float* __restrict__ h2, int n,
                                                                   const int32_t* __restrict__ env_ids,
                                                                   const uint32_t* __restrict__ visited,
                                                                   const uint8_t* __restrict__ skip, float* __restrict__ lmass,
                                                                   int n_pad, int chunks_per_wg, int item_base = 0,
                                                                   int n_items_total = 0, int env_base = 0, float* __restrict__ zstore = nullptr) {
    __shared__ __attribute__((aligned(16))) unsigned char sW[2][kTilesPerChunk][2 * kMassPlaneB];
    __shared__ __attribute__((aligned(16))) float sB[2][kChunkItems];   // 16-byte aligned: the accumulator init reads 4 consecutive biases as one ds_read_b128
    __shared__ float sM[4][2][kTileM];      // [row tile][tile half][row]: the half's maximum
    __shared__ float sS[4][2][kTileM];      // [row tile][lane half hi][row]: the first tile half's ordered half-wave sum
    const int tid = threadIdx.x;
    CIRS_MSTAMP(0);
    const int lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int rt = wv & 3, th = wv >> 2;
    const int row0 = (blockIdx.y * 4 + rt) * kTileM;
    const int I = cfg.n_items;
    const int I_tot = n_items_total > 0 ? n_items_total : I;
    const int vis_words = (I_tot + 31) / 32;
    const int n_chunks = n_chunks_of(I);
    const int c_begin = blockIdx.x * chunks_per_wg;
    const int c_end = min(n_chunks, c_begin + chunks_per_wg);
    // staging: thread tid owns unit (plane tid / 256, v = tid % 256) of each of the chunk's four tiles
    const int dst_r = (tid >> 8) * kMassPlaneB + ((tid & 255) >> 3) * kMassRowB + (tid & 7) * 16;
    // (four NAMED registers: hipcc promotes an indexed array of prefetch registers to LDS -- the loads were followed by ds_write_b128s and their waits)
    uint4 ga, gb4, gc, gd;
    float gb = 0.f;
    static_assert(kTilesPerChunk == 4, "the chunk staging names its four tiles");
#define CIRS_ISSUE(CHUNK)                                                                                  \
    do {                                                                                                   \
        const uint4* src_ = planes + (size_t)(CHUNK) * kTilesPerChunk * kMassTileU4 + tid;                 \
        ga = src_[0]; gb4 = src_[kMassTileU4]; gc = src_[2 * kMassTileU4]; gd = src_[3 * kMassTileU4];     \
        gb = ba[min((CHUNK) * kChunkItems + (tid & (kChunkItems - 1)), I - 1)];   /* unconditional, clamped: scaled / masked at the commit */ \
    } while (0)
#define CIRS_COMMIT(BUF)                                                                                   \
    do {                                                                                                   \
        *reinterpret_cast<uint4*>(sW[BUF][0] + dst_r) = ga; *reinterpret_cast<uint4*>(sW[BUF][1] + dst_r) = gb4; \
        *reinterpret_cast<uint4*>(sW[BUF][2] + dst_r) = gc; *reinterpret_cast<uint4*>(sW[BUF][3] + dst_r) = gd;  \
        if (tid < kChunkItems) sB[BUF][tid] = gb_chunk * kChunkItems + tid < I ? kMassScZ * gb : 0.f;       \
    } while (0)
    // (hipcc: a prefetch issued under a condition -- even a uniform one -- is followed by its own s_waitcnt vmcnt(0) where the paths merge; the loads below are
    //  therefore UNCONDITIONAL with clamped indices)
    int gb_chunk = min(c_begin, n_chunks - 1);
    CIRS_ISSUE(gb_chunk);
    const int jr = row0 + lo;
    const bool active = row0 < n_pad && jr < n && !(skip && skip[jr]);
    const bool wave_live = __ballot(active) != 0ull;
    const int e = active ? (env_ids ? env_ids[jr] : env_base + jr) : 0;
    Planes2 hz[4];
    {
        // a row tile's 32 x 64 hidden rows are 8 KB of consecutive memory: read coalesced by its first wave (8 x 1 KB), handed to the lanes of both its waves through LDS
        __shared__ __attribute__((aligned(16))) float sH[4][kTileM * kLdsStride];
        typedef float mass_v4 __attribute__((ext_vector_type(4)));
        if (th == 0 && wave_live) {
            mass_v4* st4 = reinterpret_cast<mass_v4*>(sH[rt]);
            mass_v4 t8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = row0 + 4 * q + (lane >> 4);
                t8[q] = r < n ? *reinterpret_cast<const mass_v4*>(h2 + (size_t)r * kH + (lane & 15) * 4) : mass_v4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) st4[(4 * q + (lane >> 4)) * (kLdsStride / 4) + (lane & 15)] = t8[q];
        }
        __shared__ int s_any;
        if (tid == 0) s_any = 0;
        __syncthreads();
        if (wave_live && lane == 0) s_any = 1;
        __syncthreads();      // (also: the hidden tiles are visible to the second wave of every row tile)
        CIRS_MSTAMP(1);
        if (s_any == 0) {   // every env of this workgroup has finished: neutral masses, no arithmetic
            if (th == 0 && row0 < n_pad && hi == 0)
                for (int c = c_begin; c < c_end; ++c) lmass[(size_t)c * n_pad + jr] = -INFINITY;
            return;
        }
        if (wave_live) {
            mass_split_hidden(sH[rt], lo, hi, hz);
        } else {
            const pk4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < 4; ++q) { hz[q].h = __builtin_bit_cast(f16x8, z4); hz[q].l = hz[q].h; }
        }
    }
    CIRS_COMMIT(0);
    __syncthreads();
    CIRS_MSTAMP(2);
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        const bool more = c + 1 < c_end;
        const bool stamp_it = c == c_begin + 1;     // (probe builds: the workgroup's second chunk)
        CIRS_MSTAMP_IF(stamp_it, 15);
        // the visited words of this wave's two tiles BEFORE the next chunk's planes: memory returns in order, so their wait does not cover the prefetch
        uint32_t vis[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tile0 = c * kChunkItems + (2 * th + u) * kTileN;
            vis[u] = (visited && active && tile0 < I) ? visited[(size_t)e * vis_words + ((item_base + tile0) >> 5)] : 0u;   // tiles beyond the catalogue: no word exists
        }
        gb_chunk = more ? c + 1 : c;       // (the last chunk re-requests itself: never committed)
        CIRS_ISSUE(gb_chunk);
        f32x16 acc[2];
        float mloc = -INFINITY;
        if (wave_live) {
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[u] = mass_logits_tile(sW[buf][2 * th + u], sB[buf] + (2 * th + u) * kTileN, hz, lo, hi);
            CIRS_MSTAMP_IF(stamp_it, 3);
            if (zstore && active) {      // the logits of this row as computed (before masking): item t * 32 + 8 q + 4 hi + (0..3) in one 16-byte store
                float* zr = zstore + ((size_t)c * n_pad + jr) * kChunkItems + 4 * hi;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(zr + (2 * th + u) * kTileN + 8 * q) = make_float4(acc[u][4 * q], acc[u][4 * q + 1], acc[u][4 * q + 2], acc[u][4 * q + 3]);
            }
            // mask, this half's maximum over the lane pair
            if (!visited && (c + 1) * kChunkItems <= I) {     // whole chunk inside the catalogue, nothing masked: no per-element tests
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, acc[u][r]);
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const int item = c * kChunkItems + (2 * th + u) * kTileN + il;
                        const bool valid = item < I && !((vis[u] >> (il & 31)) & 1u);
                        acc[u][r] = valid ? acc[u][r] : -INFINITY;
                        mloc = fmaxf(mloc, acc[u][r]);
                    }
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, CIRS_WAVE));
            if (hi == 0) sM[rt][th][lo] = mloc;
        }
        lds_barrier();      // the halves' maxima
        CIRS_MSTAMP_IF(stamp_it, 11);
        float M = -INFINITY, sl = 0.f;
        float ex[32];
        if (wave_live) {
            M = fmaxf(sM[rt][0][lo], sM[rt][1][lo]);
            if (M > -INFINITY) {        // uniform over the lane pair and the wave pair (M is shared)
                const float nml = -M * 1.4426950408889634f;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[16 * u + r] = mass_exp(acc[u][r], nml);
            } else {
#pragma unroll
                for (int q = 0; q < 32; ++q) ex[q] = 0.f;
            }
            if (th == 0) {              // the sum keeps its element order (tile, r): tiles 0, 1 here, the second wave continues with 2, 3
#pragma unroll
                for (int q = 0; q < 32; ++q) sl += ex[q];
                sS[rt][hi][lo] = sl;
            }
        }
        lds_barrier();      // the first half's ordered sum
        CIRS_MSTAMP_IF(stamp_it, 12);
        if (wave_live && th == 1) {
            sl = sS[rt][hi][lo];
#pragma unroll
            for (int q = 0; q < 32; ++q) sl += ex[q];
            float L = -INFINITY;
            const float so = __shfl_xor(sl, 32, CIRS_WAVE);
            if (M > -INFINITY) {
                const float S = hi == 0 ? sl + so : so + sl;                          // S_hi0 + S_hi1, in that order on both lanes
                L = M + mass_log(S);
            }
            if (row0 < n_pad && hi == 0) lmass[(size_t)c * n_pad + jr] = active ? L : -INFINITY;
        }
        CIRS_MSTAMP_IF(stamp_it, 13);
        // (hipcc hoists the commit's LDS stores to right behind the loads -- before this chunk's arithmetic -- unless something pins them here: the first
        //  build of this loop waited for all loads at the TOP of every chunk, 38 us per launch)
        asm volatile("" ::: "memory");
        if (more) {
            CIRS_COMMIT(buf ^ 1);      // (last read one chunk ago: every wave has passed a barrier since)
            __syncthreads();
        }
        CIRS_MSTAMP_IF(stamp_it, 14);
    }
    CIRS_MSTAMP(16);
#undef CIRS_ISSUE
#undef CIRS_COMMIT
}

// ---- the same chunk masses for FEW env rows (n_pad <= 128), one workgroup per chunk, one WAVE per (row tile, item tile) -----------------------
// actor_mass_kernel walks the four item tiles of a chunk one after the other in each wave: at 64 envs (BASELINE configs[1]) that is 26 workgroups of two live
// waves.  Here the chunk's four tiles run side by side in four waves per row tile (2 row tiles x 4 = 8 waves at 64 envs): the chunk's four plane tiles (32 KB)
// are staged at once, every wave does ONE tile's 12 MFMAs and its 16 exponentials per lane, and only what has an order is serial -- the half-wave sums
// S_hi = (((0 + e(t0, r0)) + e(t0, r1)) + ... + e(t3, r15)) are handed from tile wave to tile wave through LDS (16 adds and one barrier per hand-off).
// Bits identical to actor_mass_kernel (same logits and exponentials; max is order-free; the sum keeps its order).  grid = n_chunks, block = (n_pad / 32) * 256.
static __global__ __launch_bounds__(1024) void actor_mass_small_kernel(cirs_policy_cfg cfg, const uint4* __restrict__ planes, const float* __restrict__ ba,
                                                                       const float* __restrict__ h2, int n, const uint32_t* __restrict__ visited,
                                                                       const uint8_t* __restrict__ skip, float* __restrict__ lmass, int n_pad, int env_base,
                                                                       float* __restrict__ zstore) {
    __shared__ __attribute__((aligned(16))) unsigned char sW[4][2 * kMassPlaneB];     // the chunk's four item tiles (planes)
    __shared__ __attribute__((aligned(16))) float sB[4][kTileN];
    __shared__ __attribute__((aligned(16))) float sH[4][kTileM * kLdsStride];     // one hidden tile per row tile
    __shared__ float sM[4][4][kTileM];                                             // [row tile][item tile][row]: the tile's maximum
    __shared__ float sS[4][2][kTileM];                                             // [row tile][half][row]: the running half-wave sum
    // ~74 KB of static LDS: more than the 64 KB of gfx90a / gfx942 -- this kernel (like the whole library) is gfx950-only (160 KB per CU)
    static_assert(4 * 2 * kMassPlaneB + sizeof(float) * (4 * kTileN + 4 * kTileM * kLdsStride + 4 * 4 * kTileM + 4 * 2 * kTileM) <= 160 * 1024,
                  "actor_mass_small_kernel: LDS beyond gfx950's 160 KB");
    const int tid = threadIdx.x, n_thr = blockDim.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int rt = wv >> 2, tt = wv & 3;
    const int I = cfg.n_items, c = blockIdx.x;
    const int vis_words = (I + 31) / 32;
    typedef float mass_v4 __attribute__((ext_vector_type(4)));
    // the chunk's four plane tiles (32 KB of consecutive memory) and biases, the hidden tiles: everything requested before the first wait
    // (round 6, last hours: FIXED-count, clamped loads into registers, then the stores -- the loops over a run-time thread count were one memory round trip per
    //  iteration, four for the planes at 512 threads; n_thr = 256 .. 1024 -> 8 .. 2 of the 8 plane units and always 2 hidden units per thread are real)
    {
        const uint4* psrc = planes + (size_t)c * kTilesPerChunk * kMassTileU4;      // the chunk's four tiles are consecutive: unit f of 2048
        uint4 xr[8];
        mass_v4 hr[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) xr[q] = psrc[min(tid + q * n_thr, 4 * kMassTileU4 - 1)];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + q * n_thr, r = f >> 4, col4 = f & 15;
            hr[q] = (f < n_pad * (kH / 4) && r < n) ? *reinterpret_cast<const mass_v4*>(h2 + (size_t)r * kH + 4 * col4) : mass_v4{0.f, 0.f, 0.f, 0.f};
        }
        const float bq = tid < kChunkItems ? ba[min(c * kChunkItems + tid, I - 1)] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int f = tid + q * n_thr;
            if (f < 4 * kMassTileU4) {
                const int t = f / kMassTileU4, u = f - t * kMassTileU4, pl = u >> 8, v = u & 255;
                *reinterpret_cast<uint4*>(sW[t] + pl * kMassPlaneB + (v >> 3) * kMassRowB + (v & 7) * 16) = xr[q];
            }
        }
        if (tid < kChunkItems) sB[tid >> 5][tid & 31] = c * kChunkItems + tid < I ? kMassScZ * bq : 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + q * n_thr, r = f >> 4, col4 = f & 15;
            if (f < n_pad * (kH / 4)) *reinterpret_cast<mass_v4*>(&sH[r >> 5][(r & 31) * kLdsStride + 4 * col4]) = hr[q];
        }
    }
    const int jr = rt * kTileM + lo;
    const bool active = jr < n && !(skip && skip[jr]);
    const int e = env_base + jr;
    const int tile0 = c * kChunkItems + tt * kTileN;
    const uint32_t vis = (visited && active && tile0 < I) ? visited[(size_t)e * vis_words + (tile0 >> 5)] : 0u;
    __syncthreads();
    f32x16 acc;
    {
        Planes2 hz[4];
        mass_split_hidden(sH[rt], lo, hi, hz);
        acc = mass_logits_tile(sW[tt], sB[tt], hz, lo, hi);
    }
    if (zstore && active) {
        float* zr = zstore + ((size_t)c * n_pad + jr) * kChunkItems + tt * kTileN + 4 * hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(zr + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool valid = tile0 + il < I && !((vis >> (il & 31)) & 1u);
        acc[r] = valid ? acc[r] : -INFINITY;
        mloc = fmaxf(mloc, acc[r]);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, CIRS_WAVE));
    if (hi == 0) sM[rt][tt][lo] = mloc;
    __syncthreads();
    const float M = fmaxf(fmaxf(sM[rt][0][lo], sM[rt][1][lo]), fmaxf(sM[rt][2][lo], sM[rt][3][lo]));
    float ex[16];
    if (M > -INFINITY) {
        const float nml = -M * 1.4426950408889634f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[r] = mass_exp(acc[r], nml);      // masked elements: e^(-inf) = 0
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[r] = 0.f;
    }
    // the ordered sum, tile after tile (every wave passes every barrier)
    float sl = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (tt == t) {
            sl = t == 0 ? 0.f : sS[rt][hi][lo];
#pragma unroll
            for (int r = 0; r < 16; ++r) sl += ex[r];
            if (t < 3) sS[rt][hi][lo] = sl;
        }
        if (t < 3) __syncthreads();
    }
    if (tt == 3) {
        float L = -INFINITY;
        const float so = __shfl_xor(sl, 32, CIRS_WAVE);
        if (M > -INFINITY) {
            const float S = hi == 0 ? sl + so : so + sl;                          // S_hi0 + S_hi1, in that order on both lanes
            L = M + mass_log(S);
        }
        if (hi == 0) lmass[(size_t)c * n_pad + jr] = active ? L : -INFINITY;
    }
}

// ---- merge of the per-chunk partials of one env row (one wavefront) --------------------------------------------------------
// Candidate = (noisy score, item id, its logit); the winner is the highest score, ties -> lowest id (order independent);
// (m, s) = running max / sum-exp of the logits, folded pairwise.  Lanes first fold the chunks they own (lane, lane + 64, ...),
// then the wavefront reduces in registers: four DPP steps inside each row of 16 lanes (xor 1, xor 2, half-mirror, mirror: after
// every step both partners hold the same merged value, so the mirror partner carries exactly the other group's result) and the
// four row results are folded in row order through v_readlane.  No LDS round trips, no second look at Wa: the candidate's logit
// travels with it (it is the MFMA accumulator value the oracle's fma chain reproduces bit for bit).
struct Cand { float bs, bz, m, s; int bi; };
__device__ __forceinline__ void cand_fold(Cand& a, float os, int oi, float oz, float om, float osum) {
    if (os > a.bs || (os == a.bs && oi < a.bi)) { a.bs = os; a.bi = oi; a.bz = oz; }
    const float mn = fmaxf(a.m, om);
    if (mn > -INFINITY) {
        a.s = a.s * __expf(a.m - mn) + osum * __expf(om - mn);
        a.m = mn;
    }
}
template <int kCtrl>
__device__ __forceinline__ void cand_dpp_step(Cand& a) {
    const float os = dpp_f32<kCtrl>(a.bs), oz = dpp_f32<kCtrl>(a.bz), om = dpp_f32<kCtrl>(a.m), osum = dpp_f32<kCtrl>(a.s);
    const int oi = __builtin_amdgcn_update_dpp(0, a.bi, kCtrl, 0xF, 0xF, false);
    cand_fold(a, os, oi, oz, om, osum);
}
__device__ __forceinline__ Cand cand_wave_reduce(Cand a) {
    cand_dpp_step<0xB1>(a); cand_dpp_step<0x4E>(a); cand_dpp_step<0x141>(a); cand_dpp_step<0x140>(a);
    Cand r{row_pick(a.bs, 0), row_pick(a.bz, 0), row_pick(a.m, 0), row_pick(a.s, 0), __builtin_amdgcn_readlane(a.bi, 0)};
#pragma unroll
    for (int row = 1; row < 4; ++row)
        cand_fold(r, row_pick(a.bs, 16 * row), __builtin_amdgcn_readlane(a.bi, 16 * row), row_pick(a.bz, 16 * row), row_pick(a.m, 16 * row),
                  row_pick(a.s, 16 * row));
    return r;   // identical in every lane
}

// The two halves of a Cand reduced separately (two-level sampler: the arg-max sits on the critical path of the vector step, the
// log-sum-exp only feeds logp).  Same pairing as cand_wave_reduce; the arg-max (ties -> lowest id) does not depend on the order.
struct Best { float bs, bz; int bi; };
struct Mass { float m, s; };
__device__ __forceinline__ void best_fold(Best& a, float os, int oi, float oz) {
    if (os > a.bs || (os == a.bs && oi < a.bi)) { a.bs = os; a.bi = oi; a.bz = oz; }
}
__device__ __forceinline__ void mass_fold(Mass& a, float om, float osum) {
    const float mn = fmaxf(a.m, om);
    if (mn > -INFINITY) {
        a.s = a.s * __expf(a.m - mn) + osum * __expf(om - mn);
        a.m = mn;
    }
}
template <int kCtrl>
__device__ __forceinline__ void best_dpp_step(Best& a) {
    const float os = dpp_f32<kCtrl>(a.bs), oz = dpp_f32<kCtrl>(a.bz);
    const int oi = __builtin_amdgcn_update_dpp(0, a.bi, kCtrl, 0xF, 0xF, false);
    best_fold(a, os, oi, oz);
}
__device__ __forceinline__ Best best_wave_reduce(Best a) {
    best_dpp_step<0xB1>(a); best_dpp_step<0x4E>(a); best_dpp_step<0x141>(a); best_dpp_step<0x140>(a);
    Best r{row_pick(a.bs, 0), row_pick(a.bz, 0), __builtin_amdgcn_readlane(a.bi, 0)};
#pragma unroll
    for (int row = 1; row < 4; ++row) best_fold(r, row_pick(a.bs, 16 * row), __builtin_amdgcn_readlane(a.bi, 16 * row), row_pick(a.bz, 16 * row));
    return r;   // identical in every lane
}
template <int kCtrl>
__device__ __forceinline__ void mass_dpp_step(Mass& a) {
    const float om = dpp_f32<kCtrl>(a.m), osum = dpp_f32<kCtrl>(a.s);
    mass_fold(a, om, osum);
}
__device__ __forceinline__ Mass mass_wave_reduce(Mass a) {
    mass_dpp_step<0xB1>(a); mass_dpp_step<0x4E>(a); mass_dpp_step<0x141>(a); mass_dpp_step<0x140>(a);
    Mass r{row_pick(a.m, 0), row_pick(a.s, 0)};
#pragma unroll
    for (int row = 1; row < 4; ++row) mass_fold(r, row_pick(a.m, 16 * row), row_pick(a.s, 16 * row));
    return r;
}

// The first two chunk partials of every lane (chunks lane and lane + 64: the whole catalogue up to 128 chunks) do not depend on
// anything computed in the merging kernel: a caller may request them early (fused rollout: together with the env-state prefetch,
// before it knows whether the env still runs) and pass them in.
struct MergePre { float sc[2], z[2], m[2], s[2]; int idx[2]; };
__device__ __forceinline__ MergePre actor_merge_prefetch(int j, int lane, int n_pad, int n_chunks, const ActorPartialView& pv) {
    MergePre p;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = lane + CIRS_WAVE * q;
        const size_t o = (size_t)(c < n_chunks ? c : 0) * n_pad + j;
        p.sc[q] = pv.score[o]; p.idx[q] = pv.idx[o]; p.z[q] = pv.z[o]; p.m[q] = pv.m[o]; p.s[q] = pv.s[o];
    }
    return p;
}

__device__ __forceinline__ Cand actor_merge_chunks(int j, int lane, int n_pad, int n_chunks, const ActorPartialView& pv, const MergePre* pre) {
    Cand a{-INFINITY, 0.f, -INFINITY, 0.f, 0x7FFFFFFF};
    for (int c = lane, q = 0; c < n_chunks; c += CIRS_WAVE, ++q) {  // within a lane chunks ascend
        const size_t o = (size_t)c * n_pad + j;
        if (pre && q < 2) {
            const int k = q;   // q is 0 or 1 here: selects, no dynamic register index
            cand_fold(a, k == 0 ? pre->sc[0] : pre->sc[1], k == 0 ? pre->idx[0] : pre->idx[1], k == 0 ? pre->z[0] : pre->z[1],
                      k == 0 ? pre->m[0] : pre->m[1], k == 0 ? pre->s[0] : pre->s[1]);
        } else {
            cand_fold(a, pv.score[o], pv.idx[o], pv.z[o], pv.m[o], pv.s[o]);
        }
    }
    return cand_wave_reduce(a);
}

// action id (ties -> lowest id) and log-prob with Categorical's clamp; lane 0 writes, the action id is returned in every lane
__device__ __forceinline__ int64_t actor_merge_wave(int j, int lane, int n_pad, int n_chunks, const ActorPartialView& pv,
                                                    int64_t* __restrict__ act_out, float* __restrict__ logp_out,
                                                    const MergePre* pre = nullptr) {
    const Cand r = actor_merge_chunks(j, lane, n_pad, n_chunks, pv, pre);
    const int64_t act = r.bi == 0x7FFFFFFF ? -1 : (int64_t)r.bi;
    if (lane != 0) return act;
    act_out[j] = act;
    if (logp_out) {
        float lp = 0.f;
        if (r.bi != 0x7FFFFFFF) {
            const float lse = r.m + __logf(r.s);
            float p = __expf(r.bz - lse);  // softmax prob of the chosen item (over unmasked items)
            const float eps = 1.1920928955078125e-7f;
            p = fminf(fmaxf(p, eps), 1.0f - eps);  // torch probs_to_logits clamp
            lp = __logf(p);
        }
        logp_out[j] = lp;
    }
    return act;
}

// ---- two-level sampler, stage 2 + 3: one wavefront per env row -------------------------------------------------------------
// chunk c* = argmax_c (L_c + G1_c) (lanes stride the chunks, register-only reduction, ties -> lowest chunk), then the 128 items
// of c*: two per lane, logits as scalar fma chains in the MFMA's k-order (bias, then k = kk, 32 + kk: the very bits stage 1 saw),
// item = argmax_i (z_i + G2_i), ties -> lowest id.  The row's log-sum-exp for logp is folded from the chunk masses (hardware exp:
// tolerance-checked, it decides nothing).  hs: 64 floats of per-wave LDS scratch, stage: kPickStage more.  Result identical in
// every lane.
struct PickPre { float L[2], hv; };   // first two chunk masses of this lane + its element of the row's hidden vector
__device__ __forceinline__ PickPre actor_pick_prefetch(int j, int lane, int n_pad, int n_chunks, const float* __restrict__ lmass,
                                                       const float* __restrict__ h2) {
    PickPre p;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = lane + CIRS_WAVE * q;
        p.L[q] = lmass[(size_t)(c < n_chunks ? c : 0) * n_pad + j];
    }
    p.hv = h2[(size_t)j * kH + lane];
    return p;
}
struct PickArgs {
    const float* lmass; int n_pad, n_chunks;      // stage-1 output of this shard
    const float *wa, *ba, *h2;                   // head rows of this shard, hidden rows [n, 64]
    const uint32_t* visited; int n_items, item_base, n_items_total;
    uint64_t seed; uint32_t rng_step;
    const float* zstore;                          // (nullable) logit store of actor_mass_kernel, [n_chunks][n_pad][128]: the pick reads instead of recomputing
};
// The 128 head rows of the drawn chunk are 32 KB of CONSECUTIVE memory: the wave reads them coalesced (32 x 1 KB, all requested at
// once) and transposes them through `stage` (per-wave LDS, kPickStage floats: 64 rows of kH + 4 floats -- 16-byte rows whose b128
// reads by 16 consecutive lanes cover all 64 banks) in two halves, so that lane i then owns row i / 64 + i.  (One 256-byte row per
// lane straight from L2 costs 64 cache lines per load instruction: 20 k cycles at the head of the step kernel's critical path.)
#ifndef CIRS_PICK_STAMP
#define CIRS_PICK_STAMP(K) do { } while (0)
#endif
constexpr int kPickRow = kH + 4;
constexpr int kPickStage = 64 * kPickRow;
struct PickNoHook { __device__ __forceinline__ void operator()() const {} };
// after_issue(): called once, when the last of the chunk's rows has arrived (the memory queue is empty again) and the second
// half's dot products are about to start -- the place for a caller's own prefetches
// ZS (compile time): the logit store is present (a.zstore non-null) -- the kernel that reads the store carries neither the 128 registers of the chunk's head
// rows nor a run-time branch at the head of the pick (round 6: the nullable pointer + branch cost the C3 step kernel 1.5 us per launch).
template <bool ZS = false, class AfterIssue = PickNoHook>
__device__ __forceinline__ Cand actor_pick_wave(const PickArgs& a, int j, int e, int lane, float* hs, float* stage, const PickPre* pre,
                                                AfterIssue&& after_issue = AfterIssue()) {
    const int chunk_base = a.item_base / CIRS_SAMPLER_CHUNK;
    // the noise of this lane's first two chunks depends on nothing loaded: computed while the masses are still on their way
    float G[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = lane + CIRS_WAVE * q;
        G[q] = c < a.n_chunks ? chunk_gumbel(a.seed, a.rng_step, (uint32_t)e, (uint32_t)(chunk_base + c)) : 0.f;
    }
#ifdef CIRS_PICK_EXTRA_GUMBELS   // probe: what two more Gumbels cost at this point
    float Gx = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) Gx += chunk_gumbel(a.seed, a.rng_step, (uint32_t)e, (uint32_t)(chunk_base + 1000 + lane + CIRS_WAVE * q));
    if (Gx == 12345.678f) G[0] = Gx;
#endif
    CIRS_PICK_STAMP(26);
    Best c0{-INFINITY, 0.f, 0x7FFFFFFF};     // bi = chunk id here
    Mass ms{-INFINITY, 0.f};
    for (int c = lane, q = 0; c < a.n_chunks; c += CIRS_WAVE, ++q) {
        const float L = (pre && q < 2) ? (q == 0 ? pre->L[0] : pre->L[1]) : a.lmass[(size_t)c * a.n_pad + j];
        const float g = q == 0 ? G[0] : q == 1 ? G[1] : chunk_gumbel(a.seed, a.rng_step, (uint32_t)e, (uint32_t)(chunk_base + c));
        if (L > -INFINITY) { best_fold(c0, L + g, c, 0.f); mass_fold(ms, L, 1.0f); }
    }
    CIRS_PICK_STAMP(20);
    const Best cw = best_wave_reduce(c0);
    CIRS_PICK_STAMP(21);
    if (cw.bi == 0x7FFFFFFF) return Cand{cw.bs, 0.f, -INFINITY, 0.f, 0x7FFFFFFF};          // nothing left to recommend
    const int row0 = cw.bi * CIRS_SAMPLER_CHUNK;  // first local item of the drawn chunk
    // bias / visited words of this lane's two items first, then all 32 KB of the chunk's rows in flight at once (rows past the
    // shard's end re-read its last row: never used)
    const int I_tot = a.n_items_total > 0 ? a.n_items_total : a.n_items;
    const int vis_words = (I_tot + 31) / 32;
    float bias[2];
    uint32_t vword[2];
    bool live[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int il = row0 + lane + 64 * q;      // local item of this shard
        live[q] = il < a.n_items;
        const int ilc = live[q] ? il : a.n_items - 1;
        bias[q] = a.ba[ilc];
        vword[q] = a.visited ? a.visited[(size_t)e * vis_words + ((a.item_base + ilc) >> 5)] : 0u;
    }
    if constexpr (ZS) {      // the drawn chunk's logits were kept by actor_mass_kernel: two coalesced loads per lane, no head rows, no dot products
        float zq[2], gz[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) zq[q] = a.zstore[((size_t)cw.bi * a.n_pad + j) * kChunkItems + lane + 64 * q];
        CIRS_PICK_STAMP(36);
#pragma unroll
        for (int q = 0; q < 2; ++q) gz[q] = actor_gumbel(a.seed, a.rng_step, (uint32_t)e, (uint32_t)(a.item_base + row0 + lane + 64 * q));
        CIRS_PICK_STAMP(38);
        const Mass mwz = mass_wave_reduce(ms);
        CIRS_PICK_STAMP(22);
        CIRS_PICK_STAMP(23);
        CIRS_PICK_STAMP(37);
        after_issue();
        Best itz{-INFINITY, 0.f, 0x7FFFFFFF};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ig = a.item_base + row0 + lane + 64 * q;
            if (live[q] && !((vword[q] >> (ig & 31)) & 1u)) best_fold(itz, zq[q] + gz[q], ig, zq[q]);
        }
        CIRS_PICK_STAMP(24);
        const Best iwz = best_wave_reduce(itz);
        CIRS_PICK_STAMP(25);
        return Cand{cw.bs, iwz.bz, mwz.m, mwz.s, iwz.bi};
    } else {
    // (native vector values: a float4 struct copy becomes a memcpy through a private array that is not promoted to registers)
    typedef float pick_v4 __attribute__((ext_vector_type(4)));
    const pick_v4* wa4 = reinterpret_cast<const pick_v4*>(a.wa);
    pick_v4 g0[16], g1[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        int r = row0 + 4 * t + (lane >> 4);
        r = r < a.n_items ? r : a.n_items - 1;
        g0[t] = wa4[(size_t)r * (kH / 4) + (lane & 15)];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        int r = row0 + 64 + 4 * t + (lane >> 4);
        r = r < a.n_items ? r : a.n_items - 1;
        g1[t] = wa4[(size_t)r * (kH / 4) + (lane & 15)];
    }
    CIRS_PICK_STAMP(36);
    // underneath the loads: the item-level noise of this lane's two items and the log-sum-exp of the chunk masses
    float gi[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) gi[q] = actor_gumbel(a.seed, a.rng_step, (uint32_t)e, (uint32_t)(a.item_base + row0 + lane + 64 * q));
    CIRS_PICK_STAMP(38);
    const Mass mw = mass_wave_reduce(ms);
    hs[lane] = pre ? pre->hv : a.h2[(size_t)j * kH + lane];
    pick_v4* st4 = reinterpret_cast<pick_v4*>(stage);
    Best it{-INFINITY, 0.f, 0x7FFFFFFF};
    CIRS_PICK_STAMP(22);
#define CIRS_PICK_HALF(G4, Q)                                                                                         \
    do {                                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) st4[(4 * t + (lane >> 4)) * (kPickRow / 4) + (lane & 15)] = G4[t]; \
        __builtin_amdgcn_wave_barrier();                                                                              \
        if (Q == 1) { CIRS_PICK_STAMP(37); after_issue(); }   /* every row has arrived: the memory queue is empty */     \
        float z = bias[Q];                                                                                            \
        _Pragma("unroll") for (int k4 = 0; k4 < 8; ++k4) {                                                            \
            const pick_v4 lo4 = st4[lane * (kPickRow / 4) + k4], hi4 = st4[lane * (kPickRow / 4) + 8 + k4];           \
            z = __builtin_fmaf(hs[4 * k4 + 0], lo4.x, z); z = __builtin_fmaf(hs[32 + 4 * k4 + 0], hi4.x, z);          \
            z = __builtin_fmaf(hs[4 * k4 + 1], lo4.y, z); z = __builtin_fmaf(hs[32 + 4 * k4 + 1], hi4.y, z);          \
            z = __builtin_fmaf(hs[4 * k4 + 2], lo4.z, z); z = __builtin_fmaf(hs[32 + 4 * k4 + 2], hi4.z, z);          \
            z = __builtin_fmaf(hs[4 * k4 + 3], lo4.w, z); z = __builtin_fmaf(hs[32 + 4 * k4 + 3], hi4.w, z);          \
        }                                                                                                             \
        const int ig = a.item_base + row0 + lane + 64 * Q;                                                            \
        if (live[Q] && !((vword[Q] >> (ig & 31)) & 1u)) best_fold(it, z + gi[Q], ig, z);                              \
        __builtin_amdgcn_wave_barrier();                                                                              \
    } while (0)
    CIRS_PICK_HALF(g0, 0);
    CIRS_PICK_STAMP(23);
    CIRS_PICK_HALF(g1, 1);
    CIRS_PICK_STAMP(24);
#undef CIRS_PICK_HALF
    const Best iw = best_wave_reduce(it);
    CIRS_PICK_STAMP(25);
    return Cand{cw.bs, iw.bz, mw.m, mw.s, iw.bi};   // bs = the chunk-level noisy score (what shards are compared by), (m, s) = log-sum-exp of the shard's masses
    }
}

// log-prob of the drawn item with Categorical's clamp (torch probs_to_logits), from its logit and the row's (max, sum-exp)
__device__ __forceinline__ float cand_logp(const Cand& r) {
    if (r.bi == 0x7FFFFFFF) return 0.f;
    const float lse = r.m + __logf(r.s);
    float p = __expf(r.bz - lse);
    const float eps = 1.1920928955078125e-7f;
    p = fminf(fmaxf(p, eps), 1.0f - eps);
    return __logf(p);
}

// Column-sharded head: this shard's chunk partials of env row j as ONE tuple (score, global id, logit of that candidate, running
// max, running sum-exp) -- what a rank contributes to the cross-rank merge.
__device__ __forceinline__ void actor_shard_tuple_wave(int j, int lane, int n_pad, int n_chunks, const ActorPartialView& pv, int n,
                                                       float* __restrict__ out5) {
    const Cand r = actor_merge_chunks(j, lane, n_pad, n_chunks, pv, nullptr);
    if (lane != 0) return;
    out5[j] = r.bs;
    reinterpret_cast<int32_t*>(out5)[(size_t)n + j] = r.bi;
    out5[(size_t)2 * n + j] = r.bi != 0x7FFFFFFF ? r.bz : 0.f;
    out5[(size_t)3 * n + j] = r.m;
    out5[(size_t)4 * n + j] = r.s;
}

}  // namespace cirs
