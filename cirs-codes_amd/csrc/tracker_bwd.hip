// tracker_bwd.hip -- gradient of the PPO loss w.r.t. the state tracker (reference: the autograd graph kept alive by
// core/policy/ppo.py:215 loss.backward(retain_graph=True); structure of the tracker: core/state_tracker.py:170-250).
//
// Rows = buffer rows (env b, position p = t, p < len_b), env-major, so the keys/values an attention row needs are
// the contiguous rows offsets[b] .. offsets[b]+p.  Forward activations are recomputed layer by layer into row-major
// scratch and the backward walks the layers in reverse.  Kernels are "one thread per (row, feature)" dense ops and
// "one wavefront per row" attention ops: the tensors are tiny (<= 128 features) and the work is latency-bound, what
// matters is that everything stays on device and every reduction has a fixed order.
//   weight gradients : chunk-partial sums over 256-row chunks, then an ordered sum over chunks (two launches)
//   embedding tables : per-row contributions, then one wavefront per table row sums its rows in a fixed order
#include <algorithm>
#include <hipcub/hipcub.hpp>
#include "small_gemm.h"
#include "rng.h"
#include "internal.h"

namespace cirs {
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int tD = 32, tH = 128;
#ifdef CIRS_TBWD_PROF
// stage timestamps of one mid-grid workgroup (probe builds only: tools/probes/tbwd_prof.py)
__device__ unsigned long long g_tbwd_prof[64];
#define CIRS_BSTAMP(K) do { if ((int)blockIdx.x == 300 && threadIdx.x == 0) g_tbwd_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
// entry / exit time and hardware id of EVERY workgroup of one launch (SEL selects the launch)
__device__ unsigned long long g_tbwd_wg[2048 * 3];
#define CIRS_BWG(SEL, WHICH) do { if ((SEL) && threadIdx.x == 0 && blockIdx.x < 2048) { g_tbwd_wg[blockIdx.x * 3 + (WHICH)] = __builtin_amdgcn_s_memtime(); \
    if ((WHICH) == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); g_tbwd_wg[blockIdx.x * 3 + 2] = ((unsigned long long)xcc << 32) | hw; } } } while (0)
#else
#define CIRS_BSTAMP(K) do { } while (0)
#define CIRS_BWG(SEL, WHICH) do { } while (0)
#endif
constexpr int kChunkRows = 256;

// dropout of the recompute / backward (production mode): the masks the forward decode steps applied, regenerated from their
// counters (csrc/rng.h); on == 0: every kernel below is the dropout-free code path, bit for bit
struct DropCfg {
    int on;
    uint32_t thr;
    float inv;
    uint64_t seed;
    int env_base;
};
__device__ __forceinline__ float drop_apply(const DropCfg& dc, float v, int env, int pos, int layer, int site, int elem) {
    return dropout_keep(dc.seed, (uint32_t)(dc.env_base + env), (uint32_t)pos, (uint32_t)layer, (uint32_t)site, (uint32_t)elem, dc.thr) ? v * dc.inv : 0.f;
}
// elementwise: out[r, c] = mask(row r's (env, position), layer, site, c) * in[r, c]   (N columns per row; in == out allowed)
__global__ __launch_bounds__(256) void drop_rows(DropCfg dc, const float* __restrict__ in, const int32_t* __restrict__ row_env,
                                                 const int32_t* __restrict__ row_t, int R, int N, int layer, int site, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)R * N) return;
    const int r = (int)(i / N), c = (int)(i % N);
    out[i] = drop_apply(dc, in[i], row_env[r], row_t[r], layer, site, c);
}

// slot gather + scale + positional encoding: X0[r] = x_hist[b,p], H0 = X0*sqrt(D) + pe[p]
__global__ __launch_bounds__(256) void embed_rows(const float* __restrict__ x_hist, const float* __restrict__ pe,
                                                  const int32_t* __restrict__ row_env, const int32_t* __restrict__ row_t, int R,
                                                  int L, float* __restrict__ H0, DropCfg dc) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)R * tD) return;
    const int r = (int)(i / tD), d = (int)(i % tD);
    const int b = row_env[r], p = row_t[r];
    const float h = x_hist[((size_t)b * L + p) * tD + d] * 5.656854249492381f + pe[(size_t)p * tD + d];
    H0[i] = dc.on ? drop_apply(dc, h, b, p, 0, CIRS_DROP_POS, d) : h;
}

// causal attention forward, one wavefront per row; Q/K/V live in QKV[R,96]; P[R,NH,Lp] keeps the probabilities for
// the backward.  The probabilities are exchanged between lanes through a per-wave LDS strip (not through global).
template <int NH>
__global__ __launch_bounds__(256) void attn_fwd(const float* __restrict__ QKV, const int32_t* __restrict__ row_env,
                                                const int32_t* __restrict__ row_t, const int32_t* __restrict__ offsets, int R,
                                                int Lp, float* __restrict__ P, float* __restrict__ ATT, DropCfg dc, int layer,
                                                float* __restrict__ PM) {
    // dc.on: PM [R, NH, Lp] receives the probabilities AFTER the attention dropout (what multiplies V); P stays the softmax
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HD = tD / NH;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float* ps = smem + (size_t)(threadIdx.x >> 6) * NH * Lp;
    const int p = row_t[r], base = offsets[row_env[r]];
    const float scale = 1.0f / sqrtf((float)HD);
    // rows of QKV are 384 bytes (16-byte aligned): eight float4 loads per row instead of 32 dwords
    float q[tD];
    {
        const f32x4* q4 = reinterpret_cast<const f32x4*>(QKV + (size_t)r * 96);
#pragma unroll
        for (int d4 = 0; d4 < tD / 4; ++d4) {
            const f32x4 t = q4[d4];
            q[4 * d4] = t.x * scale; q[4 * d4 + 1] = t.y * scale; q[4 * d4 + 2] = t.z * scale; q[4 * d4 + 3] = t.w * scale;
        }
    }
    float mx[NH], sm[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) mx[h] = -INFINITY;
    float* Pr = P + (size_t)r * NH * Lp;
    for (int jp = lane; jp <= p; jp += CIRS_WAVE) {
        float k[tD];
        {
            const f32x4* k4 = reinterpret_cast<const f32x4*>(QKV + (size_t)(base + jp) * 96 + tD);
#pragma unroll
            for (int d4 = 0; d4 < tD / 4; ++d4) {
                const f32x4 t = k4[d4];
                k[4 * d4] = t.x; k[4 * d4 + 1] = t.y; k[4 * d4 + 2] = t.z; k[4 * d4 + 3] = t.w;
            }
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) sc = __builtin_fmaf(q[h * HD + d], k[h * HD + d], sc);
            ps[h * Lp + jp] = sc;
            mx[h] = fmaxf(mx[h], sc);
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) { mx[h] = wave_max_f32(mx[h]); sm[h] = 0.f; }
    for (int jp = lane; jp <= p; jp += CIRS_WAVE) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float e = expf(ps[h * Lp + jp] - mx[h]);
            ps[h * Lp + jp] = e;
            sm[h] += e;
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) sm[h] = 1.0f / wave_sum_f32(sm[h]);
    for (int jp = lane; jp <= p; jp += CIRS_WAVE) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float pr = ps[h * Lp + jp] * sm[h];
            Pr[h * Lp + jp] = pr;
            if (dc.on) {
                const float pm = drop_apply(dc, pr, row_env[r], p, layer, CIRS_DROP_ATTN, jp * NH + h);
                PM[(size_t)r * NH * Lp + h * Lp + jp] = pm;
                ps[h * Lp + jp] = pm;
            } else {
                ps[h * Lp + jp] = pr;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int half = lane >> 5, d = lane & 31, h = d / HD;
    float acc = 0.f;
    // 8 value rows in flight per pass (a load per iteration inside the fma chain is one L2 round trip per key); same chain order
    for (int j0 = half; j0 <= p; j0 += 16) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = j0 + 2 * u <= p ? QKV[(size_t)(base + j0 + 2 * u) * 96 + 2 * tD + d] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + 2 * u <= p) acc = __builtin_fmaf(ps[h * Lp + j0 + 2 * u], v8[u], acc);
    }
    acc += __shfl_xor(acc, 32, CIRS_WAVE);
    if (lane < tD) ATT[(size_t)r * tD + d] = acc;
}

// query side of the attention backward: dS (stored over P's twin buffer) and dQ
template <int NH>
__global__ __launch_bounds__(256) void attn_bwd_q(const float* __restrict__ QKV, const float* __restrict__ P, const float* __restrict__ dATT,
                                                  const int32_t* __restrict__ row_env, const int32_t* __restrict__ row_t,
                                                  const int32_t* __restrict__ offsets, int R, int Lp, float* __restrict__ dS,
                                                  float* __restrict__ dQKV, const float* __restrict__ PM, float drop_inv) {
    // PM (nullable): probabilities after the attention dropout; dP = dPM * (kept ? 1/(1-p) : 0), kept <=> PM != 0 (or P == 0)
    constexpr int HD = tD / NH;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int p = row_t[r], base = offsets[row_env[r]];
    const float scale = 1.0f / sqrtf((float)HD);
    float da[tD];
    {
        const f32x4* a4 = reinterpret_cast<const f32x4*>(dATT + (size_t)r * tD);
#pragma unroll
        for (int d4 = 0; d4 < tD / 4; ++d4) {
            const f32x4 t = a4[d4];
            da[4 * d4] = t.x; da[4 * d4 + 1] = t.y; da[4 * d4 + 2] = t.z; da[4 * d4 + 3] = t.w;
        }
    }
    const float* Pr = P + (size_t)r * NH * Lp;
    float* dSr = dS + (size_t)r * NH * Lp;
    float dot[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) dot[h] = 0.f;
    for (int jp = lane; jp <= p; jp += CIRS_WAVE) {
        float v[tD];
        {
            const f32x4* v4 = reinterpret_cast<const f32x4*>(QKV + (size_t)(base + jp) * 96 + 2 * tD);
#pragma unroll
            for (int d4 = 0; d4 < tD / 4; ++d4) {
                const f32x4 t = v4[d4];
                v[4 * d4] = t.x; v[4 * d4 + 1] = t.y; v[4 * d4 + 2] = t.z; v[4 * d4 + 3] = t.w;
            }
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) dp = __builtin_fmaf(da[h * HD + d], v[h * HD + d], dp);
            if (PM) dp = PM[(size_t)r * NH * Lp + h * Lp + jp] != 0.f ? dp * drop_inv : 0.f;
            dSr[h * Lp + jp] = dp;  // dP for now
            dot[h] = __builtin_fmaf(Pr[h * Lp + jp], dp, dot[h]);
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) dot[h] = wave_sum_f32(dot[h]);
    // softmax backward: dS = P * (dP - sum_j P dP), kept in global memory for the key-side kernel and in LDS for dQ
    extern __shared__ float smem_q[];
    float* sds = smem_q + (size_t)(threadIdx.x >> 6) * NH * Lp;
    for (int jp = lane; jp <= p; jp += CIRS_WAVE) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float ds = Pr[h * Lp + jp] * (dSr[h * Lp + jp] - dot[h]);
            dSr[h * Lp + jp] = ds;
            sds[h * Lp + jp] = ds;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // dQ[d] = scale * sum_j dS[h(d), j] * K[j, d]: lane = (half, d), the two halves take every second key (coalesced rows)
    const int half = lane >> 5, d = lane & (tD - 1), hd = d / HD;
    float acc = 0.f;
    for (int j0 = half; j0 <= p; j0 += 16) {   // 8 key rows in flight per pass, same chain order
        float k8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k8[u] = j0 + 2 * u <= p ? QKV[(size_t)(base + j0 + 2 * u) * 96 + tD + d] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + 2 * u <= p) acc = __builtin_fmaf(sds[hd * Lp + j0 + 2 * u], k8[u], acc);
    }
    acc += __shfl_xor(acc, 32, CIRS_WAVE);
    if (lane < tD) dQKV[(size_t)r * 96 + d] = acc * scale;  // scores used q*scale
}

// key side: dK[p'] = sum_{p >= p'} dS[p,p'] * q[p]*scale ; dV[p'] = sum_{p >= p'} P[p,p'] * dATT[p]
template <int NH>
__global__ __launch_bounds__(256) void attn_bwd_kv(const float* __restrict__ QKV, const float* __restrict__ P, const float* __restrict__ dS,
                                                   const float* __restrict__ dATT, const int32_t* __restrict__ row_env,
                                                   const int32_t* __restrict__ row_t, const int32_t* __restrict__ offsets,
                                                   const int32_t* __restrict__ lens, int R, int Lp, float* __restrict__ dQKV) {
    constexpr int HD = tD / NH;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int b = row_env[r], pk = row_t[r], base = offsets[b], len = lens[b];
    const float scale = 1.0f / sqrtf((float)HD);
    // lane = (which, d): lanes 0..31 own dK[d], lanes 32..63 own dV[d]; the later queries of the episode are walked in order
    // (<= max_turn of them): coalesced 128-byte rows, no cross-lane reduction
    const int d = lane & (tD - 1), which = lane >> 5, h = d / HD;
    const float* coef = (which == 0 ? dS : P) + ((size_t)base * NH + h) * Lp + pk;    // [query row][head][key position]
    const float* vec = which == 0 ? QKV + (size_t)base * 96 + d : dATT + (size_t)base * tD + d;
    const size_t cstride = (size_t)NH * Lp, vstride = which == 0 ? 96 : tD;
    const float cs = which == 0 ? scale : 1.0f;
    float acc = 0.f;
    int p = pk;
    for (; p < len; p += 8) {   // 16 independent loads in flight per pass (predicated tail), same chain order
        float c8[8], v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = p + u < len;
            c8[u] = ok ? coef[(size_t)(p + u) * cstride] : 0.f;
            v8[u] = ok ? vec[(size_t)(p + u) * vstride] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p + u < len) acc = __builtin_fmaf(c8[u] * cs, v8[u], acc);
    }
    dQKV[(size_t)r * 96 + tD + lane] = acc;   // columns [32, 64) = dK, [64, 96) = dV
}

// ---- per-episode attention (round 3): one wavefront per env, the episode's Q / K / V rows staged in LDS ONCE ---------------------------
// The per-row kernels above give every query row its own wavefront: <= max_turn of the 64 lanes hold a key, and every wavefront re-reads
// the keys / values of its episode from L2 (3 launches x 30 k wavefronts x ~40 dependent global loads at C3: 34 + 48 + 28 us per
// layer).  Here a lane owns a head group of TWO queries, p and len-1-p (the causal triangle folded: every lane walks ~len keys), its keys
// come out of LDS -- one read serves both queries and, the head groups sharing the instruction, all heads -- and soft-max statistics, dQ and
// the output need no cross-lane reduction.  One wavefront per SIMD runs at ~3.7 cycles per instruction, so the loop bodies are branch-free
// (a conditional store per query compiled to eight s_cbranch_execz blocks per iteration): every lane stores every step, lanes without a
// query write a dummy strip row, columns beyond a query's causal limit are never read, selects do the masking.
// cross-lane reduction.  In the second phase of the backward the lane owns two KEYS the same way (dK / dV accumulated over the later queries
// in order).  The probabilities are not kept between the forward and the backward pass: the backward recomputes them from Q, K (cheaper
// than 15 MB of P through HBM).  Used when max_len <= 64 and the LDS image fits 64 KB (max_len <= 50 at 4 heads); longer episodes take the
// per-row kernels.
template <int NH> struct EpGeo {
    static constexpr int NW = NH < 4 ? NH : 4;      // head groups (lane / SL)
    static constexpr int SL = 64 / NW;              // query-pair slots per head group and pass
    static constexpr int HPL = NH / NW;             // heads per lane
    static constexpr int HD = tD / NH;
    static constexpr int DPL = HPL * HD;            // dims per lane
    __host__ __device__ static constexpr int strip(int Lp) { return (HPL * Lp) | 1; }   // odd stride between queries: conflict-free both ways
    __host__ __device__ static constexpr size_t keep_words(int Lp) { return 2 * (size_t)NH * Lp; }      // dropout keep bits: [query][head] x 64 keys
    __host__ __device__ static constexpr size_t fwd_floats(int Lp) { return (size_t)Lp * 64 + (size_t)NW * (Lp + 1) * strip(Lp) + keep_words(Lp); }
    __host__ __device__ static constexpr size_t bwd_floats(int Lp) { return (size_t)Lp * 128 + (size_t)NW * (Lp + 1) * strip(Lp) + 2 * (size_t)NH * Lp + keep_words(Lp); }
};

// stage `cols` floats (multiple of 4) per row of an episode, rows `src_stride` apart, into LDS rows of `cols` floats
__device__ __forceinline__ void ep_stage(float* __restrict__ dst, const float* __restrict__ src, int rows, int src_stride, int cols, int tid, int nt) {
    const int c4n = cols >> 2;
    for (int i = tid; i < rows * c4n; i += nt) {
        const int row = i / c4n, c4 = i - row * c4n;
        *reinterpret_cast<f32x4*>(dst + (size_t)row * cols + 4 * c4) = *reinterpret_cast<const f32x4*>(src + (size_t)row * src_stride + 4 * c4);
    }
}
// The same staging in two halves, for the episode's first (at most) 32 rows: all loads requested at once into registers (unconditional, clamped: they batch),
// committed to LDS later -- whatever runs in between (the keep bits' Philox blocks) runs under the memory round trip.  NQ = 32 * COLS / 256 float4 per lane.
template <int COLS, int NQ>
__device__ __forceinline__ void ep_stage_issue(f32x4 (&r)[NQ], const float* __restrict__ src, int rows, int src_stride, int tid) {
    constexpr int c4n = COLS >> 2;
    static_assert(NQ * 64 >= 32 * c4n, "registers for 32 rows");
    const int n = rows * c4n;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = min(tid + 64 * q, n - 1), row = i / c4n, c4 = i - row * c4n;
        r[q] = *reinterpret_cast<const f32x4*>(src + (size_t)row * src_stride + 4 * c4);
    }
}
template <int COLS, int NQ>
__device__ __forceinline__ void ep_stage_commit(float* __restrict__ dst, const f32x4 (&r)[NQ], int rows, int tid) {
    constexpr int c4n = COLS >> 2;
    const int n = rows * c4n;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + 64 * q, row = i / c4n, c4 = i - row * c4n;
        if (i < n) *reinterpret_cast<f32x4*>(dst + (size_t)row * COLS + 4 * c4) = r[q];
    }
}
// one-wavefront workgroups: LDS operations of a wavefront complete in order, so "everything before is visible to every lane" is a wait for the LDS counter
// -- NOT __syncthreads(), whose release fence also waits for every global load in flight (the staged rows above)
__device__ __forceinline__ void ep_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
template <int N> __device__ __forceinline__ void ep_load(float (&v)[N], const float* __restrict__ src) {
#pragma unroll
    for (int d4 = 0; d4 < N / 4; ++d4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * d4);
        v[4 * d4] = t.x; v[4 * d4 + 1] = t.y; v[4 * d4 + 2] = t.z; v[4 * d4 + 3] = t.w;
    }
}
template <int N> __device__ __forceinline__ void ep_store(float* __restrict__ dst, const float (&v)[N], float a) {
#pragma unroll
    for (int d4 = 0; d4 < N / 4; ++d4)
        *reinterpret_cast<f32x4*>(dst + 4 * d4) = f32x4{v[4 * d4] * a, v[4 * d4 + 1] * a, v[4 * d4 + 2] * a, v[4 * d4 + 3] * a};
}
// exp(x) for x <= 0 on the transcendental unit (v_exp_f32, ~1 ulp): the per-episode kernels keep their scores in log2 units
__device__ __forceinline__ float ep_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
template <int HD> __device__ __forceinline__ float ep_dot(const float* a, const float* b) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = __builtin_fmaf(a[d], b[d], s);
    return s;
}
// The keep bits of an episode's attention-probability dropout, built ONCE per episode (round 6): sK[(p * NH + h) * 2 + (j >> 5)] bit (j & 31) = keep(query p,
// key j <= p, head h) -- the masks EP_KEEP(p, j * NH + h) of rng.h, unchanged.  One Philox block covers four consecutive (key, head) elements of a query, so
// an episode of 30 rows needs ~470 blocks = 8 per lane; evaluated where the bits are used it was one block per (query, key) PER LANE of EVERY loop (the four
// lanes that share a query recomputed each other's block): 60 per lane in the forward, 180 in the backward, ~500 cycles each -- the kernels were Philox-bound.
// Rows p and len - 1 - p are folded into one work row (together len + 1 keys) so that the lanes' item counts are even.
template <int NH>
__device__ __forceinline__ void ep_keep_build(uint32_t* __restrict__ sK, int len, int b, int layer, const DropCfg& dc, int tid, int nt) {
    // (every caller is a one-wavefront workgroup: wave syncs, so that global loads in flight stay in flight)
    for (int i = tid; i < len * NH * 2; i += nt) sK[i] = 0u;
    ep_wave_sync();
    const int half = (len + 1) >> 1;                       // folded rows
    const int bpr = (((len + 1) * NH + 3) >> 2) + 1;       // blocks per folded row (both parts rounded up)
    for (int i = tid; i < half * bpr; i += nt) {
        const int fr = i / bpr, g0 = i - fr * bpr;
        const int pA = fr, pB = len - 1 - fr;
        const int nA = ((pA + 1) * NH + 3) >> 2, nB = pB > pA ? ((pB + 1) * NH + 3) >> 2 : 0;
        const bool inA = g0 < nA;
        const int p = inA ? pA : pB, g = inA ? g0 : g0 - nA;
        if (g0 < nA + nB) {
            const u32x4 r = dropout_block(dc.seed, (uint32_t)(dc.env_base + b), (uint32_t)p, (uint32_t)layer, (uint32_t)CIRS_DROP_ATTN, (uint32_t)g);
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                const int elem = 4 * g + wd, j = elem / NH, h = elem - j * NH;
                if (j <= p && block_word(r, (uint32_t)wd) >= dc.thr) atomicOr(&sK[(p * NH + h) * 2 + (j >> 5)], 1u << (j & 31));
            }
        }
    }
    ep_wave_sync();
}
// the same bits for ONE query p (the last layer of the prefix pass only attends from the last row): sK[h * 2 + (j >> 5)]
template <int NH>
__device__ __forceinline__ void ep_keep_build_row(uint32_t* __restrict__ sK, int p, int b, int layer, const DropCfg& dc, int tid, int nt) {
    for (int i = tid; i < NH * 2; i += nt) sK[i] = 0u;
    __syncthreads();
    for (int g = tid; 4 * g < (p + 1) * NH; g += nt) {
        const u32x4 r = dropout_block(dc.seed, (uint32_t)(dc.env_base + b), (uint32_t)p, (uint32_t)layer, (uint32_t)CIRS_DROP_ATTN, (uint32_t)g);
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            const int elem = 4 * g + wd, j = elem / NH, h = elem - j * NH;
            if (j <= p && block_word(r, (uint32_t)wd) >= dc.thr) atomicOr(&sK[h * 2 + (j >> 5)], 1u << (j & 31));
        }
    }
    __syncthreads();
}
__device__ __forceinline__ bool ep_keep_bit(const uint32_t (&m)[2], int j) { return ((j < 32 ? m[0] : m[1]) >> (j & 31)) & 1u; }

template <int NH, bool kDrop>
__global__ __launch_bounds__(64) void attn_fwd_ep(const float* __restrict__ QKV, const int32_t* __restrict__ offsets,
                                                  const int32_t* __restrict__ lens, int Lp, float* __restrict__ ATT, DropCfg dc, int layer,
                                                  uint32_t* __restrict__ keep_out /* nullable: the keep bits, [row][head][2] words, for attn_bwd_ep */) {
    using G = EpGeo<NH>;
    constexpr int HPL = G::HPL, DPL = G::DPL, HD = G::HD, SL = G::SL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, len = lens[b];
    if (len <= 0) return;
    const int lane = threadIdx.x, hg = lane / SL, slot = lane - hg * SL;
    const int base = offsets[b], STR = G::strip(Lp);
    float* sKV = smem;                          // [len][K 32 | V 32]
    float* sS = smem + (size_t)Lp * 64;         // [head group][query | dummy row Lp][HPL][Lp] (query stride STR)
    uint32_t* sK = reinterpret_cast<uint32_t*>(sS + (size_t)G::NW * (Lp + 1) * STR);   // dropout keep bits [query][head][2]
    // the first 32 rows' K | V: requested now, committed behind the keep bits (their ~8 Philox blocks per lane run under the round trip)
    const int len32 = min(len, 32);
    f32x4 st_kv[8];
    ep_stage_issue<64, 8>(st_kv, QKV + (size_t)base * 96 + tD, len32, 96, lane);
    if (kDrop) {
        ep_keep_build<NH>(sK, len, b, layer, dc, lane, 64);
        if (keep_out)
            for (int i = lane; i < len * NH * 2; i += 64) keep_out[(size_t)base * NH * 2 + i] = sK[i];
    }
    ep_stage_commit<64, 8>(sKV, st_kv, len32, lane);
    if (len > 32) ep_stage(sKV + 32 * 64, QKV + (size_t)(base + 32) * 96 + tD, len - 32, 96, 64, lane, 64);
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)HD), qs = scale * 1.4426950408889634f;   // q * scale * log2(e): scores in log2 units
    for (int s0 = 0; 2 * s0 < len; s0 += SL) {
        const int pA = s0 + slot, pB = len - 1 - pA;
        const bool actA = pA <= pB, actB = pA < pB;        // the middle query of an odd episode sits in the A half only
        const int rA = actA ? pA : 0, rB = actB ? pB : 0;
        const int jn = len - s0;                           // the pass's longest key walk (slot 0's query B)
        // branch-free loops: every lane stores every step; lanes without a query write the dummy row, columns beyond a query's causal limit are never read
        float* myA = sS + ((size_t)hg * (Lp + 1) + (actA ? pA : Lp)) * STR;
        float* myB = sS + ((size_t)hg * (Lp + 1) + (actB ? pB : Lp)) * STR;
        float qA[DPL], qB[DPL];
        ep_load(qA, QKV + (size_t)(base + rA) * 96 + hg * DPL);
        ep_load(qB, QKV + (size_t)(base + rB) * 96 + hg * DPL);
#pragma unroll
        for (int d = 0; d < DPL; ++d) { qA[d] *= qs; qB[d] *= qs; }
        float mxA[HPL], mxB[HPL], smA[HPL], smB[HPL];
#pragma unroll
        for (int h = 0; h < HPL; ++h) { mxA[h] = mxB[h] = -INFINITY; smA[h] = smB[h] = 0.f; }
        float kq[DPL];
        ep_load(kq, sKV + hg * DPL);
#pragma unroll 4
        for (int j = 0; j < jn; ++j) {
            float k[DPL];      // software pipeline: key j + 1 is requested before key j is used (the strip stores below would otherwise order the loads)
#pragma unroll
            for (int d = 0; d < DPL; ++d) k[d] = kq[d];
            ep_load(kq, sKV + (size_t)(j + 1 < jn ? j + 1 : j) * 64 + hg * DPL);
            const bool inA = actA && j <= pA, inB = actB && j <= pB;
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                const float sa = ep_dot<HD>(qA + h * HD, k + h * HD), sb = ep_dot<HD>(qB + h * HD, k + h * HD);
                myA[h * Lp + j] = sa; myB[h * Lp + j] = sb;
                mxA[h] = inA ? fmaxf(mxA[h], sa) : mxA[h];
                mxB[h] = inB ? fmaxf(mxB[h], sb) : mxB[h];
            }
        }
        float accA[DPL], accB[DPL];
#pragma unroll
        for (int d = 0; d < DPL; ++d) accA[d] = accB[d] = 0.f;
        uint32_t kmA[HPL][2], kmB[HPL][2];
        if (kDrop) {
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                kmA[h][0] = sK[(rA * NH + hg * HPL + h) * 2]; kmA[h][1] = sK[(rA * NH + hg * HPL + h) * 2 + 1];
                kmB[h][0] = sK[(rB * NH + hg * HPL + h) * 2]; kmB[h][1] = sK[(rB * NH + hg * HPL + h) * 2 + 1];
            }
        }
        float vq[DPL];
        ep_load(vq, sKV + 32 + hg * DPL);
#pragma unroll 4
        for (int j = 0; j < jn; ++j) {
            float v[DPL];
#pragma unroll
            for (int d = 0; d < DPL; ++d) v[d] = vq[d];
            ep_load(vq, sKV + (size_t)(j + 1 < jn ? j + 1 : j) * 64 + 32 + hg * DPL);
            const bool inA = actA && j <= pA, inB = actB && j <= pB;
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                float ea = ep_exp2(myA[h * Lp + j] - mxA[h]), eb = ep_exp2(myB[h * Lp + j] - mxB[h]);
                ea = inA ? ea : 0.f; eb = inB ? eb : 0.f;
                smA[h] += ea; smB[h] += eb;
                if (kDrop) {
                    if (!ep_keep_bit(kmA[h], j)) ea = 0.f;
                    if (!ep_keep_bit(kmB[h], j)) eb = 0.f;
                }
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    accA[h * HD + d] = __builtin_fmaf(ea, v[h * HD + d], accA[h * HD + d]);
                    accB[h * HD + d] = __builtin_fmaf(eb, v[h * HD + d], accB[h * HD + d]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < HPL; ++h) {
            const float ia = (kDrop ? dc.inv : 1.0f) / smA[h], ib = (kDrop ? dc.inv : 1.0f) / smB[h];
#pragma unroll
            for (int d = 0; d < HD; ++d) { accA[h * HD + d] *= ia; accB[h * HD + d] *= ib; }
        }
        if (actA) ep_store(ATT + (size_t)(base + pA) * tD + hg * DPL, accA, 1.0f);
        if (actB) ep_store(ATT + (size_t)(base + pB) * tD + hg * DPL, accB, 1.0f);
    }
}

// backward of the same attention: dQKV[r] = [dQ | dK | dV] for every row of the episode.
//   phase A (lane = two queries):  scores -> soft-max statistics (m, 1/sum) -> dot = sum_j P dP -> dS = P (dP - dot) into the strip, dQ
//   phase B (lane = two keys):     dK[j] = sum_{p >= j} dS[p, j] q[p] scale,  dV[j] = sum_{p >= j} PM[p, j] dATT[p], queries in order; P is
//                                  recomputed from the statistics of phase A (one strip instead of two keeps four episodes per CU)
template <int NH, bool kDrop>
__global__ __launch_bounds__(64) void attn_bwd_ep(const float* __restrict__ QKV, const float* __restrict__ dATT,
                                                  const int32_t* __restrict__ offsets, const int32_t* __restrict__ lens, int Lp,
                                                  float* __restrict__ dQKV, DropCfg dc, int layer,
                                                  const uint32_t* __restrict__ keep_in /* nullable: attn_fwd_ep's keep bits of the same rows and key */) {
    using G = EpGeo<NH>;
    constexpr int HPL = G::HPL, DPL = G::DPL, HD = G::HD, SL = G::SL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, len = lens[b];
    if (len <= 0) return;
    const int lane = threadIdx.x, hg = lane / SL, slot = lane - hg * SL;
    const int base = offsets[b], STR = G::strip(Lp);
    float* sQKV = smem;                                  // [len][96]
    float* sdA = smem + (size_t)Lp * 96;                 // [len][32]
    float* sS = sdA + (size_t)Lp * 32;                   // [head group][query][HPL][Lp] (query stride STR): scores -> e -> dS
    float* sM = sS + (size_t)G::NW * (Lp + 1) * STR;     // [head group][query][HPL]: row max
    float* sI = sM + (size_t)NH * Lp;                    //                           1 / sum
    uint32_t* sK = reinterpret_cast<uint32_t*>(sI + (size_t)NH * Lp);   // dropout keep bits [query][head][2]
    CIRS_BWG(layer == 1, 0);
    CIRS_BSTAMP(39);
    // the first 32 rows' Q | K | V and dATT rows: all requested at once (16 float4 per lane), committed behind the keep bits -- which come from the forward
    // pass's launch when it left them (a 1 KB load instead of ~8 Philox blocks per lane), else are rebuilt under the round trip
    const int len32 = min(len, 32);
    f32x4 st_q[12], st_d[4];
    ep_stage_issue<96, 12>(st_q, QKV + (size_t)base * 96, len32, 96, lane);
    ep_stage_issue<tD, 4>(st_d, dATT + (size_t)base * tD, len32, tD, lane);
    CIRS_BSTAMP(38);
    if (kDrop) {
        if (keep_in) {
            for (int i = lane; i < len * NH * 2; i += 64) sK[i] = keep_in[(size_t)base * NH * 2 + i];
        } else {
            ep_keep_build<NH>(sK, len, b, layer, dc, lane, 64);
        }
    }
    ep_stage_commit<96, 12>(sQKV, st_q, len32, lane);
    ep_stage_commit<tD, 4>(sdA, st_d, len32, lane);
    if (len > 32) {
        ep_stage(sQKV + 32 * 96, QKV + (size_t)(base + 32) * 96, len - 32, 96, 96, lane, 64);
        ep_stage(sdA + 32 * tD, dATT + (size_t)(base + 32) * tD, len - 32, tD, tD, lane, 64);
    }
    __syncthreads();
    CIRS_BSTAMP(40);
    const float scale = 1.0f / sqrtf((float)HD), qs = scale * 1.4426950408889634f;
    const float dinv = kDrop ? dc.inv : 1.0f;
    // ---------------- phase A: lane = queries pA, pB ----------------
    for (int s0 = 0; 2 * s0 < len; s0 += SL) {
        const int pA = s0 + slot, pB = len - 1 - pA;
        const bool actA = pA <= pB, actB = pA < pB;
        const int rA = actA ? pA : 0, rB = actB ? pB : 0;
        const int jn = len - s0;
        // branch-free loops: every lane stores every step; lanes without a query write the dummy row, columns beyond a query's causal limit are never read
        float* myA = sS + ((size_t)hg * (Lp + 1) + (actA ? pA : Lp)) * STR;
        float* myB = sS + ((size_t)hg * (Lp + 1) + (actB ? pB : Lp)) * STR;
        float qA[DPL], qB[DPL], daA[DPL], daB[DPL];
        ep_load(qA, sQKV + (size_t)rA * 96 + hg * DPL);
        ep_load(qB, sQKV + (size_t)rB * 96 + hg * DPL);
        ep_load(daA, sdA + (size_t)rA * tD + hg * DPL);
        ep_load(daB, sdA + (size_t)rB * tD + hg * DPL);
#pragma unroll
        for (int d = 0; d < DPL; ++d) { qA[d] *= qs; qB[d] *= qs; }
        float mxA[HPL], mxB[HPL], smA[HPL], smB[HPL], dotA[HPL], dotB[HPL];
#pragma unroll
        for (int h = 0; h < HPL; ++h) { mxA[h] = mxB[h] = -INFINITY; smA[h] = smB[h] = 0.f; dotA[h] = dotB[h] = 0.f; }
        uint32_t kmA[HPL][2], kmB[HPL][2];
        if (kDrop) {
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                kmA[h][0] = sK[(rA * NH + hg * HPL + h) * 2]; kmA[h][1] = sK[(rA * NH + hg * HPL + h) * 2 + 1];
                kmB[h][0] = sK[(rB * NH + hg * HPL + h) * 2]; kmB[h][1] = sK[(rB * NH + hg * HPL + h) * 2 + 1];
            }
        }
        float kq[DPL], vq[DPL];
        ep_load(kq, sQKV + 32 + hg * DPL);
#pragma unroll 4
        for (int j = 0; j < jn; ++j) {
            float k[DPL];      // software pipeline: row j + 1 is requested before row j is used (the strip stores would otherwise order the loads)
#pragma unroll
            for (int d = 0; d < DPL; ++d) k[d] = kq[d];
            ep_load(kq, sQKV + (size_t)(j + 1 < jn ? j + 1 : j) * 96 + 32 + hg * DPL);
            const bool inA = actA && j <= pA, inB = actB && j <= pB;
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                const float sa = ep_dot<HD>(qA + h * HD, k + h * HD), sb = ep_dot<HD>(qB + h * HD, k + h * HD);
                myA[h * Lp + j] = sa; myB[h * Lp + j] = sb;
                mxA[h] = inA ? fmaxf(mxA[h], sa) : mxA[h];
                mxB[h] = inB ? fmaxf(mxB[h], sb) : mxB[h];
            }
        }
        CIRS_BSTAMP(45);
        ep_load(vq, sQKV + 64 + hg * DPL);
#pragma unroll 4
        for (int j = 0; j < jn; ++j) {
            float v[DPL];
#pragma unroll
            for (int d = 0; d < DPL; ++d) v[d] = vq[d];
            ep_load(vq, sQKV + (size_t)(j + 1 < jn ? j + 1 : j) * 96 + 64 + hg * DPL);
            const bool inA = actA && j <= pA, inB = actB && j <= pB;
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                float ea = ep_exp2(myA[h * Lp + j] - mxA[h]), eb = ep_exp2(myB[h * Lp + j] - mxB[h]);
                ea = inA ? ea : 0.f; eb = inB ? eb : 0.f;
                myA[h * Lp + j] = ea; myB[h * Lp + j] = eb;
                smA[h] += ea; smB[h] += eb;
                float dpa = ep_dot<HD>(daA + h * HD, v + h * HD), dpb = ep_dot<HD>(daB + h * HD, v + h * HD);
                if (kDrop) {
                    dpa = (inA && ep_keep_bit(kmA[h], j)) ? dpa * dinv : 0.f;
                    dpb = (inB && ep_keep_bit(kmB[h], j)) ? dpb * dinv : 0.f;
                }
                dotA[h] = __builtin_fmaf(ea, dpa, dotA[h]);       // sum_j e dP; normalised below
                dotB[h] = __builtin_fmaf(eb, dpb, dotB[h]);
            }
        }
        CIRS_BSTAMP(46);
        float invA[HPL], invB[HPL];
#pragma unroll
        for (int h = 0; h < HPL; ++h) {
            invA[h] = 1.0f / smA[h]; invB[h] = 1.0f / smB[h];
            dotA[h] *= invA[h]; dotB[h] *= invB[h];
            if (actA) { sM[((size_t)hg * Lp + pA) * HPL + h] = mxA[h]; sI[((size_t)hg * Lp + pA) * HPL + h] = invA[h]; }
            if (actB) { sM[((size_t)hg * Lp + pB) * HPL + h] = mxB[h]; sI[((size_t)hg * Lp + pB) * HPL + h] = invB[h]; }
        }
        float dqA[DPL], dqB[DPL];
#pragma unroll
        for (int d = 0; d < DPL; ++d) dqA[d] = dqB[d] = 0.f;
        ep_load(kq, sQKV + 32 + hg * DPL);
        ep_load(vq, sQKV + 64 + hg * DPL);
#pragma unroll 4
        for (int j = 0; j < jn; ++j) {
            float k[DPL], v[DPL];
#pragma unroll
            for (int d = 0; d < DPL; ++d) { k[d] = kq[d]; v[d] = vq[d]; }
            ep_load(kq, sQKV + (size_t)(j + 1 < jn ? j + 1 : j) * 96 + 32 + hg * DPL);
            ep_load(vq, sQKV + (size_t)(j + 1 < jn ? j + 1 : j) * 96 + 64 + hg * DPL);
            const bool inA = actA && j <= pA, inB = actB && j <= pB;
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                float dpa = ep_dot<HD>(daA + h * HD, v + h * HD), dpb = ep_dot<HD>(daB + h * HD, v + h * HD);
                if (kDrop) {
                    dpa = (inA && ep_keep_bit(kmA[h], j)) ? dpa * dinv : 0.f;
                    dpb = (inB && ep_keep_bit(kmB[h], j)) ? dpb * dinv : 0.f;
                }
                float dsa = myA[h * Lp + j] * invA[h] * (dpa - dotA[h]), dsb = myB[h * Lp + j] * invB[h] * (dpb - dotB[h]);
                dsa = inA ? dsa : 0.f; dsb = inB ? dsb : 0.f;
                myA[h * Lp + j] = dsa; myB[h * Lp + j] = dsb;
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dqA[h * HD + d] = __builtin_fmaf(dsa, k[h * HD + d], dqA[h * HD + d]);
                    dqB[h * HD + d] = __builtin_fmaf(dsb, k[h * HD + d], dqB[h * HD + d]);
                }
            }
        }
        if (actA) ep_store(dQKV + (size_t)(base + pA) * 96 + hg * DPL, dqA, scale);
        if (actB) ep_store(dQKV + (size_t)(base + pB) * 96 + hg * DPL, dqB, scale);
    }
    __syncthreads();
    CIRS_BSTAMP(41);
    // ---------------- phase B: lane = keys jA, jB ----------------
    for (int s0 = 0; 2 * s0 < len; s0 += SL) {
        const int jA = s0 + slot, jB = len - 1 - jA;
        const bool actA = jA <= jB, actB = jA < jB;
        const int rA = actA ? jA : 0, rB = actB ? jB : 0;
        float kA[DPL], kB[DPL], dKA[DPL], dVA[DPL], dKB[DPL], dVB[DPL];
        ep_load(kA, sQKV + (size_t)rA * 96 + 32 + hg * DPL);
        ep_load(kB, sQKV + (size_t)rB * 96 + 32 + hg * DPL);
#pragma unroll
        for (int d = 0; d < DPL; ++d) { kA[d] *= qs; kB[d] *= qs; dKA[d] = dVA[d] = dKB[d] = dVB[d] = 0.f; }   // scores in log2 units, like phase A
        float qn[DPL], dn[DPL];
        ep_load(qn, sQKV + (size_t)s0 * 96 + hg * DPL);
        ep_load(dn, sdA + (size_t)s0 * tD + hg * DPL);
#pragma unroll 4
        for (int pq = s0; pq < len; ++pq) {          // the pass's earliest key is s0 (slot 0's key A)
            float q[DPL], da[DPL];
#pragma unroll
            for (int d = 0; d < DPL; ++d) { q[d] = qn[d]; da[d] = dn[d]; }
            ep_load(qn, sQKV + (size_t)(pq + 1 < len ? pq + 1 : pq) * 96 + hg * DPL);
            ep_load(dn, sdA + (size_t)(pq + 1 < len ? pq + 1 : pq) * tD + hg * DPL);
            const size_t st = ((size_t)hg * Lp + pq), ss = ((size_t)hg * (Lp + 1) + pq) * STR;
            const bool inA = actA && pq >= jA, inB = actB && pq >= jB;
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                const float m = sM[st * HPL + h], iv = sI[st * HPL + h];
                float pa = ep_exp2(fminf(ep_dot<HD>(q + h * HD, kA + h * HD) - m, 0.f)) * iv;    // (beyond the causal limit the difference may be > 0: discarded below)
                float pb = ep_exp2(fminf(ep_dot<HD>(q + h * HD, kB + h * HD) - m, 0.f)) * iv;
                pa = inA ? pa : 0.f; pb = inB ? pb : 0.f;
                if (kDrop) {
                    const uint32_t* kw = sK + (pq * NH + hg * HPL + h) * 2;       // (same address in the lanes of a head group: a broadcast read)
                    pa = (inA && ((kw[rA >> 5] >> (rA & 31)) & 1u)) ? pa * dinv : 0.f;
                    pb = (inB && ((kw[rB >> 5] >> (rB & 31)) & 1u)) ? pb * dinv : 0.f;
                }
                float dsa = sS[ss + h * Lp + rA], dsb = sS[ss + h * Lp + rB];
                dsa = inA ? dsa : 0.f; dsb = inB ? dsb : 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dKA[h * HD + d] = __builtin_fmaf(dsa, q[h * HD + d], dKA[h * HD + d]);
                    dVA[h * HD + d] = __builtin_fmaf(pa, da[h * HD + d], dVA[h * HD + d]);
                    dKB[h * HD + d] = __builtin_fmaf(dsb, q[h * HD + d], dKB[h * HD + d]);
                    dVB[h * HD + d] = __builtin_fmaf(pb, da[h * HD + d], dVB[h * HD + d]);
                }
            }
        }
        if (actA) {
            float* o = dQKV + (size_t)(base + jA) * 96 + 32 + hg * DPL;
            ep_store(o, dKA, scale); ep_store(o + 32, dVA, 1.0f);
        }
        if (actB) {
            float* o = dQKV + (size_t)(base + jB) * 96 + 32 + hg * DPL;
            ep_store(o, dKB, scale); ep_store(o + 32, dVB, 1.0f);
        }
    }
    CIRS_BSTAMP(42);
    CIRS_BWG(layer == 1, 1);
}
#undef EP_KEEP

// sum over the 32 lanes of a half-wave (one LayerNorm row per half-wave), result in every lane of the half
__device__ __forceinline__ float half_sum32(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, CIRS_WAVE);
    return v;
}

// LayerNorm(32) forward, one lane per element (a row = one half-wave, coalesced 128-byte rows):
// y = Y + Y2 (residual); out = (y - mean) * rstd * g + b ; xhat and rstd kept for the backward
__global__ __launch_bounds__(256) void ln_fwd(const float* __restrict__ Y, const float* __restrict__ Y2, const float* __restrict__ g,
                                              const float* __restrict__ b, int R, float* __restrict__ xhat, float* __restrict__ rstd,
                                              float* __restrict__ out, DropCfg dc, const int32_t* __restrict__ row_env,
                                              const int32_t* __restrict__ row_t, int layer, int site) {
    static_assert(tD == 32, "one half-wave per row");
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long r = i >> 5;
    const int d = (int)(i & 31);
    const bool ok = r < R;
    float y2 = ok ? Y2[i] : 0.f;
    if (dc.on && ok) y2 = drop_apply(dc, y2, row_env[r], row_t[r], layer, site, d);   // dropout1 / dropout2 on the branch
    const float y = ok ? Y[i] + y2 : 0.f;   // the pre-LayerNorm residual sum rides along
    const float mean = half_sum32(y) * (1.0f / tD);
    const float t = y - mean;
    const float var = half_sum32(t * t) * (1.0f / tD);
    const float rs = 1.0f / sqrtf(var + 1e-5f);
    if (!ok) return;
    if (d == 0) rstd[r] = rs;
    const float xh = t * rs;
    xhat[i] = xh;
    out[i] = xh * g[d] + b[d];
}
// ---- fused row chains of the forward recompute (round 3) ------------------------------------------------------------------------
// Everything between two attention stages is row-local: out_proj + residual + LayerNorm1 + lin1 (relu) + lin2 + residual + LayerNorm2 +
// the next layer's in_proj.  One wavefront owns a 32-row tile and walks the whole chain: the GEMMs on v_mfma_f32_32x32x2_f32 with the
// same operand order as rows_gemm_body (same bits), the LayerNorms in the "lane = (row, column half)" layout the next GEMM wants as
// its A operand (tile transposed through LDS), every activation the backward pass needs written on the way.  5 launches instead of 15.
constexpr int kRowT = 132;   // LDS tile row stride (floats): 128 columns + 4 -> conflict-free b128 row reads
// (acc_row(s, hi) = row of accumulator register s: bf16x6.h, through internal.h)
// acc += A (16 k of this lane's row, lane half hi owns k in [16 hi, 16 hi + 16) of the 32-k block) x W rows (wr = &W[n][kk + 16 hi])
__device__ __forceinline__ void mm_block(sg_f32x16& acc, const float (&a)[16], const float* __restrict__ wr) {
    float bv[16];
    ep_load(bv, wr);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void acc_to_lds(float* __restrict__ sT, const sg_f32x16& acc, int col, int hi) {
#pragma unroll
    for (int s = 0; s < 16; ++s) sT[acc_row(s, hi) * kRowT + col] = acc[s];
}
// LayerNorm of the lane's half row (16 values; the other half sits in lane ^ 32)
__device__ __forceinline__ void ln_half_row(const float (&y)[16], const float* __restrict__ g, const float* __restrict__ b, int hi,
                                            float (&xh)[16], float (&out)[16], float& rs) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += y[j];
    sum += __shfl_xor(sum, 32, CIRS_WAVE);
    const float mean = sum * (1.0f / tD);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float t = y[j] - mean; q = __builtin_fmaf(t, t, q); }
    q += __shfl_xor(q, 32, CIRS_WAVE);
    rs = 1.0f / sqrtf(q * (1.0f / tD) + 1e-5f);
    float gg[16], bb[16];
    ep_load(gg, g + 16 * hi);
    ep_load(bb, b + 16 * hi);
#pragma unroll
    for (int j = 0; j < 16; ++j) { xh[j] = (y[j] - mean) * rs; out[j] = xh[j] * gg[j] + bb[j]; }
}

struct LayerFwdArgs {
    const float *ATT, *H;                                   // [R, 32] inputs: attention output, layer input (residual)
    cirs_tracker_layer y;                                   // this layer's weights
    const float *win_next, *bin_next;                       // the next layer's in_proj (null after the last layer)
    float *XH1, *RS1, *H1N, *FF1, *XH2, *RS2, *Hout, *QKVnext;
    const int32_t *row_env, *row_t;
    int R, layer;
};
// acc += A x B with the B rows already in registers
__device__ __forceinline__ void mm_regs(sg_f32x16& acc, const float (&a)[16], const float (&bv)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void ln_apply(const float (&y)[16], const float (&gg)[16], const float (&bb)[16], float (&xh)[16], float (&out)[16], float& rs) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += y[j];
    sum += __shfl_xor(sum, 32, CIRS_WAVE);
    const float mean = sum * (1.0f / tD);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float t = y[j] - mean; q = __builtin_fmaf(t, t, q); }
    q += __shfl_xor(q, 32, CIRS_WAVE);
    rs = 1.0f / sqrtf(q * (1.0f / tD) + 1e-5f);
#pragma unroll
    for (int j = 0; j < 16; ++j) { xh[j] = (y[j] - mean) * rs; out[j] = xh[j] * gg[j] + bb[j]; }
}
// One wavefront per SIMD and a strict dependency chain: what the kernel waits for is memory latency, so every weight row is requested a
// stage ahead of its MFMAs (the registers are there: one wavefront per SIMD may use all 512).
template <bool kDrop>
__global__ __launch_bounds__(64) void layer_rows_fwd(LayerFwdArgs a, DropCfg dc) {
    // 40 KB although the tile needs 17: at most four of these one-wavefront workgroups per CU, i.e. one per SIMD (the dispatcher
    // otherwise stacks up to nine on a CU while other CUs idle, and the slowest CU is the kernel)
    __shared__ __attribute__((aligned(16))) float sT[10 * 1024];
    CIRS_BSTAMP(0);
    const int lane = threadIdx.x, hi = lane >> 5, lo = lane & 31;
    const int row0 = blockIdx.x * 32, row = row0 + lo;
    const bool row_ok = row < a.R;
    const size_t rr = (size_t)(row_ok ? row : 0);
    const int env = kDrop ? a.row_env[rr] : 0, pos = kDrop ? a.row_t[rr] : 0;
    // ---- requests of the first two stages ----------------------------------------------------------------------------------------
    float x[16], hrow[16], wo[16], w1[4][16], g1[16], be1[16];
    ep_load(x, a.ATT + rr * tD + 16 * hi);
    ep_load(wo, a.y.out_proj_w + (size_t)lo * tD + 16 * hi);
    ep_load(hrow, a.H + rr * tD + 16 * hi);
    ep_load(g1, a.y.norm1_w + 16 * hi);
    ep_load(be1, a.y.norm1_b + 16 * hi);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) ep_load(w1[nt], a.y.lin1_w + (size_t)(nt * 32 + lo) * tD + 16 * hi);
    const float bo = a.y.out_proj_b[lo], bl2 = a.y.lin2_b[lo];
    float bl1[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bl1[nt] = a.y.lin1_b[nt * 32 + lo];
    if (!row_ok) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = 0.f;
    }
    // ---- out_proj -------------------------------------------------------------------------------------------------------------
    sg_f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = bo;
    CIRS_BSTAMP(1);
    mm_regs(acc, x, wo);
    CIRS_BSTAMP(2);
    acc_to_lds(sT, acc, lo, hi);
    __syncthreads();
    // ---- residual + LayerNorm1 (lane = row, column half); lin2's weight rows requested meanwhile -----------------------------------
    float w2[4][16];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) ep_load(w2[kb], a.y.lin2_w + (size_t)lo * tH + kb * 32 + 16 * hi);
    float y[16], xh[16], h1n[16], rs;
    {
        float t[16];
        ep_load(t, sT + lo * kRowT + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = t[j];
            if (kDrop) v = drop_apply(dc, v, env, pos, a.layer, CIRS_DROP_RES1, 16 * hi + j);
            y[j] = row_ok ? hrow[j] + v : 0.f;
        }
    }
    CIRS_BSTAMP(3);
    ln_apply(y, g1, be1, xh, h1n, rs);
    CIRS_BSTAMP(4);
    if (row_ok) {
        ep_store(a.XH1 + rr * tD + 16 * hi, xh, 1.0f);
        ep_store(a.H1N + rr * tD + 16 * hi, h1n, 1.0f);
        if (hi == 0) a.RS1[rr] = rs;
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) h1n[j] = 0.f;
    }
    __syncthreads();
    CIRS_BSTAMP(5);
    // ---- lin1 + relu: 4 column tiles into LDS; FF1 (after the dropout) goes to global from the row layout lin2 reads it in ------------
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = nt * 32 + lo;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = bl1[nt];
        mm_regs(acc, h1n, w1[nt]);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = fmaxf(acc[s], 0.f);
        acc_to_lds(sT, acc, n, hi);
    }
    __syncthreads();
    CIRS_BSTAMP(6);
    // ---- lin2; LayerNorm2's parameters and the next in_proj's rows requested meanwhile ------------------------------------------------
    float g2[16], be2[16], wn[3][16], bn[3];
    ep_load(g2, a.y.norm2_w + 16 * hi);
    ep_load(be2, a.y.norm2_b + 16 * hi);
    if (a.win_next) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            ep_load(wn[nt], a.win_next + (size_t)(nt * 32 + lo) * tD + 16 * hi);
            bn[nt] = a.bin_next[nt * 32 + lo];
        }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = bl2;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        float f[16];
        ep_load(f, sT + lo * kRowT + kb * 32 + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) {      // the feed-forward dropout in the row layout (lane = row: one (env, position) per lane)
            if (kDrop) f[j] = drop_apply(dc, f[j], env, pos, a.layer, CIRS_DROP_FF, kb * 32 + 16 * hi + j);
            if (!row_ok) f[j] = 0.f;
        }
        if (row_ok) ep_store(a.FF1 + rr * tH + kb * 32 + 16 * hi, f, 1.0f);     // 16 b128 stores per lane instead of 64 dword stores
        mm_regs(acc, f, w2[kb]);
    }
    CIRS_BSTAMP(7);
    __syncthreads();
    acc_to_lds(sT, acc, lo, hi);
    __syncthreads();
    // ---- residual + LayerNorm2 ---------------------------------------------------------------------------------------------------
    float hn[16];
    {
        float t[16];
        ep_load(t, sT + lo * kRowT + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = t[j];
            if (kDrop) v = drop_apply(dc, v, env, pos, a.layer, CIRS_DROP_RES2, 16 * hi + j);
            y[j] = row_ok ? h1n[j] + v : 0.f;
        }
    }
    CIRS_BSTAMP(8);
    ln_apply(y, g2, be2, xh, hn, rs);
    CIRS_BSTAMP(9);
    if (row_ok) {
        ep_store(a.XH2 + rr * tD + 16 * hi, xh, 1.0f);
        ep_store(a.Hout + rr * tD + 16 * hi, hn, 1.0f);
        if (hi == 0) a.RS2[rr] = rs;
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) hn[j] = 0.f;
    }
    CIRS_BSTAMP(10);
    // ---- the next layer's in_proj ------------------------------------------------------------------------------------------------
    if (a.win_next) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int n = nt * 32 + lo;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[s] = bn[nt];
            mm_regs(acc, hn, wn[nt]);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int r = row0 + acc_row(s, hi);
                if (r < a.R) a.QKVnext[(size_t)r * 96 + n] = acc[s];
            }
        }
    }
    CIRS_BSTAMP(11);
}

// slot gather + scale + positional encoding (embed_rows) + the first layer's in_proj, one wavefront per 32 rows
template <bool kDrop>
__global__ __launch_bounds__(64) void embed_inproj(const float* __restrict__ x_hist, const float* __restrict__ pe,
                                                   const int32_t* __restrict__ row_env, const int32_t* __restrict__ row_t, int R, int L,
                                                   const float* __restrict__ win, const float* __restrict__ bin, float* __restrict__ H0,
                                                   float* __restrict__ QKV, DropCfg dc) {
    const int lane = threadIdx.x, hi = lane >> 5, lo = lane & 31;
    const int row0 = blockIdx.x * 32, row = row0 + lo;
    const bool row_ok = row < R;
    const size_t rr = (size_t)(row_ok ? row : 0);
    const int b = row_env[rr], p = row_t[rr];
    float x[16], pp[16], h[16];
    ep_load(x, x_hist + ((size_t)b * L + p) * tD + 16 * hi);
    ep_load(pp, pe + (size_t)p * tD + 16 * hi);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float v = x[j] * 5.656854249492381f + pp[j];
        if (kDrop) v = drop_apply(dc, v, b, p, 0, CIRS_DROP_POS, 16 * hi + j);
        h[j] = row_ok ? v : 0.f;
    }
    if (row_ok) ep_store(H0 + rr * tD + 16 * hi, h, 1.0f);
    sg_f32x16 acc;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int n = nt * 32 + lo;
        const float bias = bin[n];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = bias;
        mm_block(acc, h, win + (size_t)n * tD + 16 * hi);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int r = row0 + acc_row(s, hi);
            if (r < R) QKV[(size_t)r * 96 + n] = acc[s];
        }
    }
}

// ---- fused row chain of the backward pass (round 3) ------------------------------------------------------------------------------
// Per layer, between the attention backward of the layer above and its own:  [in_proj backward of the layer above] -> LayerNorm2
// backward -> lin2 backward (relu / dropout gate) -> lin1 backward + residual -> LayerNorm1 backward -> out_proj backward.  One wavefront
// per 32-row tile, GEMMs ("NN": dX = dY W, W[k][n] read coalesced over n) on v_mfma_f32_32x32x2_f32, LayerNorm backward in the
// "lane = (row, column half)" layout; every dY / X pair a weight-gradient problem needs is left in global memory and the problems
// of the layer run as ONE batched launch afterwards (dw_batch_kernel).
__device__ __forceinline__ void load_wcols(float (&bv)[16], const float* __restrict__ W, int ldw, int k0, int n) {   // bv[j] = W[k0 + j][n]
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = W[(size_t)(k0 + j) * ldw + n];
}
// LayerNorm backward of the lane's half row: dy = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)), dxh = dout * g
__device__ __forceinline__ void ln_bwd_half_row(const float (&dout)[16], const float (&xh)[16], const float (&gg)[16], float rs, float (&dy)[16]) {
    float s1 = 0.f, s2 = 0.f, dxh[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { dxh[j] = dout[j] * gg[j]; s1 += dxh[j]; s2 = __builtin_fmaf(dxh[j], xh[j], s2); }
    s1 += __shfl_xor(s1, 32, CIRS_WAVE);
    s2 += __shfl_xor(s2, 32, CIRS_WAVE);
    const float m1 = s1 * (1.0f / tD), m2 = s2 * (1.0f / tD);
#pragma unroll
    for (int j = 0; j < 16; ++j) dy[j] = rs * (dxh[j] - m1 - xh[j] * m2);
}
struct LayerBwdArgs {
    // pre stage = in_proj backward of the layer above: dh = dY1p + dQKVp * Winp (written to dHout); without it dh = dHin
    const float *dHin, *dY1p, *dQKVp, *winp;
    float* dHout;
    int only_pre;                      // layer 0's in_proj backward: stop after the pre stage (positional-encoding dropout applied)
    cirs_tracker_layer y;
    const float *XH2, *RS2, *FF1, *XH1, *RS1;
    float *dB2, *dFF1, *dH1N, *dY1, *dB1, *dATT;
    const int32_t *row_env, *row_t;
    int R, layer;
};
template <int LDW> __device__ __forceinline__ void load_wcols_c(float (&bv)[16], const float* __restrict__ W, int k0, int n) {   // bv[j] = W[k0 + j][n]
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = W[(k0 + j) * LDW + n];
}
template <bool kDrop>
__global__ __launch_bounds__(64) void layer_rows_bwd(LayerBwdArgs a, DropCfg dc) {
    __shared__ __attribute__((aligned(16))) float sT[10 * 1024];   // 40 KB: at most one of these workgroups per SIMD (see layer_rows_fwd)
    CIRS_BSTAMP(20);
    const int lane = threadIdx.x, hi = lane >> 5, lo = lane & 31;
    const int row0 = blockIdx.x * 32, row = row0 + lo;
    const bool row_ok = row < a.R;
    const size_t rr = (size_t)(row_ok ? row : 0);
    const int env = kDrop ? a.row_env[rr] : 0, pos = kDrop ? a.row_t[rr] : 0;
    const float dinv = kDrop ? dc.inv : 1.0f;
    sg_f32x16 acc;
    float dh[16];
    if (a.dQKVp) {
        float q[3][16], res[16], bv[3][16];
        ep_load(res, a.dY1p + rr * tD + 16 * hi);
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            ep_load(q[kb], a.dQKVp + rr * 96 + kb * 32 + 16 * hi);
            load_wcols_c<tD>(bv[kb], a.winp, kb * 32 + 16 * hi, lo);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            if (!row_ok) {
#pragma unroll
                for (int j = 0; j < 16; ++j) q[kb][j] = 0.f;
            }
            mm_regs(acc, q[kb], bv[kb]);
        }
        acc_to_lds(sT, acc, lo, hi);
        __syncthreads();
        float t[16];
        ep_load(t, sT + lo * kRowT + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            dh[j] = row_ok ? res[j] + t[j] : 0.f;
            if (kDrop && a.only_pre) dh[j] = drop_apply(dc, dh[j], env, pos, 0, CIRS_DROP_POS, 16 * hi + j);
        }
        if (row_ok) ep_store(a.dHout + rr * tD + 16 * hi, dh, 1.0f);
        if (a.only_pre) return;
        __syncthreads();
    } else {
        ep_load(dh, a.dHin + rr * tD + 16 * hi);
        if (!row_ok) {
#pragma unroll
            for (int j = 0; j < 16; ++j) dh[j] = 0.f;
        }
    }
    CIRS_BSTAMP(21);
    // ---- requests: everything the LayerNorms and the lin2 stage read (unconditional loads from a clamped row: they batch) -------------
    float xh2[16], g2[16], xh1[16], g1[16], w2[4][16], ffr[4][16];
    ep_load(xh2, a.XH2 + rr * tD + 16 * hi);
    ep_load(g2, a.y.norm2_w + 16 * hi);
    const float rs2 = a.RS2[rr], rs1 = a.RS1[rr];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) load_wcols_c<tH>(w2[nt], a.y.lin2_w, 16 * hi, nt * 32 + lo);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) ep_load(ffr[kb], a.FF1 + rr * tH + kb * 32 + 16 * hi);    // the relu / dropout gate, row layout
    ep_load(xh1, a.XH1 + rr * tD + 16 * hi);
    ep_load(g1, a.y.norm1_w + 16 * hi);
    // ---- LayerNorm2 backward -------------------------------------------------------------------------------------------------------
    float dy2[16], db2[16];
    ln_bwd_half_row(dh, xh2, g2, rs2, dy2);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (!row_ok) dy2[j] = 0.f;
        db2[j] = kDrop ? drop_apply(dc, dy2[j], env, pos, a.layer, CIRS_DROP_RES2, 16 * hi + j) : dy2[j];
    }
    if (row_ok) ep_store(a.dB2 + rr * tD + 16 * hi, db2, 1.0f);
    CIRS_BSTAMP(22);
    // ---- lin2 backward: dB2 W2 into LDS (4 column tiles), gated by relu / the feed-forward dropout in the row layout -----------------------
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = 0.f;
        mm_regs(acc, db2, w2[nt]);
        acc_to_lds(sT, acc, nt * 32 + lo, hi);
    }
    float w1[4][16], wo[16];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) load_wcols_c<tD>(w1[kb], a.y.lin1_w, kb * 32 + 16 * hi, lo);
    load_wcols_c<tD>(wo, a.y.out_proj_w, 16 * hi, lo);
    __syncthreads();
    CIRS_BSTAMP(23);
    float f[4][16];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        ep_load(f[kb], sT + lo * kRowT + kb * 32 + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) f[kb][j] = (row_ok && ffr[kb][j] > 0.f) ? f[kb][j] * dinv : 0.f;
        if (row_ok) ep_store(a.dFF1 + rr * tH + kb * 32 + 16 * hi, f[kb], 1.0f);
    }
    CIRS_BSTAMP(24);
    // ---- lin1 backward + residual ----------------------------------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) mm_regs(acc, f[kb], w1[kb]);
    __syncthreads();
    acc_to_lds(sT, acc, lo, hi);
    __syncthreads();
    CIRS_BSTAMP(25);
    float dh1n[16];
    {
        float t[16];
        ep_load(t, sT + lo * kRowT + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) dh1n[j] = row_ok ? dy2[j] + t[j] : 0.f;
        if (row_ok) ep_store(a.dH1N + rr * tD + 16 * hi, dh1n, 1.0f);
    }
    CIRS_BSTAMP(26);
    // ---- LayerNorm1 backward -------------------------------------------------------------------------------------------------------
    float dy1[16], db1[16];
    ln_bwd_half_row(dh1n, xh1, g1, rs1, dy1);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (!row_ok) dy1[j] = 0.f;
        db1[j] = kDrop ? drop_apply(dc, dy1[j], env, pos, a.layer, CIRS_DROP_RES1, 16 * hi + j) : dy1[j];
    }
    if (row_ok) {
        ep_store(a.dY1 + rr * tD + 16 * hi, dy1, 1.0f);
        if (kDrop) ep_store(a.dB1 + rr * tD + 16 * hi, db1, 1.0f);
    }
    CIRS_BSTAMP(27);
    // ---- out_proj backward (dATT leaves in the row layout, through LDS) -------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.f;
    mm_regs(acc, db1, wo);
    __syncthreads();
    acc_to_lds(sT, acc, lo, hi);
    __syncthreads();
    {
        float t[16];
        ep_load(t, sT + lo * kRowT + 16 * hi);
        if (row_ok) ep_store(a.dATT + rr * tD + 16 * hi, t, 1.0f);
    }
    CIRS_BSTAMP(28);
}

// every weight-gradient problem of a layer in one launch: grid = (sum of the problems' 32 x 32 output tiles, row slabs), the slab
// partials go to each problem's own region (summed by dw_list_final at the end of the pass, like every other problem)
constexpr int kMaxDwBatch = 8;
struct DwBatchJob { const float* dY; int ldy; const float* X; int ldx; int O, K, tile_begin; float* partial; };
struct DwBatch { DwBatchJob j[kMaxDwBatch]; int n, total_tiles; };
__global__ __launch_bounds__(64) void dw_batch_kernel(DwBatch jobs, int R, int rows_per_slab) {
    int ji = 0;
    for (; ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].tile_begin; ++ji) {}
    const DwBatchJob& jb = jobs.j[ji];
    dw_gemm_body((int)blockIdx.x - jb.tile_begin, blockIdx.y, jb.dY, jb.ldy, jb.X, jb.ldx, R, jb.O, jb.K, rows_per_slab, jb.partial);
}
static inline void dw_batch_add(DwBatch& b, DwList& list, const float* dY, int ldy, const float* X, int ldx, int R, int O, int K, float* dW,
                                float* db, int diag, float* partial) {
    const int slabs = dwg_slabs(R);
    DwListJob& lj = list.j[list.n++];
    lj.O = O; lj.K = K; lj.part_off = list.part_floats; lj.diag = diag; lj.dW = dW; lj.db = db;
    DwBatchJob& jb = b.j[b.n++];
    jb.dY = dY; jb.ldy = ldy; jb.X = X; jb.ldx = ldx; jb.O = O; jb.K = K; jb.tile_begin = b.total_tiles; jb.partial = partial + lj.part_off;
    b.total_tiles += cdiv(O, 32) * cdiv(K, 32);
    list.part_floats += slabs * O * (K + 1);
    list.total_out += O * (K + 1);
}
// (R_rows < R: problems over the first R_rows rows only -- the last-row pass -- in the slab count of the R-row problems of the same pass, which is the one
// dw_list_final walks; slabs beyond the rows write zero partials)
static inline void dw_batch_launch(const DwBatch& b, int R, hipStream_t s, int R_rows = -1) {
    if (!b.n) return;
    if (R_rows < 0) R_rows = R;
    const int slabs = dwg_slabs(R);
    int rows_per_slab = (R_rows + slabs - 1) / slabs;
    rows_per_slab = (rows_per_slab + 15) & ~15;
    hipLaunchKernelGGL(dw_batch_kernel, dim3(b.total_tiles, slabs), dim3(64), 0, s, b, R_rows, rows_per_slab);
}

// LayerNorm backward, same mapping: dY = rstd * (dxh - mean(dxh) - xhat * mean(dxh*xhat)), dxh = dOut * g
// LayerNorm backward + the weight / bias gradient problem of the same LayerNorm (diag(dOut^T xhat), column sums of dOut: a dw_gemm
// problem) in ONE launch of 64-thread workgroups: the first n_ln of them are the row-wise part, the rest the dW slabs (independent of
// each other, same inputs; a launch of this size costs ~6 us whatever it computes)
__global__ __launch_bounds__(64) void ln_bwd_dw_kernel(const float* __restrict__ dOut, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                       const float* __restrict__ g, int R, float* __restrict__ dY, int n_ln, DwGemmArgs da, int dgx) {
    if ((int)blockIdx.x >= n_ln) {
        const int q = (int)blockIdx.x - n_ln;
        dw_gemm_body(q % dgx, q / dgx, da.dY, da.ldy, da.X, da.ldx, da.R, da.O, da.K, da.rows_per_slab, da.partial);
        return;
    }
    const long i = blockIdx.x * 64L + threadIdx.x;
    const long r = i >> 5;
    const int d = (int)(i & 31);
    const bool ok = r < R;
    const float dxh = ok ? dOut[i] * g[d] : 0.f;
    const float xh = ok ? xhat[i] : 0.f;
    const float m1 = half_sum32(dxh) * (1.0f / tD);
    const float m2 = half_sum32(dxh * xh) * (1.0f / tD);
    if (ok) dY[i] = rstd[r] * (dxh - m1 - xh * m2);
}

__global__ __launch_bounds__(256) void ln_bwd(const float* __restrict__ dOut, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                              const float* __restrict__ g, int R, float* __restrict__ dY) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long r = i >> 5;
    const int d = (int)(i & 31);
    const bool ok = r < R;
    const float dxh = ok ? dOut[i] * g[d] : 0.f;
    const float xh = ok ? xhat[i] : 0.f;
    const float m1 = half_sum32(dxh) * (1.0f / tD);
    const float m2 = half_sum32(dxh * xh) * (1.0f / tD);
    if (ok) dY[i] = rstd[r] * (dxh - m1 - xh * m2);
}
// (the LayerNorm weight gradient d gamma = diag(dOut^T xhat) and d beta = column sums of dOut fall out of a 32 x 32 dW
// problem with the diag flag: dw_list_final keeps the diagonal)

__global__ __launch_bounds__(256) void scale_rows(float* __restrict__ x, long n, float a) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) x[i] *= a;
}

// upstream gradient rows: G[r, s] = dstate[row_t, row_env, s]
__global__ __launch_bounds__(256) void gather_dstate(const float* __restrict__ dstate, const int32_t* __restrict__ row_env,
                                                     const int32_t* __restrict__ row_t, int R, int S, int B, float* __restrict__ G) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)R * S) return;
    const int r = (int)(i / S), s = (int)(i % S);
    G[i] = dstate[((size_t)row_t[r] * B + row_env[r]) * S + s];
}

// input slots: rows with p == 0 are the user slot, p >= 1 the gated action slot (state_tracker.py:205-215,225-242).
// Produces, per row, the "pre" gradient vectors the two small weight-gradient GEMMs need and scatters into the
// embedding tables:   user rows : DU[r] = dX0 (for ffn_user), EU[r] = Emb_user[u]
//                     item rows : DPRE[r] = dX0 * a * g(1-g), GIN[r] = [rew, a]  (for fnn_gate)
// value of lane k (a compile-time or wave-uniform k) in every lane: v_readlane_b32, not a ds_bpermute round trip
__device__ __forceinline__ float lane_bcast(float x, int k) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), k));
}

__global__ __launch_bounds__(256) void slot_bwd(cirs_tracker_weights w, const float* __restrict__ dH0, const int32_t* __restrict__ users,
                                                const int64_t* __restrict__ act, const double* __restrict__ rew,
                                                const int32_t* __restrict__ row_env, const int32_t* __restrict__ row_t, int R, int B,
                                                float* __restrict__ DU, float* __restrict__ EU, float* __restrict__ DPRE,
                                                float* __restrict__ GIN, float* __restrict__ CU, float* __restrict__ CI,
                                                int32_t* __restrict__ key_user, int32_t* __restrict__ key_item,
                                                float* __restrict__ contrib_c /* [B, 32] per-env user contribution (merged scatter) or null */,
                                                int rows_per_wave) {
    const int lane = threadIdx.x & 63;
    // the gate matrix [32][33] once per workgroup, coalesced, into LDS: lane d then walks ITS row with stride-33 reads (no bank
    // conflicts); from memory that walk is one dword per lane and load, 64 cache lines per instruction, 33 instructions per row
    __shared__ float sG[tD * (tD + 1)];
    for (int q = threadIdx.x; q < tD * (tD + 1); q += 256) sG[q] = w.gate_w[q];
    __syncthreads();
    // (rows_per_wave > 1: the call batches of the exact-redraw pass, ~0.5 M rows -- the 4 KB gate image per four rows was most of the kernel's traffic)
    for (int it_r = 0; it_r < rows_per_wave; ++it_r) {
    const int r = (blockIdx.x * rows_per_wave + it_r) * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int b = row_env[r], p = row_t[r];
    const int d = lane & 31;
    const float dx = dH0[(size_t)r * tD + d] * 5.656854249492381f;  // d/dx of x*sqrt(D)
    if (p == 0) {
        const int u = users[b];
        const float eu = w.emb_user[(size_t)u * tD + d];
        if (lane < tD) {
            DU[(size_t)r * tD + d] = dx; EU[(size_t)r * tD + d] = eu;
            DPRE[(size_t)r * tD + d] = 0.f;
        }
        if (lane <= tD) GIN[(size_t)r * (tD + 1) + lane] = 0.f;
        // dEmb_user[u, k] += sum_o dx[o] * ffn_user_w[o, k] : lane k
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < tD; ++o) acc = __builtin_fmaf(lane_bcast(dx, o), w.ffn_user_w[(size_t)o * tD + d], acc);
        if (lane < tD) {
            CU[(size_t)r * tD + d] = acc; CI[(size_t)r * tD + d] = 0.f;
            if (contrib_c) contrib_c[(size_t)b * tD + d] = acc;
        }
        if (lane == 0) { key_user[r] = u; key_item[r] = -1; }
    } else {
        const size_t ti = (size_t)(p - 1) * B + b;
        const long it = act[ti];
        const float rw = (float)rew[ti];
        const float a = w.emb_item[(size_t)it * tD + d];
        // recompute the gate for feature d (lane o = d)
        const float* gw = sG + d * (tD + 1);
        float pre = w.gate_b[d];
        pre = __builtin_fmaf(gw[0], rw, pre);
#pragma unroll
        for (int k = 0; k < tD; ++k) pre = __builtin_fmaf(gw[1 + k], lane_bcast(a, k), pre);
        const float g = 1.0f / (1.0f + expf(-pre));
        const float dpre = dx * a * g * (1.0f - g);
        if (lane < tD) {
            DPRE[(size_t)r * tD + d] = dpre;
            DU[(size_t)r * tD + d] = 0.f; EU[(size_t)r * tD + d] = 0.f;
            GIN[(size_t)r * (tD + 1) + 1 + d] = a;
        }
        if (lane == 0) GIN[(size_t)r * (tD + 1)] = rw;
        // d a[k] = dx[k]*g[k] + sum_o dpre[o] * gate_w[o, 1+k]
        float acc = dx * g;
#pragma unroll
        for (int o = 0; o < tD; ++o) acc = __builtin_fmaf(lane_bcast(dpre, o), sG[o * (tD + 1) + 1 + d], acc);
        if (lane < tD) { CI[(size_t)r * tD + d] = acc; CU[(size_t)r * tD + d] = 0.f; }
        if (lane == 0) { key_item[r] = (int32_t)it; key_user[r] = -1; }
    }
    }
}

// Deterministic embedding-gradient scatter without float atomics and without an O(rows x table) scan: the (key, row)
// pairs are sorted by key with a stable radix sort (rows stay ascending inside a key), then one 32-lane group per
// segment head adds the contribution rows of its key in row order.  Cost O(rows), independent of the table size
// (a 10^6-row catalogue costs the same as 10^4), result independent of scheduling.
__global__ __launch_bounds__(256) void scatter_keys_kernel(const int32_t* __restrict__ keys, int R, int n_table,
                                                           uint32_t* __restrict__ keys_u, int32_t* __restrict__ rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int32_t k = keys[r];
    keys_u[r] = (k < 0 || k >= n_table) ? (uint32_t)n_table : (uint32_t)k;  // rows without a contribution sort last
    rows[r] = r;
}

// Only the first row of an episode (the user slot) contributes to Emb_user: one (key, contribution) pair per env instead of one
// per buffer row -> the sort of the user-embedding scatter handles n_env pairs.  Slot = env id, so pairs of one user stay in env
// (= row) order.
__global__ __launch_bounds__(256) void compact_user_rows(const int32_t* __restrict__ row_env, const int32_t* __restrict__ row_t,
                                                         const int32_t* __restrict__ users, const float* __restrict__ CU, int R,
                                                         int32_t* __restrict__ keys_c, float* __restrict__ contrib_c) {
    const int d = threadIdx.x & 31;
    const int r = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (r >= R || row_t[r] != 0) return;
    const int b = row_env[r];
    contrib_c[(size_t)b * tD + d] = CU[(size_t)r * tD + d];
    if (d == 0) keys_c[b] = users[b];
}

// Ordered segment sums over the (key, row) pairs sorted by key, in two levels so that a very popular key (thousands of rows
// once the policy concentrates on few items) is not one sequential chain:
//   sub-runs   a sub-run starts at every segment head and at every multiple of 64 in the sorted order and ends at the next
//              such position: <= 64 rows, summed in row order by one half-wave (lane = embedding dimension) -> part[start]
//   segments   the head of a segment adds its sub-run partials in position order (every 64th position) -> g_emb[key]
// Fixed order, no atomics; rows of keys outside the table are skipped.
constexpr int kSubRun = 64;
// One half-wave (lane = embedding dimension) per sub-run start.  The keys and rows of the sub-run's 64-position block are read once
// (two per lane), the run length comes from a ballot, the row numbers travel between lanes by shuffle: every contribution row of the
// sub-run can then be requested without waiting for a key comparison (2 dependent round trips + one per 16 rows instead of two per 8).
__global__ __launch_bounds__(256) void emb_subrun_kernel(const uint32_t* __restrict__ ks, const int32_t* __restrict__ rs,
                                                         const float* __restrict__ contrib, int R, int n_table, float* __restrict__ part) {
    const int d = threadIdx.x & 31, half = (threadIdx.x >> 5) & 1;
    const int p = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool in = p < R;
    const uint32_t key = in ? ks[p] : 0xffffffffu;
    const bool start = in && key < (uint32_t)n_table && ((p % kSubRun) == 0 || ks[p - 1] != key);   // block start or segment head
    const int end = min(R, (p / kSubRun + 1) * kSubRun);
    // positions p + d and p + 32 + d of the block (the ballot needs every lane of the wavefront: no early return above)
    const int q0 = p + d, q1 = p + 32 + d;
    const bool s0 = start && q0 < end && ks[q0] == key, s1 = start && q1 < end && ks[q1] == key;
    const int r0 = s0 ? rs[q0] : 0, r1 = s1 ? rs[q1] : 0;
    const unsigned long long m0 = __ballot(s0), m1 = __ballot(s1);
    if (!start) return;
    const uint32_t h0 = (uint32_t)(m0 >> (32 * half)), h1 = (uint32_t)(m1 >> (32 * half));
    // run length: leading positions with the same key (the sort makes them contiguous)
    const int n = h0 == 0xffffffffu ? 32 + (h1 == 0xffffffffu ? 32 : __builtin_ctz(~h1)) : __builtin_ctz(~h0);
    float acc = 0.f;
    for (int q = 0; q < n; q += 16) {   // 16 rows in flight, added in row order
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int qq = q + u;
            const int row = __shfl(qq < 32 ? r0 : r1, (qq & 31) + 32 * half, CIRS_WAVE);
            t[u] = qq < n ? contrib[(size_t)row * tD + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (q + u < n) ? t[u] : 0.f;
    }
    part[(size_t)p * tD + d] = acc;
}

__global__ __launch_bounds__(256) void emb_segment_sum_kernel(const uint32_t* __restrict__ ks, const float* __restrict__ part, int R,
                                                              int n_table, float* __restrict__ g_emb, int n_first = 0x7fffffff,
                                                              float* __restrict__ g_second = nullptr) {   // keys >= n_first: rows of g_second
    const int d = threadIdx.x & 31;
    const int p = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (p >= R) return;
    const uint32_t key = ks[p];
    if (key >= (uint32_t)n_table || (p > 0 && ks[p - 1] == key)) return;  // not a segment head
    float acc = part[(size_t)p * tD + d];
    for (int q = (p / kSubRun + 1) * kSubRun; q < R; q += 8 * kSubRun) {   // the segment's later sub-runs, 8 loads in flight
        float t8[8];
        int n_ok = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int qq = q + u * kSubRun;
            const bool ok = (n_ok == u) && qq < R && ks[qq] == key;
            t8[u] = ok ? part[(size_t)qq * tD + d] : 0.f;
            n_ok += ok;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (u < n_ok) ? t8[u] : 0.f;
        if (n_ok < 8) break;
    }
    if (key < (uint32_t)n_first) g_emb[(size_t)key * tD + d] = acc;
    else g_second[(size_t)(key - (uint32_t)n_first) * tD + d] = acc;
}

static size_t emb_sort_bytes(long R) { return (size_t)R * (64 + 4 * tD) + (1u << 20); }   // keys/rows in+out, sub-run partials, sort temp

// keys [R] (< 0: no contribution), contrib [R, 32] -> g_emb [n_table, 32] (fully overwritten)
static int emb_scatter_sorted(const int32_t* keys, const float* contrib, int R, int n_table, float* g_emb, void* scratch,
                              size_t scratch_bytes, hipStream_t s) {
    static_assert(tD == 32, "emb_segment_sum_kernel maps one lane per embedding dimension");
    uint32_t* k_in = (uint32_t*)scratch;
    uint32_t* k_out = k_in + R;
    int32_t* r_in = (int32_t*)(k_out + R);
    int32_t* r_out = r_in + R;
    float* part = (float*)(((uintptr_t)(r_out + R) + 255) & ~(uintptr_t)255);   // [R, 32] sub-run partials
    char* temp = (char*)(((uintptr_t)(part + (size_t)R * tD) + 255) & ~(uintptr_t)255);
    const size_t avail = scratch_bytes - (size_t)(temp - (char*)scratch);
    int end_bit = 1;
    while ((1u << end_bit) <= (uint32_t)n_table && end_bit < 32) ++end_bit;
    size_t need = 0;
    CIRS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k_in, k_out, r_in, r_out, R, 0, end_bit, s));
    CIRS_REQUIRE(need <= avail, "embedding scatter: sort scratch too small");
    CIRS_HIP(hipMemsetAsync(g_emb, 0, (size_t)n_table * tD * sizeof(float), s));
    hipLaunchKernelGGL(scatter_keys_kernel, dim3(cdiv(R, 256)), dim3(256), 0, s, keys, R, n_table, k_in, r_in);
    CIRS_HIP(hipcub::DeviceRadixSort::SortPairs(temp, need, k_in, k_out, r_in, r_out, R, 0, end_bit, s));
    hipLaunchKernelGGL(emb_subrun_kernel, dim3(cdiv(R, 8)), dim3(256), 0, s, k_out, r_out, contrib, R, n_table, part);
    hipLaunchKernelGGL(emb_segment_sum_kernel, dim3(cdiv(R, 8)), dim3(256), 0, s, k_out, (const float*)part, R, n_table, g_emb);
    CIRS_CHECK_LAUNCH("emb_segment_sum_kernel");
    return CIRS_OK;
}

// Both embedding tables of the tracker in ONE sort: R item pairs (key = item id) followed by one user pair per env (key = n_items +
// user id, present when the env has rows in this call); contrib = [R + B, 32] (item rows, then the per-env user rows).  Same ordered
// sub-run / segment sums, half the launches of two separate scatters.
__global__ __launch_bounds__(256) void scatter_keys2_kernel(const int32_t* __restrict__ key_item, const int32_t* __restrict__ users,
                                                            const int32_t* __restrict__ lens, int R, int B, int n_items, int n_users,
                                                            uint32_t* __restrict__ keys_u, int32_t* __restrict__ rows,
                                                            float* __restrict__ zero_table /* nullable: the contiguous gradient tables, cleared here */, long zero_floats) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    // (the tables' clear rides in this launch -- emb_segment_sum_kernel, three launches on, writes the touched rows only; a memset is a launch of its own)
    if (zero_table)
        for (long i = (long)p * 4; i < zero_floats; i += (long)gridDim.x * blockDim.x * 4) *reinterpret_cast<f32x4*>(zero_table + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p >= R + B) return;
    const uint32_t sent = (uint32_t)n_items + (uint32_t)n_users;
    uint32_t k = sent;
    if (p < R) {
        const int32_t v = key_item[p];
        if (v >= 0 && v < n_items) k = (uint32_t)v;
    } else {
        const int b = p - R;
        const int32_t u = users[b];
        if (lens[b] > 0 && u >= 0 && u < n_users) k = (uint32_t)n_items + (uint32_t)u;
    }
    keys_u[p] = k;
    rows[p] = p;
}
static int emb_scatter_merged(const int32_t* key_item, const int32_t* users, const int32_t* lens, const float* contrib, int R, int B,
                              int n_items, int n_users, float* g_item, float* g_user, void* scratch, size_t scratch_bytes, hipStream_t s) {
    const int N = R + B, n_table = n_items + n_users;
    uint32_t* k_in = (uint32_t*)scratch;
    uint32_t* k_out = k_in + N;
    int32_t* r_in = (int32_t*)(k_out + N);
    int32_t* r_out = r_in + N;
    float* part = (float*)(((uintptr_t)(r_out + N) + 255) & ~(uintptr_t)255);
    char* temp = (char*)(((uintptr_t)(part + (size_t)N * tD) + 255) & ~(uintptr_t)255);
    const size_t avail = scratch_bytes - (size_t)(temp - (char*)scratch);
    int end_bit = 1;
    while ((1u << end_bit) <= (uint32_t)n_table && end_bit < 32) ++end_bit;
    size_t need = 0;
    CIRS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k_in, k_out, r_in, r_out, N, 0, end_bit, s));
    CIRS_REQUIRE(need <= avail, "embedding scatter: sort scratch too small");
    float* zero_table = nullptr;       // contiguous tables (the flat gradient buffer): cleared inside scatter_keys2_kernel (tD = 32 floats per row: float4 units)
    if (g_user + (size_t)n_users * tD == g_item) zero_table = g_user;
    else if (g_item + (size_t)n_items * tD == g_user) zero_table = g_item;
    if (zero_table && ((uintptr_t)zero_table & 15)) zero_table = nullptr;
    if (!zero_table) {
        CIRS_HIP(hipMemsetAsync(g_item, 0, (size_t)n_items * tD * sizeof(float), s));
        CIRS_HIP(hipMemsetAsync(g_user, 0, (size_t)n_users * tD * sizeof(float), s));
    }
    hipLaunchKernelGGL(scatter_keys2_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, key_item, users, lens, R, B, n_items, n_users, k_in, r_in, zero_table,
                       (long)n_table * tD);
    CIRS_HIP(hipcub::DeviceRadixSort::SortPairs(temp, need, k_in, k_out, r_in, r_out, N, 0, end_bit, s));
    hipLaunchKernelGGL(emb_subrun_kernel, dim3(cdiv(N, 8)), dim3(256), 0, s, k_out, r_out, contrib, N, n_table, part);
    hipLaunchKernelGGL(emb_segment_sum_kernel, dim3(cdiv(N, 8)), dim3(256), 0, s, k_out, (const float*)part, N, n_table, g_item, n_items, g_user);
    CIRS_CHECK_LAUNCH("emb_segment_sum_kernel (merged)");
    return CIRS_OK;
}

struct BwdScratch {
    float *G, *H[CIRS_MAX_TRACKER_LAYERS + 1];
    float *QKV[CIRS_MAX_TRACKER_LAYERS], *P[CIRS_MAX_TRACKER_LAYERS], *ATT[CIRS_MAX_TRACKER_LAYERS];
    float *XH1[CIRS_MAX_TRACKER_LAYERS], *RS1[CIRS_MAX_TRACKER_LAYERS], *H1N[CIRS_MAX_TRACKER_LAYERS], *FF1[CIRS_MAX_TRACKER_LAYERS];
    float *XH2[CIRS_MAX_TRACKER_LAYERS], *RS2[CIRS_MAX_TRACKER_LAYERS];
    float *T0, *T1, *T2, *T3, *T4, *T5, *dQKV, *dFF1, *dS, *partial, *GIN;
    float* LC;                            // last-row pass: per episode dY1 [32] | (max, 1 / sum) per head [16] | env | position
    float* PM[CIRS_MAX_TRACKER_LAYERS];   // dropout: attention probabilities after the mask
    void* sort;  // emb_sort_bytes(R)
};

constexpr int kLastFloats = tD + 16 + 2 + 2;     // BwdScratch::LC per episode (padded to a multiple of 4)

static size_t bwd_partial_floats(const cirs_tracker_cfg* cfg, long R) {
    size_t f = dw_list_floats(R, 32, tD) + dw_list_floats(R, tD, tD) + dw_list_floats(R, tD, tD + 1);   // decoder (S <= 32), ffn_user, gate
    f += (size_t)cfg->nlayers * (2 * dw_list_floats(R, tD, tD) + dw_list_floats(R, tD, tH) + dw_list_floats(R, tH, tD) +
                                 dw_list_floats(R, tD, tD) + dw_list_floats(R, 96, tD));              // LN2, LN1, lin2, lin1, out_proj, in_proj
    return f;
}

static size_t bwd_floats(const cirs_tracker_cfg* cfg, long R) {
    const long nl = cfg->nlayers, Lp = cfg->max_len, NH = cfg->nhead;
    size_t f = 0;
    f += (size_t)R * 32;                          // G (S <= 32)
    f += (size_t)(nl + 1) * R * tD;               // H
    f += (size_t)nl * R * (96 + NH * Lp + tD + tD + 1 + tD + tH + tD + 1);
    f += 6 * (size_t)R * tD + (size_t)R * 96 + (size_t)R * tH + (size_t)R * NH * Lp + 64;  // T0..5, dQKV, dFF1, dS
    if (cfg->dropout_p > 0.f) f += (size_t)nl * R * NH * Lp + 64;                      // PM
    f += bwd_partial_floats(cfg, R) + 4096;       // slab partials of every dW problem of the pass (one final launch)
    f += (size_t)R * (tD + 1);                    // GIN
    f += (size_t)R * kLastFloats;                 // LC
    f += emb_sort_bytes(R + cfg->n_env) / 4 + 64;  // (key, row) sort of the embedding scatter (item rows + one user pair per env)
    return f + 64 * 32;
}

static BwdScratch carve_bwd(void* ws, const cirs_tracker_cfg* cfg, long R) {
    float* p = (float*)ws;
    auto take = [&](size_t n) { float* r = p; p += (n + 3) & ~(size_t)3; return r; };
    const long nl = cfg->nlayers, Lp = cfg->max_len, NH = cfg->nhead;
    BwdScratch s;
    s.G = take((size_t)R * 32);
    for (int l = 0; l <= nl; ++l) s.H[l] = take((size_t)R * tD);
    for (int l = 0; l < nl; ++l) {
        s.QKV[l] = take((size_t)R * 96); s.P[l] = take((size_t)R * NH * Lp); s.ATT[l] = take((size_t)R * tD);
        s.XH1[l] = take((size_t)R * tD); s.RS1[l] = take(R); s.H1N[l] = take((size_t)R * tD); s.FF1[l] = take((size_t)R * tH);
        s.XH2[l] = take((size_t)R * tD); s.RS2[l] = take(R);
    }
    s.T0 = take((size_t)R * tD); s.T1 = take((size_t)R * tD); s.T2 = take((size_t)R * tD); s.T3 = take((size_t)R * tD);
    s.T4 = take((size_t)R * tD); s.T5 = take((size_t)R * tD);
    for (int l = 0; l < nl; ++l) s.PM[l] = cfg->dropout_p > 0.f ? take((size_t)R * NH * Lp) : nullptr;
    s.dQKV = take((size_t)R * 96); s.dFF1 = take((size_t)R * tH); s.dS = take((size_t)R * NH * Lp);
    s.partial = take(bwd_partial_floats(cfg, R) + 4096);
    s.GIN = take((size_t)R * (tD + 1));
    s.LC = take((size_t)R * kLastFloats);
    s.sort = (void*)take(emb_sort_bytes(R + cfg->n_env) / 4 + 64);
    return s;
}

}  // namespace cirs

namespace cirs {
static inline void launch_ln_bwd_dw(DwList& list, const float* dOut, const float* xhat, const float* rstd, const float* g, int R, float* dY,
                                    float* dW, float* db, float* partial, hipStream_t s) {
    const int slabs = dwg_slabs(R);
    int rows_per_slab = (R + slabs - 1) / slabs;
    rows_per_slab = (rows_per_slab + 15) & ~15;
    DwListJob& jb = list.j[list.n++];
    jb.O = tD; jb.K = tD; jb.part_off = list.part_floats; jb.diag = 1; jb.dW = dW; jb.db = db;
    list.part_floats += slabs * tD * (tD + 1);
    list.total_out += tD * (tD + 1);
    const DwGemmArgs da{dOut, tD, xhat, tD, R, tD, tD, rows_per_slab, partial + jb.part_off};
    const int n_ln = (int)cdiv((long)R * tD, 64L);
    hipLaunchKernelGGL(ln_bwd_dw_kernel, dim3(n_ln + slabs), dim3(64), 0, s, dOut, xhat, rstd, g, R, dY, n_ln, da, 1);
}
}  // namespace cirs

#ifdef CIRS_TBWD_PROF
extern "C" int cirs_debug_tbwd_wg(unsigned long long* out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(cirs::g_tbwd_wg), 2048 * 3 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
extern "C" int cirs_debug_tbwd_prof(unsigned long long* out_host64) {
    return hipMemcpyFromSymbol(out_host64, HIP_SYMBOL(cirs::g_tbwd_prof), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

namespace cirs {
// cirs_tracker_prefix_states only needs the LAST row of every env out of the last layer: everything behind that layer's attention is row-local, so its row
// chain (out_proj .. LayerNorm2: 192 MFMAs per 32 rows) runs on one row per env instead of on every row of the prefix -- at call t of a collect B rows instead
// of B (t + 1).  This kernel gathers those rows' attention outputs / layer inputs and their (env, position) (the dropout masks are keyed by them).
__global__ __launch_bounds__(256) void prefix_last_rows_kernel(const float* __restrict__ ATT, const float* __restrict__ H, const int32_t* __restrict__ row_env,
                                                               const int32_t* __restrict__ row_t, const int32_t* __restrict__ offsets,
                                                               const int32_t* __restrict__ lens, int B, float* __restrict__ ATTc, float* __restrict__ Hc,
                                                               int32_t* __restrict__ env_c, int32_t* __restrict__ t_c) {
    const int e = blockIdx.x * 8 + (threadIdx.x >> 5), c = threadIdx.x & 31;
    if (e >= B) return;
    const int n = lens[e];
    const long r = n > 0 ? (long)offsets[e] + n - 1 : -1;
    ATTc[(size_t)e * tD + c] = r >= 0 ? ATT[r * tD + c] : 0.f;
    Hc[(size_t)e * tD + c] = r >= 0 ? H[r * tD + c] : 0.f;
    if (c == 0) { env_c[e] = r >= 0 ? row_env[r] : e; t_c[e] = r >= 0 ? row_t[r] : 0; }
}
// cirs_tracker_prefix_states: state of the LAST row of every env that has rows = decoder(H_last[row]); one wavefront per env, lane j < S owns
// output j (fma chain over the 32 features in ascending order)
// compact != 0: H holds one row per env (the last layer ran on the envs' last rows only, below)
__global__ __launch_bounds__(256) void prefix_decoder_kernel(const float* __restrict__ H, const int32_t* __restrict__ offsets, const int32_t* __restrict__ lens,
                                                             int B, int S, const float* __restrict__ dec_w, const float* __restrict__ dec_b,
                                                             float* __restrict__ out, long out_stride, int compact) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
    if (e >= B || j >= S) return;
    const int n = lens[e];
    if (n <= 0) return;
    const float* h = H + (size_t)(compact ? e : offsets[e] + n - 1) * tD;
    float acc = dec_b[j];
#pragma unroll
    for (int k = 0; k < tD; ++k) acc = __builtin_fmaf(h[k], dec_w[(size_t)j * tD + k], acc);
    out[(size_t)e * out_stride + j] = acc;
}

// ---- the whole prefix pass of cirs_tracker_prefix_states from ONE launch, one wavefront per env (round 6) ----------------------------------------
// An env's prefix is at most 32 rows = one MFMA row tile, so one wavefront can walk the pass on its own: slot gather + in_proj (embed_inproj), every
// layer but the last on all rows (attn_fwd_ep's loops over the Q|K|V tile, now in LDS; layer_rows_fwd's chain, whose next in_proj refills that tile),
// the last layer for the last row only (its attention for the one query; its chain on the tile, of which only that row counts) and the decoder.
// Every stage keeps the stand-alone kernel's arithmetic -- operand order, fma chains, reduction order -- so the states are BIT-identical to the
// multi-launch pass (tests/test_gpu_dropout.py); nothing is kept for a backward pass.  Why: at one call of a collect the multi-launch pass is nine
// launches at their latency floors (~110 us for ~0.5 us of arithmetic per CU).
constexpr int kPQS = 104;   // Q|K|V tile row stride (floats): 4 rows = 32 banks apart (the two lane halves of an accumulator store)
constexpr int kPAS = 36;    // attention-output tile row stride
struct PrefixEnvArgs {
    const float *x_hist, *pe;
    const int32_t *row_env, *row_t, *offsets, *lens;
    cirs_tracker_layer layer[CIRS_MAX_TRACKER_LAYERS];
    const float *dec_w, *dec_b;
    float* state_out;
    long state_stride;
    int nl, L, S;
    TrunkFuse tf;        // on: the policy trunk of the state in the same wavefront (trunk_kernel's row: same fma chains; finished envs get its zeros)
};
template <int NH> __host__ __device__ constexpr size_t prefix_env_scratch_floats() {     // strips of the attention / the chain's 32 x kRowT tile (never live together)
    return (size_t)EpGeo<NH>::NW * 33 * EpGeo<NH>::strip(32) > (size_t)32 * kRowT ? (size_t)EpGeo<NH>::NW * 33 * EpGeo<NH>::strip(32) : (size_t)32 * kRowT;
}
constexpr int kRowBitWords = 6;   // keep bits of one row's chain: 48 Philox blocks (RES1 8 | FF 32 | RES2 8) x 4 bits
template <int NH> __host__ __device__ constexpr size_t prefix_env_lds_floats() {
    return (size_t)32 * kPQS + (size_t)32 * kPAS + prefix_env_scratch_floats<NH>() + EpGeo<NH>::keep_words(32) + 32 * kRowBitWords;
}

#define EP_KEEP(POS, ELEM) dropout_keep(dc.seed, (uint32_t)(dc.env_base + b), (uint32_t)(POS), (uint32_t)layer, (uint32_t)CIRS_DROP_ATTN, (uint32_t)(ELEM), dc.thr)
// attn_fwd_ep on the tile: every query of the episode (same loops; q / K / V come from the LDS tile, the output rows go to sA)
template <int NH, bool kDrop>
__device__ __forceinline__ void prefix_attn_all(const float* __restrict__ sQ, float* __restrict__ sS, float* __restrict__ sA, uint32_t* __restrict__ sK, int len,
                                                int b, int layer, const DropCfg& dc, int lane) {
    using G = EpGeo<NH>;
    constexpr int HPL = G::HPL, DPL = G::DPL, HD = G::HD, SL = G::SL, Lp = 32;
    const int hg = lane / SL, slot = lane - hg * SL, STR = G::strip(Lp);
    if (kDrop) ep_keep_build<NH>(sK, len, b, layer, dc, lane, 64);
    const float scale = 1.0f / sqrtf((float)HD), qs = scale * 1.4426950408889634f;
    for (int s0 = 0; 2 * s0 < len; s0 += SL) {
        const int pA = s0 + slot, pB = len - 1 - pA;
        const bool actA = pA <= pB, actB = pA < pB;
        const int rA = actA ? pA : 0, rB = actB ? pB : 0;
        const int jn = len - s0;
        float* myA = sS + ((size_t)hg * (Lp + 1) + (actA ? pA : Lp)) * STR;
        float* myB = sS + ((size_t)hg * (Lp + 1) + (actB ? pB : Lp)) * STR;
        float qA[DPL], qB[DPL];
        ep_load(qA, sQ + rA * kPQS + hg * DPL);
        ep_load(qB, sQ + rB * kPQS + hg * DPL);
#pragma unroll
        for (int d = 0; d < DPL; ++d) { qA[d] *= qs; qB[d] *= qs; }
        float mxA[HPL], mxB[HPL], smA[HPL], smB[HPL];
#pragma unroll
        for (int h = 0; h < HPL; ++h) { mxA[h] = mxB[h] = -INFINITY; smA[h] = smB[h] = 0.f; }
        float kq[DPL];
        ep_load(kq, sQ + 32 + hg * DPL);
#pragma unroll 4
        for (int j = 0; j < jn; ++j) {
            float k[DPL];
#pragma unroll
            for (int d = 0; d < DPL; ++d) k[d] = kq[d];
            ep_load(kq, sQ + (j + 1 < jn ? j + 1 : j) * kPQS + 32 + hg * DPL);
            const bool inA = actA && j <= pA, inB = actB && j <= pB;
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                const float sa = ep_dot<HD>(qA + h * HD, k + h * HD), sb = ep_dot<HD>(qB + h * HD, k + h * HD);
                myA[h * Lp + j] = sa; myB[h * Lp + j] = sb;
                mxA[h] = inA ? fmaxf(mxA[h], sa) : mxA[h];
                mxB[h] = inB ? fmaxf(mxB[h], sb) : mxB[h];
            }
        }
        float accA[DPL], accB[DPL];
#pragma unroll
        for (int d = 0; d < DPL; ++d) accA[d] = accB[d] = 0.f;
        uint32_t kmA[HPL][2], kmB[HPL][2];
        if (kDrop) {
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                kmA[h][0] = sK[(rA * NH + hg * HPL + h) * 2]; kmA[h][1] = sK[(rA * NH + hg * HPL + h) * 2 + 1];
                kmB[h][0] = sK[(rB * NH + hg * HPL + h) * 2]; kmB[h][1] = sK[(rB * NH + hg * HPL + h) * 2 + 1];
            }
        }
        float vq[DPL];
        ep_load(vq, sQ + 64 + hg * DPL);
#pragma unroll 4
        for (int j = 0; j < jn; ++j) {
            float v[DPL];
#pragma unroll
            for (int d = 0; d < DPL; ++d) v[d] = vq[d];
            ep_load(vq, sQ + (j + 1 < jn ? j + 1 : j) * kPQS + 64 + hg * DPL);
            const bool inA = actA && j <= pA, inB = actB && j <= pB;
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                float ea = ep_exp2(myA[h * Lp + j] - mxA[h]), eb = ep_exp2(myB[h * Lp + j] - mxB[h]);
                ea = inA ? ea : 0.f; eb = inB ? eb : 0.f;
                smA[h] += ea; smB[h] += eb;
                if (kDrop) {
                    if (!ep_keep_bit(kmA[h], j)) ea = 0.f;
                    if (!ep_keep_bit(kmB[h], j)) eb = 0.f;
                }
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    accA[h * HD + d] = __builtin_fmaf(ea, v[h * HD + d], accA[h * HD + d]);
                    accB[h * HD + d] = __builtin_fmaf(eb, v[h * HD + d], accB[h * HD + d]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < HPL; ++h) {
            const float ia = (kDrop ? dc.inv : 1.0f) / smA[h], ib = (kDrop ? dc.inv : 1.0f) / smB[h];
#pragma unroll
            for (int d = 0; d < HD; ++d) { accA[h * HD + d] *= ia; accB[h * HD + d] *= ib; }
        }
        if (actA) ep_store(sA + pA * kPAS + hg * DPL, accA, 1.0f);
        if (actB) ep_store(sA + pB * kPAS + hg * DPL, accB, 1.0f);
    }
}
// the same attention for the LAST query alone (what the last layer needs): scores in parallel over the keys, then lane = (head, dim) walks the keys in
// order -- the sums attn_fwd_ep forms for that query, in its order
template <int NH, bool kDrop>
__device__ __forceinline__ void prefix_attn_last(const float* __restrict__ sQ, float* __restrict__ sS, float* __restrict__ sA, uint32_t* __restrict__ sK, int len,
                                                 int b, int layer, const DropCfg& dc, int lane) {
    constexpr int HD = tD / NH;
    const int p = len - 1;
    if (kDrop) ep_keep_build_row<NH>(sK, p, b, layer, dc, lane, 64);
    const float scale = 1.0f / sqrtf((float)HD), qs = scale * 1.4426950408889634f;
    {
        const int j = lane & 31;
        for (int h = lane >> 5; h < NH; h += 2) {
            float q[HD], k[HD];
            ep_load(q, sQ + p * kPQS + h * HD);
            ep_load(k, sQ + j * kPQS + 32 + h * HD);
#pragma unroll
            for (int d = 0; d < HD; ++d) q[d] *= qs;
            sS[h * 32 + j] = ep_dot<HD>(q, k);
        }
    }
    __syncthreads();
    if (lane < 32) {
        const int h = lane / HD;
        float sc[32];
        ep_load(sc, sS + h * 32);
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 32; ++j) mx = j <= p ? fmaxf(mx, sc[j]) : mx;
        float sm = 0.f, acc = 0.f;
        const uint32_t km = kDrop ? sK[h * 2] : 0u;      // (len <= 32 here: one word)
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const bool in = j <= p;
            float e = ep_exp2(sc[j] - mx);
            e = in ? e : 0.f;
            sm += e;
            if (kDrop) {
                if (!((km >> j) & 1u)) e = 0.f;
            }
            acc = __builtin_fmaf(e, sQ[j * kPQS + 64 + lane], acc);
        }
        acc *= (kDrop ? dc.inv : 1.0f) / sm;
        sA[p * kPAS + lane] = acc;
    }
    __syncthreads();
}
#undef EP_KEEP
// The chain's dropout masks (sites RES1 / FF / RES2 of one layer) for the rows that COUNT, built across the wave: a row needs 48 Philox blocks (~500 cycles each)
// and in the chain below every lane evaluates the 24 of its half row -- also the lanes of rows beyond the prefix, and in the last layer, of which only the last
// row is read, all but two lanes.  Here the rows' blocks are dealt to the 64 lanes (n_rows * 48 / 64 each: 1 for the last layer, ~12 on average for the others
// instead of 24) and left as keep bits in LDS: sB[slot * 6 + w], block idx = 8 w + nibble (RES1 0..7 | FF 8..39 | RES2 40..47), bit = word of the block.
// Same counters, same masks (rng.h).  slot = row - r_begin; pos_of_lane = the position of the row a lane holds (lane lo = row lo).
__device__ __forceinline__ void prefix_row_bits(uint32_t* __restrict__ sB, int r_begin, int n_rows, int env, int pos_of_lane, int layer, const DropCfg& dc, int lane) {
    for (int i = lane; i < 32 * kRowBitWords; i += 64) sB[i] = 0u;
    __syncthreads();
    for (int i0 = 0; i0 < n_rows * 48; i0 += 64) {
        const int i = i0 + lane, rl = i / 48, idx = i - rl * 48;
        const int pos = __shfl(pos_of_lane, (r_begin + rl) & 31, CIRS_WAVE);
        if (i < n_rows * 48) {
            const int site = idx < 8 ? (int)CIRS_DROP_RES1 : (idx < 40 ? (int)CIRS_DROP_FF : (int)CIRS_DROP_RES2);
            const int g = idx < 8 ? idx : (idx < 40 ? idx - 8 : idx - 40);
            const u32x4 r = dropout_block(dc.seed, (uint32_t)(dc.env_base + env), (uint32_t)pos, (uint32_t)layer, (uint32_t)site, (uint32_t)g);
            uint32_t m = 0u;
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) m |= block_word(r, (uint32_t)wd) >= dc.thr ? 1u << wd : 0u;
            atomicOr(&sB[rl * kRowBitWords + (idx >> 3)], m << (4 * (idx & 7)));
        }
    }
    __syncthreads();
}
// layer_rows_fwd's chain on the tile: x = attention output (sA), hrow = layer input (registers, lane = (row, column half)) -> hrow = layer output; the
// next layer's Q|K|V into sQ when win_next is set.  myB (dropout): the six keep-bit words of this lane's row (prefix_row_bits)
template <bool kDrop>
__device__ __forceinline__ void prefix_chain(const cirs_tracker_layer& y, const float* __restrict__ win_next, const float* __restrict__ bin_next, int layer,
                                             const DropCfg& dc, const uint32_t* __restrict__ myB, bool row_ok, int lo, int hi, const float* __restrict__ sA,
                                             float* __restrict__ sT, float* __restrict__ sQ, float (&hrow)[16]) {
    uint32_t kb1 = 0u, kbf[4] = {0u, 0u, 0u, 0u}, kb2 = 0u;      // bit j = keep of element 16 hi + j of the site (FF: of hidden block kb)
    if (kDrop) {
        kb1 = (myB[0] >> (16 * hi)) & 0xFFFFu;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) kbf[kb] = (myB[1 + kb] >> (16 * hi)) & 0xFFFFu;
        kb2 = (myB[5] >> (16 * hi)) & 0xFFFFu;
    }
    float x[16], wo[16], w1[4][16], g1[16], be1[16];
    ep_load(wo, y.out_proj_w + (size_t)lo * tD + 16 * hi);
    ep_load(g1, y.norm1_w + 16 * hi);
    ep_load(be1, y.norm1_b + 16 * hi);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) ep_load(w1[nt], y.lin1_w + (size_t)(nt * 32 + lo) * tD + 16 * hi);
    const float bo = y.out_proj_b[lo], bl2 = y.lin2_b[lo];
    float bl1[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bl1[nt] = y.lin1_b[nt * 32 + lo];
    ep_load(x, sA + lo * kPAS + 16 * hi);
    if (!row_ok) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = 0.f;
    }
    sg_f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = bo;
    mm_regs(acc, x, wo);
    acc_to_lds(sT, acc, lo, hi);
    __syncthreads();
    float w2[4][16];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) ep_load(w2[kb], y.lin2_w + (size_t)lo * tH + kb * 32 + 16 * hi);
    float yv[16], xh[16], h1n[16], rs;
    {
        float t[16];
        ep_load(t, sT + lo * kRowT + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = t[j];
            if (kDrop) v = ((kb1 >> j) & 1u) ? v * dc.inv : 0.f;
            yv[j] = row_ok ? hrow[j] + v : 0.f;
        }
    }
    ln_apply(yv, g1, be1, xh, h1n, rs);
    if (!row_ok) {
#pragma unroll
        for (int j = 0; j < 16; ++j) h1n[j] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = bl1[nt];
        mm_regs(acc, h1n, w1[nt]);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = fmaxf(acc[s], 0.f);
        acc_to_lds(sT, acc, nt * 32 + lo, hi);
    }
    __syncthreads();
    float g2[16], be2[16], wn[3][16], bn[3];
    ep_load(g2, y.norm2_w + 16 * hi);
    ep_load(be2, y.norm2_b + 16 * hi);
    if (win_next) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            ep_load(wn[nt], win_next + (size_t)(nt * 32 + lo) * tD + 16 * hi);
            bn[nt] = bin_next[nt * 32 + lo];
        }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = bl2;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        float f[16];
        ep_load(f, sT + lo * kRowT + kb * 32 + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (kDrop) f[j] = ((kbf[kb] >> j) & 1u) ? f[j] * dc.inv : 0.f;
            if (!row_ok) f[j] = 0.f;
        }
        mm_regs(acc, f, w2[kb]);
    }
    __syncthreads();
    acc_to_lds(sT, acc, lo, hi);
    __syncthreads();
    {
        float t[16];
        ep_load(t, sT + lo * kRowT + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = t[j];
            if (kDrop) v = ((kb2 >> j) & 1u) ? v * dc.inv : 0.f;
            yv[j] = row_ok ? h1n[j] + v : 0.f;
        }
    }
    ln_apply(yv, g2, be2, xh, hrow, rs);
    if (!row_ok) {
#pragma unroll
        for (int j = 0; j < 16; ++j) hrow[j] = 0.f;
    }
    if (win_next) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[s] = bn[nt];
            mm_regs(acc, hrow, wn[nt]);
#pragma unroll
            for (int s = 0; s < 16; ++s) sQ[acc_row(s, hi) * kPQS + nt * 32 + lo] = acc[s];
        }
    }
    __syncthreads();
}
template <int NH, bool kDrop>
__global__ __launch_bounds__(64) void prefix_env_kernel(PrefixEnvArgs a, DropCfg dc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sQ = smem;                          // [32][Q 32 | K 32 | V 32] (row stride kPQS)
    float* sA = sQ + 32 * kPQS;                // [32][32] attention output (row stride kPAS)
    float* sX = sA + 32 * kPAS;                // attention strips | the chain's tile
    uint32_t* sK = reinterpret_cast<uint32_t*>(sX + prefix_env_scratch_floats<NH>());   // dropout keep bits of the layer's attention
    uint32_t* sB = sK + EpGeo<NH>::keep_words(32);                                       // ... of the layer's row chain (prefix_row_bits)
    const int e = blockIdx.x, len = a.lens[e];
    if (len <= 0) return;
    CIRS_BSTAMP(40);
    const int lane = threadIdx.x, hi = lane >> 5, lo = lane & 31;
    const int base = a.offsets[e];
    const bool row_ok = lo < len;
    const size_t rr = (size_t)base + (row_ok ? lo : 0);
    const int b = a.row_env[rr], p = a.row_t[rr];
    // ---- slot gather + scale + positional encoding + the first in_proj (embed_inproj) ----------------------------------------------------------
    float h[16];
    {
        float x[16], pp[16];
        ep_load(x, a.x_hist + ((size_t)b * a.L + p) * tD + 16 * hi);
        ep_load(pp, a.pe + (size_t)p * tD + 16 * hi);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = x[j] * 5.656854249492381f + pp[j];
            if (kDrop) v = drop_apply(dc, v, b, p, 0, CIRS_DROP_POS, 16 * hi + j);
            h[j] = row_ok ? v : 0.f;
        }
        sg_f32x16 acc;
        const float* win = a.layer[0].in_proj_w;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int n = nt * 32 + lo;
            const float bias = a.layer[0].in_proj_b[n];
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[s] = bias;
            mm_block(acc, h, win + (size_t)n * tD + 16 * hi);
#pragma unroll
            for (int s = 0; s < 16; ++s) sQ[acc_row(s, hi) * kPQS + n] = acc[s];
        }
    }
    __syncthreads();
    CIRS_BSTAMP(41);
    for (int l = 0; l < a.nl; ++l) {
        const bool last = l + 1 == a.nl;
        if (last) prefix_attn_last<NH, kDrop>(sQ, sX, sA, sK, len, e, l, dc, lane);
        else {
            prefix_attn_all<NH, kDrop>(sQ, sX, sA, sK, len, e, l, dc, lane);
            __syncthreads();
        }
        CIRS_BSTAMP(42 + 2 * l);
        // the chain's masks: of the prefix's rows -- in the last layer of its last row alone (the only one the decoder reads; every lane then applies that
        // row's bits: the other rows' values are never used)
        // (measured at C3, us per pass: 1 / 16 / 30 rows 37.6 -> 32.0 / 41.8 -> 38.6 / 47.4 -> 47.4 -- at 30 rows the blocks per lane are what they were and no longer
        // run underneath the chain's MFMAs; choosing per call between the bits and the in-place masks made every case slower: 44 / 50 / 60 us)
        if (kDrop) prefix_row_bits(sB, last ? len - 1 : 0, last ? 1 : len, b, p, l, dc, lane);
        prefix_chain<kDrop>(a.layer[l], last ? nullptr : a.layer[l + 1].in_proj_w, last ? nullptr : a.layer[l + 1].in_proj_b, l, dc,
                            sB + (last ? 0 : lo) * kRowBitWords, row_ok, lo, hi, sA, sX, sQ, h);
        CIRS_BSTAMP(43 + 2 * l);
    }
    // ---- decoder on the last row (prefix_decoder_kernel) -----------------------------------------------------------------------------------------
    if (lo == len - 1) ep_store(sA + 16 * hi, h, 1.0f);
    __syncthreads();
    float sval = 0.f;
    if (lane < a.S) {
        float acc = a.dec_b[lane];
#pragma unroll
        for (int k = 0; k < tD; ++k) acc = __builtin_fmaf(sA[k], a.dec_w[(size_t)lane * tD + k], acc);
        a.state_out[(size_t)e * a.state_stride + lane] = acc;
        sval = acc;
    }
    CIRS_BSTAMP(50);
    if (a.tf.on) {
        if (a.tf.skip && a.tf.skip[e]) {
            a.tf.h2[(size_t)e * kH + lane] = 0.f;
            if (lane == 0 && a.tf.value) a.tf.value[e] = 0.f;
            return;
        }
        float *xs = sX, *hs = sX + kH;
        if (lane < a.S) xs[lane] = sval;
        trunk_compute(a.tf.cfg, a.tf.w, xs, hs, lane, e, a.tf.h2, a.tf.value, nullptr);
    }
}

// ---- a pass whose upstream gradient sits on the LAST row of every episode (cirs_tracker_backward_last: the exact-redraw procedure) -------------------
// build_state call c only hands s_c = decoder(top layer, last row) to the policy (core/state_tracker.py:243-246), so in the graph of a call the top layer
// matters at ONE row per episode: its attention has one query (O(len) instead of O(len^2) per episode), its row chain, five of its six weight-gradient
// problems and the decoder's see E = n_env rows instead of R = sum of the lengths (~ E T / 2), and its attention backward is a rank-one update of
// dK / dV.  Below the top layer every row carries a gradient (through the top layer's keys and values) and the pass is the ordinary one.
//   attn_last_fwd : lane = (episode, head); the last query against keys 0 .. len-1 straight from the QKV rows (a head's 32 B per row; the NH lanes of an
//                   episode read a row's 128 B together), same loops and operand order as attn_fwd_ep's query (same bits); gathers the row's layer
//                   input and (env, position) for the row chain; keeps (max, 1 / sum) per (episode, head) for the backward
//   attn_last_bwd : the same walk twice (dot = sum_j P dP, then dS): dQ on the last row, dK_j = dS_j q, dV_j = PM_j dATT for every row; it also lays
//                   the top layer's residual gradient (non-zero on the last rows only) out over all rows for the chain of the layer below
template <int NH>
__device__ __forceinline__ void last_keep_build(uint32_t* __restrict__ sK, const int32_t* __restrict__ lens, int e0, int E, int Lp, int layer,
                                                const DropCfg& dc, int tid) {
    constexpr int EPW = 64 / NH;
    for (int i = tid; i < EPW * NH * 2; i += 64) sK[i] = 0u;
    __syncthreads();
    const int gmax = (Lp * NH + 3) >> 2;        // Philox blocks of the longest query (four (key, head) elements per block)
    for (int i = tid; i < EPW * gmax; i += 64) {
        const int el = i / gmax, g = i - el * gmax, e = e0 + el;
        const int len = e < E ? lens[e] : 0;
        if (4 * g < len * NH) {
            const int p = len - 1;
            const u32x4 r = dropout_block(dc.seed, (uint32_t)(dc.env_base + e), (uint32_t)p, (uint32_t)layer, (uint32_t)CIRS_DROP_ATTN, (uint32_t)g);
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                const int elem = 4 * g + wd, j = elem / NH, h = elem - j * NH;
                if (j <= p && block_word(r, (uint32_t)wd) >= dc.thr) atomicOr(&sK[(el * NH + h) * 2 + (j >> 5)], 1u << (j & 31));
            }
        }
    }
    __syncthreads();
}
template <int NH, bool kDrop>
__global__ __launch_bounds__(64) void attn_last_fwd(const float* __restrict__ QKV, const float* __restrict__ H, const int32_t* __restrict__ row_env,
                                                    const int32_t* __restrict__ row_t, const int32_t* __restrict__ offsets,
                                                    const int32_t* __restrict__ lens, int E, int Lp, float* __restrict__ ATTc, float* __restrict__ Hc,
                                                    float* __restrict__ stats, int32_t* __restrict__ env_c, int32_t* __restrict__ t_c, DropCfg dc, int layer) {
    constexpr int HD = tD / NH, EPW = 64 / NH;
    __shared__ uint32_t sK[EPW * NH * 2];
    const int tid = threadIdx.x, el = tid / NH, h = tid - el * NH, e0 = blockIdx.x * EPW, e = e0 + el;
    if (kDrop) last_keep_build<NH>(sK, lens, e0, E, Lp, layer, dc, tid);
    if (e >= E) return;
    const int len = lens[e];
    float acc[HD], hrow[HD];
    if (len <= 0) {      // an episode without rows: a zero row through the chain (its upstream gradient is zeroed as well: it contributes nothing)
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = 0.f;
        ep_store(ATTc + (size_t)e * tD + h * HD, acc, 1.0f);
        ep_store(Hc + (size_t)e * tD + h * HD, acc, 1.0f);
        stats[((size_t)e * NH + h) * 2] = 0.f; stats[((size_t)e * NH + h) * 2 + 1] = 0.f;
        if (h == 0) { env_c[e] = e; t_c[e] = 0; }
        return;
    }
    const int p = len - 1;
    const size_t base = (size_t)offsets[e];
    const float* rows = QKV + base * 96 + h * HD;
    const float scale = 1.0f / sqrtf((float)HD), qs = scale * 1.4426950408889634f;
    float q[HD];
    ep_load(q, rows + (size_t)p * 96);
    ep_load(hrow, H + (base + p) * tD + h * HD);
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] *= qs;
    float mx = -INFINITY;
#pragma unroll 4
    for (int j = 0; j <= p; ++j) {
        float k[HD];
        ep_load(k, rows + (size_t)j * 96 + 32);
        mx = fmaxf(mx, ep_dot<HD>(q, k));
    }
    uint32_t km[2] = {0u, 0u};
    if (kDrop) { km[0] = sK[(el * NH + h) * 2]; km[1] = sK[(el * NH + h) * 2 + 1]; }
    float sm = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll 4
    for (int j = 0; j <= p; ++j) {
        float k[HD], v[HD];
        ep_load(k, rows + (size_t)j * 96 + 32);
        ep_load(v, rows + (size_t)j * 96 + 64);
        float ex = ep_exp2(ep_dot<HD>(q, k) - mx);
        sm += ex;
        if (kDrop) {
            if (!ep_keep_bit(km, j)) ex = 0.f;
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = __builtin_fmaf(ex, v[d], acc[d]);
    }
    const float ia = (kDrop ? dc.inv : 1.0f) / sm;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] *= ia;
    ep_store(ATTc + (size_t)e * tD + h * HD, acc, 1.0f);
    ep_store(Hc + (size_t)e * tD + h * HD, hrow, 1.0f);
    stats[((size_t)e * NH + h) * 2] = mx; stats[((size_t)e * NH + h) * 2 + 1] = 1.0f / sm;
    if (h == 0) { env_c[e] = row_env[base + p]; t_c[e] = row_t[base + p]; }
}
template <int NH, bool kDrop>
__global__ __launch_bounds__(64) void attn_last_bwd(const float* __restrict__ QKV, const float* __restrict__ dATTc, const float* __restrict__ dY1c,
                                                    const float* __restrict__ stats, const int32_t* __restrict__ offsets, const int32_t* __restrict__ lens,
                                                    int E, int Lp, float* __restrict__ dQKV, float* __restrict__ dY1, DropCfg dc, int layer) {
    constexpr int HD = tD / NH, EPW = 64 / NH;
    __shared__ uint32_t sK[EPW * NH * 2];
    const int tid = threadIdx.x, el = tid / NH, h = tid - el * NH, e0 = blockIdx.x * EPW, e = e0 + el;
    if (kDrop) last_keep_build<NH>(sK, lens, e0, E, Lp, layer, dc, tid);
    if (e >= E) return;
    const int len = lens[e];
    if (len <= 0) return;
    const int p = len - 1;
    const size_t base = (size_t)offsets[e];
    const float* rows = QKV + base * 96 + h * HD;
    const float scale = 1.0f / sqrtf((float)HD), qs = scale * 1.4426950408889634f, dinv = kDrop ? dc.inv : 1.0f;
    float qr[HD], q[HD], da[HD], res[HD], zero[HD];
    ep_load(qr, rows + (size_t)p * 96);
    ep_load(da, dATTc + (size_t)e * tD + h * HD);
    ep_load(res, dY1c + (size_t)e * tD + h * HD);
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = qr[d] * qs; zero[d] = 0.f; }
    const float mx = stats[((size_t)e * NH + h) * 2], inv = stats[((size_t)e * NH + h) * 2 + 1];
    uint32_t km[2] = {0u, 0u};
    if (kDrop) { km[0] = sK[(el * NH + h) * 2]; km[1] = sK[(el * NH + h) * 2 + 1]; }
    float dot = 0.f;
#pragma unroll 4
    for (int j = 0; j <= p; ++j) {
        float k[HD], v[HD];
        ep_load(k, rows + (size_t)j * 96 + 32);
        ep_load(v, rows + (size_t)j * 96 + 64);
        const float ex = ep_exp2(ep_dot<HD>(q, k) - mx);
        float dp = ep_dot<HD>(da, v);
        if (kDrop) dp = ep_keep_bit(km, j) ? dp * dinv : 0.f;
        dot = __builtin_fmaf(ex, dp, dot);
    }
    dot *= inv;
    float dq[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
    float* orow = dQKV + base * 96 + h * HD;
    float* yrow = dY1 + base * tD + h * HD;
#pragma unroll 2
    for (int j = 0; j <= p; ++j) {
        float k[HD], v[HD];
        ep_load(k, rows + (size_t)j * 96 + 32);
        ep_load(v, rows + (size_t)j * 96 + 64);
        const float pj = ep_exp2(ep_dot<HD>(q, k) - mx) * inv;
        float dp = ep_dot<HD>(da, v), pm = pj;
        if (kDrop) {
            const bool keep = ep_keep_bit(km, j);
            dp = keep ? dp * dinv : 0.f;
            pm = keep ? pj * dinv : 0.f;
        }
        const float ds = pj * (dp - dot);
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] = __builtin_fmaf(ds, k[d], dq[d]);
        ep_store(orow + (size_t)j * 96 + 32, qr, ds * scale);
        ep_store(orow + (size_t)j * 96 + 64, da, pm);
        if (j < p) {
            ep_store(orow + (size_t)j * 96, zero, 1.0f);
            ep_store(yrow + (size_t)j * tD, zero, 1.0f);
        }
    }
    ep_store(orow + (size_t)p * 96, dq, scale);
    ep_store(yrow + (size_t)p * tD, res, 1.0f);
}
// upstream gradient of the last rows: G[e] = dstate_last[e] (zero for an episode without rows)
__global__ __launch_bounds__(256) void dstate_last_rows(const float* __restrict__ dstate_last, const int32_t* __restrict__ lens, int E, int S,
                                                        float* __restrict__ G) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)E * S) return;
    G[i] = lens[i / S] > 0 ? dstate_last[i] : 0.f;
}
// the same upstream gradient laid out over ALL rows (the general pass as cirs_tracker_backward_last's fallback): row (env b, position lens[b] - 1) gets
// dstate_last[b], every other row zero
__global__ __launch_bounds__(256) void dstate_last_full(const float* __restrict__ dstate_last, const int32_t* __restrict__ row_env,
                                                        const int32_t* __restrict__ offsets, const int32_t* __restrict__ lens, int R, int S,
                                                        float* __restrict__ G) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)R * S) return;
    const int r = (int)(i / S), s = (int)(i % S), b = row_env[r];
    G[i] = r == offsets[b] + lens[b] - 1 ? dstate_last[(size_t)b * S + s] : 0.f;
}
}  // namespace cirs

extern "C" int64_t cirs_tracker_backward_workspace_bytes(const cirs_tracker_cfg* cfg, int32_t n_rows) {
    if (!cfg || n_rows <= 0) return 0;
    return (int64_t)cirs::bwd_floats(cfg, n_rows) * 4;
}

// Few workgroups (the 64-env shape: 60 row tiles, 64 episodes): an LDS request of more than half a CU's 160 KB keeps them on DIFFERENT CUs -- the dispatcher otherwise
// stacks several one-wavefront workgroups on a CU, where they share its load path while most CUs idle (round 6: what the step kernel gained from the same change).
static inline size_t spread_dyn_lds(long n_blocks, size_t static_bytes, size_t dyn_bytes) {
    if (n_blocks > cirs::device_cu_count() || getenv("CIRS_NO_SPREAD")) return dyn_bytes;
    const size_t want = 81 * 1024;
    return static_bytes + dyn_bytes >= want ? dyn_bytes : want - static_bytes;
}
template <typename K> static inline void lds_optin(K k, size_t dyn) {
    if (dyn > 8 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
}
#define CIRS_LAUNCH_SPREAD(KERNEL, GRID, STATIC_B, DYN_B, STREAM, ...)                                  \
    do {                                                                                               \
        const size_t dyn_ = spread_dyn_lds((long)(GRID), (STATIC_B), (DYN_B));                         \
        if (dyn_ != (size_t)(DYN_B)) lds_optin(KERNEL, dyn_);                                          \
        hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(64), dyn_, STREAM, __VA_ARGS__);                   \
    } while (0)

// forward recompute over the buffer rows (+ backward unless state_out is set: then the decoder runs on the last row of every env and the call
// returns -- cirs_tracker_prefix_states)
static int tracker_rows_impl(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st,
                             const int32_t* users, const int64_t* act, const double* rew, const int32_t* row_env,
                             const int32_t* row_t, const int32_t* offsets, const int32_t* lens, int32_t n_rows,
                             const float* dstate, const cirs_tracker_grads* grads, void* workspace,
                             int64_t workspace_bytes, void* stream, float* state_out, int64_t state_stride, const float* dstate_last = nullptr,
                             const cirs::TrunkFuse* ptf = nullptr, int* ptf_fused = nullptr) {
    using namespace cirs;
    CIRS_REQUIRE(cfg && w && st && row_env && row_t && offsets && lens && workspace, "null argument");
    CIRS_REQUIRE(state_out || (users && act && rew && (dstate || dstate_last) && grads), "null argument");
    if (cfg->dim_model != tD || cfg->d_hid != tH) return fail(CIRS_E_UNSUPPORTED, "dim_model == 32 and d_hid == 128 only");
    CIRS_REQUIRE(cfg->nhead == 1 || cfg->nhead == 2 || cfg->nhead == 4 || cfg->nhead == 8, "nhead must be 1,2,4,8");
    CIRS_REQUIRE(n_rows > 0, "n_rows must be positive");
    // (the last-row pass keeps one row per env of its top layer in the row-sized scratch: sized for max(n_rows, n_env) rows)
    const int n_carve = dstate_last && cfg->n_env > n_rows ? cfg->n_env : n_rows;
    CIRS_REQUIRE(workspace_bytes >= cirs_tracker_backward_workspace_bytes(cfg, n_carve), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int R = n_rows, B = cfg->n_env, S = cfg->dim_state, L = cfg->max_len, NH = cfg->nhead, nl = cfg->nlayers;
    BwdScratch sc = carve_bwd(workspace, cfg, n_carve);
    auto g1 = [&](long n) { return dim3(cdiv(n, 256)); };

    DwList dwl{};
#define DW(dY, X, O, K, dWp, dbp) launch_dw_partial(dwl, dY, O, X, K, R, O, K, dWp, dbp, 0, sc.partial, s)
#define ATT_DISPATCH_SH(KERNEL, SHMEM, ...)                                                               \
    do {                                                                                                  \
        switch (NH) {                                                                                     \
            case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(cdiv(R, 4)), dim3(256), SHMEM, s, __VA_ARGS__); break; \
            case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(cdiv(R, 4)), dim3(256), SHMEM, s, __VA_ARGS__); break; \
            case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(cdiv(R, 4)), dim3(256), SHMEM, s, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL(KERNEL<8>, dim3(cdiv(R, 4)), dim3(256), SHMEM, s, __VA_ARGS__); break; \
        }                                                                                                 \
    } while (0)
#define ATT_DISPATCH(KERNEL, ...) ATT_DISPATCH_SH(KERNEL, 0, __VA_ARGS__)
    // per-episode attention kernels (one wavefront per env, Q/K/V in LDS) whenever their LDS image fits the default 64 KB window
    auto ep_bytes = [&](bool bwd) -> size_t {
        switch (NH) {
            case 1: return 4 * (bwd ? EpGeo<1>::bwd_floats(L) : EpGeo<1>::fwd_floats(L));
            case 2: return 4 * (bwd ? EpGeo<2>::bwd_floats(L) : EpGeo<2>::fwd_floats(L));
            case 4: return 4 * (bwd ? EpGeo<4>::bwd_floats(L) : EpGeo<4>::fwd_floats(L));
            default: return 4 * (bwd ? EpGeo<8>::bwd_floats(L) : EpGeo<8>::fwd_floats(L));
        }
    };
    const bool ep = L <= 64 && ep_bytes(true) <= 64 * 1024 && !getenv("CIRS_TRACKER_ATTN_ROWS");
#define ATT_EP1(KERNEL, N, BWD, ...)                                                                                            \
    do {                                                                                                                        \
        if (dc.on) CIRS_LAUNCH_SPREAD((KERNEL<N, true>), B, 0, ep_bytes(BWD), s, __VA_ARGS__);       \
        else CIRS_LAUNCH_SPREAD((KERNEL<N, false>), B, 0, ep_bytes(BWD), s, __VA_ARGS__);           \
    } while (0)
#define ATT_EP(KERNEL, BWD, ...)                                  \
    do {                                                          \
        switch (NH) {                                             \
            case 1: ATT_EP1(KERNEL, 1, BWD, __VA_ARGS__); break;  \
            case 2: ATT_EP1(KERNEL, 2, BWD, __VA_ARGS__); break;  \
            case 4: ATT_EP1(KERNEL, 4, BWD, __VA_ARGS__); break;  \
            default: ATT_EP1(KERNEL, 8, BWD, __VA_ARGS__); break; \
        }                                                         \
    } while (0)

    DropCfg dc{};
    if (cfg->dropout_p > 0.f) {
        CIRS_REQUIRE(cfg->dropout_p < 1.f, "dropout_p must be in [0, 1)");
        dc.on = 1; dc.thr = dropout_threshold(cfg->dropout_p); dc.inv = 1.0f / (1.0f - cfg->dropout_p);
        dc.seed = cfg->dropout_seed; dc.env_base = cfg->drop_env_base;
    }
    // ---------------- prefix states, one launch: one wavefront per env (prefixes of at most 32 rows) ----------------
    if (state_out && ep && L <= 32 && S <= 64 && !getenv("CIRS_TRACKER_ROWS_UNFUSED") && !getenv("CIRS_TRACKER_PREFIX_FULL") && !getenv("CIRS_TRACKER_PREFIX_LAUNCHES")) {
        PrefixEnvArgs pa{};
        pa.x_hist = st->x_hist; pa.pe = w->pe; pa.row_env = row_env; pa.row_t = row_t; pa.offsets = offsets; pa.lens = lens;
        for (int l = 0; l < nl; ++l) pa.layer[l] = w->layer[l];
        pa.dec_w = w->dec_w; pa.dec_b = w->dec_b; pa.state_out = state_out; pa.state_stride = (long)state_stride; pa.nl = nl; pa.L = L; pa.S = S;
        if (ptf && ptf->on) {
            CIRS_REQUIRE(ptf->cfg.hidden == kH && ptf->cfg.dim_state == S && ptf->h2, "prefix states: trunk shape mismatch");
            pa.tf = *ptf;
            if (ptf_fused) *ptf_fused = 1;
        }
#define PREFIX_ENV(N)                                                                                                                        \
    do {                                                                                                                                     \
        const size_t sh = std::max<size_t>(4 * prefix_env_lds_floats<N>(), 40 * 1024);   /* at most one of these wavefronts per SIMD */        \
        if (dc.on) hipLaunchKernelGGL((prefix_env_kernel<N, true>), dim3(B), dim3(64), sh, s, pa, dc);                                        \
        else hipLaunchKernelGGL((prefix_env_kernel<N, false>), dim3(B), dim3(64), sh, s, pa, dc);                                             \
    } while (0)
        switch (NH) {
            case 1: PREFIX_ENV(1); break;
            case 2: PREFIX_ENV(2); break;
            case 4: PREFIX_ENV(4); break;
            default: PREFIX_ENV(8); break;
        }
#undef PREFIX_ENV
        CIRS_CHECK_LAUNCH("prefix_env_kernel");
        return CIRS_OK;
    }
    // ---------------- forward recompute ----------------
    // fused row chains (embed + in_proj; out_proj .. LayerNorm2 + the next in_proj) unless CIRS_TRACKER_ROWS_UNFUSED asks for the
    // per-op launches (kept for A/B runs and for alignment-free inputs)
    const bool fused_rows = !getenv("CIRS_TRACKER_ROWS_UNFUSED");
    if (fused_rows) {
        const dim3 gt(cdiv(R, 32));
        if (dc.on) CIRS_LAUNCH_SPREAD(embed_inproj<true>, gt.x, 0, 0, s, st->x_hist, w->pe, row_env, row_t, R, L, w->layer[0].in_proj_w,
                                      w->layer[0].in_proj_b, sc.H[0], sc.QKV[0], dc);
        else CIRS_LAUNCH_SPREAD(embed_inproj<false>, gt.x, 0, 0, s, st->x_hist, w->pe, row_env, row_t, R, L, w->layer[0].in_proj_w,
                                w->layer[0].in_proj_b, sc.H[0], sc.QKV[0], dc);
    } else {
        hipLaunchKernelGGL(embed_rows, g1((long)R * tD), dim3(256), 0, s, st->x_hist, w->pe, row_env, row_t, R, L, sc.H[0], dc);
    }
    bool last_compact = false;
    // last-row pass (dstate_last): the top layer on ONE row per env -- its attention for the one query, its row chain on B compact rows
    const bool top_last = dstate_last && fused_rows && ep && nl >= 2 && !getenv("CIRS_TRACKER_LAST_FULL");
    float *dY1c = sc.LC, *att_stats = sc.LC + (size_t)B * tD;
    int32_t *env_c = reinterpret_cast<int32_t*>(att_stats + (size_t)B * 16), *t_c = env_c + B;
#define ATT_LAST1(KERNEL, N, ...)                                                                                        \
    do {                                                                                                                 \
        if (dc.on) hipLaunchKernelGGL((KERNEL<N, true>), dim3(cdiv(B, 64 / N)), dim3(64), 0, s, __VA_ARGS__);             \
        else hipLaunchKernelGGL((KERNEL<N, false>), dim3(cdiv(B, 64 / N)), dim3(64), 0, s, __VA_ARGS__);                  \
    } while (0)
#define ATT_LAST(KERNEL, ...)                                  \
    do {                                                       \
        switch (NH) {                                          \
            case 1: ATT_LAST1(KERNEL, 1, __VA_ARGS__); break;  \
            case 2: ATT_LAST1(KERNEL, 2, __VA_ARGS__); break;  \
            case 4: ATT_LAST1(KERNEL, 4, __VA_ARGS__); break;  \
            default: ATT_LAST1(KERNEL, 8, __VA_ARGS__); break; \
        }                                                      \
    } while (0)
    for (int l = 0; l < nl; ++l) {
        const cirs_tracker_layer& y = w->layer[l];
        if (!fused_rows) launch_rows_gemm(true, sc.H[l], tD, y.in_proj_w, tD, y.in_proj_b, R, tD, 96, 0, nullptr, 0, sc.QKV[l], 96, s);
        if (top_last && l == nl - 1) {
            float* Hc = sc.T1;
            ATT_LAST(attn_last_fwd, (const float*)sc.QKV[l], (const float*)sc.H[l], row_env, row_t, offsets, lens, B, L, sc.ATT[l], Hc, att_stats, env_c, t_c,
                     dc, l);
            LayerFwdArgs fa{sc.ATT[l], Hc, y, nullptr, nullptr, sc.XH1[l], sc.RS1[l], sc.H1N[l], sc.FF1[l], sc.XH2[l], sc.RS2[l], sc.H[l + 1], nullptr,
                            env_c, t_c, B, l};
            if (dc.on) CIRS_LAUNCH_SPREAD(layer_rows_fwd<true>, cdiv(B, 32), 40 * 1024, 0, s, fa, dc);
            else CIRS_LAUNCH_SPREAD(layer_rows_fwd<false>, cdiv(B, 32), 40 * 1024, 0, s, fa, dc);
            continue;
        }
        // (dropout: the forward launch leaves its keep bits in the probability scratch of the row-wise path, unused here, for the backward launch -- not in the prefix pass)
        if (ep) ATT_EP(attn_fwd_ep, false, (const float*)sc.QKV[l], offsets, lens, L, sc.ATT[l], dc, l, (dc.on && !state_out) ? reinterpret_cast<uint32_t*>(sc.P[l]) : (uint32_t*)nullptr);
        else ATT_DISPATCH_SH(attn_fwd, (size_t)4 * NH * L * sizeof(float), sc.QKV[l], row_env, row_t, offsets, R, L, sc.P[l], sc.ATT[l], dc, l, sc.PM[l]);
        if (fused_rows && state_out && l == nl - 1 && !getenv("CIRS_TRACKER_PREFIX_FULL")) {      // prefix states: the last layer's row chain on the envs' last rows only
            static_assert(tD == 32, "prefix_last_rows_kernel maps 32 columns to 32 threads");
            float *ATTc = sc.T0, *Hc = sc.T1;
            int32_t *env_c = reinterpret_cast<int32_t*>(sc.T2), *t_c = env_c + B;
            hipLaunchKernelGGL(prefix_last_rows_kernel, dim3(cdiv(B, 8)), dim3(256), 0, s, (const float*)sc.ATT[l], (const float*)sc.H[l], row_env, row_t, offsets,
                               lens, B, ATTc, Hc, env_c, t_c);
            LayerFwdArgs fa{ATTc, Hc, y, nullptr, nullptr, sc.XH1[l], sc.RS1[l], sc.H1N[l], sc.FF1[l], sc.XH2[l], sc.RS2[l], sc.H[l + 1], nullptr,
                            env_c, t_c, B, l};
            if (dc.on) CIRS_LAUNCH_SPREAD(layer_rows_fwd<true>, cdiv(B, 32), 40 * 1024, 0, s, fa, dc);
            else CIRS_LAUNCH_SPREAD(layer_rows_fwd<false>, cdiv(B, 32), 40 * 1024, 0, s, fa, dc);
            last_compact = true;
            continue;
        }
        if (fused_rows) {
            LayerFwdArgs fa{sc.ATT[l], sc.H[l], y, l + 1 < nl ? w->layer[l + 1].in_proj_w : nullptr, l + 1 < nl ? w->layer[l + 1].in_proj_b : nullptr,
                            sc.XH1[l], sc.RS1[l], sc.H1N[l], sc.FF1[l], sc.XH2[l], sc.RS2[l], sc.H[l + 1], l + 1 < nl ? sc.QKV[l + 1] : nullptr,
                            row_env, row_t, R, l};
            if (dc.on) CIRS_LAUNCH_SPREAD(layer_rows_fwd<true>, cdiv(R, 32), 40 * 1024, 0, s, fa, dc);
            else CIRS_LAUNCH_SPREAD(layer_rows_fwd<false>, cdiv(R, 32), 40 * 1024, 0, s, fa, dc);
            continue;
        }
        launch_rows_gemm(true, sc.ATT[l], tD, y.out_proj_w, tD, y.out_proj_b, R, tD, tD, 0, nullptr, 0, sc.T0, tD, s);
        hipLaunchKernelGGL(ln_fwd, g1((long)R * tD), dim3(256), 0, s, sc.H[l], sc.T0, y.norm1_w, y.norm1_b, R, sc.XH1[l], sc.RS1[l], sc.H1N[l],
                           dc, row_env, row_t, l, (int)CIRS_DROP_RES1);
        launch_rows_gemm(true, sc.H1N[l], tD, y.lin1_w, tD, y.lin1_b, R, tD, tH, 1, nullptr, 0, sc.FF1[l], tH, s);
        if (dc.on)   // FF1 holds relu(.) * mask / (1 - p): the input of lin2, the relu-and-dropout gate of the backward
            hipLaunchKernelGGL(drop_rows, g1((long)R * tH), dim3(256), 0, s, dc, (const float*)sc.FF1[l], row_env, row_t, R, tH, l, (int)CIRS_DROP_FF, sc.FF1[l]);
        launch_rows_gemm(true, sc.FF1[l], tH, y.lin2_w, tH, y.lin2_b, R, tH, tD, 0, nullptr, 0, sc.T0, tD, s);
        hipLaunchKernelGGL(ln_fwd, g1((long)R * tD), dim3(256), 0, s, sc.H1N[l], sc.T0, y.norm2_w, y.norm2_b, R, sc.XH2[l], sc.RS2[l], sc.H[l + 1],
                           dc, row_env, row_t, l, (int)CIRS_DROP_RES2);
    }
    CIRS_CHECK_LAUNCH("tracker forward recompute");
    if (state_out) {
        hipLaunchKernelGGL(prefix_decoder_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, (const float*)sc.H[nl], offsets, lens, B, S, w->dec_w, w->dec_b,
                           state_out, (long)state_stride, last_compact ? 1 : 0);
        CIRS_CHECK_LAUNCH("prefix_decoder_kernel");
        return CIRS_OK;
    }
    // ---------------- backward ----------------
    float* dH = sc.T0;  // gradient w.r.t. the current layer output
    // (each weight-gradient problem rides in the launch of the row GEMM that consumes the same dY: DW_ROWS)
#define DW_ROWS(X2, O, K, dWp, dbp, ...) launch_rows_gemm_dw(dwl, X2, K, O, K, dWp, dbp, sc.partial, __VA_ARGS__, s)
    if (top_last) {
        hipLaunchKernelGGL(dstate_last_rows, g1((long)B * S), dim3(256), 0, s, dstate_last, lens, B, S, sc.G);
        launch_rows_gemm(false, sc.G, S, w->dec_w, tD, nullptr, B, S, tD, 0, nullptr, 0, dH, tD, s);      // (the decoder's dW problem joins the top layer's batch)
    } else {
        if (dstate_last) hipLaunchKernelGGL(dstate_last_full, g1((long)R * S), dim3(256), 0, s, dstate_last, row_env, offsets, lens, R, S, sc.G);
        else hipLaunchKernelGGL(gather_dstate, g1((long)R * S), dim3(256), 0, s, dstate, row_env, row_t, R, S, B, sc.G);
        DW_ROWS(sc.H[nl], S, tD, grads->dec_w, grads->dec_b, false, sc.G, S, w->dec_w, tD, nullptr, R, S, tD, 0, nullptr, 0, dH, tD);
    }
    for (int l = nl - 1; l >= 0 && fused_rows; --l) {
        // fused row chain (layer_rows_bwd) + one batched launch of the layer's weight-gradient problems + the attention backward
        const cirs_tracker_layer& y = w->layer[l];
        const cirs_tracker_layer_grads& gy = grads->layer[l];
        if (top_last && l == nl - 1) {      // B compact rows; the slab count of its weight-gradient problems stays the R-row one (dw_list_final walks one count)
            float* dB1 = dc.on ? sc.T4 : dY1c;
            LayerBwdArgs ba{sc.T0, nullptr, nullptr, nullptr, sc.T0, 0, y, sc.XH2[l], sc.RS2[l], sc.FF1[l], sc.XH1[l], sc.RS1[l], sc.T3, sc.dFF1, sc.T1, dY1c, dB1,
                            sc.T5, env_c, t_c, B, l};
            if (dc.on) CIRS_LAUNCH_SPREAD(layer_rows_bwd<true>, cdiv(B, 32), 40 * 1024, 0, s, ba, dc);
            else CIRS_LAUNCH_SPREAD(layer_rows_bwd<false>, cdiv(B, 32), 40 * 1024, 0, s, ba, dc);
            DwBatch bt{};
            dw_batch_add(bt, dwl, sc.G, S, sc.H[nl], tD, R, S, tD, grads->dec_w, grads->dec_b, 0, sc.partial);
            dw_batch_add(bt, dwl, sc.T0, tD, sc.XH2[l], tD, R, tD, tD, gy.norm2_w, gy.norm2_b, 1, sc.partial);
            dw_batch_add(bt, dwl, sc.T3, tD, sc.FF1[l], tH, R, tD, tH, gy.lin2_w, gy.lin2_b, 0, sc.partial);
            dw_batch_add(bt, dwl, sc.dFF1, tH, sc.H1N[l], tD, R, tH, tD, gy.lin1_w, gy.lin1_b, 0, sc.partial);
            dw_batch_add(bt, dwl, sc.T1, tD, sc.XH1[l], tD, R, tD, tD, gy.norm1_w, gy.norm1_b, 1, sc.partial);
            dw_batch_add(bt, dwl, dB1, tD, sc.ATT[l], tD, R, tD, tD, gy.out_proj_w, gy.out_proj_b, 0, sc.partial);
            dw_batch_launch(bt, R, s, B);
            ATT_LAST(attn_last_bwd, (const float*)sc.QKV[l], (const float*)sc.T5, (const float*)dY1c, (const float*)att_stats, offsets, lens, B, L, sc.dQKV,
                     sc.T2, dc, l);
            continue;
        }
        const bool pre = l + 1 < nl;     // the in_proj backward of the layer above opens this layer's chain
        float* dB1 = dc.on ? sc.T4 : sc.T2;
        LayerBwdArgs ba{sc.T0, pre ? sc.T2 : nullptr, pre ? sc.dQKV : nullptr, pre ? w->layer[l + 1].in_proj_w : nullptr, sc.T0, 0, y,
                        sc.XH2[l], sc.RS2[l], sc.FF1[l], sc.XH1[l], sc.RS1[l], sc.T3, sc.dFF1, sc.T1, sc.T2, dB1, sc.T5, row_env, row_t, R, l};
        if (dc.on) CIRS_LAUNCH_SPREAD(layer_rows_bwd<true>, cdiv(R, 32), 40 * 1024, 0, s, ba, dc);
        else CIRS_LAUNCH_SPREAD(layer_rows_bwd<false>, cdiv(R, 32), 40 * 1024, 0, s, ba, dc);
        DwBatch bt{};
        if (pre) dw_batch_add(bt, dwl, sc.dQKV, 96, sc.H[l + 1], tD, R, 96, tD, grads->layer[l + 1].in_proj_w, grads->layer[l + 1].in_proj_b, 0, sc.partial);
        dw_batch_add(bt, dwl, sc.T0, tD, sc.XH2[l], tD, R, tD, tD, gy.norm2_w, gy.norm2_b, 1, sc.partial);       // diag(dH^T Xhat2), column sums
        dw_batch_add(bt, dwl, sc.T3, tD, sc.FF1[l], tH, R, tD, tH, gy.lin2_w, gy.lin2_b, 0, sc.partial);
        dw_batch_add(bt, dwl, sc.dFF1, tH, sc.H1N[l], tD, R, tH, tD, gy.lin1_w, gy.lin1_b, 0, sc.partial);
        dw_batch_add(bt, dwl, sc.T1, tD, sc.XH1[l], tD, R, tD, tD, gy.norm1_w, gy.norm1_b, 1, sc.partial);
        dw_batch_add(bt, dwl, dB1, tD, sc.ATT[l], tD, R, tD, tD, gy.out_proj_w, gy.out_proj_b, 0, sc.partial);
        dw_batch_launch(bt, R, s);
        const float* dATT = sc.T5;
        if (ep) {
            ATT_EP(attn_bwd_ep, true, (const float*)sc.QKV[l], dATT, offsets, lens, L, sc.dQKV, dc, l, dc.on ? reinterpret_cast<const uint32_t*>(sc.P[l]) : (const uint32_t*)nullptr);
        } else {
            ATT_DISPATCH_SH(attn_bwd_q, (size_t)4 * NH * L * sizeof(float), sc.QKV[l], sc.P[l], dATT, row_env, row_t, offsets, R, L, sc.dS, sc.dQKV,
                            (const float*)sc.PM[l], dc.inv);
            ATT_DISPATCH(attn_bwd_kv, sc.QKV[l], dc.on ? sc.PM[l] : sc.P[l], sc.dS, dATT, row_env, row_t, offsets, lens, R, L, sc.dQKV);
        }
    }
    if (fused_rows) {   // layer 0's in_proj backward (+ the positional-encoding dropout): the pre stage alone; its dW problem joins the slot problems below
        LayerBwdArgs ba{};
        ba.dY1p = sc.T2; ba.dQKVp = sc.dQKV; ba.winp = w->layer[0].in_proj_w; ba.dHout = sc.T0; ba.only_pre = 1;
        ba.row_env = row_env; ba.row_t = row_t; ba.R = R;
        if (dc.on) CIRS_LAUNCH_SPREAD(layer_rows_bwd<true>, cdiv(R, 32), 40 * 1024, 0, s, ba, dc);
        else CIRS_LAUNCH_SPREAD(layer_rows_bwd<false>, cdiv(R, 32), 40 * 1024, 0, s, ba, dc);
        dH = sc.T0;
    }
    for (int l = nl - 1; l >= 0 && !fused_rows; --l) {
        const cirs_tracker_layer& y = w->layer[l];
        const cirs_tracker_layer_grads& gy = grads->layer[l];
        // LN2
        float* dY2 = sc.T1;
        launch_ln_bwd_dw(dwl, dH, sc.XH2[l], sc.RS2[l], y.norm2_w, R, dY2, gy.norm2_w, gy.norm2_b, sc.partial, s);   // + diag(dH^T Xhat), column sums
        // FF: with dropout the lin2 branch sees dY2 * mask2 / (1 - p) (the residual keeps dY2) and the gate of the hidden layer is
        // relu' * mask_ff / (1 - p): FF1 > 0 already encodes "relu active and kept", the scale is applied to dFF1
        const float* dB2 = dY2;
        if (dc.on) {
            hipLaunchKernelGGL(drop_rows, g1((long)R * tD), dim3(256), 0, s, dc, (const float*)dY2, row_env, row_t, R, tD, l, (int)CIRS_DROP_RES2, sc.T3);
            dB2 = sc.T3;
        }
        DW_ROWS(sc.FF1[l], tD, tH, gy.lin2_w, gy.lin2_b, false, dB2, tD, y.lin2_w, tH, nullptr, R, tD, tH, 0, sc.FF1[l], 0, sc.dFF1, tH);
        if (dc.on) hipLaunchKernelGGL(scale_rows, g1((long)R * tH), dim3(256), 0, s, sc.dFF1, (long)R * tH, dc.inv);
        // d H1N = dY2 (residual) + dFF1 * W1
        DW_ROWS(sc.H1N[l], tH, tD, gy.lin1_w, gy.lin1_b, false, sc.dFF1, tH, y.lin1_w, tD, nullptr, R, tH, tD, 0, nullptr, 1, dY2, tD);
        // LN1
        float* dY1 = sc.T2;
        launch_ln_bwd_dw(dwl, dY2, sc.XH1[l], sc.RS1[l], y.norm1_w, R, dY1, gy.norm1_w, gy.norm1_b, sc.partial, s);
        // out_proj (its branch sees dY1 * mask1 / (1 - p); the residual keeps dY1)
        const float* dB1 = dY1;
        if (dc.on) {
            hipLaunchKernelGGL(drop_rows, g1((long)R * tD), dim3(256), 0, s, dc, (const float*)dY1, row_env, row_t, R, tD, l, (int)CIRS_DROP_RES1, sc.T3);
            dB1 = sc.T3;
        }
        float* dATT = sc.T1;
        DW_ROWS(sc.ATT[l], tD, tD, gy.out_proj_w, gy.out_proj_b, false, dB1, tD, y.out_proj_w, tD, nullptr, R, tD, tD, 0, nullptr, 0, dATT, tD);
        // attention (dropout: V is weighted by the masked probabilities PM; the softmax backward runs on P)
        if (ep) {
            ATT_EP(attn_bwd_ep, true, (const float*)sc.QKV[l], (const float*)dATT, offsets, lens, L, sc.dQKV, dc, l, dc.on ? reinterpret_cast<const uint32_t*>(sc.P[l]) : (const uint32_t*)nullptr);
        } else {
            ATT_DISPATCH_SH(attn_bwd_q, (size_t)4 * NH * L * sizeof(float), sc.QKV[l], sc.P[l], dATT, row_env, row_t, offsets, R, L, sc.dS, sc.dQKV,
                            (const float*)sc.PM[l], dc.inv);
            ATT_DISPATCH(attn_bwd_kv, sc.QKV[l], dc.on ? sc.PM[l] : sc.P[l], sc.dS, dATT, row_env, row_t, offsets, lens, R, L, sc.dQKV);
        }
        // in_proj
        // d H_l = dY1 (residual) + dQKV * W_in
        DW_ROWS(sc.H[l], 96, tD, gy.in_proj_w, gy.in_proj_b, false, sc.dQKV, 96, y.in_proj_w, tD, nullptr, R, 96, tD, 0, nullptr, 1, dY1, tD);
        dH = dY1;
        if (l > 0) {  // keep dH in T0 for the next iteration (T2 is reused as dY1)
            CIRS_HIP(hipMemcpyAsync(sc.T0, dY1, sizeof(float) * (size_t)R * tD, hipMemcpyDeviceToDevice, s));
            dH = sc.T0;
        }
    }
    CIRS_CHECK_LAUNCH("tracker backward layers");
    if (dc.on && !fused_rows)   // gradient through the PositionalEncoding dropout
        hipLaunchKernelGGL(drop_rows, g1((long)R * tD), dim3(256), 0, s, dc, (const float*)dH, row_env, row_t, R, tD, 0, (int)CIRS_DROP_POS, dH);
    // input slots + embeddings (forward activations are no longer needed: reuse their scratch)
    float* DU = sc.T1;
    float* EU = sc.H[nl];
    float* DPRE = fused_rows ? sc.H1N[0] : sc.H[0];   // (fused: H[0] is still an operand of layer 0's in_proj weight-gradient problem)
    float* CU = sc.QKV[0];                 // [R,32] per-row contribution to Emb_user
    float* CI = sc.QKV[0] + (size_t)R * tD; // [R,32] per-row contribution to Emb_item  (QKV is [R,96])
    int32_t* key_user = (int32_t*)sc.RS1[0];
    int32_t* key_item = (int32_t*)sc.RS2[0];
    const bool merged = fused_rows && B <= R;      // one sort for both tables: the per-env user rows sit behind the item rows (third part of QKV[0])
    const int slot_rpw = R >= (1 << 17) ? 8 : 1;
    hipLaunchKernelGGL(slot_bwd, dim3(cdiv(R, 4 * slot_rpw)), dim3(256), 0, s, *w, dH, users, act, rew, row_env, row_t, R, B, DU, EU, DPRE,
                       sc.GIN, CU, CI, key_user, key_item, merged ? CI + (size_t)R * tD : nullptr, slot_rpw);
    if (merged) {
        if (int rc = emb_scatter_merged(key_item, users, lens, CI, R, B, cfg->n_items, cfg->n_users, grads->emb_item, grads->emb_user, sc.sort,
                                        emb_sort_bytes(R + B), s)) return rc;
    } else {
    {   // user embeddings: one pair per env (T0 and T2 = [R, 32] each, R >= B, are free after slot_bwd)
        int32_t* keys_c = (int32_t*)sc.T0;
        float* contrib_c = sc.T2;
        CIRS_HIP(hipMemsetAsync(keys_c, 0xFF, sizeof(int32_t) * (size_t)B, s));   // -1: env without rows in this call
        hipLaunchKernelGGL(compact_user_rows, dim3(cdiv(R, 8)), dim3(256), 0, s, row_env, row_t, users, (const float*)CU, R, keys_c, contrib_c);
        if (int rc = emb_scatter_sorted(keys_c, contrib_c, B, cfg->n_users, grads->emb_user, sc.sort, emb_sort_bytes(R), s)) return rc;
    }
    if (int rc = emb_scatter_sorted(key_item, CI, R, cfg->n_items, grads->emb_item, sc.sort, emb_sort_bytes(R), s)) return rc;
    }
    if (fused_rows) {
        DwBatch bt{};
        dw_batch_add(bt, dwl, sc.dQKV, 96, sc.H[0], tD, R, 96, tD, grads->layer[0].in_proj_w, grads->layer[0].in_proj_b, 0, sc.partial);
        dw_batch_add(bt, dwl, DU, tD, EU, tD, R, tD, tD, grads->ffn_user_w, grads->ffn_user_b, 0, sc.partial);
        dw_batch_add(bt, dwl, DPRE, tD, sc.GIN, tD + 1, R, tD, tD + 1, grads->gate_w, grads->gate_b, 0, sc.partial);
        dw_batch_launch(bt, R, s);
    } else {
        DW(DU, EU, tD, tD, grads->ffn_user_w, grads->ffn_user_b);
        DW(DPRE, sc.GIN, tD, tD + 1, grads->gate_w, grads->gate_b);
    }
    launch_dw_list_final(dwl, R, sc.partial, s);   // every weight / bias gradient of the pass: slab sums in one launch
    CIRS_CHECK_LAUNCH("tracker backward slots");
#undef DW
#undef ATT_LAST
#undef ATT_LAST1
#undef ATT_DISPATCH
#undef ATT_EP
#undef ATT_EP1
#undef ATT_DISPATCH_SH
    return CIRS_OK;
}

extern "C" int cirs_tracker_backward(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st,
                                     const int32_t* users, const int64_t* act, const double* rew, const int32_t* row_env,
                                     const int32_t* row_t, const int32_t* offsets, const int32_t* lens, int32_t n_rows,
                                     const float* dstate, const cirs_tracker_grads* grads, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    CIRS_REQUIRE(users && act && rew && dstate && grads, "null argument");
    return tracker_rows_impl(cfg, w, st, users, act, rew, row_env, row_t, offsets, lens, n_rows, dstate, grads, workspace, workspace_bytes, stream,
                             nullptr, 0);
}

extern "C" int cirs_tracker_backward_last(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st,
                                          const int32_t* users, const int64_t* act, const double* rew, const int32_t* row_env,
                                          const int32_t* row_t, const int32_t* offsets, const int32_t* lens, int32_t n_rows,
                                          const float* dstate_last, const cirs_tracker_grads* grads, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
    CIRS_REQUIRE(users && act && rew && dstate_last && grads, "null argument");
    return tracker_rows_impl(cfg, w, st, users, act, rew, row_env, row_t, offsets, lens, n_rows, nullptr, grads, workspace, workspace_bytes, stream,
                             nullptr, 0, dstate_last);
}

extern "C" int cirs_tracker_prefix_states(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st,
                                          const int32_t* row_env, const int32_t* row_t, const int32_t* offsets, const int32_t* lens,
                                          int32_t n_rows, float* state_out, int64_t state_stride, void* workspace, int64_t workspace_bytes,
                                          void* stream) {
    CIRS_REQUIRE(state_out && state_stride >= (cfg ? cfg->dim_state : 0), "bad state_out");
    return tracker_rows_impl(cfg, w, st, nullptr, nullptr, nullptr, row_env, row_t, offsets, lens, n_rows, nullptr, nullptr, workspace, workspace_bytes,
                             stream, state_out, state_stride);
}

int cirs::tracker_prefix_states_trunk(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st, const int32_t* row_env,
                                      const int32_t* row_t, const int32_t* offsets, const int32_t* lens, int32_t n_rows, float* state_out, int64_t state_stride,
                                      void* workspace, int64_t workspace_bytes, void* stream, const TrunkFuse* tf, int* fused) {
    CIRS_REQUIRE(state_out && state_stride >= (cfg ? cfg->dim_state : 0) && fused, "bad state_out");
    *fused = 0;
    return tracker_rows_impl(cfg, w, st, nullptr, nullptr, nullptr, row_env, row_t, offsets, lens, n_rows, nullptr, nullptr, workspace, workspace_bytes,
                             stream, state_out, state_stride, nullptr, tf, fused);
}

// ---- row-sharded embedding tables (BASELINE configs[4]): the owner rank's ordered scatter of received gradient rows ----------------
extern "C" int64_t cirs_embedding_scatter_workspace_bytes(int64_t n_rows) { return (int64_t)cirs::emb_sort_bytes(n_rows) + 1024; }

extern "C" int cirs_embedding_scatter(const int32_t* keys, const float* contrib, int64_t n_rows, int32_t n_table_rows, float* grad_table,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace cirs;
    CIRS_REQUIRE(keys && contrib && grad_table && workspace && n_rows > 0 && n_table_rows > 0, "bad arguments");
    CIRS_REQUIRE(n_rows < (1ll << 31), "too many rows");
    CIRS_REQUIRE(workspace_bytes >= cirs_embedding_scatter_workspace_bytes(n_rows), "workspace too small");
    void* ws = (void*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    return emb_scatter_sorted(keys, contrib, (int)n_rows, n_table_rows, grad_table, ws, (size_t)workspace_bytes - 256, (hipStream_t)stream);
}

