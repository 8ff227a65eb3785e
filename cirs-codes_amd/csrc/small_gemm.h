// small_gemm.h -- the small dense layers of the learner / tracker backward on the fp32 matrix cores.
//
// All matrices here have <= 129 columns and thousands of rows; they are L1/L2 resident and latency-bound, so the
// kernels use plain (cached) global loads straight into the v_mfma_f32_32x32x2_f32 operand layout, one wavefront per
// 32 x 32 output tile:
//   rows_gemm   Y[R,N] (+)= X[R,Kd] * B  (+ bias, relu, relu-mask)   with B[kd,n] = W[n*ldw+kd] ("NT": forward linear,
//               Y = X W^T) or W[kd*ldw+n] ("NN": backward dX = dY W).  Fixed k order -> deterministic.
//   dw_gemm     dW[O,K] = sum_r dY[r,O]^T X[r,K], db[O] = sum_r dY[r,O]; rows are split into <= 128 contiguous slabs
//               (one wavefront each), slab partials are summed in slab order by dw_gemm_final: fixed order, no atomics.
#pragma once
#include "common.h"

namespace cirs {

typedef float sg_f32x16 __attribute__((ext_vector_type(16)));

// grid = (ceil(R/32), ceil(N/32)), block = 64
template <bool kNT>
__device__ __forceinline__ void rows_gemm_body(int bx, int by, const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
                                               const float* __restrict__ bias, int R, int Kd, int N, int relu,
                                               const float* __restrict__ relu_of, int accumulate, float* __restrict__ Y, int ldy,
                                               const long* __restrict__ out_row) {
    const int lane = threadIdx.x, hi = lane >> 5, lo = lane & 31;
    const int row0 = bx * 32, n0 = by * 32;
    const int row = row0 + lo, n = n0 + lo;
    const bool row_ok = row < R, n_ok = n < N;
    sg_f32x16 acc;
    const float b = (bias && n_ok) ? bias[n] : 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = b;
    const float* xr = X + (size_t)(row_ok ? row : 0) * ldx;
    if (kNT && (Kd & 31) == 0 && (ldx & 3) == 0 && ((uintptr_t)X & 15) == 0 && (ldw & 3) == 0 && ((uintptr_t)W & 15) == 0) {
        // NT form (rows of W contiguous in k): contraction index relabelled, lane half hi owns k in [kk + 16 hi, kk + 16 hi + 16)
        // -> its row of X and of W is read as four float4 per 32 k instead of 16 scalars (8.0 vs 11.0 us per launch; for the
        // NN form the strided W loads gain nothing: 13.6 vs 12.2 us, it keeps the scalar batches below)
        const float* wr = W + (size_t)(n_ok ? n : 0) * ldw;
        for (int kk = 0; kk < Kd; kk += 32) {
            float a[16], bv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(xr + kk + 16 * hi + 4 * q);
                a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(wr + kk + 16 * hi + 4 * q);
                bv[4 * q] = t.x; bv[4 * q + 1] = t.y; bv[4 * q + 2] = t.z; bv[4 * q + 3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(row_ok ? a[j] : 0.f, n_ok ? bv[j] : 0.f, acc, 0, 0, 0);
        }
    } else if (!kNT && (Kd & 31) == 0 && (ldx & 3) == 0 && ((uintptr_t)X & 15) == 0) {
        // NN form with the same relabelling of the contraction index: the lane's row of X as four float4 per 32 k (a dword per k costs
        // the address unit 64 cache lines per instruction, four times as many instructions); W[k][n] stays one coalesced dword per k
        for (int kk = 0; kk < Kd; kk += 32) {
            float a[16], bv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(xr + kk + 16 * hi + 4 * q);
                a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) bv[j] = n_ok ? W[(size_t)(kk + 16 * hi + j) * ldw + n] : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(row_ok ? a[j] : 0.f, bv[j], acc, 0, 0, 0);
        }
    } else {
        for (int kk = 0; kk < Kd; kk += 16) {  // 8 MFMA steps per batch: all 16 loads are issued before the first MFMA
            float a[8], bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kd = kk + 2 * j + hi;
                const bool k_ok = kd < Kd;
                a[j] = (row_ok && k_ok) ? xr[kd] : 0.f;
                bv[j] = 0.f;
                if (n_ok && k_ok) bv[j] = kNT ? W[(size_t)n * ldw + kd] : W[(size_t)kd * ldw + n];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[j], acc, 0, 0, 0);
        }
    }
    if (!n_ok) return;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int r = row0 + (s & 3) + 8 * (s >> 2) + 4 * hi;
        if (r >= R) continue;
        const size_t o = (size_t)(out_row ? out_row[r] : (long)r) * ldy + n;  // optional scatter of output rows
        float v = acc[s];
        if (relu) v = fmaxf(v, 0.f);
        if (relu_of && !(relu_of[o] > 0.f)) v = 0.f;
        Y[o] = accumulate ? Y[o] + v : v;
    }
}

template <bool kNT>
static __global__ __launch_bounds__(64) void rows_gemm_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
                                                              const float* __restrict__ bias, int R, int Kd, int N, int relu,
                                                              const float* __restrict__ relu_of, int accumulate,
                                                              float* __restrict__ Y, int ldy,
                                                              const long* __restrict__ out_row = nullptr) {
    rows_gemm_body<kNT>(blockIdx.x, blockIdx.y, X, ldx, W, ldw, bias, R, Kd, N, relu, relu_of, accumulate, Y, ldy, out_row);
}

static inline void launch_rows_gemm(bool nt, const float* X, int ldx, const float* W, int ldw, const float* bias, int R, int Kd, int N,
                                    int relu, const float* relu_of, int accumulate, float* Y, int ldy, hipStream_t s,
                                    const long* out_row = nullptr) {
    const dim3 grid(cdiv(R, 32), cdiv(N, 32));
    if (nt) hipLaunchKernelGGL(rows_gemm_kernel<true>, grid, dim3(64), 0, s, X, ldx, W, ldw, bias, R, Kd, N, relu, relu_of, accumulate, Y, ldy, out_row);
    else hipLaunchKernelGGL(rows_gemm_kernel<false>, grid, dim3(64), 0, s, X, ldx, W, ldw, bias, R, Kd, N, relu, relu_of, accumulate, Y, ldy, out_row);
}

constexpr int kDwMaxSlabs = 256;  // 30 k buffer rows -> ~115 rows (4 load batches of 32) per wavefront
__host__ inline int dwg_slabs(long R) {
    const long want = (R + 63) / 64;  // >= 64 rows per slab
    return (int)(want < 1 ? 1 : (want > kDwMaxSlabs ? kDwMaxSlabs : want));
}
__host__ inline size_t dwg_partial_floats(long R, int O, int K) { return (size_t)dwg_slabs(R) * O * (K + 1); }

// grid = (ceil(O/32) * ceil(K/32), n_slabs), block = 64.  partial[slab][o*(K+1)+k], k == K: bias column
__device__ __forceinline__ void dw_gemm_body(int bx, int slab, const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, int R,
                                             int O, int K, int rows_per_slab, float* __restrict__ partial) {
    const int lane = threadIdx.x, hi = lane >> 5, lo = lane & 31;
    const int k_tiles = (K + 31) / 32;
    const int o0 = (bx / k_tiles) * 32, k0 = (bx % k_tiles) * 32;
    const int r_beg = slab * rows_per_slab, r_end = min(R, r_beg + rows_per_slab);
    const int o = o0 + lo, k = k0 + lo;
    const bool o_ok = o < O, k_ok = k < K;
    sg_f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.f;
    float bsum = 0.f;
    for (int r = r_beg; r < r_end; r += 64) {  // 32 MFMA steps (64 rows) per batch: 64 loads in flight, then the MFMAs (row order unchanged)
        float a[32], b[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int rr = r + 2 * j + hi;
            const bool r_ok = rr < r_end;
            a[j] = (r_ok && o_ok) ? dY[(size_t)rr * ldy + o] : 0.f;
            b[j] = (r_ok && k_ok) ? X[(size_t)rr * ldx + k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            bsum += a[j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
        }
    }
    float* out = partial + (size_t)slab * O * (K + 1);
    if (k_ok) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int oo = o0 + (s & 3) + 8 * (s >> 2) + 4 * hi;
            if (oo < O) out[(size_t)oo * (K + 1) + k] = acc[s];
        }
    }
    if (k0 == 0) {
        bsum += __shfl_xor(bsum, 32, CIRS_WAVE);
        if (hi == 0 && o_ok) out[(size_t)o * (K + 1) + K] = bsum;
    }
}

static __global__ __launch_bounds__(64) void dw_gemm_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, int R,
                                                            int O, int K, int rows_per_slab, float* __restrict__ partial) {
    dw_gemm_body(blockIdx.x, blockIdx.y, dY, ldy, X, ldx, R, O, K, rows_per_slab, partial);
}

// A weight-gradient problem and the row GEMM that follows it in a backward chain read the same dY and do not depend on each other:
// ONE launch, the first rgx * rgy workgroups are the row GEMM's, the rest the dW problem's (each launch of this size costs ~7 us of
// boundary + cold start however little it computes).  Same per-workgroup code as the two kernels above: same bits.
struct RowsGemmArgs { const float* X; int ldx; const float* W; int ldw; const float* bias; int R, Kd, N, relu; const float* relu_of; int accumulate; float* Y; int ldy; };
struct DwGemmArgs { const float* dY; int ldy; const float* X; int ldx; int R, O, K, rows_per_slab; float* partial; };
template <bool kNT>
static __global__ __launch_bounds__(64) void rows_gemm_dw_kernel(RowsGemmArgs ra, int rgx, int rgy, DwGemmArgs da, int dgx) {
    const int b = blockIdx.x;
    if (b < rgx * rgy) {
        rows_gemm_body<kNT>(b % rgx, b / rgx, ra.X, ra.ldx, ra.W, ra.ldw, ra.bias, ra.R, ra.Kd, ra.N, ra.relu, ra.relu_of, ra.accumulate, ra.Y, ra.ldy,
                            nullptr);
    } else {
        const int q = b - rgx * rgy;
        dw_gemm_body(q % dgx, q / dgx, da.dY, da.ldy, da.X, da.ldx, da.R, da.O, da.K, da.rows_per_slab, da.partial);
    }
}

static __global__ __launch_bounds__(256) void dw_gemm_final(const float* __restrict__ partial, int n_slabs, int O, int K,
                                                            float* __restrict__ dW, float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_out = O * (K + 1);
    if (i >= n_out) return;
    float acc = 0.f;
    for (int c0 = 0; c0 < n_slabs; c0 += 16) {  // 16 loads in flight, added in slab order
        float t16[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t16[q] = (c0 + q < n_slabs) ? partial[(size_t)(c0 + q) * n_out + i] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += t16[q];
    }
    const int o = i / (K + 1), k = i % (K + 1);
    if (k < K) dW[(size_t)o * K + k] = acc;
    else if (db) db[o] = acc;
}

// dW [O,K] (row-major, ld = K) and db [O] from dY [R,O] (ld ldy) and X [R,K] (ld ldx); `partial` >= dwg_partial_floats
static inline void launch_dw_gemm(const float* dY, int ldy, const float* X, int ldx, int R, int O, int K, float* dW, float* db,
                                  float* partial, hipStream_t s) {
    const int slabs = dwg_slabs(R);
    int rows_per_slab = (R + slabs - 1) / slabs;
    rows_per_slab = (rows_per_slab + 15) & ~15;  // multiple of 16 (a batch of the kernel = 16 MFMA steps = 32 rows, the tail is masked)
    const int tiles = cdiv(O, 32) * cdiv(K, 32);
    hipLaunchKernelGGL(dw_gemm_kernel, dim3(tiles, slabs), dim3(64), 0, s, dY, ldy, X, ldx, R, O, K, rows_per_slab, partial);
    hipLaunchKernelGGL(dw_gemm_final, dim3(cdiv(O * (K + 1), 256)), dim3(256), 0, s, partial, slabs, O, K, dW, db);
}


// ---- several dW problems over the SAME rows (the PPO trunk/critic trio): row-slab partials written by
// trunk_bwd_kernel (ppo.hip), summed in slab order here ---------------------------------------------------------
struct DwJob {
    const float* dY; int ldy; const float* X; int ldx; int O; int K;
    float* dW; float* db; int tile_begin; int part_off;  // first tile id / float offset of this job's partial slabs
};
constexpr int kMaxDwJobs = 4;
struct DwJobs { DwJob j[kMaxDwJobs]; int n; int total_tiles; int total_out; };

// sum of the slab partials of flat output element i of job jb (fixed slab order); device-side twin of dw_multi_final
__device__ __forceinline__ float dw_multi_fetch(int n_out, int part_off, int n_slabs, const float* __restrict__ partial, int i) {
    float acc = 0.f;
    for (int c0 = 0; c0 < n_slabs; c0 += 32) {  // all loads of a batch in flight (a 1024-row minibatch: its 32 slabs in one round trip), added in slab order
        float t32[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) t32[q] = (c0 + q < n_slabs) ? partial[part_off + (size_t)(c0 + q) * n_out + i] : 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) acc += t32[q];
    }
    return acc;
}
__device__ __forceinline__ float dw_multi_fetch(const DwJob& jb, int n_slabs, const float* __restrict__ partial, int i) {
    return dw_multi_fetch(jb.O * (jb.K + 1), jb.part_off, n_slabs, partial, i);
}

static __global__ __launch_bounds__(256) void dw_multi_final(DwJobs jobs, int n_slabs, const float* __restrict__ partial) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= jobs.total_out) return;
    int ji = 0;
    for (; ji < jobs.n; ++ji) {
        const int n_out = jobs.j[ji].O * (jobs.j[ji].K + 1);
        if (i < n_out) break;
        i -= n_out;
    }
    const DwJob jb = jobs.j[ji];
    const int n_out = jb.O * (jb.K + 1);
    const float acc = dw_multi_fetch(jb, n_slabs, partial, i);
    const int o = i / (jb.K + 1), k = i % (jb.K + 1);
    if (k < jb.K) jb.dW[(size_t)o * jb.K + k] = acc;
    else if (jb.db) jb.db[o] = acc;
}

// ---- a whole backward pass worth of dW problems: slab partials per problem, ONE final launch -----------------------
// launch_dw_partial() runs dw_gemm_kernel into the problem's own region of `partial` and records the job; dw_list_final sums
// the slabs of every recorded problem in slab order (diag: only the diagonal of a square problem is kept -- LayerNorm weight
// gradients are the diagonal of dY^T Xhat).
constexpr int kMaxDwList = 3 + 6 * CIRS_MAX_TRACKER_LAYERS;   // the tracker backward pass: decoder, user ffn, gate + 6 per layer
struct DwListJob { int O, K, part_off, diag; float* dW; float* db; };
struct DwList { DwListJob j[kMaxDwList]; int n, total_out, part_floats; };

static inline void launch_dw_partial(DwList& list, const float* dY, int ldy, const float* X, int ldx, int R, int O, int K, float* dW,
                                     float* db, int diag, float* partial, hipStream_t s) {
    const int slabs = dwg_slabs(R);
    int rows_per_slab = (R + slabs - 1) / slabs;
    rows_per_slab = (rows_per_slab + 15) & ~15;
    DwListJob& jb = list.j[list.n++];
    jb.O = O; jb.K = K; jb.part_off = list.part_floats; jb.diag = diag; jb.dW = dW; jb.db = db;
    list.part_floats += slabs * O * (K + 1);
    list.total_out += O * (K + 1);
    const int tiles = cdiv(O, 32) * cdiv(K, 32);
    hipLaunchKernelGGL(dw_gemm_kernel, dim3(tiles, slabs), dim3(64), 0, s, dY, ldy, X, ldx, R, O, K, rows_per_slab, partial + jb.part_off);
}

// launch_dw_partial(list, dY, ..) + launch_rows_gemm(nt, ..) as one launch (see rows_gemm_dw_kernel)
static inline void launch_rows_gemm_dw(DwList& list, const float* dwX, int dw_ldx, int dwO, int dwK, float* dW, float* db, float* partial,
                                       bool nt, const float* X, int ldx, const float* W, int ldw, const float* bias, int R, int Kd, int N,
                                       int relu, const float* relu_of, int accumulate, float* Y, int ldy, hipStream_t s) {
    const int slabs = dwg_slabs(R);
    int rows_per_slab = (R + slabs - 1) / slabs;
    rows_per_slab = (rows_per_slab + 15) & ~15;
    DwListJob& jb = list.j[list.n++];
    jb.O = dwO; jb.K = dwK; jb.part_off = list.part_floats; jb.diag = 0; jb.dW = dW; jb.db = db;
    list.part_floats += slabs * dwO * (dwK + 1);
    list.total_out += dwO * (dwK + 1);
    const int tiles = cdiv(dwO, 32) * cdiv(dwK, 32);
    const RowsGemmArgs ra{X, ldx, W, ldw, bias, R, Kd, N, relu, relu_of, accumulate, Y, ldy};
    const DwGemmArgs da{X, ldx, dwX, dw_ldx, R, dwO, dwK, rows_per_slab, partial + jb.part_off};   // dY of the dW problem = X of the row GEMM
    const int rgx = cdiv(R, 32), rgy = cdiv(N, 32);
    if (nt) hipLaunchKernelGGL(rows_gemm_dw_kernel<true>, dim3(rgx * rgy + tiles * slabs), dim3(64), 0, s, ra, rgx, rgy, da, tiles);
    else hipLaunchKernelGGL(rows_gemm_dw_kernel<false>, dim3(rgx * rgy + tiles * slabs), dim3(64), 0, s, ra, rgx, rgy, da, tiles);
}

// 8 lanes per output element: lane q of the group adds slabs q*ceil(n/8) .. in slab order, the eight partial sums are then added in lane
// order (fixed shape: deterministic); one thread per element left 100 workgroups walking 256 slabs each (17 us for 25 MB)
static __global__ __launch_bounds__(256) void dw_list_final(DwList list, int n_slabs, const float* __restrict__ partial) {
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, q = threadIdx.x & 7;
    int i = g;
    const bool live = i < list.total_out;
    if (!live) i = 0;
    // the element's problem: a compile-time walk over the by-value list (a per-thread index would send the whole array through scratch)
    int O = 1, K = 0, part_off = 0, diag = 0;
    float* dW = nullptr;
    float* db = nullptr;
    bool found = false;
#pragma unroll
    for (int ji = 0; ji < kMaxDwList; ++ji) {
        if (ji < list.n && !found) {
            const int n_out_j = list.j[ji].O * (list.j[ji].K + 1);
            if (i < n_out_j) {
                O = list.j[ji].O; K = list.j[ji].K; part_off = list.j[ji].part_off; diag = list.j[ji].diag; dW = list.j[ji].dW; db = list.j[ji].db;
                found = true;
            } else {
                i -= n_out_j;
            }
        }
    }
    const int n_out = O * (K + 1);
    const int per = (n_slabs + 7) >> 3;
    const int c_beg = q * per, c_end = min(n_slabs, c_beg + per);
    float acc = 0.f;
    for (int c0 = c_beg; c0 < c_end; c0 += 16) {  // 16 loads in flight, added in slab order
        float t16[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t16[u] = (c0 + u < c_end) ? partial[part_off + (size_t)(c0 + u) * n_out + i] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += t16[u];
    }
    // the group's eight partial sums in lane order
    float tot = __shfl(acc, (threadIdx.x & 63) & 56, CIRS_WAVE);
#pragma unroll
    for (int u = 1; u < 8; ++u) tot += __shfl(acc, ((threadIdx.x & 63) & 56) | u, CIRS_WAVE);
    if (!live || q != 0) return;
    const int o = i / (K + 1), k = i % (K + 1);
    if (k < K) {
        if (!diag) dW[(size_t)o * K + k] = tot;
        else if (o == k) dW[o] = tot;
    } else if (db) {
        db[o] = tot;
    }
}

static inline void launch_dw_list_final(const DwList& list, int R, const float* partial, hipStream_t s) {
    hipLaunchKernelGGL(dw_list_final, dim3(cdiv(list.total_out * 8, 256)), dim3(256), 0, s, list, dwg_slabs(R), partial);
}

// slab-partial floats of one problem (for workspace sizing)
__host__ inline size_t dw_list_floats(long R, int O, int K) { return (size_t)dwg_slabs(R) * O * (K + 1); }

}  // namespace cirs
