// dense_small.h -- small dense helpers shared by ppo.hip and tracker_bwd.hip.
//
// dW[o,k] = sum_r dY[r,o] * X[r,k]  (k == K is the bias column: db[o] = sum_r dY[r,o]) for O, K <= 129 and thousands
// of rows.  Stage 1: each workgroup stages a 32-row slab of dY and X in LDS (coalesced loads) and every thread
// accumulates its outputs from LDS; stage 2 sums the slab partials in slab order.  Two launches, fixed order, no
// atomics -- the order (and therefore the bits) does not depend on the launch geometry.
#pragma once
#include "common.h"

namespace cirs {

constexpr int kDwRows = 32;       // rows staged in LDS at a time
constexpr int kDwMaxChunks = 64;  // at most this many partial slabs (each workgroup walks several 32-row slabs)

__host__ inline int dw_slabs(long R) { return (int)((R + kDwRows - 1) / kDwRows); }
__host__ inline int dw_chunks(long R) { const int s = dw_slabs(R); return s < kDwMaxChunks ? s : kDwMaxChunks; }
__host__ inline size_t dw_partial_floats(long R, int O, int K) { return (size_t)dw_chunks(R) * O * (K + 1); }

static __global__ __launch_bounds__(256) void dw_partial_kernel(const float* __restrict__ dY, const float* __restrict__ X, int R,
                                                                int O, int K, int slabs_per_chunk, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sY = sm;                        // [kDwRows][O]
    float* sX = sm + (size_t)kDwRows * O;  // [kDwRows][K]
    const int c = blockIdx.y;
    const int n_out = O * (K + 1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one output per thread
    const int o = i / (K + 1), k = i % (K + 1);
    float acc = 0.f;
    for (int sl = 0; sl < slabs_per_chunk; ++sl) {
        const int r0 = (c * slabs_per_chunk + sl) * kDwRows;
        if (r0 >= R) break;
        const int nr = min(kDwRows, R - r0);
        __syncthreads();
        for (int j = threadIdx.x; j < nr * O; j += blockDim.x) sY[j] = dY[(size_t)r0 * O + j];
        for (int j = threadIdx.x; j < nr * K; j += blockDim.x) sX[j] = X[(size_t)r0 * K + j];
        __syncthreads();
        if (i < n_out) {
            if (k < K) for (int r = 0; r < nr; ++r) acc = __builtin_fmaf(sY[r * O + o], sX[r * K + k], acc);
            else for (int r = 0; r < nr; ++r) acc += sY[r * O + o];
        }
    }
    if (i < n_out) partial[(size_t)c * n_out + i] = acc;
}

static __global__ __launch_bounds__(256) void dw_final_kernel(const float* __restrict__ partial, int n_chunks, int O, int K,
                                                              float* __restrict__ dW, float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_out = O * (K + 1);
    if (i >= n_out) return;
    float acc = 0.f;
    for (int c = 0; c < n_chunks; ++c) acc += partial[(size_t)c * n_out + i];
    const int o = i / (K + 1), k = i % (K + 1);
    if (k < K) dW[(size_t)o * K + k] = acc;
    else if (db) db[o] = acc;
}

// launches both stages; `partial` must hold dw_partial_floats(R, O, K) floats
static inline int launch_dw(const float* dY, const float* X, int R, int O, int K, float* dW, float* db, float* partial,
                            hipStream_t s) {
    const int n_out = O * (K + 1), chunks = dw_chunks(R);
    const int slabs_per_chunk = (dw_slabs(R) + chunks - 1) / chunks;
    const size_t shmem = sizeof(float) * (size_t)kDwRows * (O + K);
    hipLaunchKernelGGL(dw_partial_kernel, dim3(cdiv(n_out, 256), chunks), dim3(256), shmem, s, dY, X, R, O, K, slabs_per_chunk, partial);
    hipLaunchKernelGGL(dw_final_kernel, dim3(cdiv(n_out, 256)), dim3(256), 0, s, partial, chunks, O, K, dW, db);
    return CIRS_OK;
}

}  // namespace cirs
