// Scratch microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 for dependent chains, by chains per wave and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int kChains>
__global__ void k(float* out, int iters, float a, float b) {
    f32x16 acc[kChains];
    for (int c = 0; c < kChains; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < kChains; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < kChains; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int kChains>
void run(int waves_per_simd, float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * waves_per_simd), block(256);  // 4 waves per WG -> one per SIMD
    hipLaunchKernelGGL(k<kChains>, grid, block, 0, 0, d, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<kChains>, grid, block, 0, 0, d, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)iters * 8 * kChains;            // per wave
    const double flops = n_mfma * 4096 * 1024.0 * waves_per_simd;  // whole chip
    printf("chains %d waves/SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", kChains, waves_per_simd, ms,
           1e6 * ms / (n_mfma * waves_per_simd), flops / (ms * 1e-3) / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4 * 8);
    for (int w = 1; w <= 4; ++w) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
    return 0;
}
