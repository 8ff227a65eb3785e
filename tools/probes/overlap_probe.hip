// Scratch microbenchmark: does VALU work overlap with v_mfma_f32_32x32x2_f32 on gfx950 (a) inside one wave, when the
// independent VALU instructions sit between the MFMAs of a dependent chain, and (b) across two waves of one SIMD, one
// issuing only MFMAs and the other only VALU?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int kValuPerMfma, bool kMfma>
__device__ __forceinline__ void body(f32x16& acc, float (&x)[8], float a, float b, int iters) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (kMfma) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < kValuPerMfma; ++q) x[q & 7] = __builtin_fmaf(x[q & 7], a, b);
        }
    }
}

template <int kValuPerMfma, bool kMfma>
__global__ void intra(float* out, int iters, float a, float b) {
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x[8]; for (int q = 0; q < 8; ++q) x[q] = (float)q;
    body<kValuPerMfma, kMfma>(acc, x, a, b, iters);
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int q = 0; q < 8; ++q) s += x[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 8 waves per workgroup = 2 per SIMD: waves 0-3 run the MFMA chain, waves 4-7 kValuPerMfma VALU per (absent) MFMA slot
template <int kValuPerMfma, int kMode>  // mode 0: both, 1: only the MFMA waves work, 2: only the VALU waves work
__global__ __launch_bounds__(512) void inter(float* out, int iters, float a, float b) {
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x[8]; for (int q = 0; q < 8; ++q) x[q] = (float)q;
    const int wv = threadIdx.x >> 6;
    if (wv < 4) { if (kMode != 2) body<0, true>(acc, x, a, b, iters); }
    else { if (kMode != 1) body<kValuPerMfma, false>(acc, x, a, b, iters); }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int q = 0; q < 8; ++q) s += x[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10); hipDeviceSynchronize();
    hipEventRecord(e0); launch(2000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int V>
void run_intra(float* d) {
    const float both = time_ms([&](int it) { hipLaunchKernelGGL((intra<V, true>), dim3(256), dim3(256), 0, 0, d, it, 1.f, 1.f); });
    const float valu = time_ms([&](int it) { hipLaunchKernelGGL((intra<V, false>), dim3(256), dim3(256), 0, 0, d, it, 1.f, 1.f); });
    const float mfma = time_ms([&](int it) { hipLaunchKernelGGL((intra<0, true>), dim3(256), dim3(256), 0, 0, d, it, 1.f, 1.f); });
    printf("intra-wave  %2d VALU per MFMA: mfma only %.3f ms, valu only %.3f ms, interleaved %.3f ms (sum %.3f, max %.3f)\n", V, mfma,
           valu, both, mfma + valu, mfma > valu ? mfma : valu);
}
template <int V>
void run_inter(float* d) {
    const float both = time_ms([&](int it) { hipLaunchKernelGGL((inter<V, 0>), dim3(256), dim3(512), 0, 0, d, it, 1.f, 1.f); });
    const float mfma = time_ms([&](int it) { hipLaunchKernelGGL((inter<V, 1>), dim3(256), dim3(512), 0, 0, d, it, 1.f, 1.f); });
    const float valu = time_ms([&](int it) { hipLaunchKernelGGL((inter<V, 2>), dim3(256), dim3(512), 0, 0, d, it, 1.f, 1.f); });
    printf("inter-wave  %2d VALU per MFMA: mfma only %.3f ms, valu only %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n", V, mfma, valu,
           both, mfma + valu, mfma > valu ? mfma : valu);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run_intra<4>(d); run_intra<8>(d); run_intra<12>(d); run_intra<16>(d); run_intra<24>(d);
    run_inter<4>(d); run_inter<8>(d); run_inter<12>(d); run_inter<16>(d); run_inter<24>(d);
    return 0;
}
