// Scratch microbenchmark: overlap of VALU fp32 FMAs with (a) the fp32 MFMA and (b) the bf16 MFMA of gfx950, inside one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int kKind>  // 0: none, 1: f32 32x32x2, 2: bf16 32x32x16
__device__ __forceinline__ void mm(f32x16& acc, float a, float b, bf16x8 pa, bf16x8 pb) {
    if (kKind == 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    if (kKind == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0);
}

template <int kValu, int kKind, int kLds>
__global__ void intra(float* out, int iters, float a, float b) {
    __shared__ float sh[256 * 4];
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x[8]; for (int q = 0; q < 8; ++q) x[q] = (float)q;
    bf16x8 pa, pb; for (int q = 0; q < 8; ++q) { pa[q] = (__bf16)a; pb[q] = (__bf16)b; }
    sh[threadIdx.x] = a; sh[threadIdx.x + 256] = b; sh[threadIdx.x + 512] = a; sh[threadIdx.x + 768] = b;
    __syncthreads();
    float l = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            mm<kKind>(acc, a, b, pa, pb);
#pragma unroll
            for (int q = 0; q < kValu; ++q) x[q & 7] = __builtin_fmaf(x[q & 7], a, b);
#pragma unroll
            for (int q = 0; q < kLds; ++q) l += sh[(threadIdx.x + 64 * q + u * 8 + i) & 1023];
        }
    }
    float s = l;
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
template <int V, int K, int L>
float t(float* d, int wg = 256) { return time_ms([&](int it) { hipLaunchKernelGGL((intra<V, K, L>), dim3(wg), dim3(256), 0, 0, d, it, 1.f, 1.f); }); }

template <int V, int L>
void row(float* d) {
    printf("%2d VALU + %d ds_read per MFMA slot | alone %.3f | f32 mfma alone %.3f with %.3f | bf16 mfma alone %.3f with %.3f\n", V, L,
           t<V, 0, L>(d), t<0, 1, 0>(d), t<V, 1, L>(d), t<0, 2, 0>(d), t<V, 2, L>(d));
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    row<2, 0>(d); row<4, 0>(d); row<8, 0>(d); row<16, 0>(d);
    row<0, 1>(d); row<0, 2>(d); row<0, 4>(d);
    printf("two waves per SIMD (512 workgroups): f32 mfma %.3f, bf16 mfma %.3f, bf16 + 8 VALU %.3f, 8 VALU %.3f\n", t<0, 1, 0>(d, 512), t<0, 2, 0>(d, 512),
           t<8, 2, 0>(d, 512), t<8, 0, 0>(d, 512));
    return 0;
}
