// Scratch check: operand layout of v_mfma_f32_32x32x16_bf16 on gfx950 and the accuracy of an fp32 product assembled from
// three bf16 pieces per operand (6 MFMAs: hh, hm, mh, mm, hl, lh) against the fp32 MFMA and an fp64 host result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// C[32 x 32] = A[32 x 64] * B^T, B given as [32 x 64] (row n holds B[n][k]); one wave
__global__ void k(const float* A, const float* B, float* c6, float* c32) {
    const int lane = threadIdx.x, lo = lane & 31, hi = lane >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < 4; ++s) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * s + 8 * hi + j;
            __bf16 h, m, l;
            split3(A[lo * 64 + kk], h, m, l); ah[j] = h; am[j] = m; al[j] = l;
            split3(B[lo * 64 + kk], h, m, l); bh[j] = h; bm[j] = m; bl[j] = l;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    f32x16 ref; for (int r = 0; r < 16; ++r) ref[r] = 0.f;
    for (int kk = 0; kk < 32; ++kk) ref = __builtin_amdgcn_mfma_f32_32x32x2f32(A[lo * 64 + 2 * kk + hi], B[lo * 64 + 2 * kk + hi], ref, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;  // C[row][col = lo]: row indexes A, col indexes B
        c6[row * 32 + lo] = acc[r];
        c32[row * 32 + lo] = ref[r];
    }
}
int main() {
    std::vector<float> A(32 * 64), B(32 * 64), c6(1024), c32(1024);
    srand(1);
    for (auto& x : A) x = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto& x : B) x = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
    float *dA, *dB, *d6, *d32;
    hipMalloc(&dA, 8192); hipMalloc(&dB, 8192); hipMalloc(&d6, 4096); hipMalloc(&d32, 4096);
    hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d6, d32);
    hipMemcpy(c6.data(), d6, 4096, hipMemcpyDeviceToHost); hipMemcpy(c32.data(), d32, 4096, hipMemcpyDeviceToHost);
    double e6 = 0, e32 = 0, scale = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double t = 0, ta = 0;
        for (int kk = 0; kk < 64; ++kk) { t += (double)A[i * 64 + kk] * B[j * 64 + kk]; ta += fabs((double)A[i * 64 + kk] * B[j * 64 + kk]); }
        e6 = fmax(e6, fabs(c6[i * 32 + j] - t) / ta); e32 = fmax(e32, fabs(c32[i * 32 + j] - t) / ta); scale = fmax(scale, ta);
    }
    printf("max |err| / sum|a b|: bf16x6 %.3e   fp32 mfma %.3e   (2^-24 = %.3e)\n", e6, e32, ldexp(1.0, -24));
    return 0;
}
