// Does hipExtAnyOrderLaunch let a kernel start before its predecessor in the same stream has finished on gfx950?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int* out) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main() {
    int* out; (void)hipMalloc(&out, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int flag : {0, 1}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, 0, 100000LL, out);
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, 0, nullptr, nullptr, flag, 100000LL, out + 1);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("second launch flags=%d: both kernels take %.1f us\n", flag, best * 1e3);
    }
    return 0;
}
