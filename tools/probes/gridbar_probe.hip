// Scratch microbenchmark: cost of a software grid barrier (atomic counter + spin) in a cooperative launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void grid_barrier(unsigned long long* counter, unsigned long long target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1ull);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k(unsigned long long* counter, unsigned long long base, int n_bar, float* out) {
    float acc = 0.f;
    for (int b = 0; b < n_bar; ++b) {
        acc += out[(blockIdx.x * 256 + threadIdx.x + b) & 65535];
        grid_barrier(counter, base + (unsigned long long)gridDim.x * (b + 1));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    unsigned long long* c; float* out;
    (void)hipMalloc(&c, 8); (void)hipMemset(c, 0, 8); (void)hipMalloc(&out, 65536 * 4 * 4); (void)hipMemset(out, 0, 65536 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    unsigned long long base = 0;
    for (int grid : {256, 512, 768}) for (int n_bar : {0, 1, 10, 100}) {
        void* args[] = {&c, &base, &n_bar, &out};
        (void)hipLaunchCooperativeKernel((void*)k, dim3(grid), dim3(256), args, 0, 0); base += (unsigned long long)grid * n_bar;
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipError_t rc = hipLaunchCooperativeKernel((void*)k, dim3(grid), dim3(256), args, 0, 0); base += (unsigned long long)grid * n_bar;
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("grid %d barriers %d: %.1f us total (rc=%d)\n", grid, n_bar, ms * 1e3, (int)rc);
    }
    return 0;
}
