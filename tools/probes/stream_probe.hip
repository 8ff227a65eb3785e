// Scratch microbenchmark for the "one workgroup per env, all T steps in one launch" rollout: what does it cost a workgroup to
// stream its share of the actor head's weight table (C2: 3327 x 64 fp32 = 852 KB, L2 resident) once per vector step, T steps in a row, with a dependent
// reduction between the steps?  Loads are float4, `depth` of them in flight per thread; `split` workgroups share one env's table (213 KB each at 4).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/probes/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int kDepth>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ table, int n_vec4, int split, int steps, float* __restrict__ out) {
    const int part = blockIdx.x % split;
    const int per = (n_vec4 + split - 1) / split;
    const int lo = part * per, hi = min(n_vec4, lo + per);
    float carry = 0.f;
    __shared__ float red[4];
    for (int t = 0; t < steps; ++t) {
        float4 acc = make_float4(carry, 0.f, 0.f, 0.f);
        float4 buf[kDepth];
        int i = lo + threadIdx.x;
#pragma unroll
        for (int d = 0; d < kDepth; ++d) buf[d] = (i + d * 256 < hi) ? table[i + d * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (; i < hi; i += kDepth * 256) {
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                const float4 v = buf[d];
                const int nx = i + (kDepth + d) * 256;
                buf[d] = nx < hi ? table[nx] : make_float4(0.f, 0.f, 0.f, 0.f);
                acc.x = fmaf(v.x, 1.0001f, acc.x); acc.y = fmaf(v.y, 1.0001f, acc.y); acc.z = fmaf(v.z, 1.0001f, acc.z); acc.w = fmaf(v.w, 1.0001f, acc.w);
            }
        }
        float s = (acc.x + acc.y) + (acc.z + acc.w);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        carry = ((red[0] + red[1]) + (red[2] + red[3])) * 1e-9f;      // the next step depends on this one (as the sampled action does)
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = carry;
}
template <int kDepth>
static void run(const float4* table, int n_vec4, int envs, int split, int steps, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(stream_kernel<kDepth>, dim3(envs * split), dim3(256), 0, 0, table, n_vec4, split, steps, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            const double bytes_per_wg_step = 16.0 * ((n_vec4 + split - 1) / split);
            const double us_step = ms * 1e3 / steps;
            printf("envs %3d x split %d, depth %2d: %7.2f us per step, %6.1f KB per workgroup and step -> %5.1f B/cycle per CU at 2.4 GHz, %6.2f TB/s over %d CUs\n", envs, split,
                   kDepth, us_step, bytes_per_wg_step / 1024, bytes_per_wg_step / (us_step * 2400.0), bytes_per_wg_step * envs * split / (us_step * 1e6), envs * split);
        }
    }
}
int main() {
    const int n_items = 3327, kH = 64;
    const int n_vec4 = n_items * kH / 4;
    std::vector<float> h((size_t)n_vec4 * 4, 0.001f);
    float4* table; float* out;
    (void)hipMalloc(&table, (size_t)n_vec4 * 16); (void)hipMemcpy(table, h.data(), (size_t)n_vec4 * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 4096 * 4);
    const int steps = 30;
    for (int split : {1, 2, 4}) {
        run<2>(table, n_vec4, 64, split, steps, out);
        run<4>(table, n_vec4, 64, split, steps, out);
        run<8>(table, n_vec4, 64, split, steps, out);
        run<16>(table, n_vec4, 64, split, steps, out);
    }
    // the larger catalogue at 1024 envs: one workgroup per env cannot be co-resident 4 per CU and stream 2.7 MB each -- shown for scale at 256 envs
    return 0;
}
