// hashing.hip -- feature hashing for open-vocabulary ids (BASELINE configs[4] "hashed features").
// The reference has no hashing (SparseFeat.use_hash only prints a notice, DeepCTR-Torch/deepctr_torch/inputs.py:31-33), so the
// semantics are this build's own: bucket = splitmix64(id) mod n_buckets, a fixed 64-bit finaliser.
#include "common.h"

namespace cirs {

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void hash_ids_kernel(const int64_t* __restrict__ ids, long n, uint64_t n_buckets, int64_t* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)(splitmix64((uint64_t)ids[i]) % n_buckets);
}

}  // namespace cirs

extern "C" int cirs_hash_ids(const int64_t* ids, int64_t n, int64_t n_buckets, int64_t* out, void* stream) {
    using namespace cirs;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(ids && out && n_buckets > 0, "bad arguments");
    hipLaunchKernelGGL(hash_ids_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, (long)n, (uint64_t)n_buckets, out);
    CIRS_CHECK_LAUNCH("hash_ids_kernel");
    return CIRS_OK;
}
