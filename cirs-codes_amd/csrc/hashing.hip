// hashing.hip -- feature hashing for open-vocabulary ids (BASELINE configs[4] "hashed features").
// The reference has no hashing (SparseFeat.use_hash only prints a notice, DeepCTR-Torch/deepctr_torch/inputs.py:31-33), so the
// semantics are this build's own: bucket = splitmix64(id) mod n_buckets, a fixed 64-bit finaliser.
#include "common.h"
#include "permutation.h"

namespace cirs {

__global__ __launch_bounds__(256) void hash_ids_kernel(const int64_t* __restrict__ ids, long n, uint64_t n_buckets, int64_t* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)(splitmix64((uint64_t)ids[i]) % n_buckets);
}

__global__ __launch_bounds__(256) void permutation_kernel(long n, int h, uint64_t key, int32_t* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)permute_index(i, n, h, key);
}

// `count` permutations of [0, n) (keys by value), out[c][i]: one launch for all the repeats of an update
__global__ __launch_bounds__(256) void permutations_kernel(long n, int h, PermKeys keys, int32_t* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const uint64_t key = perm_key_of(keys, c);
    if (i < n) out[(size_t)c * n + i] = (int32_t)permute_index(i, n, h, key);
}

}  // namespace cirs

extern "C" int cirs_random_permutations(int64_t n, uint64_t seed, uint64_t tag0, int32_t count, int32_t* out, void* stream) {
    using namespace cirs;
    if (n <= 0 || count <= 0) return CIRS_OK;
    CIRS_REQUIRE(out && n <= 0x7FFFFFFF, "bad arguments");
    const int h = perm_half_bits((uint64_t)n);
    for (int c0 = 0; c0 < count; c0 += kMaxPermKeys) {
        PermKeys keys{};
        const int nc = count - c0 < kMaxPermKeys ? count - c0 : kMaxPermKeys;
        for (int q = 0; q < nc; ++q) keys.k[q] = perm_key(seed, tag0 + (uint64_t)(c0 + q));
        hipLaunchKernelGGL(permutations_kernel, dim3(cdiv(n, 256), nc), dim3(256), 0, (hipStream_t)stream, (long)n, h, keys, out + (size_t)c0 * n);
    }
    CIRS_CHECK_LAUNCH("permutations_kernel");
    return CIRS_OK;
}

extern "C" int cirs_random_permutation(int64_t n, uint64_t seed, uint64_t tag, int32_t* out, void* stream) {
    using namespace cirs;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(out && n <= 0x7FFFFFFF, "bad arguments");
    const int h = perm_half_bits((uint64_t)n);
    const uint64_t key = perm_key(seed, tag);
    hipLaunchKernelGGL(permutation_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (long)n, h, key, out);
    CIRS_CHECK_LAUNCH("permutation_kernel");
    return CIRS_OK;
}

extern "C" int cirs_hash_ids(const int64_t* ids, int64_t n, int64_t n_buckets, int64_t* out, void* stream) {
    using namespace cirs;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(ids && out && n_buckets > 0, "bad arguments");
    hipLaunchKernelGGL(hash_ids_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, (long)n, (uint64_t)n_buckets, out);
    CIRS_CHECK_LAUNCH("hash_ids_kernel");
    return CIRS_OK;
}
