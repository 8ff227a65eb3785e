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

// ---- pseudo-random permutation of [0, n) without a sort ------------------------------------------------------------
// out[i] = P(i) where P is a keyed bijection: a 6-round balanced Feistel network on 2h bits (2^(2h) >= n, < 4n) with a
// splitmix64 round function, restricted to [0, n) by cycle walking (re-encrypt while the value is >= n; <= 4 expected rounds).
// One thread per element, no host round trip: replaces np.random.permutation / torch.randperm for the minibatch shuffle.
__host__ __device__ __forceinline__ uint64_t feistel_encrypt(uint64_t x, int h, uint64_t key) {
    const uint64_t mask = (1ull << h) - 1;
    uint64_t l = x >> h, r = x & mask;
#pragma unroll
    for (int round = 0; round < 6; ++round) {
        const uint64_t f = splitmix64(r ^ (key + 0x632BE59BD9B4E019ull * (uint64_t)(round + 1))) & mask;
        const uint64_t nl = r;
        r = l ^ f;
        l = nl;
    }
    return (l << h) | r;
}
__host__ __device__ __forceinline__ int64_t permute_index(int64_t i, int64_t n, int h, uint64_t key) {
    uint64_t x = (uint64_t)i;
    do { x = feistel_encrypt(x, h, key); } while (x >= (uint64_t)n);
    return (int64_t)x;
}
__global__ __launch_bounds__(256) void permutation_kernel(long n, int h, uint64_t key, int32_t* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)permute_index(i, n, h, key);
}

// `count` permutations of [0, n) (keys by value), out[c][i]: one launch for all the repeats of an update
constexpr int kMaxPermKeys = 8;
struct PermKeys { uint64_t k[kMaxPermKeys]; };
__global__ __launch_bounds__(256) void permutations_kernel(long n, int h, PermKeys keys, int32_t* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    uint64_t key = keys.k[0];
#pragma unroll
    for (int q = 1; q < kMaxPermKeys; ++q) key = c == q ? keys.k[q] : key;      // (a select chain: an indexed by-value array would go through scratch)
    if (i < n) out[(size_t)c * n + i] = (int32_t)permute_index(i, n, h, key);
}

}  // namespace cirs

extern "C" int cirs_random_permutations(int64_t n, uint64_t seed, uint64_t tag0, int32_t count, int32_t* out, void* stream) {
    using namespace cirs;
    if (n <= 0 || count <= 0) return CIRS_OK;
    CIRS_REQUIRE(out && n <= 0x7FFFFFFF, "bad arguments");
    int h = 1;
    while ((1ull << (2 * h)) < (uint64_t)n) ++h;
    for (int c0 = 0; c0 < count; c0 += kMaxPermKeys) {
        PermKeys keys{};
        const int nc = count - c0 < kMaxPermKeys ? count - c0 : kMaxPermKeys;
        for (int q = 0; q < nc; ++q) keys.k[q] = splitmix64(seed ^ splitmix64(tag0 + (uint64_t)(c0 + q)));
        hipLaunchKernelGGL(permutations_kernel, dim3(cdiv(n, 256), nc), dim3(256), 0, (hipStream_t)stream, (long)n, h, keys, out + (size_t)c0 * n);
    }
    CIRS_CHECK_LAUNCH("permutations_kernel");
    return CIRS_OK;
}

extern "C" int cirs_random_permutation(int64_t n, uint64_t seed, uint64_t tag, int32_t* out, void* stream) {
    using namespace cirs;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(out && n <= 0x7FFFFFFF, "bad arguments");
    int h = 1;
    while ((1ull << (2 * h)) < (uint64_t)n) ++h;
    const uint64_t key = splitmix64(seed ^ splitmix64(tag));
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
