// permutation.h -- the keyed pseudo-random permutation behind the minibatch shuffle (Batch.split(shuffle=True), tianshou/data/batch.py:734-744) and the
// splitmix64 finaliser of the feature hashing: device / host functions shared by hashing.hip (stand-alone entry points) and ppo.hip (the update's
// permutations inside process_fn's last launch).  The oracle restates them bit for bit (oracle/cirs_oracle.c).
#pragma once
#include "common.h"

namespace cirs {

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
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
// keys of up to kMaxPermKeys permutations by value (an indexed by-value array would go through scratch: select chain at the use)
constexpr int kMaxPermKeys = 8;
struct PermKeys { uint64_t k[kMaxPermKeys]; };
__host__ __device__ __forceinline__ uint64_t perm_key_of(const PermKeys& keys, int c) {
    uint64_t key = keys.k[0];
#pragma unroll
    for (int q = 1; q < kMaxPermKeys; ++q) key = c == q ? keys.k[q] : key;
    return key;
}
__host__ __device__ __forceinline__ int perm_half_bits(uint64_t n) {
    int h = 1;
    while ((1ull << (2 * h)) < n) ++h;
    return h;
}
__host__ inline uint64_t perm_key(uint64_t seed, uint64_t tag) { return splitmix64(seed ^ splitmix64(tag)); }

}  // namespace cirs
