// common.h -- shared device/host helpers for libcirs_hip (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>

#include "../../include/cirs_hip.h"

#define CIRS_WAVE 64

namespace cirs {

void set_error(const std::string& msg);

inline int fail(int code, const std::string& msg) {
    set_error(msg);
    return code;
}

#define CIRS_REQUIRE(cond, msg)                                             \
    do {                                                                    \
        if (!(cond)) return ::cirs::fail(CIRS_E_INVALID, std::string(msg)); \
    } while (0)

#define CIRS_CHECK_LAUNCH(what)                                                                          \
    do {                                                                                                 \
        hipError_t _e = hipGetLastError();                                                               \
        if (_e != hipSuccess)                                                                            \
            return ::cirs::fail(CIRS_E_LAUNCH, std::string(what) + ": " + hipGetErrorString(_e));        \
    } while (0)

#define CIRS_HIP(call)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (call);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return ::cirs::fail(CIRS_E_LAUNCH, std::string(#call) + ": " + hipGetErrorString(_e));       \
    } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- wavefront (64-lane) reductions -------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, CIRS_WAVE);
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, CIRS_WAVE);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, CIRS_WAVE));
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, CIRS_WAVE);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, CIRS_WAVE);
    return v;
}

// ---- register-only wavefront reductions (DPP within a row of 16 lanes, v_readlane across the four rows): no LDS round trips.
// Pairing order: xor 1, xor 2, then the two quads / the two halves of a row (mirror permutations: after the xor steps all lanes
// of a quad / half hold the same value, so the mirror partner carries exactly the other group's sum), then (r0 + r1) + (r2 + r3).
// A balanced tree like the __shfl_xor butterfly above, but with a different pairing: sums may differ in the last bit, max is exact.
template <int kCtrl>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_pick(float v, int lane_id) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_id));
}
__device__ __forceinline__ float wave_sum_f32_dpp(float v) {
    v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);   // row_half_mirror
    v += dpp_f32<0x140>(v);   // row_mirror
    return (row_pick(v, 0) + row_pick(v, 16)) + (row_pick(v, 32) + row_pick(v, 48));
}
__device__ __forceinline__ float wave_max_f32_dpp(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return fmaxf(fmaxf(row_pick(v, 0), row_pick(v, 16)), fmaxf(row_pick(v, 32), row_pick(v, 48)));
}

// reductions inside a ROW of 16 lanes (four DPP steps, no cross-row traffic): afterwards every lane of the row holds the row's result
__device__ __forceinline__ float row16_sum_f32(float v) {
    v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); v += dpp_f32<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max_f32(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v)); v = fmaxf(v, dpp_f32<0x4E>(v)); v = fmaxf(v, dpp_f32<0x141>(v)); v = fmaxf(v, dpp_f32<0x140>(v));
    return v;
}
// sum over lanes 0..31 (the caller passes 0 in lanes >= 32 or ignores them): two rows
__device__ __forceinline__ float half_sum_f32_dpp(float v) {
    v = row16_sum_f32(v);
    return row_pick(v, 0) + row_pick(v, 16);
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it stalls every wave until its
// outstanding global stores/loads have completed (~1-2 us for a store).  Use where the data exchanged is in LDS and
// the global accesses in flight are either write-only results or prefetches consumed behind a later register dependency.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// category bitmask of an item's packed 4 x u8 category ids (CIRS_CAT_NONE = empty slot)
__device__ __forceinline__ unsigned long long cat_mask(uint32_t packed) {
    unsigned long long m = 0;
#pragma unroll
    for (int k = 0; k < CIRS_MAX_CATS_PER_ITEM; ++k) {
        uint32_t c = (packed >> (8 * k)) & 0xFFu;
        if (c != CIRS_CAT_NONE) m |= 1ull << (c & 63u);
    }
    return m;
}

// 1/Jaccard exactly as the reference computes it in float64: sim = |A&B|/|A|B|, dist = 1.0/sim (inf if disjoint)
__device__ __forceinline__ double jaccard_dist(uint32_t pa, uint32_t pb) {
    unsigned long long ma = cat_mask(pa), mb = cat_mask(pb);
    double inter = (double)__popcll(ma & mb);
    double uni = (double)__popcll(ma | mb);
    double sim = inter / uni;
    return 1.0 / sim;
}

// By-value kernel arguments are read through the scalar cache, which is cold at every launch; the compiler fetches them piecemeal
// at first use, so a kernel with a few hundred bytes of arguments starts with several DEPENDENT scalar-cache misses (~0.8 us
// each on MI355X) before its first vector load goes out.  One dword of every 64-byte line of the first kBytes of the kernarg
// segment, requested back to back at kernel entry, turns that into one round trip.  kBytes must not exceed the kernel's explicit
// argument bytes.
template <int kBytes>
__device__ __forceinline__ void kernarg_warm() {
    typedef const uint32_t __attribute__((address_space(4))) * kernarg_words;
    kernarg_words ka = (kernarg_words)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t warm = 0;
#pragma unroll
    for (int o = 0; o < kBytes / 4; o += 16) warm ^= ka[o];
    asm volatile("" ::"s"(warm));
}

}  // namespace cirs
