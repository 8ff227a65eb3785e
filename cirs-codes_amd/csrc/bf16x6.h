// bf16x6.h -- fp32 products on the bf16 matrix pipe of gfx950 (shared by the PPO head kernels and the DeepFM sweep).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cirs {

typedef float f32x16_b __attribute__((ext_vector_type(16)));

// ---- fp32 products on the bf16 matrix pipe ("bf16x6") -------------------------------------------------------------
// On gfx950 v_mfma_f32_32x32x2_f32 retires 2 k per 64 cycles and v_mfma_f32_32x32x16_bf16 16 k per 32 cycles, and neither
// overlaps with VALU work of any wave of the SIMD (tools/probes/overlap_probe.hip): every cycle spent in an MFMA is a cycle
// of the kernel.  An fp32 value is the exact sum of three bf16 pieces h + m + l (8 significand bits each), so
//   a * b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh) + O(2^-24 |a b|)
// is six bf16 MFMAs per 16 k (192 cycles) instead of eight fp32 MFMAs (512 cycles), accumulated in fp32 by the matrix
// core, with the error of an fp32 product chain (tools/probes/bf16x6_probe.hip: 1.2e-7 vs 1.0e-7 of sum|a b|).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t pk4 __attribute__((ext_vector_type(4)));
struct Planes { bf16x8 h, m, l; };

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo_val, float hi_val) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_val), "v"(hi_val));
    return r;
}
// two fp32 values -> packed (h, m, l) pieces; the first value sits in the low half.  The two residual subtractions of a level
// are written on a 2-vector so that they become ONE v_pk_add_f32 (9 VALU ops per pair instead of 11).
typedef float f32x2_b __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = cvt_pk_bf16(a, b);
    const f32x2_b x = {a, b}, hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    const f32x2_b r = x - hf;
    m = cvt_pk_bf16(r.x, r.y);
    const f32x2_b mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    const f32x2_b t = r - mf;
    l = cvt_pk_bf16(t.x, t.y);
}
__device__ __forceinline__ Planes split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
    // the four pairs advance level by level: every v_cvt_pk_bf16_f32 has three independent ones behind it before its result is
    // consumed (written pair after pair the compiler left an s_nop behind each conversion)
    const f32x2_b x[4] = {{x0, x1}, {x2, x3}, {x4, x5}, {x6, x7}};
    uint32_t hh[4], mm[4], ll[4];
    f32x2_b r[4], t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hh[i] = cvt_pk_bf16(x[i].x, x[i].y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_b hf = {__uint_as_float(hh[i] << 16), __uint_as_float(hh[i] & 0xffff0000u)};
        r[i] = x[i] - hf;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) mm[i] = cvt_pk_bf16(r[i].x, r[i].y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_b mf = {__uint_as_float(mm[i] << 16), __uint_as_float(mm[i] & 0xffff0000u)};
        t[i] = r[i] - mf;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ll[i] = cvt_pk_bf16(t[i].x, t[i].y);
    const pk4 h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
    Planes o;
    o.h = __builtin_bit_cast(bf16x8, h); o.m = __builtin_bit_cast(bf16x8, m); o.l = __builtin_bit_cast(bf16x8, l);
    return o;
}
__device__ __forceinline__ Planes split8(const f32x16_b& v, int base) {
    return split8(v[base], v[base + 1], v[base + 2], v[base + 3], v[base + 4], v[base + 5], v[base + 6], v[base + 7]);
}
// c += a * b over 16 k; small terms first
__device__ __forceinline__ f32x16_b mfma_bf16x6(const Planes& a, const Planes& b, f32x16_b c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}
// The same product with the accumulation split BY MAGNITUDE: `big` takes the h*h term of every k-step, `small` the five cross terms (<= 2^-8 of
// the big ones).  Every MFMA rounds its result into the fp32 accumulator; in one chain that is 6 roundings per 16 k at the magnitude of the
// running sum, here one (the cross terms round at their own, 2^-8 smaller, magnitude): over K = 64 the logits carry ~2 instead of ~5 ulp of
// |z| -- what separates a float32 soft-max of a sharp policy from a worse one (tests/test_gpu_head_precision.py, sharp = 6).  The caller adds
// big + small once at the end.
__device__ __forceinline__ void mfma_bf16x6_split(const Planes& a, const Planes& b, f32x16_b& big, f32x16_b& small) {
    big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, big, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, small, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, small, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, small, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, small, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, small, 0, 0, 0);
}
// Two k-steps at once with TWO cross-term accumulators, issued so that no MFMA follows one it depends on (a dependent bf16 MFMA waits ~8 cycles
// longer for its predecessor): for kernels whose MFMA groups run back to back without vector work in between.  z = big + (sm0 + sm1).
__device__ __forceinline__ void mfma_bf16x6_split2(const Planes& a0, const Planes& b0, const Planes& a1, const Planes& b1, f32x16_b& big, f32x16_b& sm0,
                                                   f32x16_b& sm1) {
    sm0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.l, b0.h, sm0, 0, 0, 0);
    sm1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.l, b1.h, sm1, 0, 0, 0);
    big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b0.h, big, 0, 0, 0);
    sm0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b0.l, sm0, 0, 0, 0);
    sm1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b1.l, sm1, 0, 0, 0);
    big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b1.h, big, 0, 0, 0);
    sm0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.m, b0.m, sm0, 0, 0, 0);
    sm1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.m, b1.m, sm1, 0, 0, 0);
    sm0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.m, b0.h, sm0, 0, 0, 0);
    sm1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.m, b1.h, sm1, 0, 0, 0);
    sm0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b0.m, sm0, 0, 0, 0);
    sm1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b1.m, sm1, 0, 0, 0);
}
// two independent accumulators sharing the A operand, issued alternately: a dependent bf16 MFMA waits ~40 cycles for its
// predecessor, an independent one issues after 32
__device__ __forceinline__ void mfma_bf16x6_pair(const Planes& a, const Planes& b0, const Planes& b1, f32x16_b& c0, f32x16_b& c1) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b0.h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b1.h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b0.l, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b1.l, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b0.m, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b1.m, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b0.h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b1.h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b0.m, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b1.m, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b0.h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b1.h, c1, 0, 0, 0);
}
// two independent accumulators sharing the B operand
__device__ __forceinline__ void mfma_bf16x6_pair_b(const Planes& a0, const Planes& a1, const Planes& b, f32x16_b& c0, f32x16_b& c1) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.l, b.h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.l, b.h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b.l, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b.l, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.m, b.m, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.m, b.m, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.m, b.h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.m, b.h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b.m, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b.m, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b.h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b.h, c1, 0, 0, 0);
}

// ---- fp32 products from TWO fp16 pieces per operand ("f16x3", round 6 experiment) ------------------------------------
// x = h + l with h = fp16(x), l = fp16(x - h): round-to-nearest pieces of 11 significant bits each leave |x - h - l| <= 2^-24 |x| -- fp32's own half ulp --
// as long as both pieces are NORMAL fp16 numbers (|piece| >= 6.1e-5; below that a piece keeps 6e-8 of ABSOLUTE resolution).  a b = ah bh + (ah bl + al bh)
// + O(2^-24 |a b|): THREE MFMAs per 16 k instead of six, one split level instead of two.  The price is the 5-bit exponent: operands are pre-scaled by
// exact powers of two into [2^-2, 2^15] (callers: ppo.hip kScWa / kScH2 / the per-row-block dZ scale) and results unscaled exactly.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_b __attribute__((ext_vector_type(2)));
struct Planes2 { f16x8 h, l; };
__device__ __forceinline__ uint32_t cvt_pk_f16(float lo_val, float hi_val) {
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_val), "v"(hi_val));
    return r;
}
__device__ __forceinline__ Planes2 split8h(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
    const f32x2_b x[4] = {{x0, x1}, {x2, x3}, {x4, x5}, {x6, x7}};
    uint32_t hh[4], ll[4];
    f32x2_b r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hh[i] = cvt_pk_f16(x[i].x, x[i].y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2_b hv = __builtin_bit_cast(f16x2_b, hh[i]);
        const f32x2_b hf = {(float)hv.x, (float)hv.y};
        r[i] = x[i] - hf;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ll[i] = cvt_pk_f16(r[i].x, r[i].y);
    const pk4 h = {hh[0], hh[1], hh[2], hh[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
    Planes2 o;
    o.h = __builtin_bit_cast(f16x8, h); o.l = __builtin_bit_cast(f16x8, l);
    return o;
}
__device__ __forceinline__ Planes2 split8h(const f32x16_b& v, int base) {
    return split8h(v[base], v[base + 1], v[base + 2], v[base + 3], v[base + 4], v[base + 5], v[base + 6], v[base + 7]);
}
// two k-steps, accumulation split by magnitude like mfma_bf16x6_split2: big = the h*h terms, sm0 / sm1 = the cross terms of k-step 0 / 1
__device__ __forceinline__ void mfma_f16x3_split2(const Planes2& a0, const Planes2& b0, const Planes2& a1, const Planes2& b1, f32x16_b& big, f32x16_b& sm0,
                                                  f32x16_b& sm1) {
    sm0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.l, b0.h, sm0, 0, 0, 0);
    sm1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.l, b1.h, sm1, 0, 0, 0);
    big = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, b0.h, big, 0, 0, 0);
    sm0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, b0.l, sm0, 0, 0, 0);
    sm1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, b1.l, sm1, 0, 0, 0);
    big = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, b1.h, big, 0, 0, 0);
}
// two independent accumulators sharing the A operand (small terms first)
__device__ __forceinline__ void mfma_f16x3_pair(const Planes2& a, const Planes2& b0, const Planes2& b1, f32x16_b& c0, f32x16_b& c1) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l, b0.h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l, b1.h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b0.l, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b1.l, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b0.h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b1.h, c1, 0, 0, 0);
}
// accumulator register r of a 32 x 32 tile, lane half hi -> row (C/D layout of the 32x32 MFMAs)
__device__ __forceinline__ constexpr int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }


}  // namespace cirs
