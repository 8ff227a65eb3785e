// rng.h -- counter-based sampler noise (device).  Philox4x32-10 (Salmon et al., SC'11) + a branch-free,
// fmaf-only natural log so that the Gumbel noise is BIT-REPRODUCIBLE on any IEEE-754 machine: the CPU oracle
// restates the same formulas and must produce identical bits (tests/test_gpu_policy.py).
//
// Why Gumbel-max: torch.multinomial(probs, 1) draws q_i ~ Exp(1) and returns argmax_i probs_i / q_i
// (the "exponential race").  argmax_i (logit_i - log q_i) is the same draw expressed on logits, so the sampler
// needs no normalised probabilities and fuses into the head GEMM epilogue (reference: core/policy/ppo.py:148-155).
#pragma once
#include <stdint.h>

namespace cirs {

struct u32x4 {
    uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return u32x4{c0, c1, c2, c3};
}

// natural log of a positive normal float; Cephes-style minimax polynomial evaluated with explicit fmaf in a fixed
// order (relative error ~1e-7).  No table, no division, no library call -> identical bits on CPU and GPU.
__host__ __device__ __forceinline__ float det_logf(float x) {
    uint32_t bits = __builtin_bit_cast(uint32_t, x);
    int e = (int)(bits >> 23) - 127;
    float m = __builtin_bit_cast(float, (bits & 0x007FFFFFu) | 0x3F800000u);  // [1,2)
    if (m > 1.41421356237f) {
        m = m * 0.5f;
        e += 1;
    }
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, f, -1.1514610310e-1f);
    p = __builtin_fmaf(p, f, 1.1676998740e-1f);
    p = __builtin_fmaf(p, f, -1.2420140846e-1f);
    p = __builtin_fmaf(p, f, 1.4249322787e-1f);
    p = __builtin_fmaf(p, f, -1.6668057665e-1f);
    p = __builtin_fmaf(p, f, 2.0000714765e-1f);
    p = __builtin_fmaf(p, f, -2.4999993993e-1f);
    p = __builtin_fmaf(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    const float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = f + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

// uniform in (0,1) with 23 random bits, exactly representable: (k + 0.5) * 2^-23
__host__ __device__ __forceinline__ float u01_from_bits(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-7f; }

__host__ __device__ __forceinline__ float gumbel_from_bits(uint32_t x) {
    const float u = u01_from_bits(x);
    return -det_logf(-det_logf(u));
}

// e^x for x <= 0, bit-reproducible like det_logf: y = x * log2(e) (one rounding), n = floor(y), g = (y - n) - 1/2 in [-1/2, 1/2),
// 2^(n + 1/2 + g) = 2^n * sqrt(2) * e^(g ln 2) with the degree-6 Taylor polynomial in g (coefficients sqrt(2) (ln 2)^k / k!, fmaf
// Horner), scaled by an exact ldexp.  Relative error <= 3e-6 (dominated by the rounding of y for |x| ~ 30), x < -87 -> 0.
__host__ __device__ __forceinline__ float det_expf_neg(float x) {
    // branch-free: out-of-range inputs (x < -87, -inf, NaN) run the polynomial on -87 and select 0 at the end -- a per-element branch
    // serialises the 64 evaluations of a chunk (one wavefront per SIMD: ~100 cycles each instead of ~40)
    const bool ok = x >= -87.0f;
    const float xc = fmaxf(x, -87.0f);   // NaN -> -87 as well
    const float y = xc * 1.4426950408889634f;
    const float n = floorf(y);
    const float g = (y - n) - 0.5f;
    float p = 0.00021783880947623402f;
    p = __builtin_fmaf(p, g, 0.0018856498645618558f);
    p = __builtin_fmaf(p, g, 0.013602088205516338f);
    p = __builtin_fmaf(p, g, 0.07849466055631638f);
    p = __builtin_fmaf(p, g, 0.3397315740585327f);
    p = __builtin_fmaf(p, g, 0.9802581667900085f);
    p = __builtin_fmaf(p, g, 1.4142135381698608f);
    const float r = ldexpf(p, (int)n);
    return ok ? r : 0.f;
}

// two evaluations at once on the packed-fp32 pipe (v_pk_mul / v_pk_add / v_pk_fma: IEEE like their scalar forms, same bits)
typedef float det_f2 __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ det_f2 det_expf_neg2(det_f2 x) {
    const bool ok0 = x.x >= -87.0f, ok1 = x.y >= -87.0f;
    det_f2 xc;
    xc.x = fmaxf(x.x, -87.0f); xc.y = fmaxf(x.y, -87.0f);
    const det_f2 y = xc * 1.4426950408889634f;
    det_f2 n;
    n.x = floorf(y.x); n.y = floorf(y.y);
    const det_f2 g = (y - n) - 0.5f;
    det_f2 p = 0.00021783880947623402f;
    p = __builtin_elementwise_fma(p, g, (det_f2)0.0018856498645618558f);
    p = __builtin_elementwise_fma(p, g, (det_f2)0.013602088205516338f);
    p = __builtin_elementwise_fma(p, g, (det_f2)0.07849466055631638f);
    p = __builtin_elementwise_fma(p, g, (det_f2)0.3397315740585327f);
    p = __builtin_elementwise_fma(p, g, (det_f2)0.9802581667900085f);
    p = __builtin_elementwise_fma(p, g, (det_f2)1.4142135381698608f);
    det_f2 r;
    r.x = ok0 ? ldexpf(p.x, (int)n.x) : 0.f;
    r.y = ok1 ? ldexpf(p.y, (int)n.y) : 0.f;
    return r;
}

// four pairs in lock-step: the six dependent packed fmas of one pair leave the pipe half idle (one wavefront per SIMD), four
// independent chains written stage by stage fill it.  Same arithmetic per element as det_expf_neg.
__device__ __forceinline__ void det_expf_neg8(const det_f2 (&x)[4], det_f2 (&out)[4]) {
    det_f2 y[4], n[4], g[4], p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        det_f2 xc;
        xc.x = fmaxf(x[i].x, -87.0f); xc.y = fmaxf(x[i].y, -87.0f);
        y[i] = xc * 1.4426950408889634f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { n[i].x = floorf(y[i].x); n[i].y = floorf(y[i].y); }
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = (y[i] - n[i]) - 0.5f;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma((det_f2)0.00021783880947623402f, g[i], (det_f2)0.0018856498645618558f);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], g[i], (det_f2)0.013602088205516338f);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], g[i], (det_f2)0.07849466055631638f);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], g[i], (det_f2)0.3397315740585327f);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], g[i], (det_f2)0.9802581667900085f);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], g[i], (det_f2)1.4142135381698608f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[i].x = x[i].x >= -87.0f ? ldexpf(p[i].x, (int)n[i].x) : 0.f;
        out[i].y = x[i].y >= -87.0f ? ldexpf(p[i].y, (int)n[i].y) : 0.f;
    }
}

#define CIRS_RNG_STREAM_ACTOR 0x43495253u /* 'CIRS' */
#define CIRS_RNG_STREAM_CHUNK 0x43484E4Bu /* 'CHNK' */
#define CIRS_SAMPLER_CHUNK 128             /* items per chunk of the two-level sampler (4 MFMA tiles of 32) */

// noise for (env e, item i) at rng_step: Philox counter (i>>2, e, rng_step, stream), key = seed; output word i&3.
// Four consecutive items of one env share a Philox block (the head kernel holds 4 consecutive items per lane).
__host__ __device__ __forceinline__ float actor_gumbel(uint64_t seed, uint32_t rng_step, uint32_t env, uint32_t item) {
    const u32x4 r = philox4x32_10(item >> 2, env, rng_step, CIRS_RNG_STREAM_ACTOR, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t sel = item & 3u;
    const uint32_t x = sel == 0 ? r.x : sel == 1 ? r.y : sel == 2 ? r.z : r.w;
    return gumbel_from_bits(x);
}

// ---- dropout masks of the state tracker (production mode) ---------------------------------------------------
// The reference never puts the tracker in eval(): nn.Dropout(p = 0.1) is live at five kinds of sites (core/state_tracker.py:155-156,
// 176; torch TransformerEncoderLayer): 0 = PositionalEncoding output, 1 = attention probabilities, 2 = attention-branch residual
// (dropout1), 3 = feed-forward hidden (dropout), 4 = feed-forward residual (dropout2).  Here every keep / drop decision is a pure
// function of (seed, env, position, layer, site, element): Philox counter (element >> 2, env, position | site << 12 | layer << 16,
// 'DROP'), key = seed, word element & 3; keep iff word >= thr with thr = floor(p * 2^32).  A position keeps its masks for the
// rest of the episode, which is what makes the K/V-cached decode step and the row-parallel backward recompute consistent.
#define CIRS_RNG_STREAM_DROPOUT 0x44524F50u /* 'DROP' */
enum { CIRS_DROP_POS = 0, CIRS_DROP_ATTN = 1, CIRS_DROP_RES1 = 2, CIRS_DROP_FF = 3, CIRS_DROP_RES2 = 4 };

__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) { return (uint32_t)((double)p * 4294967296.0); }

__host__ __device__ __forceinline__ u32x4 dropout_block(uint64_t seed, uint32_t env, uint32_t pos, uint32_t layer, uint32_t site,
                                                        uint32_t elem_group) {
    return philox4x32_10(elem_group, env, pos | (site << 12) | (layer << 16), CIRS_RNG_STREAM_DROPOUT, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__host__ __device__ __forceinline__ uint32_t block_word(const u32x4& r, uint32_t sel) {
    return sel == 0 ? r.x : sel == 1 ? r.y : sel == 2 ? r.z : r.w;
}
__host__ __device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t env, uint32_t pos, uint32_t layer, uint32_t site,
                                                      uint32_t elem, uint32_t thr) {
    return block_word(dropout_block(seed, env, pos, layer, site, elem >> 2), elem & 3u) >= thr;
}

// Two-level ("chunked") Gumbel-max sampler -- the counter-based sampler of the rollout.
//   Categorical(softmax(z)).sample() is drawn in two exact stages: a chunk c of 128 consecutive items with probability
//   mass_c / sum mass, mass_c = sum_{i in c} e^{z_i} -- realised as argmax_c (L_c + G1_c), L_c = log mass_c -- and then an item
//   inside the chosen chunk, argmax_{i in c} (z_i + G2_i).  Both arg-maxes are Gumbel-max draws, so the pair is an exact sample of
//   the categorical distribution, but only n_chunks + 128 noise values are needed per draw instead of one per catalogue item
//   (10728 items: 84 + 128).  L_c must be identical on every implementation (it decides the chunk): L_c = M_c + det_logf(S_c),
//   M_c = max of the chunk's valid logits, S_c = sum of det_expf_neg(z_i - M_c) in the order fixed below.
//   Noise: G1_c from Philox counter (c >> 2, env, rng_step, 'CHNK'), word c & 3; G2_i = actor_gumbel (counter (i >> 2, env,
//   rng_step, 'CIRS')).  Ties -> lowest chunk / lowest item.  Masked (already recommended) items are invalid in both stages.
__host__ __device__ __forceinline__ float chunk_gumbel(uint64_t seed, uint32_t rng_step, uint32_t env, uint32_t chunk) {
    const u32x4 r = philox4x32_10(chunk >> 2, env, rng_step, CIRS_RNG_STREAM_CHUNK, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t sel = chunk & 3u;
    const uint32_t x = sel == 0 ? r.x : sel == 1 ? r.y : sel == 2 ? r.z : r.w;
    return gumbel_from_bits(x);
}

}  // namespace cirs
