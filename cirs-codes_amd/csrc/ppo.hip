// ppo.hip -- the PPO learner for gfx950 (reference: core/policy/ppo.py:96-246, tianshou a2c/base/statistics).
//
// prepare   gae_kernel          one thread per env: float64 reverse scan delta_t + gamma*lambda*gae (base.py:380-396),
//                               v_s/v_s_ un-normalised by sqrt(ret_rms.var + eps) (a2c.py:95-97), compaction to buffer order
//           returns_kernel      single workgroup: mean / population variance of the un-normalised returns (float64, fixed
//                               order), normalised returns, RunningMeanStd merge (statistics.py:80-95)
// minibatch gather -> adv norm -> trunk fwd -> head stats (MFMA, transposed tile, no logits in HBM) -> row scalars
//           -> head backward (one fused kernel): the logits tile is recomputed on the matrix cores instead of reading a
//              B x I probability matrix; dZ feeds the d h2 product from the accumulator registers (Z^T layout) and,
//              after a 4 KB transpose through LDS, the dWa product (Z layout)
//           -> trunk / critic backward (small dense kernels) -> d obs scatter (gradient into the state tracker)
//           -> clip_grad_norm_ (trunk counted twice) -> Adam (trunk: coefficient squared, two sub-steps)
// Every reduction has a fixed order: two runs (or two ranks of a replicated learner) produce identical bits.
//
// Roofline of one minibatch step (mb x I x 64): algorithmic 3 * 2*mb*I*64 flop (forward + two backward products,
// SURVEY 8(d)) = 4.2 GFLOP at mb = 1024, I = 10728; executed 8*mb*I*64 (forward statistics pass + one recompute of the
// logits in the backward kernel) on the fp32 MFMA pipe (157 TF peak).  HBM traffic is Wa (2.7 MB) + dWa partials
// (n_row_blocks x 2.7 MB) + dH2 partials (n_chunks x mb x 256 B = 22 MB): MFMA-bound.
#include "internal.h"
#include "bf16x6.h"
#include "small_gemm.h"
#include "permutation.h"
#include "policy_kernels.h"

namespace cirs {
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: a float4 struct array goes through memcpy / scratch


constexpr int kBwdWaves = 4;  // row tiles per workgroup of the fused head backward kernel (= rows/32 per dWa slab)
__host__ __device__ inline int n_row_blocks_of(int n_pad) { return (n_pad / kTileM + kBwdWaves - 1) / kBwdWaves; }
// floats between consecutive dWa|dba partial slabs (16 B aligned for the float4 stores)
__host__ __device__ inline size_t dwa_slab_stride(int I) { return (((size_t)I * 64 + I) + 3) & ~(size_t)3; }
// dWa slabs the workspace holds: one per row block of the fused backward kernel
__host__ __device__ inline int n_dwa_slabs_of(int n_pad) { return n_row_blocks_of(n_pad); }

struct PpoLayout {  // offsets (floats) into the flat parameter buffer
    long w1, b1, w2, b2, wa, ba, wc, bc, total, trunk;
};
__host__ __device__ inline PpoLayout ppo_layout(int I, int S) {
    PpoLayout L;
    L.w1 = 0; L.b1 = L.w1 + (long)kH * S; L.w2 = L.b1 + kH; L.b2 = L.w2 + (long)kH * kH;
    L.trunk = L.b2 + kH;
    L.wa = L.trunk; L.ba = L.wa + (long)I * kH; L.wc = L.ba + I; L.bc = L.wc + kH; L.total = L.bc + 1;
    return L;
}

// ------------------------------------------------------------------------------------------------------------
// prepare
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gae_kernel(cirs_ppo_cfg cfg, cirs_traj traj, const int32_t* __restrict__ lens,
                                                  const int32_t* __restrict__ offsets, int B, int S,
                                                  const double* __restrict__ rms_state, cirs_ppo_batch out,
                                                  double* __restrict__ unnorm_ret) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int L = lens[b], off = offsets[b];
    const double scale = cfg.rew_norm ? sqrt(rms_state[1] + 1e-8) : 1.0;  // a2c.py:95-97, pg.py:60 (_eps = 1e-8)
    const double gamma = (double)cfg.gamma, gl = (double)cfg.gamma * (double)cfg.gae_lambda;
    double gae = 0.0;
    // the recurrence runs backwards over the episode, eight steps per pass: the 8 x 5 loads of the NEXT pass are requested before this
    // pass is computed (they do not depend on it) -- with one step of look-ahead every step still cost a cold memory round trip
    // (44 us per update for 30 steps); the arithmetic and its order are unchanged
    struct StepIn { bool done; float value, logp; double rew; long act; };
    auto load_step = [&](int t) {
        if (t < 0) return StepIn{};
        const size_t ti = (size_t)t * B + b;
        return StepIn{traj.done[ti] != 0, traj.value[ti], traj.logp[ti], traj.rew[ti], (long)traj.act[ti]};
    };
    StepIn cur8[8], nxt8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cur8[u] = load_step(L - 1 - u);
    float value_next = 0.f;   // V(s_{t+1}) as recorded at step t+1
    for (int t0 = L - 1; t0 >= 0; t0 -= 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt8[u] = load_step(t0 - 8 - u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 - u;
            if (t < 0) break;
            const StepIn cur = cur8[u];
            const bool done = cur.done;
            const double v_s = (double)cur.value * scale;
            // value_mask (base.py:264): V(s') is zeroed on done; otherwise V(s_{t+1}) recorded at the next step
            const double v_ns = (done || t + 1 >= L) ? 0.0 : (double)value_next * scale;
            const double end_flag = (done || t == L - 1) ? 1.0 : 0.0;  // done OR unfinished_index (base.py:307-308)
            const double delta = cur.rew + v_ns * gamma - v_s;
            gae = delta + (1.0 - end_flag) * gl * gae;
            const int row = off + t;
            out.adv[row] = (float)gae;
            unnorm_ret[row] = gae + v_s;
            out.v_s[row] = cur.value;
            out.logp_old[row] = cur.logp;
            out.act[row] = (int32_t)cur.act;
            out.row_env[row] = b;
            out.row_t[row] = t;
            value_next = cur.value;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) cur8[u] = nxt8[u];
    }
}

// exclusive prefix sum of the episode lengths (the buffer offsets of the envs) + the row count, one workgroup (n_env <= 2^20)
__device__ __forceinline__ void offsets_block(const int32_t* __restrict__ lens, int B, int32_t* __restrict__ offsets, int32_t* __restrict__ n_out) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (B + 1023) / 1024, b0 = tid * per, b1 = min(B, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += lens[b];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {     // inclusive scan of the per-thread sums
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int b = b0; b < b1; ++b) { offsets[b] = run; run += lens[b]; }
    if (tid == 1023) *n_out = part[1023];
}
__global__ __launch_bounds__(1024) void offsets_kernel(const int32_t* __restrict__ lens, int B, int32_t* __restrict__ offsets, int32_t* __restrict__ n_out) {
    offsets_block(lens, B, offsets, n_out);
}

__device__ __forceinline__ void compact_obs_elem(const cirs_traj& traj, const cirs_ppo_batch& out, long i, int N, int B, int S) {
    if (i >= (long)N * S) return;
    const int row = (int)(i / S), k = (int)(i % S);
    out.obs[i] = traj.obs[((size_t)out.row_t[row] * B + out.row_env[row]) * S + k];
}
__global__ __launch_bounds__(256) void compact_obs_kernel(cirs_traj traj, cirs_ppo_batch out, int N_arg, int B, int S, const int32_t* __restrict__ n_dev = nullptr) {
    const int N = n_dev ? *n_dev : N_arg;     // n_dev: the row count lives on the device (cirs_ppo_prepare_async: the host does not know it yet)
    compact_obs_elem(traj, out, blockIdx.x * (long)blockDim.x + threadIdx.x, N, B, S);
}

// single workgroup, fixed-order float64 reductions
// (one workgroup of 1024 threads; red: 1024 doubles of LDS, s_mean: one more)
__device__ __forceinline__ void returns_block(const cirs_ppo_cfg& cfg, const double* __restrict__ unnorm_ret, int N, double* __restrict__ rms_state,
                                              float* __restrict__ ret_out, double* red, double& s_mean) {
    const int tid = threadIdx.x;
    double acc = 0.0;
    // eight loads in flight per pass, added in the same (index) order as a plain loop
    for (int i0 = tid; i0 < N; i0 += 8 * 1024) {
        double x8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x8[u] = i0 + u * 1024 < N ? unnorm_ret[i0 + u * 1024] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 1024 < N) acc += x8[u];
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) s_mean = red[0] / (double)N;
    __syncthreads();
    const double mean = s_mean;
    acc = 0.0;
    for (int i0 = tid; i0 < N; i0 += 8 * 1024) {
        double x8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x8[u] = i0 + u * 1024 < N ? unnorm_ret[i0 + u * 1024] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 1024 < N) { const double d = x8[u] - mean; acc += d * d; }
    }
    __syncthreads();
    red[tid] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double var = red[0] / (double)N;  // np.var: population variance
    const double scale = cfg.rew_norm ? sqrt(rms_state[1] + 1e-8) : 1.0;  // OLD variance (a2c.py:101-103)
    for (int i0 = tid; i0 < N; i0 += 8 * 1024) {
        double x8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x8[u] = i0 + u * 1024 < N ? unnorm_ret[i0 + u * 1024] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 1024 < N) ret_out[i0 + u * 1024] = (float)(x8[u] / scale);
    }
    __syncthreads();
    if (tid == 0 && cfg.rew_norm) {  // RunningMeanStd.update (statistics.py:80-95)
        const double o_mean = rms_state[0], o_var = rms_state[1], o_cnt = rms_state[2];
        const double bc = (double)N, delta = mean - o_mean, tot = o_cnt + bc;
        const double new_mean = o_mean + delta * bc / tot;
        const double m2 = o_var * o_cnt + var * bc + delta * delta * o_cnt * bc / tot;
        rms_state[0] = new_mean; rms_state[1] = m2 / tot; rms_state[2] = tot;
    }
}
__global__ __launch_bounds__(1024) void returns_kernel(cirs_ppo_cfg cfg, const double* __restrict__ unnorm_ret, int N_arg,
                                                       double* __restrict__ rms_state, float* __restrict__ ret_out,
                                                       const int32_t* __restrict__ n_dev = nullptr) {
    __shared__ double red[1024];
    __shared__ double s_mean;
    returns_block(cfg, unnorm_ret, n_dev ? *n_dev : N_arg, rms_state, ret_out, red, s_mean);
}
// process_fn's last launch (cirs_ppo_prepare_async): three independent jobs behind gae_kernel on disjoint workgroups of 1024 threads -- the return
// normalisation (workgroup 0: returns_kernel's code), the compaction of the observations (compact_obs_kernel's, 1024 elements per workgroup) and, n_perm > 0,
// the update's minibatch permutations (cirs_random_permutations' with the row count read from the device: out[c][i], i < n, row stride n).  Same bits as the
// three launches; two launches (~10-15 us) fewer per update.
__global__ __launch_bounds__(1024) void prepare_tail_kernel(cirs_ppo_cfg cfg, cirs_traj traj, cirs_ppo_batch out, const double* __restrict__ unnorm_ret,
                                                            double* __restrict__ rms_state, const int32_t* __restrict__ n_dev, int B, int S, int n_compact,
                                                            PermKeys keys, int n_perm, long perm_upper, int32_t* __restrict__ perm_out) {
    __shared__ double red[1024];
    __shared__ double s_mean;
    const int N = *n_dev;
    const int b = blockIdx.x;
    if (b == 0) { returns_block(cfg, unnorm_ret, N, rms_state, out.ret, red, s_mean); return; }
    if (b <= n_compact) { compact_obs_elem(traj, out, (long)(b - 1) * 1024 + threadIdx.x, N, B, S); return; }
    const long g = (long)(b - 1 - n_compact) * 1024 + threadIdx.x;
    const int c = (int)(g / perm_upper);
    const long i = g - (long)c * perm_upper;
    if (c < n_perm && i < N) perm_out[(size_t)c * N + i] = (int32_t)permute_index(i, N, perm_half_bits((uint64_t)N), perm_key_of(keys, c));
}

// ------------------------------------------------------------------------------------------------------------
// minibatch: gather + advantage normalisation
// ------------------------------------------------------------------------------------------------------------
// ---- operand pieces of the actor-head kernels -----------------------------------------------------------------------------------------------------
// Since round 6 every product of the head kernels is an fp32 product from TWO fp16 pieces per operand, three MFMAs per 16 k ("f16x3", bf16x6.h; rounds 2-5:
// three bf16 pieces, six MFMAs).  Operands are pre-scaled by exact powers of two so that both pieces stay in fp16's normal range: Wa by kScWa, H2 by kScH2, dZ
// by a power of two per workgroup (its rows' largest coefficient -> [2^13, 2^14)); logits / dH2 / dWa are unscaled exactly where they leave the accumulators.
// Measured against rounds 5's bf16x6 on one box: minibatch step 75.5 -> 69.8 us, head_bwd_fused_kernel 36.6 -> 31.4 us (rocprofv3), gradient error vs float64 at
// torch-fp32's level in the sharp regime (tests/test_gpu_head_precision.py), golden bars used <= 0.10 (profiles/r06_margins_head_f16.json).
// The plane storage keeps the three-plane tile layout of the earlier bf16 form (the middle plane is neither written nor read).
typedef Planes2 HPl;
constexpr float kScWa = 256.f, kScH2 = 64.f;
#define HPL_SET(P, H, L) do { (P).h = __builtin_bit_cast(f16x8, H); (P).l = __builtin_bit_cast(f16x8, L); } while (0)
#define hsplit8 split8h
#define hmfma_split2 mfma_f16x3_split2
#define hmfma_pair mfma_f16x3_pair
constexpr float kScZ = kScWa * kScH2, kScZi = 1.0f / kScZ;      // scale of the logits accumulators (bias pre-scaled at staging) and its inverse
constexpr int kPlaneTileU4 = 1536;  // uint4 per item tile of the fp16 planes of Wa (wa_planes_kernel)
struct MbView {  // contiguous minibatch arrays carved from the workspace (n_pad rows)
    float *obs, *adv, *ret, *v_s, *logp_old;  // gathered
    int32_t* act;
    float *h1, *h2, *value;                   // trunk forward
    float *lse, *ez, *za;                     // head stats: log-sum-exp, E_p[z], logit of the taken action
    float *c_logp, *c_ent, *h_ent, *dvalue;   // row coefficients for the backward pass
    float *ent_row;                           // entropy per row (from the backward pass)
    float *clip_row, *vf_row;                 // per-row surrogate / value-loss terms (summed in fixed order later)
    long *dst_row;                            // row of the [T+1,B] tracker-gradient tensor each minibatch row scatters to
    float *da2, *da1;                         // [n_pad,64] pre-activation gradients
    float *dh2p;                              // [n_chunks, n_pad, 64] partial d h2
    float *entp;                              // [n_chunks, n_pad]
    float *dwap;                              // [n_row_blocks, I*64 + I] partial dWa | dba
    float *red;                               // [16] scalars: adv mean/std, loss sums, grad norm coef
    float *normp;                             // [256] sum-of-squares partials
    float *dwp;                               // weight-gradient slab partials
    uint4 *wa_planes;                         // fp16 planes of Wa per item tile (wa_planes_kernel)
    uint4 *h2z, *h2b;                         // fp16 planes of H2 in the head kernels' register order (written by trunk_adv_kernel):
                                              // [32-row tile][12][lane] uint4, unit q of a lane = its q-th operand register quad
    int *sync;                                // [kSyncInts] arrival flags (see kSyncInts)
    void* head_ws;                            // workspace of the head kernel (h2 copy + partials)
};

// arrival FLAGS of the workgroups that other workgroups of the same launch wait for (one word per producer, zeroed by head_bwd_fused_kernel earlier in the
// step): [0, 256) the R workgroups of trunk_rows_kernel, [256, 288) the A0 workgroups of adam_next_kernel.  A flag per producer instead of one counter:
// 22 - 128 read-modify-writes of ONE address are serialised at the memory side (a few hundred ns each), a flag is a plain write-through store, and the
// consumer's threads poll one flag each (one cache line per instruction).
constexpr int kSyncInts = 288, kSyncA0 = 256;
__device__ __forceinline__ void flag_arrive(int* flags, int b) {      // the caller has drained its stores (s_waitcnt vmcnt(0)) and synchronised the workgroup
    if (threadIdx.x == 0) __hip_atomic_store(flags + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A lost or late arrival must neither hang the device nor pass silently (round 6): on giving up the workgroup counts itself in a STICKY device word that
// cirs_ppo_handoff_status copies out -- the host checks it at the update's read-back and raises (cirs_hip/learner.py: check_handoffs).  Assumptions of the
// hand-off, in one place: (i) producers have the LOWEST block ids of the launch and are therefore dispatched before (or together with) their consumers -- not
// a HIP guarantee, which is exactly why the wait is bounded and loud; (ii) payload and flag are `sc1` (write-through) stores, the payload drained with
// `s_waitcnt vmcnt(0)` before the flag leaves, and consumers read both with `sc1` loads (L2 / fabric, never a stale L1 line) -- MI355X_MICROARCH.md's
// "sc1 stores AND sc1 loads" form, which needs no release / acquire fence.
__device__ int g_handoff_lost = 0;
__device__ __forceinline__ void flags_wait(const int* flags, int n) {      // every thread of the workgroup calls it; bounded
    int spins = 0;
    for (;;) {
        int mine = 1;
        for (int q = threadIdx.x; q < n; q += blockDim.x) mine &= __hip_atomic_load(flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (__syncthreads_and(mine)) break;
        if (++spins >= (1 << 20)) {
            if (threadIdx.x == 0) atomicAdd(&g_handoff_lost, 1);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
__global__ void handoff_status_kernel(int* __restrict__ out, int reset) {
    out[0] = g_handoff_lost;
    if (reset) g_handoff_lost = 0;
}
// the one read-back of an update as ONE launch: the episode lengths and the sticky hand-off count written straight to pinned host memory (instead of
// a status kernel + two device-to-host copies, each a blit launch of ~5 us)
__global__ __launch_bounds__(256) void update_readback_kernel(const int32_t* __restrict__ lens, int n, int32_t* __restrict__ lens_host, int32_t* __restrict__ lost_host) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lens_host[i] = lens[i];
    if (i == 0 && lost_host) lost_host[0] = g_handoff_lost;
}
// ... and with process_fn's first job in the same launch (one workgroup of 1024 threads): offsets_kernel's scan, then the copy
__global__ __launch_bounds__(1024) void update_readback_offsets_kernel(const int32_t* __restrict__ lens, int n, int32_t* __restrict__ lens_host,
                                                                       int32_t* __restrict__ lost_host, int32_t* __restrict__ offsets, int32_t* __restrict__ n_out) {
    for (int i = threadIdx.x; i < n; i += 1024) lens_host[i] = lens[i];
    if (threadIdx.x == 0 && lost_host) lost_host[0] = g_handoff_lost;
    offsets_block(lens, n, offsets, n_out);
}
__host__ __device__ inline int snap_floats(int S) { return kH * (S + 66) + kH + 1; }      // trunk (w1 | b1 | w2 | b2) + wc | bc
__host__ __device__ inline int snap_stride(int S) { return (snap_floats(S) + 3) & ~3; }       // (16-byte aligned arrays)
__host__ inline size_t dwp_floats(int n_pad, int S) {
    const size_t a = (size_t)(n_pad / kTileM) * (kH * (kH + 1) + kH * (S + 1) + (kH + 1)) + 64, b = (size_t)(n_pad / 8) * snap_stride(S) + 64;
    return a > b ? a : b;
}
__host__ inline size_t mb_ws_floats(int n_pad, int I, int S) {
    const size_t nch = n_chunks_of(I);
    size_t f = 0;
    f += (size_t)n_pad * S + 4 * (size_t)n_pad;       // obs, adv, ret, v_s, logp_old
    f += n_pad;                                        // act
    f += 2 * (size_t)n_pad * kH + n_pad;               // h1, h2, value
    f += 3 * (size_t)n_pad;                            // lse, ez, za
    f += 4 * (size_t)n_pad + n_pad;                    // c_logp, c_ent, h_ent, dvalue, ent_row
    f += 2 * (size_t)n_pad + 2 * (size_t)n_pad + 8;    // clip_row, vf_row, dst_row (int64)
    f += 2 * (size_t)n_pad * kH;                       // da2, da1
    f += nch * (size_t)n_pad * kH + nch * (size_t)n_pad;  // dh2p, entp
    f += (size_t)n_dwa_slabs_of(n_pad) * dwa_slab_stride(I);     // dwap
    f += kSyncInts + 4;         // sync
    f += 64 + 1024;                                    // red + sum-of-squares partials (kNormBlocks)
    f += dwp_floats(n_pad, S);                         // dW row slabs (one per 32 rows: trunk_bwd_kernel; one per 8 rows in flat order: trunk_rows_kernel)
    f += (size_t)cdiv(I, kTileN) * kPlaneTileU4 * 4;   // wa_planes
    f += 2 * (size_t)n_pad * 96;                       // h2z, h2b: 24 uint4 per row each
    f += (size_t)n_pad * kH + 4 * nch * (size_t)n_pad; // head workspace
    return f;
}

__host__ inline MbView carve(void* ws, int n_pad, int I, int S) {
    float* p = (float*)ws;
    const size_t nch = n_chunks_of(I);
    MbView v;
    auto take = [&](size_t n) { float* r = p; p += (n + 3) & ~(size_t)3; return r; };
    v.obs = take((size_t)n_pad * S); v.adv = take(n_pad); v.ret = take(n_pad); v.v_s = take(n_pad); v.logp_old = take(n_pad);
    v.act = (int32_t*)take(n_pad);
    v.h1 = take((size_t)n_pad * kH); v.h2 = take((size_t)n_pad * kH); v.value = take(n_pad);
    v.lse = take(n_pad); v.ez = take(n_pad); v.za = take(n_pad);
    v.c_logp = take(n_pad); v.c_ent = take(n_pad); v.h_ent = take(n_pad); v.dvalue = take(n_pad); v.ent_row = take(n_pad);
    v.clip_row = take(n_pad); v.vf_row = take(n_pad);
    v.dst_row = (long*)take(2 * (size_t)n_pad + 4);
    v.da2 = take((size_t)n_pad * kH); v.da1 = take((size_t)n_pad * kH);
    v.dh2p = take(nch * (size_t)n_pad * kH); v.entp = take(nch * (size_t)n_pad);
    v.dwap = take((size_t)n_dwa_slabs_of(n_pad) * dwa_slab_stride(I));
    v.sync = (int*)take(kSyncInts);
    v.red = take(64);
    v.normp = take(1024);
    v.dwp = take(dwp_floats(n_pad, S));
    v.wa_planes = (uint4*)take((size_t)cdiv(I, kTileN) * kPlaneTileU4 * 4);
    v.h2z = (uint4*)take((size_t)n_pad * 96); v.h2b = (uint4*)take((size_t)n_pad * 96);
    v.head_ws = (void*)p;
    return v;
}

// advantage statistics of the (global) minibatch: mean and unbiased std (torch.Tensor.std, ppo.py:185-186) of
// adv_flat[idx[0..m)].  One workgroup of 256 threads, fixed order.  red[0] = mean, red[1] = std (0, 1 when
// normalisation is off).
__device__ __forceinline__ void adv_stats_block(const float* __restrict__ adv_flat, const int32_t* __restrict__ idx, int m,
                                                int enable, float* __restrict__ red, float* sh /* [256] */) {
    __shared__ float s_mean;
    const int tid = threadIdx.x;
    if (!enable) {
        if (tid == 0) { red[0] = 0.f; red[1] = 1.f; }
        return;
    }
    float acc = 0.f;
    for (int i = tid; i < m; i += 256) acc += adv_flat[idx[i]];
    sh[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) s_mean = sh[0] / (float)m;
    __syncthreads();
    const float mean = s_mean;
    acc = 0.f;
    for (int i = tid; i < m; i += 256) {
        const float d = adv_flat[idx[i]] - mean;
        acc += d * d;
    }
    __syncthreads();
    sh[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) { red[0] = mean; red[1] = sqrtf(sh[0] / (float)(m - 1)); }
}

// One minibatch row's loss terms and backward coefficients (ppo.py:183-212) from its logit statistics: z = logit of the taken action,
// lse = log-sum-exp, ez = E_p[z].  Shared by head_stats_merge_kernel (item-sharded learner) and the prologue of head_bwd_fused_kernel.
struct RowTerms { float clip_row, c_logp, vf_row, dvalue, c_ent, h_ent; };
__device__ __forceinline__ RowTerms ppo_row_terms(const cirs_ppo_cfg& cfg, float z, float lse, float ez, float logp_old, float adv, float adv_mean,
                                                  float adv_std, float val, float vs, float ret, float inv_mb) {
    RowTerms o;
    const float eps = 1.1920928955078125e-7f;
    const float praw = __expf(z - lse);
    const bool clamped = praw < eps || praw > 1.0f - eps;  // probs_to_logits clamp blocks the gradient
    const float logp = __logf(fminf(fmaxf(praw, eps), 1.0f - eps));
    const float ratio = __expf(logp - logp_old);
    const float A = (adv - adv_mean) / adv_std;            // per-minibatch advantage normalisation (ppo.py:184-186)
    const float s1 = ratio * A;
    const float s2 = fminf(fmaxf(ratio, 1.0f - cfg.eps_clip), 1.0f + cfg.eps_clip) * A;
    float surr = fminf(s1, s2);
    // d(-min(s1,s2))/d logp: s1 path when s1 <= s2 (a tie passes the full gradient), else clamp blocks it
    float c_logp = (s1 <= s2 && !clamped) ? -inv_mb * A * ratio : 0.f;
    if (cfg.dual_clip > 0.f) {   // -max(min(s1, s2), dual_clip * A) (core/policy/ppo.py:190-193): the constant branch has no gradient
        const float dcl = cfg.dual_clip * A;
        if (dcl > surr) { surr = dcl; c_logp = 0.f; }
        else if (dcl == surr) c_logp *= 0.5f;                // torch.max splits the gradient of a tie
    }
    o.clip_row = -surr;
    o.c_logp = c_logp;
    const float d1 = ret - val;
    float vf = d1 * d1, dv = -2.0f * d1;
    if (cfg.value_clip) {
        const float dlt = val - vs;
        const float vclip = vs + fminf(fmaxf(dlt, -cfg.eps_clip), cfg.eps_clip);
        const float d2 = ret - vclip;
        const float vf2 = d2 * d2;
        const float dv2 = (dlt >= -cfg.eps_clip && dlt <= cfg.eps_clip) ? -2.0f * d2 : 0.f;
        if (vf2 > vf) { vf = vf2; dv = dv2; }
        else if (vf2 == vf) dv = 0.5f * (dv + dv2);  // torch.max splits ties
    }
    o.vf_row = vf;
    o.dvalue = cfg.vf_coef * inv_mb * dv;
    // entropy gradient coefficient: dL/dz_i += c_ent * p_i * (z_i - lse + H), H = lse - E_p[z]
    o.c_ent = cfg.ent_coef * inv_mb;
    o.h_ent = lse - ez;
    return o;
}

// One wavefront per minibatch row: merge the head-stats partials (lse, E_p[z]), recompute the taken action's logit
// with the MFMA k-order, then (lane 0) the row's loss terms and backward coefficients (ppo.py:183-212).  Rows are read
// from the buffer-order batch through idx (no separate gather pass); padded rows get neutral coefficients.
__global__ __launch_bounds__(256) void head_stats_merge_kernel(cirs_ppo_cfg cfg, cirs_ppo_batch b, const int32_t* __restrict__ idx,
                                                               int mb, int mb_norm, int n_pad, int n_chunks, int n_env,
                                                               ActorPartialView pv, const float* __restrict__ wa,
                                                               const float* __restrict__ ba, MbView v,
                                                               const float* __restrict__ za_in, int item_base) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_pad) return;
    if (j >= mb) {
        if (lane == 0) {
            v.c_logp[j] = 0.f; v.c_ent[j] = 0.f; v.h_ent[j] = 0.f; v.dvalue[j] = 0.f; v.lse[j] = 1e30f;  // p = exp(z - lse) = 0
            v.clip_row[j] = 0.f; v.vf_row[j] = 0.f; v.act[j] = 0; v.dst_row[j] = 0;
        }
        return;
    }
    // The row's gathers form a chain of dependent round trips (idx -> action -> head row of the action).  The first two are
    // requested by every lane before the merge of the chunk partials, the head row and the hidden row arrive as ONE coalesced load
    // each (lane i holds element i) and reach lane 0's fma chain through LDS -- one lane reading 128 floats from memory on its own is
    // 128 memory instructions behind two serial round trips.
    __shared__ __attribute__((aligned(16))) float sRow[4][2][kH];
    // Item-sharded head (za_in != null): the partials are one (m, s, t) triple per shard, folded by head_tp_fold_kernel and
    // all-gathered; za_in [n_chunks][n_pad] carries the action's logit from the shard that owns the item (NaN elsewhere), and the
    // action is stored relative to THIS shard's first item (item_base) for the backward kernel's `item == action` test.
    const int src = idx[j];
    const int a = b.act[src];
    const float hl = v.h2[(size_t)j * kH + lane], wl = za_in ? 0.f : wa[(size_t)a * kH + lane];
    float m = -INFINITY, s = 0.f, t = 0.f;
    for (int c = lane; c < n_chunks; c += CIRS_WAVE) {
        const size_t o = (size_t)c * n_pad + j;
        const float om = pv.m[o], os = pv.s[o], ot = pv.score[o];
        const float mn = fmaxf(m, om);
        if (mn > -INFINITY) {
            const float fa = __expf(m - mn), fb = __expf(om - mn);
            s = s * fa + os * fb; t = t * fa + ot * fb; m = mn;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, CIRS_WAVE), os = __shfl_xor(s, off, CIRS_WAVE), ot = __shfl_xor(t, off, CIRS_WAVE);
        const float mn = fmaxf(m, om);
        if (mn > -INFINITY) {
            const float fa = __expf(m - mn), fb = __expf(om - mn);
            s = s * fa + os * fb; t = t * fa + ot * fb; m = mn;
        }
    }
    sRow[threadIdx.x >> 6][0][lane] = hl;
    sRow[threadIdx.x >> 6][1][lane] = wl;
    __builtin_amdgcn_wave_barrier();
    if (lane != 0) return;
    const float* hr = sRow[threadIdx.x >> 6][0];
    const float* wr = sRow[threadIdx.x >> 6][1];
    float z;
    if (za_in) {
        z = 0.f;
        for (int c = 0; c < n_chunks; ++c) {
            const float zc = za_in[(size_t)c * n_pad + j];
            if (zc == zc) z = zc;      // exactly one shard owns the action
        }
    } else {
        z = ba[a];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            z = __builtin_fmaf(hr[kk], wr[kk], z);
            z = __builtin_fmaf(hr[32 + kk], wr[32 + kk], z);
        }
    }
    const float lse = m + __logf(s);
    v.lse[j] = lse;
    v.act[j] = a - item_base;
    v.dst_row[j] = (long)b.row_t[src] * n_env + b.row_env[src];
    // ---- row losses + backward coefficients; every mean is over the (global) minibatch of mb_norm rows --------------
    const RowTerms rt = ppo_row_terms(cfg, z, lse, t / s, b.logp_old[src], b.adv[src], v.red[0], v.red[1], v.value[j], b.v_s[src], b.ret[src],
                                      1.0f / (float)mb_norm);
    v.clip_row[j] = rt.clip_row; v.c_logp[j] = rt.c_logp; v.vf_row[j] = rt.vf_row; v.dvalue[j] = rt.dvalue;
    v.c_ent[j] = rt.c_ent; v.h_ent[j] = rt.h_ent;
}

__device__ __forceinline__ float dz_of(float z, float lse, float c_logp, float c_ent, float h_ent, bool is_act, float& p_out) {
    const float p = __expf(z - lse);
    p_out = p;
    return c_logp * ((is_act ? 1.0f : 0.0f) - p) + c_ent * p * (z - lse + h_ent);
}

// Wa as fp16 planes, per item tile of 32 (items beyond I are zero rows), 24576 B per tile:
//   [0, 12288)      row-major  R[p][item 32][col 64]    : A operand of Z^T = Wa H2^T (lane = item, 8 consecutive cols)
//   [12288, 24576)  col-major  C[p][col 64][slot 32]    : B operand of dH2 = dZ Wa  (lane = col, 8 consecutive slots);
//                   slot 16 t + 8 hi + j holds item acc_row(8 t + j, hi): the order in which the logit accumulators
//                   of a lane enumerate the items, so the dZ registers are the A operand as they are.
__device__ __forceinline__ void wa_planes_from_lds(int tile, uint4* __restrict__ planes, const float* sw);
__device__ __forceinline__ void wa_planes_block(int tile, int I, const float* __restrict__ wa, uint4* __restrict__ planes, float* sw) {
    const int tid = threadIdx.x, tile0 = tile * kTileN;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = tid + 256 * q, item = f >> 4, col = (f & 15) * 4;
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tile0 + item < I) t4 = *reinterpret_cast<const float4*>(wa + (size_t)(tile0 + item) * kH + col);
        float* d = &sw[item * 65 + col];
        d[0] = t4.x; d[1] = t4.y; d[2] = t4.z; d[3] = t4.w;
    }
    __syncthreads();
    wa_planes_from_lds(tile, planes, sw);
}
// the six planes of item tile `tile` from its fp32 image sw[item][65] (all 256 threads; the caller has synchronised the image)
__device__ __forceinline__ void wa_planes_from_lds(int tile, uint4* __restrict__ planes, const float* sw) {
    const int tid = threadIdx.x;
    uint4* out = planes + (size_t)tile * kPlaneTileU4;
    {
        const float* r = &sw[(tid >> 3) * 65 + 8 * (tid & 7)];
        const HPl pl = hsplit8(kScWa * r[0], kScWa * r[1], kScWa * r[2], kScWa * r[3], kScWa * r[4], kScWa * r[5], kScWa * r[6], kScWa * r[7]);
        out[tid] = __builtin_bit_cast(uint4, pl.h); out[512 + tid] = __builtin_bit_cast(uint4, pl.l);
    }
    {
        const int n = tid >> 2, t = (tid >> 1) & 1, hi = tid & 1;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = kScWa * sw[acc_row(8 * t + j, hi) * 65 + n];
        const HPl pl = hsplit8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
        out[768 + tid] = __builtin_bit_cast(uint4, pl.h); out[1280 + tid] = __builtin_bit_cast(uint4, pl.l);
    }
}

// ---- trunk forward of minibatch rows (first launch of a stand-alone minibatch step; inside a learn() loop: the Adam launch of the step before) ----
// What a row leaves behind for the later kernels of its step.
struct TrunkRowOut {
    float *h2, *value, *h1, *obs_copy;     // [n_pad, 64], [n_pad], [n_pad, 64], [n_pad, S]
    int32_t* act; long* dst;               // the row's action / its row of the [T+1, B] tracker-gradient tensor, in minibatch order
    float* row4;                           // [4][n_pad]: adv, logp_old, ret, v_s of the minibatch rows
    uint4 *h2z, *h2b;                      // fp16 planes of H2 in the head kernels' register order
};
// The weights as the row job reads them: the three matrices from LDS copies, the biases through `w` (global memory in trunk_adv_kernel, LDS in
// adam_next_kernel, whose workgroups form the updated values themselves).
struct TrunkLds { const float* w1; int ld1; const float* w2; const float* wc; };

// row j's gathers, requested by the whole wavefront before anything waits
struct TrunkRowIn { int ri; float x; };
__device__ __forceinline__ TrunkRowIn trunk_row_gather(int j, int mb, const int32_t* __restrict__ idx, const float* __restrict__ obs_flat, long stride, int S,
                                                       int lane) {
    TrunkRowIn in;
    in.ri = idx[j < mb ? j : mb - 1];                     // rows >= mb: never used -- clamped, so the load stays inside the index array (mb >= 1)
    in.x = (j < mb && lane < S) ? obs_flat[(size_t)in.ri * stride + lane] : 0.f;    // rows >= mb: zeros (as trunk_rows)
    return in;
}
// a row's input into LDS + what the row leaves for the later kernels besides the trunk's own outputs
__device__ __forceinline__ void trunk_row_pre(int S, const TrunkRowIn& in, int j, int mb, int n_pad, int lane, float* xs, const cirs_ppo_batch& bt, int n_env,
                                              const TrunkRowOut& o) {
    if (lane < S) {
        xs[lane] = in.x;
        o.obs_copy[(size_t)j * S + lane] = in.x;
    }
    if (o.act) {   // the row's action, its row of the [T+1, B] tracker-gradient tensor and its four scalars in minibatch order (what the merge
                   // kernel used to gather through idx: the backward kernel's prologue reads them coalesced, without a dependent round trip)
        if (lane == 0) {
            o.act[j] = j < mb ? bt.act[in.ri] : 0;
            o.dst[j] = j < mb ? (long)bt.row_t[in.ri] * n_env + bt.row_env[in.ri] : 0;
        } else if (lane <= 4) {
            const float* srcp = lane == 1 ? bt.adv : lane == 2 ? bt.logp_old : lane == 3 ? bt.ret : bt.v_s;
            o.row4[(size_t)(lane - 1) * n_pad + j] = j < mb ? srcp[in.ri] : 0.f;
        }
    }
}
// The head kernels want H2 as fp16 planes in THEIR register order (hz: the lane's row, 8 consecutive columns per register quad; hb: 8
// rows of one column per quad).  Splitting here, once per row, replaces an LDS round trip + 8 split8 per wavefront in the prologue of every
// head workgroup (31 chunks x 8 row blocks re-split the same rows); each lane owns one element and drops its three 2-byte pieces
// into both layouts (same arithmetic as split_pair: same bits).  a = h2[j][lane].
__device__ __forceinline__ void trunk_row_planes(float a, int j, int lane, const TrunkRowOut& o) {
    a *= kScH2;
    const uint32_t hp = cvt_pk_f16(a, 0.f) & 0xffffu;
    const float r1 = a - (float)__builtin_bit_cast(f16x2_b, hp).x;
    const uint32_t lp = cvt_pk_f16(r1, 0.f) & 0xffffu;
    const int tile = j >> 5, rr = j & 31, c = lane;
    // hz: k-step s = c / 16, lane half hi = (c / 8) & 1, element c & 7; the consumer's lane is (hi, lo = row)
    unsigned short* z = reinterpret_cast<unsigned short*>(o.h2z + ((size_t)(tile * 12 + (c >> 4) * 3) * 64 + ((c >> 3) & 1) * 32 + rr)) + (c & 7);
    z[0] = (unsigned short)hp; z[2 * 64 * 8] = (unsigned short)lp;
    // hb[c / 32][t]: element jb of the consumer lane (hi_b, lo = c % 32), accumulator row rr = acc_row(8 t + jb, hi_b)
    const int hi_b = (rr >> 2) & 1, sb = (rr & 3) + 4 * (rr >> 3);
    unsigned short* bq = reinterpret_cast<unsigned short*>(o.h2b + ((size_t)(tile * 12 + ((c >> 5) * 2 + (sb >> 3)) * 3) * 64 + hi_b * 32 + (c & 31))) + (sb & 7);
    bq[0] = (unsigned short)hp; bq[2 * 64 * 8] = (unsigned short)lp;
}
// one minibatch row by one wavefront: trunk (the rollout's fma chains), the gathered row scalars, H2 as fp16 planes
__device__ __forceinline__ void trunk_row_job(const cirs_policy_cfg& cfg, const cirs_policy_weights& w, const TrunkLds& L, const TrunkRowIn& in, int j, int mb,
                                              int n_pad, int lane, float* xs, float* hs, const cirs_ppo_batch& bt, int n_env, const TrunkRowOut& o) {
    trunk_row_pre(cfg.dim_state, in, j, mb, n_pad, lane, xs, bt, n_env, o);
    trunk_compute(cfg, w, xs, hs, lane, j, o.h2, o.value, o.h1, L.w1, L.ld1, L.w2, L.wc);
    trunk_row_planes(xs[lane], j, lane, o);        // (h2[j][lane]: left in xs by trunk_compute)
}
// First launch of a stand-alone minibatch step, three independent jobs by workgroup index:
//   [0, n_row_wgs)                trunk forward of the minibatch rows (same fma chains as the rollout)
//   n_row_wgs                     advantage statistics of the (global) minibatch
//   (n_row_wgs, n_row_wgs + tiles] fp16 planes of one Wa item tile (operands of the two head kernels)
// Inside cirs_ppo_learn's loop only the first step of an update needs it: adam_next_kernel does the same three jobs for the step after it.
__global__ __launch_bounds__(256) void trunk_adv_kernel(cirs_policy_cfg cfg, cirs_policy_weights w, const float* __restrict__ obs_flat,
                                                        long stride, int n_pad, const int32_t* __restrict__ idx, int mb,
                                                        const float* __restrict__ adv_flat, const int32_t* __restrict__ sidx,
                                                        int m_stats, int enable, float* __restrict__ red, int n_row_wgs,
                                                        uint4* __restrict__ planes, cirs_ppo_batch bt, int n_env, TrunkRowOut out) {
    __shared__ float lds_raw[kTileN * 65];
    static_assert(kTileN * 65 >= 4 * 2 * kH, "the trunk rows use 4 x 2 x 64 floats of the same buffer");
    if ((int)blockIdx.x > n_row_wgs) {
        wa_planes_block((int)blockIdx.x - n_row_wgs - 1, cfg.n_items, w.wa, planes, lds_raw);
        return;
    }
    if ((int)blockIdx.x == n_row_wgs) {
        adv_stats_block(adv_flat, sidx, m_stats, enable, red, lds_raw);
        return;
    }
    // the four rows of this workgroup share the trunk weights: staged once, coalesced (W2 as float4 rows, W1 / wc as dwords), while
    // the row's own gather (idx -> obs row) is on its way
    __shared__ __attribute__((aligned(16))) float sW2[kH * kLdsRow2];
    __shared__ float sW1[kH * 33];
    __shared__ float sWc[kH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, S = cfg.dim_state;
    const int j = blockIdx.x * 4 + wv;                     // < n_pad: the grid has n_pad / 4 row workgroups
    const int ld1 = S | 1;     // odd row stride: lane o reads sW1[o * ld1 + k] without bank conflicts
    const bool al16 = (reinterpret_cast<uintptr_t>(w.w2) & 15) == 0;
    f32x4 t2[4];
    float t1[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* p = w.w2 + (size_t)(tid + 256 * q) * 4;
        t2[q] = al16 ? *reinterpret_cast<const f32x4*>(p) : f32x4{p[0], p[1], p[2], p[3]};
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) t1[q] = tid + 256 * q < kH * S ? w.w1[tid + 256 * q] : 0.f;
    const float tc = tid < kH ? w.wc[tid] : 0.f;
    const TrunkRowIn in = trunk_row_gather(j, mb, idx, obs_flat, stride, S, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i4 = tid + 256 * q;
        *reinterpret_cast<f32x4*>(&sW2[(i4 >> 4) * kLdsRow2 + (i4 & 15) * 4]) = t2[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = tid + 256 * q;
        if (i < kH * S) sW1[(i / S) * ld1 + (i % S)] = t1[q];
    }
    if (tid < kH) sWc[tid] = tc;
    float (*rows)[2][kH] = reinterpret_cast<float (*)[2][kH]>(lds_raw);
    __syncthreads();
    trunk_row_job(cfg, w, TrunkLds{sW1, ld1, sW2, sWc}, in, j, mb, n_pad, lane, rows[wv][0], rows[wv][1], bt, n_env, out);
}

constexpr int kTStride = 36;                 // transpose buffer row stride (floats): 16 B aligned, conflict-free b128 reads
constexpr int kRSize = kTileN * kH + 2 * kTileN; // per-wave dWa partial tile + the two half-waves' dba partials (no cross-half shuffle)
constexpr int kRowB = 144, kColB = 80;       // LDS row strides (bytes) of the R and C planes
constexpr int kRPlaneB = kTileN * kRowB, kCPlaneB = kH * kColB;
constexpr int kWBufB = 3 * kRPlaneB + 3 * kCPlaneB;

// ---- head statistics (forward): log-sum-exp and sum exp(z - m) z per row, f16x3 logits ---------------------------
// grid = (n_chunks, ceil(n_pad/32/4)); workgroup = 4 waves = 4 row tiles walking the item tiles of one chunk; the R planes
// of a Wa tile (12 KB) are staged once per workgroup, double-buffered.  Output: the per-chunk partials (m, s, t) of each
// row in the ActorPartialView arrays (score = t), merged by head_stats_merge_kernel.
#ifdef CIRS_HEAD_PROF
// stage timestamps of chosen workgroups of the small kernels of a minibatch step (probe builds only: tools/probes/step_prof.py)
__device__ unsigned long long g_step_prof[64];
#define CIRS_PSTAMP(COND, K) do { if ((COND) && threadIdx.x == 0) g_step_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CIRS_PSTAMP(COND, K) do { } while (0)
#endif
#ifdef CIRS_HEAD_PROF
// per-tile stage timestamps of workgroup (0, 0) / wave 0 at its third tile (probe builds only: tools/probes/head_prof.py)
__device__ unsigned long long g_head_prof[64];
#define CIRS_HSTAMP(K) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && it == 2) g_head_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
#define CIRS_TSTAMP(B, K) do { if ((int)blockIdx.x == (B) && threadIdx.x == 0) g_head_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
#define CIRS_SSTAMP(K) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_head_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CIRS_HSTAMP(K) do { } while (0)
#define CIRS_TSTAMP(B, K) do { } while (0)
#define CIRS_SSTAMP(K) do { } while (0)
#endif
__global__ __launch_bounds__(256, 2) void head_stats_kernel(int I, int mb, int n_pad, int tiles_per_chunk,
                                                            const uint4* __restrict__ planes, const float* __restrict__ ba,
                                                            const uint4* __restrict__ h2z, ActorPartialView pv,
                                                            const int32_t* __restrict__ act_rows, float* __restrict__ za_out) {
    __shared__ __attribute__((aligned(16))) unsigned char sW[2][3 * kRPlaneB];
    __shared__ __attribute__((aligned(16))) float sB[2][kTileN];   // 16-byte aligned: the accumulator init reads 4 consecutive biases as one ds_read_b128
    const int tid = threadIdx.x;
    CIRS_SSTAMP(40);
    const int lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int row0 = (blockIdx.y * 4 + wv) * kTileM;
    const int chunk = blockIdx.x;
    const int jr = row0 + lo;
    const bool wave_ok = row0 < n_pad && row0 < mb;   // some row of this tile belongs to the minibatch
    const bool active = jr < mb;
    HPl hz[4];  // B operand: this lane's row of H2, element j of k-step s = column 16 s + 8 hi + j
    {   // pre-split by trunk_adv_kernel in exactly this order: 12 coalesced 16-byte loads per lane, no LDS, no splits
        const uint4* zp = h2z + (size_t)((wave_ok ? row0 : 0) >> 5) * 12 * 64 + lane;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            HPL_SET(hz[s4], zp[(3 * s4) * 64], zp[(3 * s4 + 2) * 64]);
        }
    }
    float run_m = -INFINITY, run_s = 0.f, run_t = 0.f;
    // the logit of the row's taken action is one of this kernel's accumulator values (in exactly one chunk, tile, lane half): it is kept
    // for the backward kernel's prologue, which then needs neither the head row of the action nor a 64-term chain
    const int act_r = (act_rows && active) ? act_rows[jr] : -1;
    float za_val = 0.f;
    bool za_have = false;
    const int first_tile = chunk * tiles_per_chunk * kTileN;
    const int n_tiles = max(0, min(tiles_per_chunk, (I - first_tile + kTileN - 1) / kTileN));
    const int dst_r = (tid >> 3) * kRowB + (tid & 7) * 16;
    uint4 g0, g2;
    float gb = 0.f;
#define CIRS_ISSUE(TILE0)                                                                                  \
    do {                                                                                                   \
        const uint4* src_ = planes + (size_t)((TILE0) / kTileN) * kPlaneTileU4 + tid;                      \
        g0 = src_[0]; g2 = src_[512];                                             \
        if (tid < kTileN) gb = ((TILE0) + tid) < I ? kScZ * ba[(TILE0) + tid] : 0.f;   /* the accumulators hold kScZ z */ \
    } while (0)
#define CIRS_COMMIT(BUF)                                                                                   \
    do {                                                                                                   \
        unsigned char* base_ = sW[BUF];                                                                    \
        *reinterpret_cast<uint4*>(base_ + dst_r) = g0;                                                     \
        \
        *reinterpret_cast<uint4*>(base_ + 2 * kRPlaneB + dst_r) = g2;                                      \
        if (tid < kTileN) sB[BUF][tid] = gb;                                                               \
    } while (0)
    CIRS_SSTAMP(41);
    if (n_tiles > 0) { CIRS_ISSUE(first_tile); CIRS_COMMIT(0); }
    __syncthreads();
    CIRS_SSTAMP(42);
    for (int it = 0; it < n_tiles; ++it) {
        const int buf = it & 1;
        const int tile0 = first_tile + it * kTileN;
        if (it == 2) CIRS_SSTAMP(43);
        if (it + 1 < n_tiles) CIRS_ISSUE(tile0 + kTileN);
        if (wave_ok) {
            const unsigned char* tw = sW[buf];
            HPl za[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const unsigned char* ap = tw + lo * kRowB + (16 * s4 + 8 * hi) * 2;
                HPL_SET(za[s4], *reinterpret_cast<const uint4*>(ap), *reinterpret_cast<const uint4*>(ap + 2 * kRPlaneB));
            }
            f32x16 acc, accs, acct;      // bias + the h*h terms | the cross terms, two chains (bf16x6.h: the logits round once per k-step at their own magnitude)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = sB[buf][acc_row(r, hi)]; accs[r] = 0.f; acct[r] = 0.f; }
            if (it == 2) CIRS_SSTAMP(44);
            hmfma_split2(za[0], hz[0], za[1], hz[1], acc, accs, acct);
            hmfma_split2(za[2], hz[2], za[3], hz[3], acc, accs, acct);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = (acc[r] + (accs[r] + acct[r])) * kScZi;
            if (it == 2) CIRS_SSTAMP(45);
            {
                const int arel = act_r - tile0;
                const bool mine = act_r >= 0 && arel >= 0 && arel < kTileN && ((arel >> 2) & 1) == hi;
                if (__any(mine)) {       // (wave-uniform: most tiles hold no action of the wave's rows)
                    const int rsel = (arel & 3) + 4 * (arel >> 3);
                    float zsel = acc[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) zsel = rsel == r ? acc[r] : zsel;
                    if (mine) { za_val = zsel; za_have = true; }
                }
            }
            // log-sum-exp per tile: the lane's maximum first, then ONE rescale of the running sums and one exp per element;
            // items beyond I (last tile only) carry -inf and add exp(-inf) = 0
            f32x16 zt = acc;
            if (tile0 + kTileN > I) {
#pragma unroll
                for (int r = 0; r < 16; ++r) zt[r] = tile0 + acc_row(r, hi) < I ? acc[r] : -INFINITY;
            }
            float tmax = zt[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, zt[r]);
            if (tmax > -INFINITY) {
                const float mn = fmaxf(run_m, tmax);
                const float keep = __expf(run_m - mn);      // run_m = -inf on the first tile: keep = 0
                float ss = 0.f, tt = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ex = __expf(zt[r] - mn);
                    ss += ex;
                    tt = __builtin_fmaf(ex, acc[r], tt);   // acc is finite where ex is 0
                }
                run_s = __builtin_fmaf(run_s, keep, ss);
                run_t = __builtin_fmaf(run_t, keep, tt);
                run_m = mn;
            }
        }
        if (it == 2) CIRS_SSTAMP(46);
        if (it + 1 < n_tiles) CIRS_COMMIT(buf ^ 1);
        __syncthreads();
        if (it == 2) CIRS_SSTAMP(47);
    }
    CIRS_SSTAMP(48);
#undef CIRS_ISSUE
#undef CIRS_COMMIT
    if (row0 >= n_pad) return;
    if (!active) { run_m = -INFINITY; run_s = 0.f; run_t = 0.f; }
    {   // combine the two half-waves (same row, disjoint items)
        const float om = __shfl_xor(run_m, 32, CIRS_WAVE), osum = __shfl_xor(run_s, 32, CIRS_WAVE);
        const float ot = __shfl_xor(run_t, 32, CIRS_WAVE);
        const float mn = fmaxf(run_m, om);
        if (mn > -INFINITY) {
            const float fa = __expf(run_m - mn), fb = __expf(om - mn);
            run_s = run_s * fa + osum * fb;
            run_t = run_t * fa + ot * fb;
            run_m = mn;
        }
    }
    if (hi == 0) {
        const size_t po = (size_t)chunk * n_pad + jr;
        pv.score[po] = run_t; pv.m[po] = run_m; pv.s[po] = run_s;
    }
    if (za_have) za_out[jr] = za_val;
}

// ---- head backward (fused): dWa, dba, d h2, entropy correction ------------------------------------------------
// grid = (n_chunks, ceil(n_pad/32/kBwdWaves)), ONE workgroup per CU; workgroup = kBwdWaves waves = that many ROW tiles
// walking the item tiles of one chunk.  Per (row tile, item tile) the logits are recomputed ONCE; all three products run
// as f16x3 products (bf16x6.h):
//   ZT[32 items x 32 rows] = Wa_tile * H2_tile^T             (A = Wa planes R from LDS, B = this lane's H2 row, registers)
//   dZ in place (lane owns a ROW: its lse / coefficients / action are scalars)
//   dH2_tile[32 rows x 64] += dZ[rows x items] * Wa_tile      (A = the dZT registers split in place, B = Wa planes C, LDS)
//   dZ -> LDS -> registers in the transposed (Z) layout: lane owns an ITEM, registers are rows (4 KB per wave)
//   dWa_tile[32 items x 64]  = dZ^T[items x rows] * H2_tile   (A = transposed dZ registers split, B = H2 planes, registers)
// d h2 stays in the accumulators across the chunk (one partial slab per chunk, summed by trunk_bwd_kernel); the dWa
// tile of each wave covers only its 32 rows, so the kBwdWaves partial tiles are summed through LDS in wave order and
// written to the slab of this ROW BLOCK (n_row_blocks slabs, summed in slab order by the extra workgroups of the trunk-backward launch).
// The Wa planes of a tile (24 KB, written by wa_planes_kernel) are staged in LDS once per workgroup, double-buffered with
// the next tile's global loads in flight; row strides 144 B / 80 B keep the ds_read_b128 of the operands conflict-free.


// schedule knobs of head_bwd_fused_kernel (A/B-measured on one box with tools/ab_headbwd.py; defaults = the measured best)
#ifndef CIRS_BWD_COMMIT_EARLY
#define CIRS_BWD_COMMIT_EARLY 1   // 1: next tile's planes -> LDS before the dH2 product; 0: at the end of the iteration
#endif
#ifndef CIRS_BWD_REDUCE_POS
#define CIRS_BWD_REDUCE_POS 0     // where the previous tile's dWa sum runs: 0 around the second logits MFMA group, 1 around the dH2 product
#endif
// kEnt: the entropy term of dZ is compiled in (ent_coef != 0); the reference's scripts train with ent_coef = 0 (CIRS-RL-kuaishou.py:97),
// where dZ = c_logp (delta - p) and the entropy is only reported.
// kMerge: the merge of the head-statistics partials and the row's loss terms / backward coefficients run in THIS kernel's prologue (every
// workgroup for its own rows, the workgroups of chunk 0 store what later kernels read) instead of a launch of their own between the two head
// kernels: a 6.5 us launch on the critical path of every minibatch step becomes ~1.5 us of prologue.  The item-sharded learner keeps the
// merge kernel (its statistics cross the ranks first).
struct HeadMergeArgs { cirs_ppo_cfg cfg; cirs_ppo_batch b; const int32_t* idx; int mb_norm, n_schunks; ActorPartialView pv; int t_all; /* A/B switch: every workgroup loads the E_p[z] partials */ };
template <bool kEnt, bool kMerge>
__global__ __launch_bounds__(kBwdWaves * 64, 1) void head_bwd_fused_kernel(int I, int mb, int n_pad, int tiles_per_chunk,
                                                                         const uint4* __restrict__ planes,
                                                                         const float* __restrict__ ba, MbView v,
                                                                         float* __restrict__ dwap, HeadMergeArgs ma) {
    constexpr int kThreads = kBwdWaves * 64;
    static_assert(kThreads == 256, "the plane staging maps one 16-byte unit per thread and plane");
    __shared__ __attribute__((aligned(16))) unsigned char sW[2][kWBufB];
    __shared__ __attribute__((aligned(16))) float sB[2][kTileN];   // 16-byte aligned: the accumulator init reads 4 consecutive biases as one ds_read_b128
    __shared__ __attribute__((aligned(16))) float sT[kBwdWaves][kTileN * kTStride];
    __shared__ __attribute__((aligned(16))) float sR[2][kBwdWaves][kRSize];   // double-buffered: the sum of tile t runs inside iteration t + 1
    const int tid = threadIdx.x;
    CIRS_SSTAMP(30);
    if (blockIdx.x == 0 && blockIdx.y == 0) {      // arrival flags of this step's trunk-backward / optimiser launches
        v.sync[tid] = 0;
        if (tid < kSyncInts - 256) v.sync[256 + tid] = 0;
    }
    const int lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int row0 = (blockIdx.y * kBwdWaves + wv) * kTileM;
    const bool wave_ok = row0 < n_pad;   // all waves take part in the staging, barriers and the slab reduction
    const int chunk = blockIdx.x;
    // gridDim.x is padded to a multiple of 8: workgroups are dealt round-robin to the 8 XCDs, so linear id % 8 = chunk % 8
    // and the row blocks that walk the same chunk share one L2 (the Wa planes of a tile leave HBM / MALL once, not 8 times)
    if (chunk * tiles_per_chunk * kTileN >= I) return;
    const int jr = wave_ok ? row0 + lo : 0;
    // The wave's 32 x 64 tile of H2 is 8 KB of consecutive memory: read coalesced (8 x 1 KB per wave) and handed to the lanes through
    // LDS -- a lane reading its own 256-byte row costs 64 cache lines per load instruction (the prologue was 6.1 k of the kernel's
    // 77 k ticks).  The row scalars are requested first; they travel while the tile does.
    const float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    const int act = v.act[jr];
    HPl hz[4];      // B operand of ZT: this lane's row of H2, element j of k-step s = column 16 s + 8 hi + j
    HPl hb[2][2];   // B operand of the dWa product: element j of k-step t = H2[row acc_row(8 t + j, hi)][32 c + lo]
    {   // both pre-split by trunk_adv_kernel in register order: 24 coalesced 16-byte loads per lane (the LDS round trip + 8 split8 of
        // the round-2 prologue were 4.4 k of the kernel's 78 k ticks)
        const size_t tb = (size_t)((wave_ok ? row0 : 0) >> 5) * 12 * 64 + lane;
        const uint4* zp = v.h2z + tb;
        const uint4* bp = v.h2b + tb;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            HPL_SET(hz[s4], zp[(3 * s4) * 64], zp[(3 * s4 + 2) * 64]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                HPL_SET(hb[c][t], bp[(3 * (2 * c + t)) * 64], bp[(3 * (2 * c + t) + 2) * 64]);
            }
    }
    const bool row_ok = wave_ok && jr < mb;
    f32x16 dh0, dh1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dh0[r] = 0.f; dh1[r] = 0.f; }
    float ent = 0.f;  // clamp correction of the entropy (see below)
    const float eps = 1.1920928955078125e-7f;
    const float kLogEps = -15.942385152878742f, kLog1mEps = -1.1920929665620834e-7f;
    if (!wave_ok) {
        for (int q = lane; q < kRSize; q += 64) { sR[0][wv][q] = 0.f; sR[1][wv][q] = 0.f; }
    }

    const int first_tile = chunk * tiles_per_chunk * kTileN;
    const int n_tiles = max(0, min(tiles_per_chunk, (I - first_tile + kTileN - 1) / kTileN));
    // staging: unit `tid` of each of the six planes; destination offsets inside a buffer
    const int dst_r = (tid >> 3) * kRowB + (tid & 7) * 16;
    const int dst_c = 3 * kRPlaneB + (tid >> 2) * kColB + (tid & 3) * 16;
    uint4 gr0, gr2, gc0, gc2;
    float gb = 0.f;
#define CIRS_ISSUE(TILE0)                                                                                  \
    do {                                                                                                   \
        const uint4* src_ = planes + (size_t)((TILE0) / kTileN) * kPlaneTileU4 + tid;                      \
        gr0 = src_[0]; gr2 = src_[512]; gc0 = src_[768]; gc2 = src_[1280]; \
        if (tid < kTileN) gb = ((TILE0) + tid) < I ? kScZ * ba[(TILE0) + tid] : 0.f;   /* the accumulators hold kScZ z */ \
    } while (0)
#define CIRS_COMMIT(BUF)                                                                                   \
    do {                                                                                                   \
        unsigned char* base_ = sW[BUF];                                                                    \
        *reinterpret_cast<uint4*>(base_ + dst_r) = gr0;                                                    \
        \
        *reinterpret_cast<uint4*>(base_ + 2 * kRPlaneB + dst_r) = gr2;                                     \
        *reinterpret_cast<uint4*>(base_ + dst_c) = gc0;                                                    \
        \
        *reinterpret_cast<uint4*>(base_ + 2 * kCPlaneB + dst_c) = gc2;                                     \
        if (tid < kTileN) sB[BUF][tid] = gb;                                                               \
    } while (0)
    // the first tile's planes and the H2 planes above are on their way while the row statistics are merged
    if (n_tiles > 0) CIRS_ISSUE(first_tile);
    float lse, c_logp, c_ent, h_ent;
    if (!kMerge) {
        lse = v.lse[jr]; c_logp = v.c_logp[jr]; c_ent = kEnt ? v.c_ent[jr] : 0.f; h_ent = kEnt ? v.h_ent[jr] : 0.f;
    } else {
        // lane (row lo, half hi) folds every second chunk partial of its row -- first the maxima, then the sums against the row maximum: two
        // batches of independent, unconditional, coalesced loads -- and the halves meet by one shuffle
        const bool real = wave_ok && jr < mb;
        const int jc = real ? jr : 0;
        // v.adv / v.ret / v.v_s / v.logp_old are consecutive [n_pad] arrays filled by trunk_adv_kernel in minibatch order
        const float za = v.za[jc], adv = v.adv[jc], lpo = v.adv[n_pad + jc], ret = v.adv[2 * (size_t)n_pad + jc], vs = v.adv[3 * (size_t)n_pad + jc],
                    val = v.value[jc];
        const float red0 = v.red[0], red1 = v.red[1];
        // the chunk partials (m, s, t) of the row, 32 per lane half and batch, ALL requested before the first is used (a plain loop waits for
        // every load: 56 dependent round trips were 13 us of prologue); indices beyond the last chunk are clamped and weighted 0
        const int nsc = ma.n_schunks;
        const float* __restrict__ pm_ = ma.pv.m + jc;
        const float* __restrict__ ps_ = ma.pv.s + jc;
        const float* __restrict__ pt_ = ma.pv.score + jc;
        float M = -INFINITY, ssum = 0.f, tsum = 0.f;
        // E_p[z]'s partials only where somebody uses them: every workgroup when the entropy term is in dZ, else the chunk-0 workgroups (they store the reported
        // entropy) -- a third of the prologue's 86 KB per workgroup for the other 30 of 31 (a prologue costs what its bytes cost: ~11 B/cycle per CU)
        const bool need_t = kEnt || blockIdx.x == 0 || ma.t_all;
        for (int cb = 0; cb < nsc; cb += 64) {          // (wave-uniform trip count: the two halves meet in a shuffle inside)
            const int c0 = cb + hi;
            float m32[32], s32[32], u32[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int c = c0 + 2 * q;
                const size_t o = (size_t)(c < nsc ? c : nsc - 1) * n_pad;
                m32[q] = pm_[o]; s32[q] = ps_[o];
            }
            if (need_t) {
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const int c = c0 + 2 * q;
                    u32[q] = pt_[(size_t)(c < nsc ? c : nsc - 1) * n_pad];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 32; ++q) u32[q] = 0.f;
            }
            float mb_ = -INFINITY;
#pragma unroll
            for (int q = 0; q < 32; ++q) mb_ = fmaxf(mb_, m32[q]);
            mb_ = fmaxf(mb_, __shfl_xor(mb_, 32, CIRS_WAVE));      // both halves rescale to the same maximum
            const float Mn = fmaxf(M, mb_);
            const float keep = __expf(M - Mn);                     // first batch: exp(-inf) = 0
            ssum *= keep; tsum *= keep;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const float f = c0 + 2 * q < nsc ? __expf(m32[q] - Mn) : 0.f;     // (-inf, 0, 0) partials of chunks beyond the catalogue: f = 0
                ssum = __builtin_fmaf(s32[q], f, ssum);
                tsum = __builtin_fmaf(u32[q], f, tsum);
            }
            M = Mn;
        }
        ssum += __shfl_xor(ssum, 32, CIRS_WAVE);
        tsum += __shfl_xor(tsum, 32, CIRS_WAVE);
        const float lse_r = M + __logf(ssum);
        const RowTerms rt = ppo_row_terms(ma.cfg, za, lse_r, tsum / ssum, lpo, adv, red0, red1, val, vs, ret, 1.0f / (float)ma.mb_norm);
        lse = real ? lse_r : 1e30f;          // rows beyond the minibatch: p = exp(z - lse) = 0, neutral coefficients (as the merge kernel writes)
        c_logp = real ? rt.c_logp : 0.f;
        c_ent = (kEnt && real) ? rt.c_ent : 0.f;
        h_ent = (kEnt && real) ? rt.h_ent : 0.f;
        if (blockIdx.x == 0 && hi == 0 && wave_ok) {      // one writer per row: what trunk_bwd_kernel and the loss sums read
            v.dvalue[jr] = real ? rt.dvalue : 0.f;
            v.clip_row[jr] = real ? rt.clip_row : 0.f;
            v.vf_row[jr] = real ? rt.vf_row : 0.f;
            v.h_ent[jr] = real ? rt.h_ent : 0.f;       // the reported entropy (dh2_sum_kernel: ent_row = h_ent + clamp correction)
        }
    }
    // f16x3: dZ is formed ALREADY SCALED by a power of two per workgroup -- its 128 rows' largest |c_logp| + 16 |c_ent| (a bound on |dZ|) lands in [2^13, 2^14) --
    // so that its fp16 pieces are normal numbers; dH2 / dWa / dba are unscaled exactly where they leave the kernel.
    float dz_scale = 1.0f, dz_inv = 1.0f;
    __shared__ float s_cmax[kBwdWaves];
    {
        const float cm = wave_max_f32_dpp(fabsf(c_logp) + 16.0f * fabsf(c_ent));
        if (lane == 0) s_cmax[wv] = cm;
    }
    CIRS_SSTAMP(31);
    if (n_tiles > 0) CIRS_COMMIT(0);
    __syncthreads();
    {
        float cm = s_cmax[0];
#pragma unroll
        for (int q = 1; q < kBwdWaves; ++q) cm = fmaxf(cm, s_cmax[q]);
        int ex = 0;
        if (cm > 0.f) { (void)frexpf(cm, &ex); ex = max(ex, -80); dz_scale = ldexpf(1.0f, 14 - ex); dz_inv = ldexpf(1.0f, ex - 14); }
        c_logp *= dz_scale; c_ent *= dz_scale;
    }
    const float nlse2 = -(lse * kLog2e), ncl = -c_logp;   // p = exp2(z log2e - lse log2e); padded rows: lse = 1e30 -> p = 0
    CIRS_SSTAMP(32);
    float* slab = dwap + (size_t)blockIdx.y * dwa_slab_stride(I);
    // sum of the kBwdWaves partial dWa tiles of one item tile in wave order -> slab of this row block (coalesced float4 stores).
    // It runs one iteration late, between the operand reads and the MFMAs of the NEXT tile: its LDS round trip and its stores hide
    // behind that tile's operand latency / matrix work instead of standing alone behind a second workgroup barrier.
    // two halves, so that the caller can put matrix work between them: the LDS reads of all partial tiles, then the adds + stores
    constexpr int kRedQ = (kTileN * kH / 4) / kThreads;
    const float dw_unscale = dz_inv * (1.0f / kScH2), dh_unscale = dz_inv * (1.0f / kScWa);      // (exact powers of two)
    struct RedRegs { f32x4 t[kRedQ][kBwdWaves]; float b[kBwdWaves]; };
    auto reduce_load = [&](int rb, RedRegs& rg) {
#pragma unroll
        for (int q = 0; q < kRedQ; ++q)
#pragma unroll
            for (int w2 = 0; w2 < kBwdWaves; ++w2) rg.t[q][w2] = reinterpret_cast<const f32x4*>(sR[rb][w2])[tid + kThreads * q];
        if (wv == 0) {    // wave-uniform: the bias partials are 64 floats per wave (two half-wave sums per item), lanes 0..31 combine them
#pragma unroll
            for (int w2 = 0; w2 < kBwdWaves; ++w2) rg.b[w2] = sR[rb][w2][kTileN * kH + lo] + sR[rb][w2][kTileN * kH + kTileN + lo];
        }
    };
    auto reduce_store = [&](const RedRegs& rg, int t0) {
        const bool full = t0 + kTileN <= I;     // wave-uniform: only the catalogue's last tile masks its stores per lane
#pragma unroll
        for (int q = 0; q < kRedQ; ++q) {
            const int f = tid + kThreads * q;  // float4 index within the 32 x 64 tile
            f32x4 t = rg.t[q][0];
#pragma unroll
            for (int w2 = 1; w2 < kBwdWaves; ++w2) t += rg.t[q][w2];
            t *= dw_unscale;
            if (full || t0 + (f >> 4) < I) *reinterpret_cast<f32x4*>(slab + (size_t)t0 * kH + 4 * f) = t;
        }
        if (wv == 0) {
            float t = rg.b[0];
#pragma unroll
            for (int w2 = 1; w2 < kBwdWaves; ++w2) t += rg.b[w2];
            if (hi == 0 && (full || t0 + lo < I)) slab[(size_t)I * kH + t0 + lo] = t * dz_inv;
        }
    };
    auto reduce_tile = [&](int rb, int t0) { RedRegs rg; reduce_load(rb, rg); reduce_store(rg, t0); };
    for (int it = 0; it < n_tiles; ++it) {
        const int buf = it & 1;
        const int tile0 = first_tile + it * kTileN;
        CIRS_HSTAMP(0);
        if (it + 1 < n_tiles) CIRS_ISSUE(tile0 + kTileN);  // global loads in flight during the MFMAs below
        f32x16 dw0, dw1;
        float db = 0.f;
        if (!wave_ok) {     // idle row tile (padding of the last row block): staging and the reduction only
            if (it + 1 < n_tiles) CIRS_COMMIT(buf ^ 1);
            if (it > 0) reduce_tile(buf ^ 1, tile0 - kTileN);
        }
        if (wave_ok) {
            const unsigned char* tw = sW[buf];
            // every LDS operand of this tile up front: one latency exposure instead of one per k-step
            HPl za[4], cb[2][2];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const unsigned char* ap = tw + lo * kRowB + (16 * s4 + 8 * hi) * 2;
                HPL_SET(za[s4], *reinterpret_cast<const uint4*>(ap), *reinterpret_cast<const uint4*>(ap + 2 * kRPlaneB));
            }
            f32x16 acc, acc1, acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = sB[buf][acc_row(r, hi)]; acc1[r] = 0.f; acc2[r] = 0.f; }
            CIRS_HSTAMP(1);
            // (acc: bias + the h*h terms, acc1 / acc2: the cross terms -- same split as head_stats_kernel, so that p = exp(z - lse) sums to one)
            hmfma_split2(za[0], hz[0], za[1], hz[1], acc, acc1, acc2);
            RedRegs rg;
            if (CIRS_BWD_REDUCE_POS == 0 && it > 0) reduce_load(buf ^ 1, rg);
            hmfma_split2(za[2], hz[2], za[3], hz[3], acc, acc1, acc2);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] += acc2[r];
            if (CIRS_BWD_REDUCE_POS == 0 && it > 0) reduce_store(rg, tile0 - kTileN);
            // the B planes of the dH2 product are requested only now: the A planes of the logits are dead (the two sets never coexist:
            // the kernel runs at the 256-VGPR limit and every value beyond it costs an AGPR copy per use) and the dZ arithmetic below
            // covers their LDS latency
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const unsigned char* bp = tw + 3 * kRPlaneB + (32 * c + lo) * kColB + (16 * t + 8 * hi) * 2;
                    HPL_SET(cb[c][t], *reinterpret_cast<const uint4*>(bp), *reinterpret_cast<const uint4*>(bp + 2 * kCPlaneB));
                }
            CIRS_HSTAMP(2);
            float* tt = sT[wv];
            // dZ.  t = (z - lse) log2 e as ONE fma, p = exp2(t), d = -c_logp p (+ the entropy term when compiled in); the action's
            // element gets + c_logp in a second pass that only tiles containing some row's action execute (wave-uniform branch: ~9 % of
            // the tiles at 32 rows x 336 tiles).  Padded rows carry zero coefficients and lse = 1e30 (p = 0, d = 0); items beyond I exist
            // only in the last tile (zero weights -> finite z) and are masked there.  Categorical.entropy uses log(clamp(p, eps, 1-eps)):
            // the un-clamped entropy lse - E_p[z] comes from the forward statistics and only clamped elements (p < eps or p > 1 - eps)
            // contribute a correction, applied below when the wave has any (t is kept for that; p is recomputed there).
            // Clamp correction of the entropy, low side (p < eps = 2^-23 <=> t < -23): corr = p (log eps - (z - lse)) = ln2 p (-23 - t),
            // accumulated branch-free as p * max(-23 - t, 0) -- 3 VALU per element; a wave-level "any clamped element?" branch is taken
            // in most tiles once the policy has sharpened (measured: 35.3 us untrained vs 40.6 us after 65 updates).  High side
            // (p > 1 - eps: one item holds the whole row) stays a rare branch.
            f32x16 tk;
            float pmax = 0.f, elo = 0.f;
            f32x2_b elo2 = {0.f, 0.f};
            const float kTEps = -23.0f;
            const bool last_tile = tile0 + kTileN > I;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float t0 = __builtin_fmaf(acc[r] + acc1[r], kLog2e * kScZi, nlse2), t1 = __builtin_fmaf(acc[r + 1] + acc1[r + 1], kLog2e * kScZi, nlse2);
                const float p0 = __builtin_amdgcn_exp2f(t0), p1 = __builtin_amdgcn_exp2f(t1);
                tk[r] = t0; tk[r + 1] = t1;
                pmax = __builtin_fmaxf(__builtin_fmaxf(pmax, p0), p1);      // v_max3_f32
                if (!last_tile) {    // two independent partial sums as one packed subtract + packed fma (the max has no packed form)
                    const f32x2_b d2 = f32x2_b{kTEps, kTEps} - f32x2_b{t0, t1};
                    const f32x2_b w2 = {__builtin_fmaxf(d2.x, 0.f), __builtin_fmaxf(d2.y, 0.f)};
                    elo2 = f32x2_b{p0, p1} * w2 + elo2;
                }
                if (kEnt) {   // + c_ent p (z - lse + H)
                    acc[r] = __builtin_fmaf(c_ent * p0, __builtin_fmaf(t0, kLn2, h_ent), ncl * p0);
                    acc[r + 1] = __builtin_fmaf(c_ent * p1, __builtin_fmaf(t1, kLn2, h_ent), ncl * p1);
                } else {
                    acc[r] = ncl * p0; acc[r + 1] = ncl * p1;
                }
            }
            if (__any(act >= tile0 && act < tile0 + kTileN)) {
                const int rel = act - tile0 - 4 * hi;     // accumulator register r holds item (r & 3) + 8 (r >> 2) + 4 hi of the tile
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += (rel == (r & 3) + 8 * (r >> 2)) ? c_logp : 0.f;
            }
            if (last_tile) {     // items beyond I (zero weights -> finite z): no gradient, no entropy term
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool in = tile0 + acc_row(r, hi) < I;
                    acc[r] = in ? acc[r] : 0.f;
                    elo += in ? __builtin_amdgcn_exp2f(tk[r]) * __builtin_fmaxf(kTEps - tk[r], 0.f) : 0.f;
                }
            }
            ent -= kLn2 * (elo + (elo2.x + elo2.y));      // padded rows: p = 0 exactly (lse = 1e30), so they add nothing
            CIRS_HSTAMP(3);
#pragma unroll
            for (int r = 0; r < 16; ++r) tt[acc_row(r, hi) * kTStride + lo] = acc[r];   // transposed exchange: T[item][row]
            if (__any(row_ok && pmax > 1.0f - eps)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(tk[r]);
                    const bool ok = row_ok && (!last_tile || tile0 + acc_row(r, hi) < I);
                    ent -= (ok && p > 1.0f - eps) ? p * (kLog1mEps - tk[r] * kLn2) : 0.f;
                }
            }
            CIRS_HSTAMP(4);
            // the wave's own LDS writes above are read back by other lanes of the same wave.  The read-back is requested BEFORE the dH2
            // product: its latency and the second split (VALU) then sit in the issue gaps of the 24 dH2 MFMAs, which do not depend on them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            f32x16 dzt;  // lane owns item lo; register r' = row acc_row(r', hi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t4 = *reinterpret_cast<const float4*>(&tt[lo * kTStride + 8 * g + 4 * hi]);
                dzt[4 * g] = t4.x; dzt[4 * g + 1] = t4.y; dzt[4 * g + 2] = t4.z; dzt[4 * g + 3] = t4.w;
            }
            // next tile's planes -> the other LDS buffer (last read in iteration it - 1, one barrier ago): the prefetch was requested at
            // the top of this iteration, and no store of this iteration has been issued yet (the wait is for loads only)
            if (CIRS_BWD_COMMIT_EARLY && it + 1 < n_tiles) CIRS_COMMIT(buf ^ 1);
            if (CIRS_BWD_REDUCE_POS == 1 && it > 0) reduce_load(buf ^ 1, rg);      // the previous tile's partial dWa tiles: their LDS round trip hides behind the dH2 product
            {
                const HPl a0 = hsplit8(acc, 0), a1 = hsplit8(acc, 8);   // element j: dZ[row lo][item acc_row(8 t + j, hi)]
                hmfma_pair(a0, cb[0][0], cb[1][0], dh0, dh1);
                hmfma_pair(a1, cb[0][1], cb[1][1], dh0, dh1);
            }
            if (CIRS_BWD_REDUCE_POS == 1 && it > 0) reduce_store(rg, tile0 - kTileN);
            CIRS_HSTAMP(5);
            CIRS_HSTAMP(6);
#pragma unroll
            for (int r = 0; r < 16; ++r) { dw0[r] = 0.f; dw1[r] = 0.f; db += dzt[r]; }
            {
                const HPl b0 = hsplit8(dzt, 0), b1 = hsplit8(dzt, 8);   // element j: dZ[row acc_row(8 t + j, hi)][item lo]
                hmfma_pair(b0, hb[0][0], hb[1][0], dw0, dw1);
                hmfma_pair(b1, hb[0][1], hb[1][1], dw0, dw1);
            }
        }
        CIRS_HSTAMP(7);
        if (wave_ok) {
            float* rr = sR[buf][wv];    // this buffer was last read in iteration it - 1 (the sum of tile it - 2), one barrier ago
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = acc_row(r, hi);
                rr[il * kH + lo] = dw0[r];
                rr[il * kH + 32 + lo] = dw1[r];
            }
            rr[kTileN * kH + lane] = db;     // slot 32 hi + lo: the reducer adds the two halves
        }
        if (!CIRS_BWD_COMMIT_EARLY && wave_ok && it + 1 < n_tiles) CIRS_COMMIT(buf ^ 1);
        CIRS_HSTAMP(9);
        lds_barrier();   // ONE workgroup barrier per tile: partial tiles + next tile's planes are visible
        CIRS_HSTAMP(10);
    }
    CIRS_SSTAMP(33);
    if (n_tiles > 0) reduce_tile((n_tiles - 1) & 1, first_tile + (n_tiles - 1) * kTileN);
    CIRS_SSTAMP(34);
#undef CIRS_ISSUE
#undef CIRS_COMMIT
    if (!wave_ok) return;
    float* hslab = v.dh2p + (size_t)chunk * n_pad * kH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + acc_row(r, hi);
        hslab[(size_t)row * kH + lo] = dh0[r] * dh_unscale;
        hslab[(size_t)row * kH + 32 + lo] = dh1[r] * dh_unscale;
    }
    ent += __shfl_xor(ent, 32, CIRS_WAVE);
    if (hi == 0) v.entp[(size_t)chunk * n_pad + jr] = ent;
    CIRS_SSTAMP(35);
}

// ---- slab sums of the wa|ba gradient (one of kWaSumBlocks workgroups of 512 threads) ------------------------------------
// The n_slabs row-block partials written by head_bwd_fused_kernel are summed in slab order into the flat gradient, one
// float4 per thread and slab (16-byte aligned: wa_beg and slab_stride are multiples of 4), 8 independent loads in flight;
// the workgroup's sum of squares goes to partial_out (clip_grad_norm_ stage 1).  Runs as extra workgroups of the
// trunk-backward launch: it only depends on the head backward kernel.
constexpr int kNormBlocks = 256;   // workgroups of sumsq_partial_kernel = its slots of the norm partials
constexpr int kWaSumBlocks = 176;  // slots [kNormBlocks, kNormBlocks + kWaSumBlocks) of the norm partials
__device__ __forceinline__ void wa_slab_sum_block(float* __restrict__ g, long wa_beg, long wa_len, const float* __restrict__ dwap,
                                                  long slab_stride, int n_slabs, int b, float* __restrict__ partial_out, float* sh) {
    const int tid = threadIdx.x;  // blockDim.x == 512
    float acc = 0.f;
    const long n4 = wa_len >> 2;
    // two float4 per thread and pass, all 2 x 8 slab loads in flight together; kWaSumBlocks x 1024 float4 cover wa|ba of the C3
    // catalogue (174 k float4) in ONE pass = one memory round trip per workgroup
    for (long q4 = b * 1024L + tid; q4 < n4; q4 += (long)kWaSumBlocks * 1024) {
        const long q4b = q4 + 512;
        const bool has_b = q4b < n4;
        f32x4 ta[8], tb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ta[q] = q < n_slabs ? *reinterpret_cast<const f32x4*>(dwap + (size_t)q * slab_stride + 4 * q4) : f32x4{0.f, 0.f, 0.f, 0.f};
            tb[q] = (has_b && q < n_slabs) ? *reinterpret_cast<const f32x4*>(dwap + (size_t)q * slab_stride + 4 * q4b) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q) { x += ta[q]; y += tb[q]; }
        for (int s0 = 8; s0 < n_slabs; ++s0) {   // more than 8 row blocks (minibatches beyond 1024 rows): the rest, in slab order
            x += *reinterpret_cast<const f32x4*>(dwap + (size_t)s0 * slab_stride + 4 * q4);
            if (has_b) y += *reinterpret_cast<const f32x4*>(dwap + (size_t)s0 * slab_stride + 4 * q4b);
        }
        *reinterpret_cast<f32x4*>(g + wa_beg + 4 * q4) = x;
        acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
        if (has_b) {
            *reinterpret_cast<f32x4*>(g + wa_beg + 4 * q4b) = y;
            acc += (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
        }
    }
    const long wi = (n4 << 2) + b * 512L + tid;   // the (< 4) elements after the last whole float4 (b = 0 covers them)
    if (wi < wa_len) {
        float x = 0.f;
        for (int s0 = 0; s0 < n_slabs; ++s0) x += dwap[(size_t)s0 * slab_stride + wi];
        g[wa_beg + wi] = x;
        acc += x * x;
    }
    sh[tid] = acc;
    __syncthreads();
    for (int st = 256; st > 0; st >>= 1) {
        if (tid < st) sh[tid] += sh[tid + st];
        __syncthreads();
    }
    if (tid == 0) *partial_out = sh[0];
}

// The same sums for trunk_rows_kernel: ONE float4 per thread and pass (8 slab loads, 64 KB in flight per workgroup) and as many workgroups as it takes
// (340 at C3 instead of 176 x 128 KB): a CU pulls ~11 B/cycle at kernel start whatever is asked of it, so the slab sums take what the SLOWEST CU needs
// for its share -- many small workgroups let the dispatcher hand the work to whichever CU is free (the row workgroups of the same launch sit on 128 of them).
constexpr int kWaSlotsMax = 768;   // slots [kNormBlocks, kNormBlocks + 768) of the norm partials: all of them are folded, unused ones hold 0
__device__ __forceinline__ void wa_slab_sum_block1(float* __restrict__ g, long wa_beg, long wa_len, const float* __restrict__ dwap,
                                                   long slab_stride, int n_slabs, int b, int n_w, float* __restrict__ partial_out, float* sh) {
    const int tid = threadIdx.x;  // blockDim.x == 512
    float acc = 0.f;
    const long n4 = wa_len >> 2;
    for (long q4 = b * 512L + tid; q4 < n4; q4 += (long)n_w * 512) {
        f32x4 ta[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ta[q] = q < n_slabs ? *reinterpret_cast<const f32x4*>(dwap + (size_t)q * slab_stride + 4 * q4) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q) x += ta[q];
        for (int s0 = 8; s0 < n_slabs; ++s0) x += *reinterpret_cast<const f32x4*>(dwap + (size_t)s0 * slab_stride + 4 * q4);   // (minibatches beyond 1024 rows)
        *reinterpret_cast<f32x4*>(g + wa_beg + 4 * q4) = x;
        acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    }
    const long wi = (n4 << 2) + b * 512L + tid;   // the (< 4) elements after the last whole float4 (b = 0 covers them)
    if (wi < wa_len) {
        float x = 0.f;
        for (int s0 = 0; s0 < n_slabs; ++s0) x += dwap[(size_t)s0 * slab_stride + wi];
        g[wa_beg + wi] = x;
        acc += x * x;
    }
    sh[tid] = acc;
    __syncthreads();
    for (int st = 256; st > 0; st >>= 1) {
        if (tid < st) sh[tid] += sh[tid + st];
        __syncthreads();
    }
    if (tid == 0) *partial_out = sh[0];
}

// one 32 x 32 tile of dW = dY^T X (+ the bias column sum of dY when k0 == 0) over the 32 rows of a trunk-backward workgroup:
// A = dY from its LDS copy (sY[row][o]), B = X rows from global memory; written as this workgroup's row slab of the job
// (partial layout of dw_multi_final / dw_multi_fetch: out[o * (K + 1) + k], k == K the bias column)
// B operand (the X rows) of a dw_tile_32rows tile: depends on nothing computed in the kernel, so a caller requests it at kernel entry
__device__ __forceinline__ void dw_tile_load_x(const float* __restrict__ X, int ldx, int K, int row0, int n_rows, int k0, int lane, float (&b)[16]) {
    const int hi = lane >> 5, k = k0 + (lane & 31);
    const bool k_ok = k < K;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        b[j] = (k_ok && row0 + 2 * j + hi < n_rows) ? X[(size_t)(row0 + 2 * j + hi) * ldx + k] : 0.f;  // rows beyond the minibatch: dY = 0
}
// the same operand from an LDS copy of the tile's X rows (sX[row][k], row stride ldx)
__device__ __forceinline__ void dw_tile_x_lds(const float* sX, int ldx, int K, int row0, int n_rows, int k0, int lane, float (&b)[16]) {
    const int hi = lane >> 5, k = k0 + (lane & 31);
    const bool k_ok = k < K;
#pragma unroll
    for (int j = 0; j < 16; ++j) b[j] = (k_ok && row0 + 2 * j + hi < n_rows) ? sX[(2 * j + hi) * ldx + k] : 0.f;
}
// Values that cross workgroups INSIDE one launch are written through (sc1 stores, completed before the arrival is counted) and read around
// the non-coherent caches (sc1 loads): the write-through form of MI355X_MICROARCH.md's inter-workgroup visibility rules -- no L2 write-back fence.
__device__ __forceinline__ void st_sc1(float* p, float a) { __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float2 ld_sc1x2(const float* p) {      // (p 8-byte aligned)
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}
__device__ __forceinline__ void dw_tile_32rows(const float* sY, const float (&b)[16], int K, int o0, int k0, float* __restrict__ out, int lane) {
    const int hi = lane >> 5, lo = lane & 31;
    const int o = o0 + lo, k = k0 + lo;
    const bool k_ok = k < K;
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = sY[(2 * j + hi) * kLdsStride + o];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        bsum += a[j];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
    }
    if (k_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(size_t)(o0 + acc_row(r, hi)) * (K + 1) + k] = acc[r];
    }
    if (k0 == 0) {
        bsum += __shfl_xor(bsum, 32, CIRS_WAVE);
        if (hi == 0) out[(size_t)o * (K + 1) + K] = bsum;
    }
}

// ---- trunk backward, one workgroup (8 waves) per 32 minibatch rows ------------------------------------------------
//   d a2 = (sum_chunks d h2 partial + dvalue * wc) * relu'(h2)      512 threads, one float4 each, chunk order fixed
//   entropy per row = (lse - E_p[z]) + sum_chunks clamp correction
//   d a1 = (d a2 * W2) * relu'(h1)                                  waves 0,1: one 32 x 32 MFMA tile each, A from LDS
//   d obs = d a1 * W1  -> scattered to the [T+1,B,S] tracker-gradient tensor at dst_row (wave 2, only when requested)
// d a2 / d a1 are also written to global memory for the weight-gradient GEMMs.
// ---- chunk-slab sums of d h2 and of the entropy partials, on EVERY CU ------------------------------------------------------------
// head_bwd_fused_kernel leaves one [n_pad, 64] slab of d h2 per item chunk (31 at C3: 8 MB).  Summed inside the trunk-backward
// kernel that was 254 KB per workgroup through 32 CUs -- and one CU pulls only ~30-60 GB/s from far memory: 6-8 us of that kernel.
// Here one wavefront per 64 float4: 256 workgroups, 32 KB each, all slabs of an element in flight at once, added in chunk order
// (the order the trunk-backward kernel used); the sum replaces slab 0.  The last n_pad / 64 workgroups do the same for the
// entropy partials (one float per row and chunk) and finish ent_row.
// tp_dh2 / tp_ent (item-sharded head): the sums go to the exchange buffer instead (they are PARTIAL over this rank's items: the caller
// all-reduces them, tp_post_kernel finishes ent_row).
__global__ __launch_bounds__(64) void dh2_sum_kernel(int mb, int n_pad, int n_chunks, MbView v, float* __restrict__ tp_dh2 = nullptr,
                                                     float* __restrict__ tp_ent = nullptr) {
    const int n_f4_wgs = n_pad * (kH / 4) / 64;
    if ((int)blockIdx.x < n_f4_wgs) {
        const size_t q4 = (size_t)blockIdx.x * 64 + threadIdx.x;
        float* p = v.dh2p + q4 * 4;
        const size_t cstride = (size_t)n_pad * kH;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < n_chunks; c0 += 32) {
            f32x4 t32[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) t32[u] = (c0 + u < n_chunks) ? *reinterpret_cast<const f32x4*>(p + (size_t)(c0 + u) * cstride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += t32[u];
        }
        *reinterpret_cast<f32x4*>(tp_dh2 ? tp_dh2 + q4 * 4 : p) = acc;
        return;
    }
    const int r = ((int)blockIdx.x - n_f4_wgs) * 64 + threadIdx.x;
    if (r >= n_pad) return;
    float e = 0.f;
    for (int c0 = 0; c0 < n_chunks; c0 += 32) {
        float t32[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) t32[u] = (c0 + u < n_chunks) ? v.entp[(size_t)(c0 + u) * n_pad + r] : 0.f;
#pragma unroll
        for (int u = 0; u < 32; ++u) e += t32[u];
    }
    if (tp_ent) tp_ent[r] = r < mb ? e : 0.f;
    else v.ent_row[r] = r < mb ? v.h_ent[r] + e : 0.f;  // (lse - E_p[z]) + clamp correction
}

// ---- item-sharded head (tensor-parallel learner; BASELINE configs[4], SURVEY 8(e) last paragraph) ----------------------------
// head_tp_fold_kernel: this rank's chunk partials of the forward statistics -> ONE (m, s, t) triple per minibatch row, plus the
// action's logit when this shard owns the action's item (NaN otherwise): out4 [4][n_pad], the 16-byte-per-row message of the
// all-gather.  One wavefront per row; the logit is the fma chain of head_stats_merge_kernel.
__global__ __launch_bounds__(256) void head_tp_fold_kernel(cirs_ppo_batch b, const int32_t* __restrict__ idx, int mb, int n_pad, int n_chunks,
                                                           ActorPartialView pv, const float* __restrict__ wa_shard,
                                                           const float* __restrict__ ba_shard, int item_base, int n_items_shard,
                                                           const float* __restrict__ h2, float* __restrict__ out4) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_pad) return;
    if (j >= mb) {
        if (lane == 0) { out4[j] = -INFINITY; out4[n_pad + j] = 0.f; out4[2 * n_pad + j] = 0.f; out4[3 * n_pad + j] = __builtin_nanf(""); }
        return;
    }
    __shared__ __attribute__((aligned(16))) float sRow[4][2][kH];
    const int a = b.act[idx[j]] - item_base;
    const bool mine = a >= 0 && a < n_items_shard;
    const float hl = h2[(size_t)j * kH + lane], wl = mine ? wa_shard[(size_t)a * kH + lane] : 0.f;
    float m = -INFINITY, s = 0.f, t = 0.f;
    for (int c = lane; c < n_chunks; c += CIRS_WAVE) {
        const size_t o = (size_t)c * n_pad + j;
        const float om = pv.m[o], os = pv.s[o], ot = pv.score[o];
        const float mn = fmaxf(m, om);
        if (mn > -INFINITY) {
            const float fa = __expf(m - mn), fb = __expf(om - mn);
            s = s * fa + os * fb; t = t * fa + ot * fb; m = mn;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, CIRS_WAVE), os = __shfl_xor(s, off, CIRS_WAVE), ot = __shfl_xor(t, off, CIRS_WAVE);
        const float mn = fmaxf(m, om);
        if (mn > -INFINITY) {
            const float fa = __expf(m - mn), fb = __expf(om - mn);
            s = s * fa + os * fb; t = t * fa + ot * fb; m = mn;
        }
    }
    sRow[threadIdx.x >> 6][0][lane] = hl;
    sRow[threadIdx.x >> 6][1][lane] = wl;
    __builtin_amdgcn_wave_barrier();
    if (lane != 0) return;
    float z = __builtin_nanf("");
    if (mine) {
        const float* hr = sRow[threadIdx.x >> 6][0];
        const float* wr = sRow[threadIdx.x >> 6][1];
        z = ba_shard[a];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            z = __builtin_fmaf(hr[kk], wr[kk], z);
            z = __builtin_fmaf(hr[32 + kk], wr[32 + kk], z);
        }
    }
    out4[j] = m; out4[n_pad + j] = s; out4[2 * n_pad + j] = t; out4[3 * n_pad + j] = z;
}

__global__ __launch_bounds__(512) void trunk_bwd_kernel(int mb, int n_pad, int n_chunks, int S, const float* __restrict__ w1,
                                                        const float* __restrict__ w2, const float* __restrict__ wc, MbView v,
                                                        float* __restrict__ dobs_accum, float* __restrict__ g, long wa_beg, long wa_len,
                                                        long slab_stride, int n_slabs, DwJobs jobs, float* __restrict__ dwp) {
    __shared__ __attribute__((aligned(16))) float sA[kTileM * kLdsStride];
    __shared__ __attribute__((aligned(16))) float sD[kTileM * kLdsStride];
    __shared__ __attribute__((aligned(16))) float sW2[kH * kLdsStride];        // W2 [k][n]
    __shared__ __attribute__((aligned(16))) float sH1[kTileM * kLdsStride];    // h1 rows of the tile
    __shared__ __attribute__((aligned(16))) float sH2[kTileM * kLdsStride];    // h2 rows of the tile
    __shared__ float sW1[kH * 32];                                             // W1 [k][S], S <= 32
    __shared__ float sObs[kTileM * 32];                                        // obs rows of the tile [32][S]
    __shared__ float sDv[kTileM];
    CIRS_TSTAMP(0, 16);
    CIRS_TSTAMP(n_pad / kTileM, 24);
    if ((int)blockIdx.x >= n_pad / kTileM) {   // extra workgroups (single-rank path): slab sums of the wa|ba gradient
        const int b = (int)blockIdx.x - n_pad / kTileM;
        wa_slab_sum_block(g, wa_beg, wa_len, v.dwap, slab_stride, n_slabs, b, v.normp + kNormBlocks + b, sA);
        CIRS_TSTAMP(n_pad / kTileM, 25);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int row0 = blockIdx.x * kTileM;
    // Everything the later stages read besides stage 1's own result -- W2, W1, the h1 / h2 / obs rows of the tile, dvalue -- depends on
    // nothing computed here.  A wavefront's memory instruction costs the CU's address unit ~16 cycles whatever its width, and eight
    // wavefronts share that unit: per-lane operand loads (32 + 16 dwords per lane, 64 more for the critic row) took longer to ISSUE
    // than the 31 slab loads below.  So: whole tiles with coalesced 16-byte loads (one or two per thread) into LDS, operands from there.
    const int f_rl = tid >> 4, f_c4 = (tid & 15) * 4;          // this thread's float4 of a 32 x 64 tile
    f32x4 w2a, w2b;                                                                        // W2 [64][64]: float4 tid and tid + 512
    if ((reinterpret_cast<uintptr_t>(w2) & 15) == 0) {
        w2a = *reinterpret_cast<const f32x4*>(w2 + (size_t)tid * 4);
        w2b = *reinterpret_cast<const f32x4*>(w2 + (size_t)(tid + 512) * 4);
    } else {   // a caller's weight pointer need not be 16-byte aligned
        const float* pa = w2 + (size_t)tid * 4; const float* pb = w2 + (size_t)(tid + 512) * 4;
        w2a = f32x4{pa[0], pa[1], pa[2], pa[3]}; w2b = f32x4{pb[0], pb[1], pb[2], pb[3]};
    }
    const f32x4 h1t = *reinterpret_cast<const f32x4*>(v.h1 + (size_t)(row0 + f_rl) * kH + f_c4);
    float w1t[4], obt[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) w1t[q] = (dobs_accum && tid + 512 * q < kH * S) ? w1[tid + 512 * q] : 0.f;   // W1 [64][S], S <= 32
#pragma unroll
    for (int q = 0; q < 2; ++q) obt[q] = (dwp && tid + 512 * q < kTileM * S) ? v.obs[(size_t)row0 * S + tid + 512 * q] : 0.f;
    {
        const int f = tid;  // one float4 column group of the 32 x 64 tile per thread
        const int rl = f >> 4, c4 = (f & 15) * 4;
        const int r = row0 + rl;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = v.dh2p + (size_t)r * kH + c4;
        {   // slab 0 holds the sum over the item chunks (dh2_sum_kernel)
            const f32x4 t = *reinterpret_cast<const f32x4*>(src);
            acc.x = t.x; acc.y = t.y; acc.z = t.z; acc.w = t.w;
        }
        const float dv = v.dvalue[r];
        const float4 wc4 = *reinterpret_cast<const float4*>(wc + c4);
        const float4 h4 = *reinterpret_cast<const float4*>(v.h2 + (size_t)r * kH + c4);
        const bool ok = r < mb;
        float4 t;
        t.x = (ok && h4.x > 0.f) ? __builtin_fmaf(dv, wc4.x, acc.x) : 0.f;
        t.y = (ok && h4.y > 0.f) ? __builtin_fmaf(dv, wc4.y, acc.y) : 0.f;
        t.z = (ok && h4.z > 0.f) ? __builtin_fmaf(dv, wc4.z, acc.z) : 0.f;
        t.w = (ok && h4.w > 0.f) ? __builtin_fmaf(dv, wc4.w, acc.w) : 0.f;
        *reinterpret_cast<float4*>(v.da2 + (size_t)r * kH + c4) = t;
        *reinterpret_cast<float4*>(&sA[rl * kLdsStride + c4]) = t;
        *reinterpret_cast<float4*>(&sH2[rl * kLdsStride + c4]) = h4;
        if ((f & 15) == 0) sDv[rl] = ok ? dv : 0.f;
    }
    *reinterpret_cast<f32x4*>(&sW2[(tid >> 4) * kLdsStride + f_c4]) = w2a;
    *reinterpret_cast<f32x4*>(&sW2[((tid + 512) >> 4) * kLdsStride + f_c4]) = w2b;
    *reinterpret_cast<f32x4*>(&sH1[f_rl * kLdsStride + f_c4]) = h1t;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (tid + 512 * q < kH * S) sW1[tid + 512 * q] = w1t[q];
#pragma unroll
    for (int q = 0; q < 2; ++q) if (tid + 512 * q < kTileM * S) sObs[tid + 512 * q] = obt[q];
    CIRS_TSTAMP(0, 17);
    __syncthreads();
    CIRS_TSTAMP(0, 18);
    if (wv < 2) {  // d a1 tile: columns wv*32 .. +32
        const int n = wv * 32 + lo;
        float arow[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(&sA[lo * kLdsStride + hi * 32 + 4 * q]);
            arow[4 * q] = t4.x; arow[4 * q + 1] = t4.y; arow[4 * q + 2] = t4.z; arow[4 * q + 3] = t4.w;
        }
        float bcol[32], h1g[16];    // W2 column n, the relu gate's h1 values: from the LDS tiles
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) bcol[kk] = sW2[(hi * 32 + kk) * kLdsStride + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) h1g[r] = sH1[((r & 3) + 8 * (r >> 2) + 4 * hi) * kLdsStride + n];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[kk], bcol[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const size_t o = (size_t)(row0 + rl) * kH + n;
            const float d = h1g[r] > 0.f ? acc[r] : 0.f;
            v.da1[o] = d;
            sD[rl * kLdsStride + n] = d;
        }
    } else if (dwp && wv < 6) {  // d W2 | d b2 row slab of this workgroup: four 32 x 32 tiles, waves 2..5
        const int t = wv - 2;
        float xrow[16];
        dw_tile_x_lds(sH1, kLdsStride, kH, row0, mb, (t & 1) * 32, lane, xrow);
        dw_tile_32rows(sA, xrow, kH, (t >> 1) * 32, (t & 1) * 32, dwp + jobs.j[1].part_off + (size_t)blockIdx.x * (kH * (kH + 1)), lane);
    } else if (dwp && wv == 6) {  // d wc | d bc row slab: lane = column of h2
        float acc = 0.f, bs = 0.f;
        float dv32[kTileM], h32[kTileM];    // from the LDS tiles (dvalue is 0 there for rows beyond the minibatch)
#pragma unroll
        for (int r = 0; r < kTileM; ++r) { dv32[r] = sDv[r]; h32[r] = sH2[r * kLdsStride + lane]; }
#pragma unroll
        for (int r = 0; r < kTileM; ++r) {
            if (row0 + r < mb) {
                acc = __builtin_fmaf(dv32[r], h32[r], acc);
                bs += dv32[r];
            }
        }
        float* out = dwp + jobs.j[0].part_off + (size_t)blockIdx.x * (kH + 1);
        out[lane] = acc;
        if (lane == 0) out[kH] = bs;
    }
    CIRS_TSTAMP(0, 19);
    __syncthreads();
    CIRS_TSTAMP(0, 20);
    if (dwp && wv < 2) {  // d W1 | d b1 row slab: two 32 x 32 tiles (S <= 32 columns), waves 0, 1
        float xrow[16];
        dw_tile_x_lds(sObs, S, S, row0, mb, 0, lane, xrow);
        dw_tile_32rows(sD, xrow, S, wv * 32, 0, dwp + jobs.j[2].part_off + (size_t)blockIdx.x * (kH * (S + 1)), lane);
    }
    CIRS_TSTAMP(0, 21);
    if (!dobs_accum) return;
    if (wv == 2) {
        float arow[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(&sD[lo * kLdsStride + hi * 32 + 4 * q]);
            arow[4 * q] = t4.x; arow[4 * q + 1] = t4.y; arow[4 * q + 2] = t4.z; arow[4 * q + 3] = t4.w;
        }
        float bcol[32];   // W1 column lo
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) bcol[kk] = lo < S ? sW1[(hi * 32 + kk) * S + lo] : 0.f;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[kk], bcol[kk], acc, 0, 0, 0);
        if (lo < S) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < mb) dobs_accum[(size_t)v.dst_row[row] * S + lo] = acc[r];
            }
        }
    }
}

// ---- trunk backward of the single-rank step (round 5): chunk-slab sums of d h2 + trunk backward + weight-gradient sums + squared norm, ONE launch ----
// Until round 5 this was three launches (dh2_sum_kernel, trunk_bwd_kernel, sumsq_partial_kernel: 4.9 + 8.5 + 5.2 us, each within 1-3 us of the
// floor of a dependent launch).  What forced the first boundary was bytes per workgroup: a 32-row MFMA tile needs the 31 chunk slabs of its rows,
// 254 KB through ONE CU.  Here a row workgroup owns 8 rows (64 KB of slabs) and does its small products on the vector ALUs in the matrix cores'
// order (k = kk, 32 + kk: same bits as trunk_bwd_kernel); 128 workgroups instead of 32 pull the same bytes.  Roles by workgroup index:
//   R [0, n_r)            8 rows each: d a2 (chunk slabs summed in chunk order + the critic's term), entropy per row, d a1, d obs scatter, and the
//                         rows' share of every trunk / critic weight gradient as ONE slab in the flat gradient's own order [w1|b1|w2|b2|wc|bc],
//                         written through (sc1) before the workgroup counts its arrival
//   W next n_w            slab sums of the wa|ba gradient + their squared-norm partials (depend on the head backward kernel only)
//   F last n_f            wait for the R arrivals, then 128 outputs each: four threads per output sum a quarter of the slabs in slab order and
//                         meet in quarter order; flat gradient + squared-norm partial.  The hand-off runs beside the W workgroups.
// R has the lowest indices: everything it needs is dispatched before anything that waits for it.
constexpr int kRR = 8;                       // rows of an R workgroup
constexpr int kFOut = 128;                   // outputs of an F workgroup
struct RowsLds {
    __attribute__((aligned(16))) float w2[kH * kLdsStride];      // W2 [k][n]
    __attribute__((aligned(16))) float a2[kRR * kH];             // d a2
    __attribute__((aligned(16))) float a1[kRR * kH];             // d a1
    __attribute__((aligned(16))) float h1[kRR * kH];
    __attribute__((aligned(16))) float h2[kRR * kH];
    __attribute__((aligned(16))) float obs[kRR * 32];
    float w1[kH * 32];                                           // W1 [k][S]
    float dv[kRR];
    __attribute__((aligned(16))) float ps[3][kRR * kH];          // chunk-slab sums of quarters 1..3 of the chunks (quarter 0 stays in its threads' registers)
};
__device__ __forceinline__ void st_sc1x2(float* p, float a, float b) {      // 8-byte write-through store (p 8-byte aligned)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(512) void trunk_rows_kernel(int mb, int n_pad, int n_chunks, int S, const float* __restrict__ w1, const float* __restrict__ w2,
                                                         const float* __restrict__ wc, MbView v, float* __restrict__ dobs_accum, float* __restrict__ g,
                                                         long wa_beg, long wa_len, long slab_stride, int n_slabs, int n_r, int rslab /* floats per R slab */,
                                                         int w_delay /* W workgroups start this many x 1024 cycles late: the R workgroups' requests go first */,
                                                         float* __restrict__ tail_out /* data-parallel phase 1: {clip, vf, ent, 0} partials of this rank */, int mb_norm,
                                                         int n_w /* W workgroups */) {
    __shared__ RowsLds L;
    __shared__ float sRed[512];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int n_tr = kH * S + kH + kH * kH + kH, n_dw = n_tr + kH + 1;      // [w1 | b1 | w2 | b2] + [wc | bc]
    const int n_f_wgs = (n_dw + kFOut - 1) / kFOut;
    if (b >= n_r + n_w + n_f_wgs) {
        // ---- loss partials of this rank (data-parallel step; the single-rank step forms them in its optimiser launch): needs the rows' entropies
        flags_wait(v.sync, n_r);
        __shared__ float sL[3][512];
        float e = 0.f, c = 0.f, f = 0.f;
        for (int r = tid; r < mb; r += 512) { e += ld_sc1(v.ent_row + r); c += v.clip_row[r]; f += v.vf_row[r]; }
        sL[0][tid] = c; sL[1][tid] = f; sL[2][tid] = e;
        __syncthreads();
        for (int st = 256; st > 0; st >>= 1) {
            if (tid < st) { sL[0][tid] += sL[0][tid + st]; sL[1][tid] += sL[1][tid + st]; sL[2][tid] += sL[2][tid + st]; }
            __syncthreads();
        }
        if (tid == 0) {
            const float inv = 1.0f / (float)mb_norm;
            tail_out[0] = sL[0][0] * inv; tail_out[1] = sL[1][0] * inv; tail_out[2] = sL[2][0] * inv; tail_out[3] = 0.f;
        }
        return;
    }
    if (b >= n_r + n_w) {
        // ---- F ----------------------------------------------------------------------------------------------------------------
        const int f = b - n_r - n_w;
        CIRS_PSTAMP(f == 0, 30);
        flags_wait(v.sync, n_r);
        CIRS_PSTAMP(f == 0, 31);
        const int o = tid & (kFOut - 1), q = tid >> 7;          // output of this workgroup, slab quarter
        const int e = f * kFOut + o;
        const int per = (n_r + 3) >> 2, s_lo = q * per, s_hi = min(n_r, s_lo + per);
        float x = 0.f;
        if (e < n_dw) {
            const float* src = v.dwp + e;
            for (int c0 = s_lo; c0 < s_hi; c0 += 32) {          // all loads of a batch in flight, added in slab order
                float t32[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) t32[u] = (c0 + u < s_hi) ? ld_sc1(src + (size_t)(c0 + u) * rslab) : 0.f;
#pragma unroll
                for (int u = 0; u < 32; ++u) x += t32[u];
            }
        }
        sRed[tid] = x;
        __syncthreads();
        CIRS_PSTAMP(f == 0, 32);
        float sq = 0.f;
        if (tid < kFOut && e < n_dw) {
            const float tot = ((sRed[tid] + sRed[kFOut + tid]) + sRed[2 * kFOut + tid]) + sRed[3 * kFOut + tid];
            g[e < n_tr ? e : wa_beg + wa_len + (e - n_tr)] = tot;
            sq = (e < n_tr ? 2.0f : 1.0f) * tot * tot;          // trunk parameters appear twice in the reference's list
        }
        __syncthreads();
        sRed[tid] = sq;
        __syncthreads();
        for (int st = 64; st > 0; st >>= 1) {
            if (tid < st) sRed[tid] += sRed[tid + st];
            __syncthreads();
        }
        if (tid == 0) v.normp[f] = sRed[0];
        const int n_f = (n_dw + kFOut - 1) / kFOut;
        if (f == 0) {      // the slots no workgroup owns
            if (tid >= n_f && tid < kNormBlocks) v.normp[tid] = 0.f;
            for (int q = n_w + tid; q < kWaSlotsMax; q += 512) v.normp[kNormBlocks + q] = 0.f;
        }
        CIRS_PSTAMP(f == 0, 33);
        return;
    }
    if (b >= n_r) {
        CIRS_PSTAMP(b == n_r, 26);
        for (int q = 0; q < w_delay; ++q) __builtin_amdgcn_s_sleep(16);
        wa_slab_sum_block1(g, wa_beg, wa_len, v.dwap, slab_stride, n_slabs, b - n_r, n_w, v.normp + kNormBlocks + (b - n_r), sRed);
        CIRS_PSTAMP(b == n_r, 27);
        return;
    }
    // ---- R: thread (row r = tid / 64, column n = tid % 64) ----------------------------------------------------------------------
    CIRS_PSTAMP(b == 0, 20);
    const int r = tid >> 6, n = tid & 63;
    // requests: the chunk slabs of the rows first -- as float4 (thread = 4 columns of a row and a quarter of the chunks: a quarter of the load
    // instructions of one column per thread, and the CU's address unit takes ~16 cycles per instruction whatever its width) --, then the operands
    // every later stage reads from LDS
    const int r4 = (tid >> 4) & (kRR - 1), c4 = (tid & 15) * 4, row4 = b * kRR + r4;      // row4 < n_pad (n_pad is a multiple of 32)
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f}, h1q = acc4, h2q = acc4, wcq = acc4;
    float dvr = 0.f;
    // (round 6, second half) all four thread quarters take a quarter of the chunks each -- 8 x 16 bytes in ONE batch per thread at 31 chunks -- and the quarters'
    // sums meet in LDS in quarter order: with the first quarter alone walking the 31 slabs it was two dependent round trips of 16 (measured: the second batch
    // was 0.95 us of the 70.7 us step).  Fixed order (chunks ascending inside a quarter, quarters ascending): deterministic, same on every rank.
    {
        const int qd = tid >> 7, per = (n_chunks + 3) >> 2, c_lo = qd * per, c_hi = min(n_chunks, c_lo + per);
        const float* src = v.dh2p + (size_t)row4 * kH + c4;
        const size_t cstride = (size_t)n_pad * kH;
        for (int c0 = c_lo; c0 < c_hi; c0 += 8) {
            f32x4 t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t8[u] = (c0 + u < c_hi) ? *reinterpret_cast<const f32x4*>(src + (size_t)(c0 + u) * cstride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) acc4 += t8[u];
        }
        if (qd > 0) *reinterpret_cast<f32x4*>(&L.ps[qd - 1][r4 * kH + c4]) = acc4;
    }
    if (tid < kRR * 16) {
        h1q = *reinterpret_cast<const f32x4*>(v.h1 + (size_t)row4 * kH + c4);
        h2q = *reinterpret_cast<const f32x4*>(v.h2 + (size_t)row4 * kH + c4);
        wcq = *reinterpret_cast<const f32x4*>(wc + c4);
        dvr = v.dvalue[row4];
    }
    const f32x4 w2a = *reinterpret_cast<const f32x4*>(w2 + (size_t)tid * 4), w2b = *reinterpret_cast<const f32x4*>(w2 + (size_t)(tid + 512) * 4);
    float w1t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w1t[q] = tid + 512 * q < kH * S ? w1[tid + 512 * q] : 0.f;   // W1 [64][S], S <= 32
    const float obv = tid < kRR * S ? v.obs[(size_t)b * kRR * S + tid] : 0.f;
    float ent = 0.f, hent = 0.f;
    if (tid < kRR) {            // entropy per row = (lse - E_p[z]) + the chunks' clamp corrections, chunk order
        const int rr = b * kRR + tid;
        for (int c0 = 0; c0 < n_chunks; c0 += 32) {
            float t32[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) t32[u] = (c0 + u < n_chunks) ? v.entp[(size_t)(c0 + u) * n_pad + rr] : 0.f;
#pragma unroll
            for (int u = 0; u < 32; ++u) ent += t32[u];
        }
        hent = v.h_ent[rr];
    }
    __syncthreads();
    if (tid < kRR * 16) {
        acc4 = ((acc4 + *reinterpret_cast<const f32x4*>(&L.ps[0][r4 * kH + c4])) + *reinterpret_cast<const f32x4*>(&L.ps[1][r4 * kH + c4])) +
               *reinterpret_cast<const f32x4*>(&L.ps[2][r4 * kH + c4]);
        const bool ok = row4 < mb;
        f32x4 d4;
#pragma unroll
        for (int u = 0; u < 4; ++u) d4[u] = (ok && h2q[u] > 0.f) ? __builtin_fmaf(dvr, wcq[u], acc4[u]) : 0.f;
        *reinterpret_cast<f32x4*>(&L.a2[r4 * kH + c4]) = d4;
        *reinterpret_cast<f32x4*>(&L.h1[r4 * kH + c4]) = h1q;
        *reinterpret_cast<f32x4*>(&L.h2[r4 * kH + c4]) = h2q;
        if ((tid & 15) == 0) L.dv[r4] = ok ? dvr : 0.f;
    }
    *reinterpret_cast<f32x4*>(&L.w2[(tid >> 4) * kLdsStride + (tid & 15) * 4]) = w2a;
    *reinterpret_cast<f32x4*>(&L.w2[((tid + 512) >> 4) * kLdsStride + (tid & 15) * 4]) = w2b;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (tid + 512 * q < kH * S) L.w1[tid + 512 * q] = w1t[q];
    if (tid < kRR * S) L.obs[(tid / S) * 32 + tid % S] = obv;
    if (tid < kRR) st_sc1(v.ent_row + b * kRR + tid, (b * kRR + tid) < mb ? hent + ent : 0.f);      // (written through: the loss workgroup of a data-parallel step reads it inside this launch)
    __syncthreads();
    CIRS_PSTAMP(b == 0, 21);
    // d a1 = (d a2 W2) relu'(h1): the matrix cores' order k = kk, 32 + kk
    {
        float t = 0.f;
        const float* ar = L.a2 + r * kH;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            t = __builtin_fmaf(ar[kk], L.w2[kk * kLdsStride + n], t);
            t = __builtin_fmaf(ar[32 + kk], L.w2[(32 + kk) * kLdsStride + n], t);
        }
        L.a1[r * kH + n] = L.h1[r * kH + n] > 0.f ? t : 0.f;
    }
    __syncthreads();
    CIRS_PSTAMP(b == 0, 22);
    float* slab = v.dwp + (size_t)b * rslab;
    const int o_b1 = kH * S, o_w2 = o_b1 + kH, o_b2 = o_w2 + kH * kH;
    {   // d W2 [o][k]: thread (o = tid / 8, k = 8 (tid % 8) .. + 8), rows in order
        const int o = tid >> 3, kb = (tid & 7) * 8;
        float w[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < kRR; ++rr) {
            const float a = L.a2[rr * kH + o];
            const f32x4 ha = *reinterpret_cast<const f32x4*>(&L.h1[rr * kH + kb]), hb = *reinterpret_cast<const f32x4*>(&L.h1[rr * kH + kb + 4]);
#pragma unroll
            for (int u = 0; u < 4; ++u) { w[u] = __builtin_fmaf(a, ha[u], w[u]); w[4 + u] = __builtin_fmaf(a, hb[u], w[4 + u]); }
        }
        float* d = slab + o_w2 + o * kH + kb;       // (o_w2 = 64 (S + 1): 8-byte aligned)
#pragma unroll
        for (int u = 0; u < 8; u += 2) st_sc1x2(d + u, w[u], w[u + 1]);
    }
    {   // d W1 [o][s]: thread (o = tid / 8, s = 4 (tid % 8) .. + 4)
        const int o = tid >> 3, sb = (tid & 7) * 4;
        float w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < kRR; ++rr) {
            const float a = L.a1[rr * kH + o];
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(&L.obs[rr * 32 + sb]);
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = __builtin_fmaf(a, x4[u], w[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (sb + u < S) st_sc1(slab + o * S + sb + u, w[u]);
    }
    if (tid < kH) {             // bias columns and the critic's row
        float s2 = 0.f, s1 = 0.f, wcg = 0.f;
#pragma unroll
        for (int rr = 0; rr < kRR; ++rr) { s2 += L.a2[rr * kH + tid]; s1 += L.a1[rr * kH + tid]; wcg = __builtin_fmaf(L.dv[rr], L.h2[rr * kH + tid], wcg); }
        st_sc1(slab + o_b2 + tid, s2); st_sc1(slab + o_b1 + tid, s1); st_sc1(slab + n_tr + tid, wcg);
    } else if (tid == kH) {
        float s0 = 0.f;
#pragma unroll
        for (int rr = 0; rr < kRR; ++rr) s0 += L.dv[rr];
        st_sc1(slab + n_tr + kH, s0);
    }
    CIRS_PSTAMP(b == 0, 23);
    if (dobs_accum && tid < kRR * 32) {         // d obs = d a1 W1, scattered to the [T+1, B, S] tracker-gradient tensor; thread (row tid / 32, s = tid % 32)
        const int rr = tid >> 5, sc = tid & 31, rw = b * kRR + rr;
        if (sc < S && rw < mb) {
            float t = 0.f;
            const float* ar = L.a1 + rr * kH;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                t = __builtin_fmaf(ar[kk], L.w1[kk * S + sc], t);
                t = __builtin_fmaf(ar[32 + kk], L.w1[(32 + kk) * S + sc], t);
            }
            dobs_accum[(size_t)v.dst_row[rw] * S + sc] = t;
        }
    }
    // every slab store of this workgroup has left before its arrival is counted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CIRS_PSTAMP(b == 0, 24);
    flag_arrive(v.sync, b);
    CIRS_PSTAMP(b == 0, 25);
}

// ---- item-sharded head: pieces around the all-reduce of the d h2 partials ------------------------------------------------------
// wa_slab_sum_kernel: the slab sums of THIS shard's wa|ba gradient (complete: the shard saw every row of the minibatch) into the flat
// gradient + its kWaSumBlocks sum-of-squares partials into the rank's slots of the exchange buffer (the other ranks' slots are zero:
// the sum over ranks hands every rank all partials).
__global__ __launch_bounds__(512) void wa_slab_sum_kernel(float* __restrict__ g, long wa_beg, long wa_len, const float* __restrict__ dwap,
                                                          long slab_stride, int n_slabs, float* __restrict__ slots) {
    __shared__ float sh[512];
    wa_slab_sum_block(g, wa_beg, wa_len, dwap, slab_stride, n_slabs, (int)blockIdx.x, slots + blockIdx.x, sh);
}
// tp_post_kernel (after the all-reduce): entropy per row = (lse - E_p[z]) + the clamp corrections of all shards; the squared-norm
// partials of the wa|ba gradient of ALL shards folded in (rank, block) order into the slots adam2_kernel reads.
__global__ __launch_bounds__(256) void tp_post_kernel(int mb, int n_pad, int world, const float* __restrict__ red_ent,
                                                      const float* __restrict__ red_slots, MbView v) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_pad) v.ent_row[r] = r < mb ? v.h_ent[r] + red_ent[r] : 0.f;
    if (blockIdx.x == 0 && threadIdx.x < kWaSumBlocks) {
        float acc = 0.f;
        for (int q = 0; q < world; ++q) acc += red_slots[(size_t)q * kWaSumBlocks + threadIdx.x];
        v.normp[kNormBlocks + threadIdx.x] = acc;
    }
}

// clip_grad_norm_: total norm over [trunk, wa, ba, trunk, wc, bc] -> coef = min(max_norm/(norm+1e-6), 1).
// stage 1: kNormBlocks workgroups, fixed-order tree -> partial sums of squares.  Single-rank path (dw_partial set): the wa|ba
// squares come from the slab-sum workgroups of the trunk-backward launch (slots kNormBlocks..), here the trunk / critic
// gradients are summed from their row slabs into the flat gradient on the way; data-parallel phase 2 (dw_partial null):
// every element of the all-reduced gradient is read from the flat buffer.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(float* __restrict__ g, long n_trunk, long n_total, long wa_beg, long wa_len,
                                                            int wa_fused,
                                                            DwJobs jobs, int n_dw_slabs, const float* __restrict__ dw_partial, int S,
                                                            float* __restrict__ partial) {
    __shared__ float sh[256];
    const int tid = threadIdx.x;
    const long per = (n_total + kNormBlocks - 1) / kNormBlocks;
    const long lo = blockIdx.x * per, hi = min(n_total, lo + per);
    float acc = 0.f;
    // slots of the wa|ba slab-sum workgroups (trunk-backward launch): theirs when wa_fused, zero otherwise
    // (adam_next_kernel folds all kWaSlotsMax slots; blocks 0..255 own slots b, b + 256, b + 512 of them)
    if (tid < 3) {
        const int q = blockIdx.x + 256 * tid;
        if (!wa_fused || q >= kWaSumBlocks) partial[kNormBlocks + q] = 0.f;
    }
    // everything else: already summed in g unless dw_partial holds it (handled below)
    for (long i = lo + tid; i < hi && !dw_partial; i += 256) {
        const float x = g[i];
        acc += (i < n_trunk ? 2.0f : 1.0f) * x * x;  // trunk parameters appear twice in the reference's list
    }
    if (dw_partial) {
        // trunk / critic gradients still live in row-slab partials (trunk_bwd_kernel): job 2 = W1|b1, 1 = W2|b2, 0 = wc|bc.
        // Element e of [trunk | wc | bc] belongs to thread e of the grid (fixed assignment -> fixed summation order).
        const long e = blockIdx.x * 256L + tid;
        const long n_dw = n_trunk + kH + 1;
        if (e < n_dw) {
            const long i = e < n_trunk ? e : wa_beg + wa_len + (e - n_trunk);
            int ji, q;
            if (e < (long)kH * S) { ji = 2; q = (int)(e / S) * (S + 1) + (int)(e % S); }
            else if (e < (long)kH * S + kH) { ji = 2; q = (int)(e - (long)kH * S) * (S + 1) + S; }
            else if (e < (long)kH * S + kH + (long)kH * kH) { const int r = (int)(e - ((long)kH * S + kH)); ji = 1; q = (r / kH) * (kH + 1) + (r % kH); }
            else if (e < n_trunk) { ji = 1; q = (int)(e - ((long)kH * S + kH + (long)kH * kH)) * (kH + 1) + kH; }
            else { ji = 0; q = (int)(e - n_trunk); }  // wc[0..63] then bc: row 0 of a [1, 64 + 1] problem
            // (a per-thread index into the by-value job table would send the whole table through scratch memory: select the two fields)
            const int n_out = ji == 2 ? jobs.j[2].O * (jobs.j[2].K + 1) : ji == 1 ? jobs.j[1].O * (jobs.j[1].K + 1) : jobs.j[0].O * (jobs.j[0].K + 1);
            const int p_off = ji == 2 ? jobs.j[2].part_off : ji == 1 ? jobs.j[1].part_off : jobs.j[0].part_off;
            const float x = dw_multi_fetch(n_out, p_off, n_dw_slabs, dw_partial, q);
            g[i] = x;
            acc += (e < n_trunk ? 2.0f : 1.0f) * x * x;
        }
    }
    sh[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) partial[blockIdx.x] = sh[0];
}
// stage 2: one workgroup: norm, clip coefficient, entropy mean, total loss
// loss partials of this rank: {clip, vf, ent} already divided by the global minibatch size -> grads tail
__device__ __forceinline__ void loss_partials_block(int mb, int mb_norm, const MbView& v, float* __restrict__ tail, float* sh3) {
    const int tid = threadIdx.x;  // blockDim.x == 256, sh3 = float[3][256]
    float e = 0.f, c = 0.f, f = 0.f;
    // (round 6: the first 2048 rows' terms requested in one batch -- a loop with a run-time trip count is one round trip per iteration, and this workgroup's chain
    // ended adam_next_kernel 0.9 us after the T workgroups'; same order of additions)
    {
        float e8[8], c8[8], f8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = tid + 256 * q, rc = r < mb ? r : 0;
            e8[q] = v.ent_row[rc]; c8[q] = v.clip_row[rc]; f8[q] = v.vf_row[rc];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (tid + 256 * q < mb) { e += e8[q]; c += c8[q]; f += f8[q]; }
    }
    for (int r = tid + 2048; r < mb; r += 256) { e += v.ent_row[r]; c += v.clip_row[r]; f += v.vf_row[r]; }
    sh3[tid] = c; sh3[256 + tid] = f; sh3[512 + tid] = e;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sh3[tid] += sh3[tid + s]; sh3[256 + tid] += sh3[256 + tid + s]; sh3[512 + tid] += sh3[512 + tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const float inv = 1.0f / (float)mb_norm;
        tail[0] = sh3[0] * inv; tail[1] = sh3[256] * inv; tail[2] = sh3[512] * inv; tail[3] = 0.f;
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void loss_partials_kernel(int mb, int mb_norm, MbView v, float* __restrict__ tail) {
    __shared__ float sh3[768];
    loss_partials_block(mb, mb_norm, v, tail, sh3);
}

// torch.optim.Adam (_single_tensor_adam): lerp_, mul_/addcmul_, bias corrections from the step count
struct AdamSeg { int n_sub; int scale_pow; float step_size0, bc2s0, step_size1, bc2s1; float rbc2s0, rbc2s1; /* 1 / bc2s (adam_next_kernel) */ };

// torch's update p -= step_size * m / (sqrt(v) / sqrt(1 - beta2^t) + eps) with the hardware's 1-ulp square root and reciprocal (v_sqrt_f32,
// v_rcp_f32) instead of the correctly rounded expansions (2 divisions + 1 square root = ~45 instructions per sub-step, which made the trunk's
// 5.6 k redundant updates per T workgroup a 5 us chain): relative error 2e-7 of a step of size lr, i.e. 1e-10 of a parameter.  The
// denominator is >= eps, never denormal; a denormal v (|g| < 1e-19) is below eps^2 by 22 orders of magnitude either way.
// <kSub, kPow> = (2, 2) for the trunk (the reference's parameter list holds it twice: coefficient squared, two sub-steps), (1, 1) for the heads
template <int kSub, int kPow>
__device__ __forceinline__ void adam_elem(float& pi, float& mi, float& vi, float gi, const AdamSeg& sg, float c, float beta1, float beta2, float eps) {
#pragma unroll
    for (int q = 0; q < kPow; ++q) gi *= c;
#pragma unroll
    for (int sub = 0; sub < kSub; ++sub) {
        mi = mi + (1.0f - beta1) * (gi - mi);
        vi = vi * beta2 + (1.0f - beta2) * gi * gi;
        const float ss = sub == 0 ? sg.step_size0 : sg.step_size1;
        const float rb2 = sub == 0 ? sg.rbc2s0 : sg.rbc2s1;
        pi = pi - ss * (mi * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vi) * rb2 + eps));
    }
}
// one launch over the whole flat buffer: elements [0, n_first) use segment a (trunk), the rest segment b (heads).
// clip_grad_norm_ stage 2 rides along: EVERY workgroup sums the kNormBlocks partial sums of squares in the same fixed
// tree order (identical coefficient everywhere); workgroup 0 also forms the loss terms (mb > 0: single-rank path, the
// loss partials are reduced here; mb == 0: they already are in `tail`) and publishes norm / coefficient.
__global__ __launch_bounds__(256) void adam2_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, long n_first, AdamSeg sa, AdamSeg sb, float beta1,
                                                    float beta2, float eps, cirs_ppo_cfg cfg, const float* __restrict__ partial,
                                                    float* __restrict__ tail, MbView mv, float* __restrict__ loss_out, int mb, int mb_norm) {
    __shared__ float sh[256];
    __shared__ float sh3[768];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && mb > 0) loss_partials_block(mb, mb_norm, mv, tail, sh3);
    static_assert(kNormBlocks == 256 && kWaSumBlocks <= 256, "each thread folds one slot of each kind");
    sh[tid] = partial[tid] + (tid < kWaSumBlocks ? partial[kNormBlocks + tid] : 0.f);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const float total_norm = sqrtf(sh[0]);
    float c = 1.0f;
    if (cfg.max_grad_norm > 0.f) c = fminf(cfg.max_grad_norm / (total_norm + 1e-6f), 1.0f);
    if (blockIdx.x == 0 && tid == 0) {
        mv.red[4] = c;
        mv.red[5] = total_norm;
        const float clip = tail[0], vf = tail[1], ent = tail[2];
        loss_out[0] = clip + cfg.vf_coef * vf - cfg.ent_coef * ent;
        loss_out[1] = clip; loss_out[2] = vf; loss_out[3] = ent;
    }
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AdamSeg sg = i < n_first ? sa : sb;
    float gi = g[i];
    float pi = p[i], mi = m[i], vi = v[i];
    if (sg.n_sub == 2) adam_elem<2, 2>(pi, mi, vi, gi, sg, c, beta1, beta2, eps);      // (the arithmetic of adam_next_kernel: every learner mode takes the same step)
    else adam_elem<1, 1>(pi, mi, vi, gi, sg, c, beta1, beta2, eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
}

// ---- the optimiser launch of a single-rank minibatch step (round 5) ---------------------------------------------------------------
// adam2_kernel's arithmetic (adam_elem / norm_coef_block are the same statements, so the same bits), re-tiled so that the launch can also do
// the three jobs trunk_adv_kernel did at the head of the NEXT step -- that launch (8 us, all of it latency on the critical path of every
// minibatch step) disappears from cirs_ppo_learn's loop:
//   [0, n_a0)           A0: Adam on [trunk | wc | bc], one element per thread, the parameters written through + an arrival count (what T waits for)
//   next n_t            T: trunk forward of 32 rows of the next minibatch on the updated trunk (waits for A0, reads 22 KB)
//   next n_s            S: advantage statistics of the next minibatch (no dependency on the update)
//   next n_p            P: Adam on one 32-item tile of Wa (8 consecutive elements of a head row per thread) + the tile's fp16 planes from
//                          the updated registers (what wa_planes_block re-read from memory)
//   rest                A: Adam on ba; the first one reduces the loss terms and publishes them
// Without a next step (n_t = n_s = 0) it is the step's Adam launch and nothing else.
struct AdamNext {
    int n_t, n_s, n_p;
    cirs_policy_cfg pcfg;
    const float* obs_flat; const int32_t* idx; int mb, n_pad;      // next minibatch: rows idx[0 .. mb) of the buffer-order batch
    const float* adv_flat; const int32_t* sidx; int m_stats, enable; float* red;
    uint4* planes;
    int n_a0;                                                      // A0 workgroups (set with or without a next step)
    int s_magic;                                                   // ceil(65536 / S): i / S = (i * s_magic) >> 16 for i < 2048
    int pa_delay;                                                  // P / A workgroups start this many x 1024 cycles late (the T workgroups' requests go first)
    int drop_arrival;                                              // TEST HOOK (CIRS_PPO_TEST_DROP_ARRIVAL=1): A0 workgroup 0 never raises its flag -> the T workgroups' wait must time out LOUDLY
    cirs_ppo_batch bt; int n_env;
    TrunkRowOut out;
};
// clip_grad_norm_ stage 2 in every workgroup (256 threads): the same fixed tree over the same slots -> the same coefficient everywhere
__device__ __forceinline__ float norm_coef_block(float part_a, float part_b, const cirs_ppo_cfg& cfg, float* sh, float& total_norm) {
    const int tid = threadIdx.x;
    static_assert(kNormBlocks == 256 && kWaSumBlocks <= 256, "each thread folds one slot of each kind");
    sh[tid] = part_a + part_b;      // slot tid of sumsq_partial_kernel / the F workgroups + slot tid of the wa|ba slab-sum workgroups
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    total_norm = sqrtf(sh[0]);
    float c = 1.0f;
    if (cfg.max_grad_norm > 0.f) c = fminf(cfg.max_grad_norm / (total_norm + 1e-6f), 1.0f);
    return c;
}
__device__ __forceinline__ float norm_coef_block(const float* __restrict__ partial, const cirs_ppo_cfg& cfg, float* sh, float& total_norm) {
    const int tid = threadIdx.x;
    return norm_coef_block(partial[tid], (partial[kNormBlocks + tid] + partial[2 * kNormBlocks + tid]) + partial[3 * kNormBlocks + tid], cfg, sh, total_norm);
}
constexpr int kTrunkQ2 = 13;         // 256 x 13 float2 >= 64 (S + 66) floats, S <= 32: the trunk's parameters, two elements per thread and pass
constexpr int kTrunkRowsPerWg = kTileM;   // rows of a T workgroup: one MFMA row tile
struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    PpoLayout L; AdamSeg sa, sb; float beta1, beta2, eps; cirs_ppo_cfg cfg;
    const float* partial; float* tail; float* loss_out; int mb, mb_norm;
};
struct AdamLds {
    float sh[256];
    float sh3[768];
    __attribute__((aligned(16))) float lt[12288];               // T: TrunkTileLds (weights + the tiles of its 32 rows);  P: the item tile's fp32 image [32][65]
};
// T: the trunk forward of 32 rows of the NEXT minibatch on the weights this launch forms.
//   The rows' gathers (idx -> obs row, row scalars) are requested first; then the workgroup waits for the A0 workgroups of the same launch -- the 22
//   lowest block ids, one trunk / critic parameter per thread: their Adam update is a few hundred cycles behind one memory round trip, their stores are
//   written through and counted in sync[1] -- and reads the 5.6 k updated parameters (22 KB, sc1) into LDS.  (The first version formed the update in every T
//   workgroup from gradient + a snapshot of p / m / v: 89 KB of operands and 1.3 k instructions of Adam per workgroup were 20 k of its 27 k cycles.)
//   Both layers run on the fp32 matrix cores: D[feature][row] = W X^T as v_mfma_f32_32x32x2_f32 k-steps in ascending k from the bias -- bit for bit the
//   sequential fma chain of trunk_compute (MI355X_MICROARCH.md: the fp32 MFMA is an fma chain; the lane with hi = 0 supplies k = 2 s, hi = 1
//   k = 2 s + 1), so the rows equal what trunk_adv_kernel / the rollout compute.  Waves 0, 1 own one 32-feature tile each; the critic chains run
//   one row per lane; the outputs leave as whole tiles (coalesced float4 rows, H2's fp16 planes as 16-byte units of the head kernels' layout).
constexpr int kTS = kH + 1;          // LDS row stride of the T role's tiles (odd: lane = row reads are conflict-free)
struct TrunkTileLds {                // overlays AdamLds::lt
    float w2[kH * kTS];
    float w1[kH * 33];
    float vec[4 * kH];               // b1 | b2 | wc | bc
    float x[kTileM * 33];
    float h1[kTileM * kTS];
    float h2[kTileM * kTS];
    int ri[kTileM];
    float dump[32];
};
constexpr int kTW1 = kH * kTS, kTVec = kTW1 + kH * 33, kTDump = kTVec + 4 * kH + kTileM * 33 + 2 * kTileM * kTS + kTileM;
__device__ __forceinline__ void adam_next_trunk(const AdamArgs& a, const AdamNext& nx, AdamLds& l, int b, const int* mv_sync) {
    const int tid = threadIdx.x, S = a.cfg.dim_state;
    const int lane = tid & 63, wv = tid >> 6, hi = lane >> 5, lo = lane & 31;
    const int j0 = b * kTileM;
    TrunkTileLds& t = *reinterpret_cast<TrunkTileLds*>(l.lt);
    const int ldx = S | 1;
    // ---- requests: the rows' gathers (idx -> obs row / row scalars) travel while this workgroup waits for the trunk's updated parameters ------
    const int n_tr = (int)a.L.trunk;
    // obs elements of the 32 rows: element i = row * S + k, three passes of 256 threads cover S <= 24; the rest (S <= 32) in a fourth
    int xrow[4], xk[4], xri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + 256 * q;
        const int row = (i * nx.s_magic) >> 16;                 // i / S (s_magic = ceil(65536 / S), exact for i < 2048)
        const bool ok = i < kTileM * S;
        xrow[q] = ok ? row : 0; xk[q] = ok ? i - row * S : -1;
        xri[q] = nx.idx[min(j0 + xrow[q], nx.mb - 1)];      // (rows >= mb are never used: clamped so that the load stays inside the index array)
    }
    // row scalars: thread (field = tid / 32, row = tid % 32): act, row_t, row_env, adv, logp_old, ret, v_s
    const int fr = tid & 31, ff = tid >> 5;
    const bool frow_ok = j0 + fr < nx.mb;
    const int fri = nx.idx[min(j0 + fr, nx.mb - 1)];
    float xq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xq[q] = (xk[q] >= 0 && j0 + xrow[q] < nx.mb) ? nx.obs_flat[(size_t)xri[q] * S + xk[q]] : 0.f;      // rows >= mb: zeros (as trunk_rows)
    int fi = 0; float fv = 0.f;
    {
        const void* srcp = ff == 0 ? (const void*)nx.bt.act : ff == 1 ? (const void*)nx.bt.row_t : ff == 2 ? (const void*)nx.bt.row_env
                         : ff == 3 ? (const void*)nx.bt.adv : ff == 4 ? (const void*)nx.bt.logp_old : ff == 5 ? (const void*)nx.bt.ret : (const void*)nx.bt.v_s;
        const int w32 = (frow_ok && ff < 7) ? reinterpret_cast<const int*>(srcp)[fri] : 0;       // (all seven fields are 4-byte arrays)
        fi = w32; fv = __int_as_float(w32);
    }
    // ---- the trunk's updated parameters: written through by the A0 workgroups of this launch (22 workgroups, one element per thread, the lowest
    // block ids: dispatched first), counted in sync[1]; read around the caches as 8-byte units -> LDS ---------------------------------------------------
    flags_wait(mv_sync + kSyncA0, nx.n_a0);
    CIRS_PSTAMP(b == 0, 1);
    const int ld1 = S | 1, o_b1 = kH * S, o_w2 = o_b1 + kH, o_b2 = o_w2 + kH * kH;
    float2 pw[kTrunkQ2];
#pragma unroll
    for (int q = 0; q < kTrunkQ2; ++q) {
        const int e2 = tid + 256 * q;
        pw[q] = ld_sc1x2(a.p + 2 * (e2 < (n_tr >> 1) ? e2 : 0));
    }
    const float pc = ld_sc1(a.p + a.L.wc + (tid < kH + 1 ? tid : 0));                     // wc | bc
#pragma unroll
    for (int q = 0; q < kTrunkQ2; ++q) {
        const int e2 = tid + 256 * q;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = 2 * e2 + u;
            const int r1 = (e * nx.s_magic) >> 16, r2 = e - o_w2;          // row of a W1 element (e < 64 S <= 2048: exact)
            // offset inside TrunkTileLds as integer selects (a select between pointers compiled to a branch per region and element)
            const int d_w1 = kTW1 + r1 * ld1 + (e - r1 * S), d_b1 = kTVec + (e - o_b1), d_w2 = (r2 >> 6) * kTS + (r2 & 63), d_b2 = kTVec + kH + (e - o_b2);
            int d = e < o_b2 ? d_w2 : d_b2;
            d = e < o_w2 ? d_b1 : d;
            d = e < o_b1 ? d_w1 : d;
            d = e2 < (n_tr >> 1) ? d : kTDump + (tid & 31);                // (threads beyond the trunk store to a dump row)
            l.lt[d] = u ? pw[q].y : pw[q].x;
        }
    }
    if (tid < kH + 1) t.vec[2 * kH + tid] = pc;               // wc[0..63], bc at [3 * kH]
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (xk[q] >= 0) { t.x[xrow[q] * ldx + xk[q]] = xq[q]; nx.out.obs_copy[(size_t)(j0 + xrow[q]) * S + xk[q]] = xq[q]; }
    if (nx.out.act) {
        if (ff == 0) nx.out.act[j0 + fr] = fi;
        else if (ff == 1) t.ri[fr] = fi;                      // row_t; combined with row_env below
        else if (ff >= 3 && ff < 7) nx.out.row4[(size_t)(ff - 3) * nx.n_pad + j0 + fr] = fv;
    }
    __syncthreads();
    if (nx.out.act && ff == 2) nx.out.dst[j0 + fr] = frow_ok ? (long)t.ri[fr] * nx.n_env + fi : 0;
    CIRS_PSTAMP(b == 0, 2);
    // ---- layer 1: waves 0, 1 (feature tile wv) --------------------------------------------------------------------------------
    if (wv < 2) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = t.vec[32 * wv + acc_row(r, hi)];              // b1
        const float* wr = t.w1 + (32 * wv + lo) * ld1 + hi;
        const float* xr = t.x + lo * ldx + hi;
        const int ks = (S + 1) >> 1;
        float av[16], bv[16];                                  // S <= 32: at most 16 k-steps, operands requested up front
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const bool kin = 2 * s2 + hi < S;                  // (odd S: the last k-step's upper half multiplies zeros)
            av[s2] = kin ? wr[2 * s2] : 0.f; bv[s2] = kin ? xr[2 * s2] : 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2)
            if (s2 < ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);      // (wave-uniform)
#pragma unroll
        for (int r = 0; r < 16; ++r) t.h1[lo * kTS + 32 * wv + acc_row(r, hi)] = fmaxf(acc[r], 0.f);
    }
    __syncthreads();
    // ---- layer 2 ----------------------------------------------------------------------------------------------------------------
    if (wv < 2) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = t.vec[kH + 32 * wv + acc_row(r, hi)];         // b2
        float av[32], bv[32];
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) { av[s2] = t.w2[(32 * wv + lo) * kTS + 2 * s2 + hi]; bv[s2] = t.h1[lo * kTS + 2 * s2 + hi]; }
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) t.h2[lo * kTS + 32 * wv + acc_row(r, hi)] = fmaxf(acc[r], 0.f);
    }
    __syncthreads();
    CIRS_PSTAMP(b == 0, 18);
    // ---- outputs ----------------------------------------------------------------------------------------------------------------
    if (wv == 3 && lane < kTileM) {      // critic: one row per lane, sequential chain (bit-reproducible)
        float v = t.vec[3 * kH];
        const float* hr = t.h2 + lane * kTS;
#pragma unroll
        for (int k = 0; k < kH; ++k) v = __builtin_fmaf(t.vec[2 * kH + k], hr[k], v);
        nx.out.value[j0 + lane] = v;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {        // h1 / h2 tiles as float4 rows
        const int f = tid + 256 * q, row = f >> 4, c4 = (f & 15) * 4;
        const float* s1 = t.h1 + row * kTS + c4; const float* s2 = t.h2 + row * kTS + c4;
        *reinterpret_cast<f32x4*>(nx.out.h1 + (size_t)(j0 + row) * kH + c4) = f32x4{s1[0], s1[1], s1[2], s1[3]};
        *reinterpret_cast<f32x4*>(nx.out.h2 + (size_t)(j0 + row) * kH + c4) = f32x4{s2[0], s2[1], s2[2], s2[3]};
    }
    {   // H2 planes, 16-byte units of the head kernels' register order (trunk_row_planes: the same pieces, element by element)
        const int tile = j0 >> 5;
        // hz: unit (k-step s = tid / 64, lane' = (hi', row)) = columns 16 s + 8 hi' .. + 8 of the row
        const int s4 = tid >> 6, hz_hi = (tid >> 5) & 1, hz_row = tid & 31;
        const float* r = t.h2 + hz_row * kTS + 16 * s4 + 8 * hz_hi;
        const HPl pz = hsplit8(kScH2 * r[0], kScH2 * r[1], kScH2 * r[2], kScH2 * r[3], kScH2 * r[4], kScH2 * r[5], kScH2 * r[6], kScH2 * r[7]);
        uint4* z = nx.out.h2z + (size_t)(tile * 12 + s4 * 3) * 64 + (tid & 63);
        z[0] = __builtin_bit_cast(uint4, pz.h); z[128] = __builtin_bit_cast(uint4, pz.l);
        // hb: unit (column half ch = tid / 128, t = (tid / 64) % 2, lane'' = (hi_b, column % 32)) = rows acc_row(8 t + jb, hi_b), jb < 8, of one column
        const int ch = tid >> 7, tt = (tid >> 6) & 1, hb_hi = (tid >> 5) & 1, col = 32 * ch + (tid & 31);
        float xb[8];
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) xb[jb] = kScH2 * t.h2[acc_row(8 * tt + jb, hb_hi) * kTS + col];
        const HPl pb = hsplit8(xb[0], xb[1], xb[2], xb[3], xb[4], xb[5], xb[6], xb[7]);
        uint4* bq = nx.out.h2b + (size_t)(tile * 12 + (ch * 2 + tt) * 3) * 64 + (tid & 63);
        bq[0] = __builtin_bit_cast(uint4, pb.h); bq[128] = __builtin_bit_cast(uint4, pb.l);
    }
}
// P: thread = 8 consecutive elements of one head row (two float4: f = tid, tid + 256 of the tile's 512)
__device__ __forceinline__ void adam_next_planes(const AdamArgs& a, const AdamNext& nx, AdamLds& l, int bp) {
    const int tid = threadIdx.x, I = a.cfg.n_items;
    const int tile0 = bp * kTileN;
    f32x4 g4[2], p4[2], m4[2], v4[2];
    size_t off[2];
    bool ok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = tid + 256 * q, item = f >> 4, col = (f & 15) * 4;
        ok[q] = tile0 + item < I;
        off[q] = (size_t)a.L.wa + (size_t)(ok[q] ? tile0 + item : 0) * kH + col;
        g4[q] = *reinterpret_cast<const f32x4*>(a.g + off[q]); p4[q] = *reinterpret_cast<const f32x4*>(a.p + off[q]);
        m4[q] = *reinterpret_cast<const f32x4*>(a.m + off[q]); v4[q] = *reinterpret_cast<const f32x4*>(a.v + off[q]);
    }
    float tn;
    const float c = norm_coef_block(a.partial, a.cfg, l.sh, tn);
    CIRS_PSTAMP(nx.n_t > 0 && bp == 0, 9);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = tid + 256 * q, item = f >> 4, col = (f & 15) * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float pi = p4[q][u], mi = m4[q][u], vi = v4[q][u];
            adam_elem<1, 1>(pi, mi, vi, g4[q][u], a.sb, c, a.beta1, a.beta2, a.eps);
            p4[q][u] = ok[q] ? pi : 0.f; m4[q][u] = mi; v4[q][u] = vi;          // (items beyond the catalogue: zero rows in the planes)
        }
        if (ok[q]) {
            *reinterpret_cast<f32x4*>(a.p + off[q]) = p4[q]; *reinterpret_cast<f32x4*>(a.m + off[q]) = m4[q]; *reinterpret_cast<f32x4*>(a.v + off[q]) = v4[q];
        }
        float* d = &l.lt[item * 65 + col];
        d[0] = p4[q][0]; d[1] = p4[q][1]; d[2] = p4[q][2]; d[3] = p4[q][3];
    }
    __syncthreads();
    CIRS_PSTAMP(nx.n_t > 0 && bp == 0, 10);
    wa_planes_from_lds(bp, nx.planes, l.lt);
}
// A0: element e of [trunk | wc | bc] (nothing else: the T workgroups wait for these)
__device__ __forceinline__ void adam_next_trunk_params(const AdamArgs& a, const MbView& mv, AdamLds& l, int b0, int drop_arrival) {
    const int tid = threadIdx.x;
    const long e = b0 * 256L + tid;
    const long i = e < a.L.trunk ? e : a.L.wc + (e - a.L.trunk);
    const long ic = i < a.L.total ? i : 0;
    float gi = a.g[ic], pi = a.p[ic], mi = a.m[ic], vi = a.v[ic];      // (requested before the norm's tree: one round trip for both)
    float total_norm;
    const float c = norm_coef_block(a.partial, a.cfg, l.sh, total_norm);
    if (i < a.L.total) {
        if (i < a.L.trunk) adam_elem<2, 2>(pi, mi, vi, gi, a.sa, c, a.beta1, a.beta2, a.eps);
        else adam_elem<1, 1>(pi, mi, vi, gi, a.sb, c, a.beta1, a.beta2, a.eps);
        st_sc1(a.p + i, pi);        // (written through: the T workgroups of this launch read it)
        a.m[i] = mi; a.v[i] = vi;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!(drop_arrival && b0 == 0)) flag_arrive(mv.sync + kSyncA0, b0);
}
// A: ba (the trunk / critic belong to A0, the Wa matrix to the P workgroups); its first workgroup reduces the loss terms and publishes them
__device__ __forceinline__ void adam_next_rest(const AdamArgs& a, const MbView& mv, AdamLds& l, int ba_) {
    const int tid = threadIdx.x;
    // (the norm's slots and this thread's ba element are requested before the loss terms are folded: one round trip for all of them)
    const float np_a = a.partial[tid], np_b = (a.partial[kNormBlocks + tid] + a.partial[2 * kNormBlocks + tid]) + a.partial[3 * kNormBlocks + tid];
    float total_norm, c;
    float lt[3] = {0.f, 0.f, 0.f};          // (thread 0 of the first workgroup: the step's loss terms)
    if (ba_ == 0 && a.mb > 0) {
        // the loss terms' three sums and the norm's tree share their barriers (nine instead of eighteen): the same pairs are added in the same order as
        // loss_partials_block / norm_coef_block add them (the coefficient must be the same bits in every workgroup)
        float e = 0.f, cl = 0.f, f = 0.f;
        {
            float e8[8], c8[8], f8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = tid + 256 * q, rc = r < a.mb ? r : 0;
                e8[q] = mv.ent_row[rc]; c8[q] = mv.clip_row[rc]; f8[q] = mv.vf_row[rc];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (tid + 256 * q < a.mb) { e += e8[q]; cl += c8[q]; f += f8[q]; }
        }
        for (int r = tid + 2048; r < a.mb; r += 256) { e += mv.ent_row[r]; cl += mv.clip_row[r]; f += mv.vf_row[r]; }
        float* sh3 = l.sh3;
        sh3[tid] = cl; sh3[256 + tid] = f; sh3[512 + tid] = e; l.sh[tid] = np_a + np_b;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (tid < s2) { sh3[tid] += sh3[tid + s2]; sh3[256 + tid] += sh3[256 + tid + s2]; sh3[512 + tid] += sh3[512 + tid + s2]; l.sh[tid] += l.sh[tid + s2]; }
            __syncthreads();
        }
        total_norm = sqrtf(l.sh[0]);
        c = 1.0f;
        if (a.cfg.max_grad_norm > 0.f) c = fminf(a.cfg.max_grad_norm / (total_norm + 1e-6f), 1.0f);
        if (tid == 0) {
            const float inv = 1.0f / (float)a.mb_norm;
            lt[0] = sh3[0] * inv; lt[1] = sh3[256] * inv; lt[2] = sh3[512] * inv;
            a.tail[0] = lt[0]; a.tail[1] = lt[1]; a.tail[2] = lt[2]; a.tail[3] = 0.f;
        }
    } else {
        c = norm_coef_block(np_a, np_b, a.cfg, l.sh, total_norm);
        if (ba_ == 0 && tid == 0) { lt[0] = a.tail[0]; lt[1] = a.tail[1]; lt[2] = a.tail[2]; }
    }
    if (ba_ == 0 && tid == 0) {
        mv.red[4] = c;
        mv.red[5] = total_norm;
        const float clip = lt[0], vf = lt[1], ent = lt[2];
        a.loss_out[0] = clip + a.cfg.vf_coef * vf - a.cfg.ent_coef * ent;
        a.loss_out[1] = clip; a.loss_out[2] = vf; a.loss_out[3] = ent;
    }
    const long i = a.L.ba + ba_ * 256L + tid;
    if (i < a.L.wc) {
        float gi = a.g[i], pi = a.p[i], mi = a.m[i], vi = a.v[i];
        adam_elem<1, 1>(pi, mi, vi, gi, a.sb, c, a.beta1, a.beta2, a.eps);
        a.p[i] = pi; a.m[i] = mi; a.v[i] = vi;
    }
}
__global__ __launch_bounds__(256) void adam_next_kernel(AdamArgs a, MbView mv, AdamNext nx) {
    __shared__ AdamLds l;
    const int b = blockIdx.x;
    const int b_t = nx.n_a0, b_s = b_t + nx.n_t, b_p = b_s + nx.n_s, b_a = b_p + nx.n_p;
    const bool pn = nx.n_t > 0;      // (probe builds stamp the launches that have a next step)
    CIRS_PSTAMP(pn && b == b_t, 0); CIRS_PSTAMP(pn && b == b_s - 1, 4); CIRS_PSTAMP(pn && b == b_s, 6); CIRS_PSTAMP(pn && b == b_p, 8);
    CIRS_PSTAMP(pn && b == b_a - 1, 12); CIRS_PSTAMP(pn && b == 0, 14); CIRS_PSTAMP(pn && b == nx.n_a0 - 1, 16);
    if (b >= b_p) for (int q = 0; q < nx.pa_delay; ++q) __builtin_amdgcn_s_sleep(16);
    if (b < b_t) adam_next_trunk_params(a, mv, l, b, nx.drop_arrival);
    else if (b < b_s) adam_next_trunk(a, nx, l, b - b_t, mv.sync);
    else if (b < b_p) adv_stats_block(nx.adv_flat, nx.sidx, nx.m_stats, nx.enable, nx.red, l.sh);
    else if (b < b_a) adam_next_planes(a, nx, l, b - b_p);
    else adam_next_rest(a, mv, l, b - b_a);
    CIRS_PSTAMP(pn && b == b_t, 3); CIRS_PSTAMP(pn && b == b_s - 1, 5); CIRS_PSTAMP(pn && b == b_s, 7); CIRS_PSTAMP(pn && b == b_p, 11);
    CIRS_PSTAMP(pn && b == b_a - 1, 13); CIRS_PSTAMP(pn && b == 0, 15); CIRS_PSTAMP(pn && b == nx.n_a0 - 1, 17);
    (void)pn;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, int n_sub, float beta1, float beta2, float eps,
                                                   float step_size0, float bc2s0, float step_size1, float bc2s1,
                                                   const float* __restrict__ grad_scale, int scale_pow) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    if (grad_scale) {
        const float c = grad_scale[0];
        for (int q = 0; q < scale_pow; ++q) gi *= c;  // clip coefficient applied once per occurrence in the param list
    }
    float pi = p[i], mi = m[i], vi = v[i];
    for (int sub = 0; sub < n_sub; ++sub) {
        mi = mi + (1.0f - beta1) * (gi - mi);                 // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * beta2 + (1.0f - beta2) * gi * gi;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float ss = sub == 0 ? step_size0 : step_size1;
        const float b2 = sub == 0 ? bc2s0 : bc2s1;
        const float denom = sqrtf(vi) / b2 + eps;
        pi = pi - ss * (mi / denom);                          // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
    p[i] = pi; m[i] = mi; v[i] = vi;
}

// ---- sharded optimiser of the data-parallel learner (reduce-scatter -> this rank's gradient shard -> Adam -> all-gather) --------
// stats layout per rank (kShardStatFloats): [0, kShardNormBlocks) partial sums of squares of the rank's shard (trunk elements
// weighted twice, as sumsq_partial_kernel), [kShardNormBlocks, +4) the loss partials {clip, vf, ent, 0} that live in the
// gradient tail [P, P+4) -- only where the shard holds them, 0 elsewhere, so the sum over ranks is exact.
constexpr int kShardNormBlocks = 64;
constexpr int kShardStatFloats = kShardNormBlocks + 8;
__global__ __launch_bounds__(256) void shard_sumsq_kernel(const float* __restrict__ g_shard, long begin, long len, long n_trunk, long P,
                                                          float* __restrict__ stats) {
    __shared__ float sh[256];
    const int tid = threadIdx.x;
    const long per = (len + kShardNormBlocks - 1) / kShardNormBlocks;
    const long lo = blockIdx.x * per, hi = min(len, lo + per);
    float acc = 0.f;
    for (long i = lo + tid; i < hi; i += 256) {
        const long gi = begin + i;
        if (gi < P) {
            const float x = g_shard[i];
            acc += (gi < n_trunk ? 2.0f : 1.0f) * x * x;
        }
    }
    sh[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) stats[blockIdx.x] = sh[0];
    if (blockIdx.x == 0 && tid < 8) {
        const long gi = P + tid;
        stats[kShardNormBlocks + tid] = (tid < 4 && gi >= begin && gi < begin + len) ? g_shard[gi - begin] : 0.f;
    }
}
// Adam over the shard from the all-gathered stats of every rank: the norm is the sum of the W x kShardNormBlocks partials in
// (rank, block) order -- identical on every rank --, the clip coefficient and the two segments as in adam2_kernel.
__global__ __launch_bounds__(256) void shard_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long begin, long len, long n_first, long P, AdamSeg sa,
                                                         AdamSeg sb, float beta1, float beta2, float eps, cirs_ppo_cfg cfg,
                                                         const float* __restrict__ stats_all, int world, float* __restrict__ loss_out) {
    __shared__ float sh[256];
    const int tid = threadIdx.x;
    float acc = 0.f;
    const int n_part = world * kShardNormBlocks;
    for (int q = tid; q < n_part; q += 256) acc += stats_all[(size_t)(q / kShardNormBlocks) * kShardStatFloats + (q % kShardNormBlocks)];
    sh[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const float total_norm = sqrtf(sh[0]);
    float c = 1.0f;
    if (cfg.max_grad_norm > 0.f) c = fminf(cfg.max_grad_norm / (total_norm + 1e-6f), 1.0f);
    if (blockIdx.x == 0 && tid == 0 && loss_out) {
        float t3[3] = {0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r)
            for (int j = 0; j < 3; ++j) t3[j] += stats_all[(size_t)r * kShardStatFloats + kShardNormBlocks + j];
        loss_out[0] = t3[0] + cfg.vf_coef * t3[1] - cfg.ent_coef * t3[2];
        loss_out[1] = t3[0]; loss_out[2] = t3[1]; loss_out[3] = t3[2];
    }
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= len || begin + i >= P) return;
    const AdamSeg sg = begin + i < n_first ? sa : sb;
    float gi = g[i];
    float pi = p[i], mi = m[i], vi = v[i];
    if (sg.n_sub == 2) adam_elem<2, 2>(pi, mi, vi, gi, sg, c, beta1, beta2, eps);      // (the arithmetic of adam_next_kernel: every learner mode takes the same step)
    else adam_elem<1, 1>(pi, mi, vi, gi, sg, c, beta1, beta2, eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
}

static int launch_adam(float* p, const float* g, float* m, float* v, long n, long step_before, int n_sub, float lr, float b1,
                       float b2, float eps, const float* grad_scale, int scale_pow, hipStream_t s) {
    CIRS_REQUIRE(n_sub == 1 || n_sub == 2, "n_sub must be 1 or 2");
    double ss[2] = {0, 0}, bs[2] = {1, 1};
    for (int q = 0; q < n_sub; ++q) {
        const double t = (double)(step_before + 1 + q);
        ss[q] = (double)lr / (1.0 - pow((double)b1, t));
        bs[q] = sqrt(1.0 - pow((double)b2, t));
    }
    hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, g, m, v, n, n_sub, b1, b2, eps, (float)ss[0],
                       (float)bs[0], (float)ss[1], (float)bs[1], grad_scale, scale_pow);
    CIRS_CHECK_LAUNCH("adam_kernel");
    return CIRS_OK;
}

#ifdef CIRS_HEAD_PROF
extern "C" int cirs_debug_step_prof(unsigned long long* out_host64) {
    return hipMemcpyFromSymbol(out_host64, HIP_SYMBOL(cirs::g_step_prof), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
extern "C" int cirs_debug_head_prof(unsigned long long* out_host32) {
    return hipMemcpyFromSymbol(out_host32, HIP_SYMBOL(cirs::g_head_prof), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

static int validate_ppo(const cirs_ppo_cfg* cfg) {
    CIRS_REQUIRE(cfg, "ppo cfg null");
    if (cfg->hidden != kH) return fail(CIRS_E_UNSUPPORTED, "this build supports hidden == 64 only");
    CIRS_REQUIRE(cfg->n_items > 0 && cfg->dim_state > 0 && cfg->dim_state <= 64, "bad n_items/dim_state");
    return CIRS_OK;
}

}  // namespace cirs

extern "C" int64_t cirs_ppo_param_count(const cirs_ppo_cfg* cfg) {
    if (!cfg) return 0;
    return cirs::ppo_layout(cfg->n_items, cfg->dim_state).total;
}

extern "C" int64_t cirs_ppo_workspace_bytes(const cirs_ppo_cfg* cfg, int32_t max_minibatch) {
    using namespace cirs;
    if (!cfg || max_minibatch <= 0) return 0;
    return (int64_t)(mb_ws_floats(n_pad_of(max_minibatch), cfg->n_items, cfg->dim_state) + 64 * 32) * 4;
}

extern "C" int cirs_ppo_prepare(const cirs_ppo_cfg* cfg, const cirs_traj* traj, const int32_t* lens,
                                const int32_t* offsets, int32_t n_env, int32_t max_turn, int32_t n_rows,
                                double* rms_state, const cirs_ppo_batch* out, void* stream) {
    using namespace cirs;
    if (int rc = validate_ppo(cfg)) return rc;
    CIRS_REQUIRE(traj && lens && offsets && rms_state && out, "null argument");
    CIRS_REQUIRE(out->obs && out->act && out->adv && out->ret && out->v_s && out->logp_old && out->row_env && out->row_t, "batch pointer null");
    CIRS_REQUIRE(n_env > 0 && max_turn > 0 && n_rows > 0, "bad sizes");
    hipStream_t s = (hipStream_t)stream;
    // float64 scratch for the un-normalised returns lives in the ret buffer's shadow: allocate on the stream
    double* unnorm = nullptr;
    CIRS_HIP(hipMallocAsync((void**)&unnorm, sizeof(double) * (size_t)n_rows, s));
    // one wavefront per workgroup: the per-env row stores touch 64 cache lines per instruction, so the wavefronts are spread over as
    // many CUs (address units) as there are
    hipLaunchKernelGGL(gae_kernel, dim3(cdiv(n_env, 64)), dim3(64), 0, s, *cfg, *traj, lens, offsets, n_env, cfg->dim_state,
                       rms_state, *out, unnorm);
    hipLaunchKernelGGL(compact_obs_kernel, dim3(cdiv((long)n_rows * cfg->dim_state, 256)), dim3(256), 0, s, *traj, *out, n_rows, n_env, cfg->dim_state);
    CIRS_CHECK_LAUNCH("gae_kernel");
    hipLaunchKernelGGL(returns_kernel, dim3(1), dim3(1024), 0, s, *cfg, unnorm, n_rows, rms_state, out->ret);
    CIRS_CHECK_LAUNCH("returns_kernel");
    CIRS_HIP(hipFreeAsync(unnorm, s));
    return CIRS_OK;
}

// process_fn without a host-side row count: the offsets and N are formed on the device from the episode lengths, so the whole
// preparation can be enqueued right behind the rollout while the host is still waiting for the lengths (the one read-back of an update
// then overlaps with these kernels instead of leaving the GPU idle behind it).  offsets_out [n_env], n_rows_out [1]: device outputs;
// scratch: n_env * max_turn doubles; the batch arrays must hold n_env * max_turn rows.  Same kernels as cirs_ppo_prepare: same bits.
extern "C" int cirs_ppo_prepare_async_perms(const cirs_ppo_cfg* cfg, const cirs_traj* traj, const int32_t* lens, int32_t n_env, int32_t max_turn,
                                            int32_t* offsets_out, int32_t* n_rows_out, double* rms_state, const cirs_ppo_batch* out,
                                            double* scratch, uint64_t perm_seed, uint64_t perm_tag0, int32_t n_perm, int32_t* perm_out, int32_t offsets_ready,
                                            void* stream) {
    using namespace cirs;
    if (int rc = validate_ppo(cfg)) return rc;
    CIRS_REQUIRE(traj && lens && offsets_out && n_rows_out && rms_state && out && scratch, "null argument");
    CIRS_REQUIRE(out->obs && out->act && out->adv && out->ret && out->v_s && out->logp_old && out->row_env && out->row_t, "batch pointer null");
    CIRS_REQUIRE(n_env > 0 && n_env <= (1 << 20) && max_turn > 0, "bad sizes");
    CIRS_REQUIRE(n_perm >= 0 && n_perm <= kMaxPermKeys && (n_perm == 0 || perm_out), "permutations: at most 8 per call, perm_out non-null");
    hipStream_t s = (hipStream_t)stream;
    if (!offsets_ready) hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(1024), 0, s, lens, n_env, offsets_out, n_rows_out);      // (else: cirs_ppo_update_readback formed them)
    hipLaunchKernelGGL(gae_kernel, dim3(cdiv(n_env, 64)), dim3(64), 0, s, *cfg, *traj, lens, (const int32_t*)offsets_out, n_env, cfg->dim_state,
                       rms_state, *out, scratch);
    const long upper = (long)n_env * max_turn;
    const int n_compact = (int)cdiv(upper * cfg->dim_state, 1024L);
    PermKeys keys{};
    for (int q = 0; q < n_perm; ++q) keys.k[q] = perm_key(perm_seed, perm_tag0 + (uint64_t)q);
    hipLaunchKernelGGL(prepare_tail_kernel, dim3(1 + n_compact + (int)cdiv((long)n_perm * upper, 1024L)), dim3(1024), 0, s, *cfg, *traj, *out, (const double*)scratch,
                       rms_state, (const int32_t*)n_rows_out, n_env, cfg->dim_state, n_compact, keys, (int)n_perm, upper, perm_out);
    CIRS_CHECK_LAUNCH("cirs_ppo_prepare_async");
    return CIRS_OK;
}
extern "C" int cirs_ppo_prepare_async(const cirs_ppo_cfg* cfg, const cirs_traj* traj, const int32_t* lens, int32_t n_env, int32_t max_turn,
                                      int32_t* offsets_out, int32_t* n_rows_out, double* rms_state, const cirs_ppo_batch* out,
                                      double* scratch, void* stream) {
    return cirs_ppo_prepare_async_perms(cfg, traj, lens, n_env, max_turn, offsets_out, n_rows_out, rms_state, out, scratch, 0, 0, 0, nullptr, 0, stream);
}

extern "C" int cirs_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int64_t step_before,
                              int32_t n_sub, float lr, float beta1, float beta2, float eps, const float* grad_scale,
                              int32_t scale_pow, void* stream) {
    using namespace cirs;
    CIRS_REQUIRE(params && grads && m && v && n > 0, "bad arguments");
    return launch_adam(params, grads, m, v, n, step_before, n_sub, lr, beta1, beta2, eps, grad_scale, scale_pow, (hipStream_t)stream);
}

// ---- the launches of a minibatch step as separate pieces, so that cirs_ppo_minibatch (one step), cirs_ppo_minibatch_dp (the step cut at the
// gradient all-reduce) and cirs_ppo_learn (all steps of an update from one call) issue the SAME kernels on the same data ---------------
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
struct PpoRun {     // what does not change between the steps of a call
    const cirs_ppo_cfg* cfg;
    float *params, *grads, *adam_m, *adam_v;
    const cirs_ppo_batch* batch;
    int n_env;
    void* workspace;
    int n_pad_carve;        // the workspace is carved for this many rows (>= every step's n_pad), so consecutive steps of different size do not overlap
    hipStream_t s;
    int I, S;
    cirs::PpoLayout L;
    cirs_policy_cfg pcfg;
    cirs_policy_weights w;
    cirs::MbView v;
    float* tail;            // {clip, vf, ent, 0} partials of this rank (behind the flat gradient)
    bool merge_launch;      // CIRS_PPO_MERGE_KERNEL=1: the round-2 sequence with the merge of the statistics partials as a launch of its own (A/B runs)
};
struct PpoStep { const int32_t* idx; int mb; const int32_t* sidx; int mb_norm; float* dobs; float* loss_out; long opt_step; };

static PpoRun ppo_run(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v, const cirs_ppo_batch* batch, int n_env,
                      void* workspace, int max_mb, hipStream_t s) {
    using namespace cirs;
    PpoRun r;
    r.cfg = cfg; r.params = params; r.grads = grads; r.adam_m = adam_m; r.adam_v = adam_v; r.batch = batch; r.n_env = n_env; r.workspace = workspace;
    r.n_pad_carve = n_pad_of(max_mb); r.s = s; r.I = cfg->n_items; r.S = cfg->dim_state;
    r.L = ppo_layout(r.I, r.S);
    r.pcfg = cirs_policy_cfg{r.I, r.S, kH};
    r.w = cirs_policy_weights{params + r.L.w1, params + r.L.b1, params + r.L.w2, params + r.L.b2, params + r.L.wa, params + r.L.ba, params + r.L.wc,
                              params + r.L.bc};
    r.v = carve(workspace, r.n_pad_carve, r.I, r.S);
    r.tail = grads + r.L.total;
    const char* mk_ = getenv("CIRS_PPO_MERGE_KERNEL");      // (read per call: tests toggle it)
    r.merge_launch = mk_ && atoi(mk_) != 0;
    return r;
}
static cirs::TrunkRowOut trunk_out_of(const cirs::MbView& v) { return cirs::TrunkRowOut{v.h2, v.value, v.h1, v.obs, v.act, v.dst_row, v.adv, v.h2z, v.h2b}; }

// 1+2. advantage statistics of the (global) minibatch, the trunk forward (same fma chains as the rollout -> ratio == 1 exactly while the
//      weights are unchanged; rows gathered from the buffer-order batch through idx, v.obs keeps the copy for d W1) and the fp16 planes of Wa
static int launch_trunk_adv(const PpoRun& r, const PpoStep& st) {
    using namespace cirs;
    const int n_pad = n_pad_of(st.mb);
    hipLaunchKernelGGL(trunk_adv_kernel, dim3(cdiv(n_pad, 4) + 1 + cdiv(r.I, kTileN)), dim3(256), 0, r.s, r.pcfg, r.w, (const float*)r.batch->obs, (long)r.S,
                       n_pad, st.idx, st.mb, (const float*)r.batch->adv, st.sidx, st.mb_norm, (int)r.cfg->norm_adv, r.v.red, (int)cdiv(n_pad, 4),
                       r.v.wa_planes, *r.batch, r.n_env, trunk_out_of(r.v));
    CIRS_CHECK_LAUNCH("trunk_adv_kernel");
    return CIRS_OK;
}
// 3-5. head statistics, head backward (+ the merge of the statistics partials in its prologue), chunk-slab sums of d h2 (unless the trunk-backward
//      launch of the single-rank step sums them itself)
static int launch_head(const PpoRun& r, const PpoStep& st, int* n_bchunks_out, bool with_dh2_sum) {
    using namespace cirs;
    const int I = r.I, mb = st.mb, n_pad = n_pad_of(mb), n_slabs = n_row_blocks_of(n_pad);
    const MbView& v = r.v;
    ActorPartialView pv = partial_view(v.head_ws, n_pad, I);
    const int n_item_tiles = cdiv(I, kTileN);
    // head statistics (log-sum-exp, E_p[z]) on the matrix cores from the fp16 planes of Wa; all workgroups co-resident (2 per CU) with equal tile counts
    const int tpc_s = head_tiles_per_chunk(n_item_tiles, n_slabs, 2);   // measured: 1 / 2 / 3 workgroups per CU = 17.4 / 14.5 / 16.7 us
    const int n_schunks = cdiv(n_item_tiles, tpc_s);   // <= n_chunks: the partial arrays fit
    CIRS_PROF_LAUNCH(2, r.s, hipLaunchKernelGGL(head_stats_kernel, dim3(n_schunks, n_slabs), dim3(256), 0, r.s, I, mb, n_pad, tpc_s,
                                                (const uint4*)v.wa_planes, r.w.ba, (const uint4*)v.h2z, pv, (const int32_t*)v.act, v.za));
    CIRS_CHECK_LAUNCH("head_stats_kernel");
    // head backward; the merge of the statistics partials + row losses + backward coefficients (means over the global minibatch) run in
    // its prologue (CIRS_PPO_MERGE_KERNEL=1: as a launch of their own, the round-2 sequence, for A/B runs)
    const HeadMergeArgs hma{*r.cfg, *r.batch, st.idx, st.mb_norm, n_schunks, pv, env_int("CIRS_PPO_MERGE_T_ALL", 0)};
    if (r.merge_launch) {
        hipLaunchKernelGGL(head_stats_merge_kernel, dim3(cdiv(n_pad, 4)), dim3(256), 0, r.s, *r.cfg, *r.batch, st.idx, mb, st.mb_norm, n_pad, n_schunks,
                           r.n_env, pv, r.w.wa, r.w.ba, v, (const float*)nullptr, 0);
        CIRS_CHECK_LAUNCH("head_stats_merge_kernel");
    }
    // chunking of the backward kernel: all workgroups co-resident (1 per CU) with equal tile counts -> no tail round
    const int tpc = head_tiles_per_chunk(n_item_tiles, n_slabs, 1);
    const int n_bchunks = cdiv(n_item_tiles, tpc);  // <= n_chunks: the d h2 / entropy partial slabs fit
    const dim3 bgrid((n_bchunks + 7) & ~7, n_slabs), bblock(kBwdWaves * 64);
    if (r.cfg->ent_coef != 0.f) {
        if (r.merge_launch) CIRS_PROF_LAUNCH(1, r.s, hipLaunchKernelGGL((head_bwd_fused_kernel<true, false>), bgrid, bblock, 0, r.s, I, mb, n_pad, tpc, (const uint4*)v.wa_planes, r.w.ba, v, v.dwap, hma));
        else CIRS_PROF_LAUNCH(1, r.s, hipLaunchKernelGGL((head_bwd_fused_kernel<true, true>), bgrid, bblock, 0, r.s, I, mb, n_pad, tpc, (const uint4*)v.wa_planes, r.w.ba, v, v.dwap, hma));
    } else {
        if (r.merge_launch) CIRS_PROF_LAUNCH(1, r.s, hipLaunchKernelGGL((head_bwd_fused_kernel<false, false>), bgrid, bblock, 0, r.s, I, mb, n_pad, tpc, (const uint4*)v.wa_planes, r.w.ba, v, v.dwap, hma));
        else CIRS_PROF_LAUNCH(1, r.s, hipLaunchKernelGGL((head_bwd_fused_kernel<false, true>), bgrid, bblock, 0, r.s, I, mb, n_pad, tpc, (const uint4*)v.wa_planes, r.w.ba, v, v.dwap, hma));
    }
    CIRS_CHECK_LAUNCH("head_bwd_fused_kernel");
    if (with_dh2_sum) {
        hipLaunchKernelGGL(dh2_sum_kernel, dim3(n_pad * (kH / 4) / 64 + cdiv(n_pad, 64)), dim3(64), 0, r.s, mb, n_pad, n_bchunks, v);
        CIRS_CHECK_LAUNCH("dh2_sum_kernel");
    }
    *n_bchunks_out = n_bchunks;
    return CIRS_OK;
}
// d wc/d bc, d W2/d b2, d W1/d b1 as one row slab per trunk-backward workgroup (32 rows)
static cirs::DwJobs trunk_dw_jobs(const PpoRun& r, int n_pad) {
    using namespace cirs;
    const MbView& v = r.v;
    DwJobs jobs{};
    jobs.n = 3;
    jobs.j[0] = DwJob{v.dvalue, 1, v.h2, kH, 1, kH, r.grads + r.L.wc, r.grads + r.L.bc, 0, 0};
    jobs.j[1] = DwJob{v.da2, kH, v.h1, kH, kH, kH, r.grads + r.L.w2, r.grads + r.L.b2, 0, 0};
    jobs.j[2] = DwJob{v.da1, kH, v.obs, r.S, kH, r.S, r.grads + r.L.w1, r.grads + r.L.b1, 0, 0};
    const int n_dw_slabs = n_pad / kTileM;
    int off = 0, out = 0;
    for (int q = 0; q < 3; ++q) {
        jobs.j[q].part_off = off;
        off += n_dw_slabs * jobs.j[q].O * (jobs.j[q].K + 1);
        out += jobs.j[q].O * (jobs.j[q].K + 1);
    }
    jobs.total_out = out;
    return jobs;
}
// 6. trunk backward of the data-parallel step (and of CIRS_PPO_ROWS_KERNEL=0): d a2, d a1, d obs (scattered to the tracker-gradient tensor) over 32-row
//    MFMA tiles + the slab sums of the wa|ba gradient as extra workgroups of the same launch; the weight gradients stay in row slabs
static int launch_trunk_bwd(const PpoRun& r, const PpoStep& st, int n_bchunks) {
    using namespace cirs;
    static_assert(kH == 64, "trunk_bwd_kernel tiles assume hidden == 64");
    CIRS_REQUIRE(r.S <= 32, "dim_state > 32 is not supported by the trunk backward kernel");
    const int n_pad = n_pad_of(st.mb), n_slabs = n_row_blocks_of(n_pad);
    const long seg = (long)r.I * kH + r.I;
    const DwJobs jobs = trunk_dw_jobs(r, n_pad);
    hipLaunchKernelGGL(trunk_bwd_kernel, dim3(n_pad / kTileM + kWaSumBlocks), dim3(512), 0, r.s, st.mb, n_pad, n_bchunks, r.S, r.w.w1, r.w.w2, r.w.wc, r.v, st.dobs,
                       r.grads, (long)r.L.wa, seg, (long)dwa_slab_stride(r.I), n_slabs, jobs, r.v.dwp);
    CIRS_CHECK_LAUNCH("trunk_bwd_kernel");
    return CIRS_OK;
}
// Which trunk backward a step runs.  trunk_rows_kernel (one launch: chunk-slab sums, trunk backward, every gradient sum, squared-norm / loss
// partials) or the round-4 sequence dh2_sum_kernel + trunk_bwd_kernel (+ sumsq_partial_kernel / dw_multi_final + loss_partials_kernel).  Measured on
// one box at C3, 1024 rows (tools/ab_step.py, round 5): with ONE arrival counter 78.5 us per step against 77.2 us for the sequence; with one arrival FLAG
// per row workgroup (128 read-modify-writes of one address were serialised at the memory side) 75.75 against 76.1 us.  CIRS_PPO_ROWS_KERNEL=0/1 forces
// either (A/B runs, tests).
static bool rows_kernel_wanted(int phase, int mb) {
    if (cirs::n_pad_of(mb) / cirs::kRR > cirs::kSyncA0) return false;      // (one arrival flag per row workgroup)
    const char* e = getenv("CIRS_PPO_ROWS_KERNEL");
    if (e) return atoi(e) != 0;
    (void)phase;
    return true;
}
static int launch_trunk_rows(const PpoRun& r, const PpoStep& st, int n_bchunks, bool with_loss_partials) {
    using namespace cirs;
    CIRS_REQUIRE(r.S <= 32, "dim_state > 32 is not supported by the trunk backward kernel");
    const int n_pad = n_pad_of(st.mb), n_slabs = n_row_blocks_of(n_pad);
    const long seg = (long)r.I * kH + r.I;
    const int n_r = n_pad / kRR, n_f = cdiv(snap_floats(r.S), kFOut);
    CIRS_REQUIRE(n_r <= kSyncA0, "trunk_rows_kernel: more than 2048 rows in a minibatch (one arrival flag per 8 rows)");
    const long n4 = seg >> 2;
    int n_w = (int)cdiv(n4, 512L);
    n_w = n_w > kWaSlotsMax ? kWaSlotsMax : n_w;
    // (round 6, second half) no more W workgroups than CUs the R workgroups leave free: a W workgroup that shares a CU with an R workgroup competes with its
    // slab loads -- same box, 1024 rows: 340 / 256 / 170 / 128 / 84 W workgroups = 70.1 / 70.2 / 69.5 / 69.0 / 69.3 us per step; CIRS_PPO_W_WGS overrides
    {
        const int free_cus = std::max(64, device_cu_count() - n_r), wv = env_int("CIRS_PPO_W_WGS", 0);
        if (wv > 0) n_w = std::min(n_w, wv);
        else n_w = std::min(n_w, free_cus);
    }
    hipLaunchKernelGGL(trunk_rows_kernel, dim3(n_r + n_w + n_f + (with_loss_partials ? 1 : 0)), dim3(512), 0, r.s, st.mb, n_pad, n_bchunks, r.S, r.w.w1,
                       r.w.w2, r.w.wc, r.v, st.dobs, r.grads, (long)r.L.wa, seg, (long)dwa_slab_stride(r.I), n_slabs, n_r, snap_stride(r.S),
                       env_int("CIRS_PPO_W_DELAY", 0), with_loss_partials ? r.tail : (float*)nullptr, st.mb_norm, n_w);
    CIRS_CHECK_LAUNCH("trunk_rows_kernel");
    return CIRS_OK;
}
static cirs::AdamSeg adam_seg_of(const cirs_ppo_cfg* cfg, long step_before, int n_sub, int scale_pow) {
    cirs::AdamSeg sg{n_sub, scale_pow, 0.f, 1.f, 0.f, 1.f, 1.f, 1.f};
    for (int q = 0; q < n_sub; ++q) {
        const double t = (double)(step_before + 1 + q);
        const float ss = (float)((double)cfg->lr / (1.0 - pow((double)cfg->beta1, t)));
        const float bs = (float)sqrt(1.0 - pow((double)cfg->beta2, t));
        const float rbs = (float)(1.0 / sqrt(1.0 - pow((double)cfg->beta2, t)));
        if (q == 0) { sg.step_size0 = ss; sg.bc2s0 = bs; sg.rbc2s0 = rbs; } else { sg.step_size1 = ss; sg.bc2s1 = bs; sg.rbc2s1 = rbs; }
    }
    return sg;
}
// 7. clip_grad_norm_ + Adam (trunk: coefficient squared, two sub-steps 2k+1, 2k+2; heads: one step, coefficient once).
//    adam_next_kernel -- with `next` it also runs the head of the next step (trunk forward, advantage statistics, Wa planes); data-parallel
//    phase 2: the squared norm of the all-reduced flat gradient first (sumsq_partial_kernel).
static int launch_norm_adam(const PpoRun& r, const PpoStep& st, int phase, bool folded, const PpoStep* next) {
    using namespace cirs;
    const int n_pad = n_pad_of(st.mb);
    const long seg = (long)r.I * kH + r.I;
    const MbView& v = r.v;
    if (!(phase == 0 && folded)) {
        const DwJobs jobs = phase == 0 ? trunk_dw_jobs(r, n_pad) : DwJobs{};
        hipLaunchKernelGGL(sumsq_partial_kernel, dim3(kNormBlocks), dim3(256), 0, r.s, r.grads, r.L.trunk, r.L.total, r.L.wa, seg, (int)(phase == 0), jobs,
                           phase == 0 ? n_pad / kTileM : 0, phase == 0 ? (const float*)v.dwp : (const float*)nullptr, r.S, v.normp);
        CIRS_CHECK_LAUNCH("sumsq_partial_kernel");
    }
    const AdamSeg sa = adam_seg_of(r.cfg, 2 * st.opt_step, 2, 2), sb = adam_seg_of(r.cfg, st.opt_step, 1, 1);
    AdamNext nx{};
    nx.n_p = cdiv(r.I, kTileN);
    nx.planes = v.wa_planes;
    if (next) {
        const int np = n_pad_of(next->mb);
        nx.n_t = np / kTrunkRowsPerWg; nx.n_s = 1;
        nx.pcfg = r.pcfg; nx.obs_flat = r.batch->obs; nx.idx = next->idx; nx.mb = next->mb; nx.n_pad = np;
        nx.adv_flat = r.batch->adv; nx.sidx = next->sidx; nx.m_stats = next->mb_norm; nx.enable = (int)r.cfg->norm_adv; nx.red = v.red;
        nx.bt = *r.batch; nx.n_env = r.n_env; nx.out = trunk_out_of(v); nx.s_magic = (65536 + r.S - 1) / r.S;
        nx.pa_delay = env_int("CIRS_PPO_PA_DELAY", 0);      // (A/B on one box: 0 / 4 -> 78.0-78.6 / 78.5-78.9 us per step)
        nx.drop_arrival = env_int("CIRS_PPO_TEST_DROP_ARRIVAL", 0);
    }
    nx.n_a0 = cdiv(r.L.trunk + kH + 1, 256);
    const int n_a = nx.n_a0 + cdiv(r.I, 256);
    // (phase 2: the loss partials were all-reduced with the gradient: mb = 0 leaves `tail` as it is)
    const AdamArgs aa{r.params, r.grads, r.adam_m, r.adam_v, r.L, sa, sb, r.cfg->beta1, r.cfg->beta2, r.cfg->adam_eps, *r.cfg, v.normp, r.tail, st.loss_out,
                      phase == 0 ? st.mb : 0, st.mb_norm};
    hipLaunchKernelGGL(adam_next_kernel, dim3(nx.n_t + nx.n_s + nx.n_p + n_a), dim3(256), 0, r.s, aa, v, nx);
    CIRS_CHECK_LAUNCH("adam_next_kernel");
    return CIRS_OK;
}

// options of a data-parallel step that is part of a chain of steps (cirs_ppo_minibatch_dp_chain): the workspace is carved for the update's largest
// local minibatch, phase 1 skips the head launch when the previous step's phase 2 already ran it, phase 2 runs the head of the next step
struct DpChain { int max_mb; int head_done; const int32_t* next_idx; int next_mb; const int32_t* next_idx_global; int next_mb_global; };
static int ppo_minibatch_impl(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v, int64_t opt_step,
                              const cirs_ppo_batch* batch, const int32_t* idx, int32_t mb, const int32_t* idx_global,
                              int32_t mb_global, float* dobs_accum, int32_t n_env, float* loss_out, void* workspace,
                              int64_t workspace_bytes, int phase, void* stream, const DpChain* ch = nullptr) {
    using namespace cirs;
    if (int rc = validate_ppo(cfg)) return rc;
    CIRS_REQUIRE(params && grads && adam_m && adam_v && batch && loss_out && workspace, "null argument");
    CIRS_REQUIRE(phase >= 0 && phase <= 2, "phase must be 0, 1 or 2");
    CIRS_REQUIRE(mb >= 1 && mb_global >= 2 && mb <= mb_global, "bad minibatch sizes (need mb_global >= 2 for the unbiased std)");
    const int carve_mb = ch ? (ch->max_mb > mb ? ch->max_mb : mb) : mb;
    CIRS_REQUIRE(!ch || !ch->next_idx || ch->next_mb <= carve_mb, "next minibatch larger than max_mb");
    CIRS_REQUIRE(workspace_bytes >= cirs_ppo_workspace_bytes(cfg, carve_mb), "workspace too small");
    const PpoRun r = ppo_run(cfg, params, grads, adam_m, adam_v, batch, n_env, workspace, carve_mb, (hipStream_t)stream);
    const PpoStep st{idx, (int)mb, idx_global ? idx_global : idx, (int)(idx_global ? mb_global : mb), dobs_accum, loss_out, (long)opt_step};
    bool folded = false;
    const bool rows = rows_kernel_wanted(phase, mb);
    if (phase == 0 || phase == 1) {
        CIRS_REQUIRE(idx != nullptr || (ch && ch->head_done), "idx is null");
        int n_bchunks = 0;
        if (!(ch && ch->head_done)) { if (int rc = launch_trunk_adv(r, st)) return rc; }
        if (int rc = launch_head(r, st, &n_bchunks, !rows)) return rc;
        if (rows) {     // phase 1: the flat gradient and this rank's loss partials are complete in `grads` after this launch (the caller's all-reduce follows)
            if (int rc = launch_trunk_rows(r, st, n_bchunks, phase == 1)) return rc;
            folded = true;
            if (phase == 1) return CIRS_OK;
        } else {
            if (int rc = launch_trunk_bwd(r, st, n_bchunks)) return rc;
            if (phase == 1) {
                const DwJobs jobs = trunk_dw_jobs(r, n_pad_of(mb));
                hipLaunchKernelGGL(dw_multi_final, dim3(cdiv(jobs.total_out, 256)), dim3(256), 0, r.s, jobs, n_pad_of(mb) / kTileM, (const float*)r.v.dwp);
                CIRS_CHECK_LAUNCH("dw_multi_final");
                hipLaunchKernelGGL(loss_partials_kernel, dim3(1), dim3(256), 0, r.s, (int)mb, st.mb_norm, r.v, r.tail);
                CIRS_CHECK_LAUNCH("loss_partials_kernel");
                return CIRS_OK;
            }
        }
    }
    if (ch && ch->next_idx && phase == 2) {
        const PpoStep nxt{ch->next_idx, ch->next_mb, ch->next_idx_global ? ch->next_idx_global : ch->next_idx,
                          ch->next_idx_global ? ch->next_mb_global : ch->next_mb, nullptr, nullptr, (long)opt_step + 1};
        return launch_norm_adam(r, st, phase, folded, &nxt);
    }
    return launch_norm_adam(r, st, phase, folded, nullptr);
}

extern "C" int cirs_ppo_minibatch(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v,
                                  int64_t opt_step, const cirs_ppo_batch* batch, const int32_t* idx, int32_t mb,
                                  float* dobs_accum, int32_t n_env, float* loss_out, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    if (mb < 2) return cirs::fail(CIRS_E_INVALID, "minibatch needs >= 2 rows (unbiased std)");
    return ppo_minibatch_impl(cfg, params, grads, adam_m, adam_v, opt_step, batch, idx, mb, nullptr, mb, dobs_accum, n_env, loss_out,
                              workspace, workspace_bytes, 0, stream);
}

// Batch.split(size, merge_last=True) (tianshou/data/batch.py:734-744): minibatch k of n rows -> [begin, end)
static int ppo_slice_count(int n, int bs) {
    int k = 0;
    const bool merge_last = n % bs > 0;
    for (int s0 = 0; s0 < n; s0 += bs) { ++k; if (merge_last && s0 + 2 * bs >= n) break; }
    return k;
}
static void ppo_slice(int n, int bs, int k, int* b, int* e) {
    const bool merge_last = n % bs > 0;
    const int s0 = k * bs;
    *b = s0;
    *e = (merge_last && s0 + 2 * bs >= n) ? n : (s0 + bs < n ? s0 + bs : n);
}
extern "C" int cirs_ppo_handoff_status(int32_t* lost_out, int32_t reset, void* stream) {
    CIRS_REQUIRE(lost_out, "cirs_ppo_handoff_status: lost_out is null");
    hipLaunchKernelGGL(cirs::handoff_status_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)lost_out, (int)reset);
    CIRS_CHECK_LAUNCH("handoff_status_kernel");
    return CIRS_OK;
}

extern "C" int cirs_ppo_update_readback(const int32_t* lens, int32_t n_env, int32_t* lens_host_pinned, int32_t* lost_host_pinned, int32_t* offsets_out,
                                        int32_t* n_rows_out, void* stream) {
    CIRS_REQUIRE(lens && lens_host_pinned && n_env > 0 && (!offsets_out == !n_rows_out), "cirs_ppo_update_readback: bad arguments");
    if (offsets_out) {
        CIRS_REQUIRE(n_env <= (1 << 20), "cirs_ppo_update_readback: n_env");
        hipLaunchKernelGGL(cirs::update_readback_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lens, (int)n_env, lens_host_pinned, lost_host_pinned,
                           offsets_out, n_rows_out);
    } else
    hipLaunchKernelGGL(cirs::update_readback_kernel, dim3(cirs::cdiv(n_env, 256)), dim3(256), 0, (hipStream_t)stream, lens, (int)n_env, lens_host_pinned,
                       lost_host_pinned);
    CIRS_CHECK_LAUNCH("update_readback_kernel");
    return CIRS_OK;
}

extern "C" int32_t cirs_ppo_learn_steps(int32_t n_rows, int32_t batch_size, int32_t n_repeat) {
    if (n_rows < 1 || batch_size < 1 || n_repeat < 1) return 0;
    return n_repeat * ppo_slice_count(n_rows, batch_size);
}
extern "C" int cirs_ppo_learn(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v, int64_t opt_step,
                              const cirs_ppo_batch* batch, const int32_t* perms, int32_t n_rows, int32_t batch_size, int32_t n_repeat,
                              float* dobs_accum, int64_t dobs_floats, int32_t n_env, float* losses, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    using namespace cirs;
    if (int rc = validate_ppo(cfg)) return rc;
    CIRS_REQUIRE(params && grads && adam_m && adam_v && batch && perms && losses && workspace, "null argument");
    CIRS_REQUIRE(n_rows >= 2 && batch_size >= 2 && n_repeat >= 1, "need n_rows >= 2, batch_size >= 2, n_repeat >= 1");
    CIRS_REQUIRE(!dobs_accum || dobs_floats > 0, "dobs_floats must be the size of dobs_accum");
    const int n_sl = ppo_slice_count(n_rows, batch_size);
    int max_mb = 0;
    for (int k = 0; k < n_sl; ++k) { int b, e; ppo_slice(n_rows, batch_size, k, &b, &e); CIRS_REQUIRE(e - b >= 2, "a minibatch of one row has no unbiased std"); max_mb = e - b > max_mb ? e - b : max_mb; }
    CIRS_REQUIRE(workspace_bytes >= cirs_ppo_workspace_bytes(cfg, max_mb), "workspace too small");
    const PpoRun r = ppo_run(cfg, params, grads, adam_m, adam_v, batch, n_env, workspace, max_mb, (hipStream_t)stream);
    const char* pf_ = getenv("CIRS_PPO_LEARN_PREFETCH");      // =0: every step starts with its own trunk_adv_kernel launch (A/B runs, tests)
    const bool prefetch = !(pf_ && atoi(pf_) == 0) && !r.merge_launch;
    const int n_steps = n_repeat * n_sl;
    auto step_of = [&](int k) {
        const int rep = k / n_sl;
        int b, e; ppo_slice(n_rows, batch_size, k % n_sl, &b, &e);
        const int32_t* idx = perms + (size_t)rep * n_rows + b;
        return PpoStep{idx, e - b, idx, e - b, (dobs_accum && rep == n_repeat - 1) ? dobs_accum : nullptr, losses + 4 * (size_t)k, (long)(opt_step + k)};
    };
    bool have_head = false;     // the head of step k (trunk forward, statistics, planes) already ran inside step k - 1's Adam launch
    for (int k = 0; k < n_steps; ++k) {
        const PpoStep st = step_of(k);
        if (dobs_accum && k == (n_repeat - 1) * n_sl) {      // optim.zero_grad() at the top of the last repeat: only its d loss / d obs reaches the tracker
            if (hipMemsetAsync(dobs_accum, 0, sizeof(float) * (size_t)dobs_floats, r.s) != hipSuccess) return fail(CIRS_E_LAUNCH, "hipMemsetAsync(dobs_accum)");
        }
        int n_bchunks = 0;
        bool folded = false;
        const bool rows = rows_kernel_wanted(0, st.mb);
        if (!have_head) { if (int rc = launch_trunk_adv(r, st)) return rc; }
        if (int rc = launch_head(r, st, &n_bchunks, !rows)) return rc;
        if (rows) { if (int rc = launch_trunk_rows(r, st, n_bchunks, false)) return rc; folded = true; }
        else if (int rc = launch_trunk_bwd(r, st, n_bchunks)) return rc;
        const bool has_next = prefetch && k + 1 < n_steps;
        const PpoStep nxt = has_next ? step_of(k + 1) : PpoStep{};
        if (int rc = launch_norm_adam(r, st, 0, folded, has_next ? &nxt : nullptr)) return rc;
        have_head = has_next;
    }
    return CIRS_OK;
}

extern "C" int cirs_ppo_minibatch_dp(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v,
                                     int64_t opt_step, const cirs_ppo_batch* batch, const int32_t* idx_local, int32_t mb_local,
                                     const int32_t* idx_global, int32_t mb_global, float* dobs_accum, int32_t n_env,
                                     float* loss_out, void* workspace, int64_t workspace_bytes, int32_t phase, void* stream) {
    if (phase != 2 && !idx_global) return cirs::fail(CIRS_E_INVALID, "idx_global is null");
    return ppo_minibatch_impl(cfg, params, grads, adam_m, adam_v, opt_step, batch, idx_local, mb_local, idx_global, mb_global,
                              dobs_accum, n_env, loss_out, workspace, workspace_bytes, phase, stream);
}

extern "C" int cirs_ppo_minibatch_dp_chain(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v, int64_t opt_step,
                                           const cirs_ppo_batch* batch, const int32_t* idx_local, int32_t mb_local, const int32_t* idx_global,
                                           int32_t mb_global, float* dobs_accum, int32_t n_env, float* loss_out, void* workspace,
                                           int64_t workspace_bytes, int32_t phase, int32_t max_mb_local, int32_t head_done,
                                           const int32_t* next_idx_local, int32_t next_mb_local, const int32_t* next_idx_global,
                                           int32_t next_mb_global, void* stream) {
    if (phase != 1 && phase != 2) return cirs::fail(CIRS_E_INVALID, "phase must be 1 or 2");
    if (phase == 1 && !idx_global) return cirs::fail(CIRS_E_INVALID, "idx_global is null");
    if (next_idx_local && (!next_idx_global || next_mb_local < 1 || next_mb_global < 2)) return cirs::fail(CIRS_E_INVALID, "bad next minibatch");
    const DpChain ch{(int)max_mb_local, (int)head_done, next_idx_local, (int)next_mb_local, next_idx_global, (int)next_mb_global};
    return ppo_minibatch_impl(cfg, params, grads, adam_m, adam_v, opt_step, batch, idx_local, mb_local, idx_global, mb_global, dobs_accum, n_env,
                              loss_out, workspace, workspace_bytes, phase, stream, &ch);
}

// ---- tensor-parallel (item-sharded head) minibatch step ------------------------------------------------------------------------
extern "C" int64_t cirs_ppo_tp_exchange_floats(int32_t n_rows, int32_t world) {
    return (int64_t)cirs::n_pad_of(n_rows) * (cirs::kH + 1) + (int64_t)world * cirs::kWaSumBlocks;
}

extern "C" int cirs_ppo_minibatch_tp(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v, int64_t opt_step,
                                     const cirs_ppo_batch* batch, const int32_t* idx, int32_t mb, int32_t item_base, int32_t rank,
                                     int32_t world, float* stats4, const float* stats_all, float* red, float* dobs_accum, int32_t n_env,
                                     float* loss_out, void* workspace, int64_t workspace_bytes, int32_t phase, void* stream) {
    using namespace cirs;
    if (int rc = validate_ppo(cfg)) return rc;
    CIRS_REQUIRE(params && grads && adam_m && adam_v && batch && idx && workspace && red, "null argument");
    CIRS_REQUIRE(phase >= 1 && phase <= 3, "phase must be 1, 2 or 3");
    CIRS_REQUIRE(mb >= 2 && world >= 1 && rank >= 0 && rank < world && item_base >= 0, "bad minibatch / rank arguments");
    CIRS_REQUIRE(workspace_bytes >= cirs_ppo_workspace_bytes(cfg, mb), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int I = cfg->n_items, S = cfg->dim_state;      // I = items of THIS shard
    CIRS_REQUIRE(S <= 32, "dim_state > 32 is not supported by the trunk backward kernel");
    const int n_pad = n_pad_of(mb);
    const PpoLayout L = ppo_layout(I, S);
    MbView v = carve(workspace, n_pad, I, S);
    cirs_policy_cfg pcfg{I, S, kH};
    cirs_policy_weights w{params + L.w1, params + L.b1, params + L.w2, params + L.b2, params + L.wa, params + L.ba, params + L.wc, params + L.bc};
    const long seg = (long)I * kH + I;
    const int n_slabs = n_row_blocks_of(n_pad);
    const int n_item_tiles = cdiv(I, kTileN);
    float* red_dh2 = red;
    float* red_ent = red + (size_t)n_pad * kH;
    float* red_slots = red_ent + n_pad;
    float* tail = grads + L.total;
    if (phase == 1) {
        CIRS_REQUIRE(stats4, "stats4 is null");
        hipLaunchKernelGGL(trunk_adv_kernel, dim3(cdiv(n_pad, 4) + 1 + cdiv(I, kTileN)), dim3(256), 0, s, pcfg, w, (const float*)batch->obs, (long)S, n_pad,
                           idx, (int)mb, (const float*)batch->adv, idx, (int)mb, (int)cfg->norm_adv, v.red, (int)cdiv(n_pad, 4), v.wa_planes, *batch, (int)n_env,
                           TrunkRowOut{v.h2, v.value, v.h1, v.obs, nullptr, nullptr, nullptr, v.h2z, v.h2b});
        CIRS_CHECK_LAUNCH("trunk_adv_kernel");
        ActorPartialView pv = partial_view(v.head_ws, n_pad, I);
        const int tpc_s = head_tiles_per_chunk(n_item_tiles, n_slabs, 2);
        const int n_schunks = cdiv(n_item_tiles, tpc_s);
        hipLaunchKernelGGL(head_stats_kernel, dim3(n_schunks, n_slabs), dim3(256), 0, s, I, (int)mb, n_pad, tpc_s, (const uint4*)v.wa_planes, w.ba,
                           (const uint4*)v.h2z, pv, (const int32_t*)nullptr, (float*)nullptr);
        CIRS_CHECK_LAUNCH("head_stats_kernel");
        hipLaunchKernelGGL(head_tp_fold_kernel, dim3(cdiv(n_pad, 4)), dim3(256), 0, s, *batch, idx, (int)mb, n_pad, n_schunks, pv, (const float*)w.wa,
                           (const float*)w.ba, (int)item_base, I, (const float*)v.h2, stats4);
        CIRS_CHECK_LAUNCH("head_tp_fold_kernel");
        return CIRS_OK;
    }
    if (phase == 2) {
        CIRS_REQUIRE(stats_all, "stats_all is null");
        // stats_all [4][world][n_pad]: the triples of all shards in rank order are the chunk partials of the merge (fixed order: every
        // rank computes identical row statistics and coefficients)
        ActorPartialView pa;
        pa.m = const_cast<float*>(stats_all); pa.s = pa.m + (size_t)world * n_pad; pa.score = pa.s + (size_t)world * n_pad;
        pa.idx = nullptr; pa.z = nullptr;
        const float* za = pa.score + (size_t)world * n_pad;
        hipLaunchKernelGGL(head_stats_merge_kernel, dim3(cdiv(n_pad, 4)), dim3(256), 0, s, *cfg, *batch, idx, (int)mb, (int)mb, n_pad, (int)world,
                           (int)n_env, pa, (const float*)nullptr, (const float*)nullptr, v, za, (int)item_base);
        CIRS_CHECK_LAUNCH("head_stats_merge_kernel");
        const int tpc = head_tiles_per_chunk(n_item_tiles, n_slabs, 1);
        const int n_bchunks = cdiv(n_item_tiles, tpc);
        const HeadMergeArgs no_merge{};    // the row coefficients come from head_stats_merge_kernel above (statistics of every shard)
        if (cfg->ent_coef != 0.f) {
            hipLaunchKernelGGL((head_bwd_fused_kernel<true, false>), dim3((n_bchunks + 7) & ~7, n_slabs), dim3(kBwdWaves * 64), 0, s, I, mb, n_pad, tpc,
                               (const uint4*)v.wa_planes, w.ba, v, v.dwap, no_merge);
        } else {
            hipLaunchKernelGGL((head_bwd_fused_kernel<false, false>), dim3((n_bchunks + 7) & ~7, n_slabs), dim3(kBwdWaves * 64), 0, s, I, mb, n_pad, tpc,
                               (const uint4*)v.wa_planes, w.ba, v, v.dwap, no_merge);
        }
        CIRS_CHECK_LAUNCH("head_bwd_fused_kernel");
        CIRS_HIP(hipMemsetAsync(red_slots, 0, sizeof(float) * (size_t)world * kWaSumBlocks, s));
        hipLaunchKernelGGL(dh2_sum_kernel, dim3(n_pad * (kH / 4) / 64 + cdiv(n_pad, 64)), dim3(64), 0, s, (int)mb, n_pad, n_bchunks, v, red_dh2, red_ent);
        CIRS_CHECK_LAUNCH("dh2_sum_kernel");
        hipLaunchKernelGGL(wa_slab_sum_kernel, dim3(kWaSumBlocks), dim3(512), 0, s, grads, (long)L.wa, seg, (const float*)v.dwap,
                           (long)dwa_slab_stride(I), n_slabs, red_slots + (size_t)rank * kWaSumBlocks);
        CIRS_CHECK_LAUNCH("wa_slab_sum_kernel");
        return CIRS_OK;
    }
    // phase 3: `red` holds the sums over the ranks
    CIRS_REQUIRE(loss_out, "loss_out is null");
    hipLaunchKernelGGL(tp_post_kernel, dim3(cdiv(n_pad, 256)), dim3(256), 0, s, (int)mb, n_pad, (int)world, (const float*)red_ent,
                       (const float*)red_slots, v);
    CIRS_CHECK_LAUNCH("tp_post_kernel");
    DwJobs jobs;
    jobs.n = 3;
    jobs.j[0] = DwJob{v.dvalue, 1, v.h2, kH, 1, kH, grads + L.wc, grads + L.bc, 0, 0};
    jobs.j[1] = DwJob{v.da2, kH, v.h1, kH, kH, kH, grads + L.w2, grads + L.b2, 0, 0};
    jobs.j[2] = DwJob{v.da1, kH, v.obs, S, kH, S, grads + L.w1, grads + L.b1, 0, 0};
    const int n_dw_slabs = n_pad / kTileM;
    {
        int off = 0, out = 0;
        for (int q = 0; q < 3; ++q) {
            jobs.j[q].part_off = off;
            off += n_dw_slabs * jobs.j[q].O * (jobs.j[q].K + 1);
            out += jobs.j[q].O * (jobs.j[q].K + 1);
        }
        jobs.total_out = out;
    }
    MbView v3 = v;
    v3.dh2p = red_dh2;      // the trunk backward reads the summed d h2 (slab 0 position) from the exchange buffer
    hipLaunchKernelGGL(trunk_bwd_kernel, dim3(n_pad / kTileM), dim3(512), 0, s, (int)mb, n_pad, 1, S, w.w1, w.w2, w.wc, v3, dobs_accum, grads,
                       (long)L.wa, seg, (long)dwa_slab_stride(I), n_slabs, jobs, v.dwp);
    CIRS_CHECK_LAUNCH("trunk_bwd_kernel");
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(kNormBlocks), dim3(256), 0, s, grads, L.trunk, L.total, L.wa, seg, 1, jobs, n_dw_slabs,
                       (const float*)v.dwp, S, v.normp);
    CIRS_CHECK_LAUNCH("sumsq_partial_kernel");
    auto seg_of = [&](long step_before, int n_sub, int scale_pow) { return adam_seg_of(cfg, step_before, n_sub, scale_pow); };
    hipLaunchKernelGGL(adam2_kernel, dim3(cdiv(L.total, 256)), dim3(256), 0, s, params, grads, adam_m, adam_v, L.total, L.trunk,
                       seg_of(2 * opt_step, 2, 2), seg_of(opt_step, 1, 1), cfg->beta1, cfg->beta2, cfg->adam_eps, *cfg, v.normp, tail, v3, loss_out,
                       (int)mb, (int)mb);
    CIRS_CHECK_LAUNCH("adam2_kernel");
    return CIRS_OK;
}

// ---- sharded optimiser step of the data-parallel learner -------------------------------------------------------------------
extern "C" int32_t cirs_ppo_shard_stat_floats(void) { return cirs::kShardStatFloats; }

extern "C" int cirs_ppo_shard_norm(const cirs_ppo_cfg* cfg, const float* grads_shard, int64_t shard_begin, int64_t shard_len,
                                   float* stats_out, void* stream) {
    using namespace cirs;
    if (int rc = validate_ppo(cfg)) return rc;
    CIRS_REQUIRE(grads_shard && stats_out && shard_begin >= 0 && shard_len > 0, "bad shard arguments");
    const PpoLayout L = ppo_layout(cfg->n_items, cfg->dim_state);
    hipLaunchKernelGGL(shard_sumsq_kernel, dim3(kShardNormBlocks), dim3(256), 0, (hipStream_t)stream, grads_shard, (long)shard_begin,
                       (long)shard_len, L.trunk, L.total, stats_out);
    CIRS_CHECK_LAUNCH("shard_sumsq_kernel");
    return CIRS_OK;
}

extern "C" int cirs_ppo_shard_adam(const cirs_ppo_cfg* cfg, float* params_shard, const float* grads_shard, float* adam_m_shard,
                                   float* adam_v_shard, int64_t shard_begin, int64_t shard_len, int64_t opt_step,
                                   const float* stats_all, int32_t world, float* loss_out, void* stream) {
    using namespace cirs;
    if (int rc = validate_ppo(cfg)) return rc;
    CIRS_REQUIRE(params_shard && grads_shard && adam_m_shard && adam_v_shard && stats_all, "null argument");
    CIRS_REQUIRE(shard_begin >= 0 && shard_len > 0 && world >= 1, "bad shard arguments");
    const PpoLayout L = ppo_layout(cfg->n_items, cfg->dim_state);
    auto seg_of = [&](long step_before, int n_sub, int scale_pow) { return adam_seg_of(cfg, step_before, n_sub, scale_pow); };
    hipLaunchKernelGGL(shard_adam_kernel, dim3(cdiv(shard_len, 256)), dim3(256), 0, (hipStream_t)stream, params_shard, grads_shard,
                       adam_m_shard, adam_v_shard, (long)shard_begin, (long)shard_len, L.trunk, L.total, seg_of(2 * opt_step, 2, 2),
                       seg_of(opt_step, 1, 1), cfg->beta1, cfg->beta2, cfg->adam_eps, *cfg, stats_all, (int)world, loss_out);
    CIRS_CHECK_LAUNCH("shard_adam_kernel");
    return CIRS_OK;
}
