// tracker.hip -- incremental StateTrackerTransformer for gfx950 (reference: core/state_tracker.py:170-250).
//
// The reference recomputes a 2-layer causal TransformerEncoder over the whole episode prefix at every step and keeps
// only the last position.  With a causal mask and dropout off that equals a decode step with per-layer K/V caches,
// which is what this kernel does: per env and step it
//   1. builds the new input slot      x = ffn_user(Emb_user[u])                     (state_tracker.py:205-215)
//                                  or x = sigmoid(fnn_gate([r, a])) * a, a = Emb_item[item]   (:225-242)
//   2. h = x*sqrt(D) + pe[pos]                                                      (:180-181, PositionalEncoding)
//   3. for each post-norm encoder layer: q,k,v = in_proj(h); append k,v to the cache; causal attention of q over
//      cache[0..pos]; out_proj; LayerNorm; FF(relu); LayerNorm                       (torch.nn.TransformerEncoderLayer)
//   4. s = decoder(h)                                                                (:183-186)
// One 64-lane wavefront per env (4 envs per workgroup).  Lane o owns output feature o of every mat-vec; the input
// vector is broadcast from a per-wave LDS scratch; attention lanes stride over cached positions.  fp32 throughout.
//
// Bytes per env-step: 128 B embedding row + 2 layers x (pos x 256 B K/V reads + 256 B K/V writes) + 128 B x_hist
// + 80 B state; weights (~118 KB) are shared by all envs and stay in L2.  ~60 kFLOP per env-step: latency-bound.
#include <hip/hip_runtime.h>
#ifdef CIRS_TRK_PROF
// probe builds: sub-stage stamps inside the sampler's pick (slots 20..), see CIRS_STAMP below
namespace cirs { extern __device__ unsigned long long g_trk_prof[64]; }
#define CIRS_PICK_STAMP(K) do { if (blockIdx.x == 0 && threadIdx.x == 0) cirs::g_trk_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "internal.h"
#include "policy_kernels.h"
#include "env_kernels.h"

namespace cirs {

constexpr int kD = 32;    // dim_model
constexpr int kHid = 128; // d_hid

template <int K>
__device__ __forceinline__ float dot_row(const float* __restrict__ wrow, const float* xs, float acc) {
    const float4* w4 = reinterpret_cast<const float4*>(wrow);
#pragma unroll
    for (int k4 = 0; k4 < K / 4; ++k4) {
        const float4 w = w4[k4];
        acc = __builtin_fmaf(w.x, xs[4 * k4 + 0], acc);
        acc = __builtin_fmaf(w.y, xs[4 * k4 + 1], acc);
        acc = __builtin_fmaf(w.z, xs[4 * k4 + 2], acc);
        acc = __builtin_fmaf(w.w, xs[4 * k4 + 3], acc);
    }
    return acc;
}

// A weight row held in registers: issued as loads BEFORE the stage that consumes it, so the L2 latency of the next
// mat-vec overlaps the current one (one wavefront per SIMD here: nothing else hides it).  dot_pre uses the same fma order
// as dot_row.
template <int K>
struct RowRegs { float4 q[K / 4]; };
template <int K>
__device__ __forceinline__ RowRegs<K> load_row(const float* __restrict__ wrow) {
    RowRegs<K> r;
    const float4* w4 = reinterpret_cast<const float4*>(wrow);
#pragma unroll
    for (int k4 = 0; k4 < K / 4; ++k4) r.q[k4] = w4[k4];
    return r;
}
template <int K>
__device__ __forceinline__ float dot_pre(const RowRegs<K>& r, const float* xs, float acc) {
#pragma unroll
    for (int k4 = 0; k4 < K / 4; ++k4) {
        acc = __builtin_fmaf(r.q[k4].x, xs[4 * k4 + 0], acc);
        acc = __builtin_fmaf(r.q[k4].y, xs[4 * k4 + 1], acc);
        acc = __builtin_fmaf(r.q[k4].z, xs[4 * k4 + 2], acc);
        acc = __builtin_fmaf(r.q[k4].w, xs[4 * k4 + 3], acc);
    }
    return acc;
}

// LayerNorm over the 32 values held by lanes 0..31 (lanes >= 32 pass 0 and get garbage they never use); wo / bo = this lane's
// weight and bias, loaded by the caller well before (a load here would wait behind every weight row requested in between)
__device__ __forceinline__ float layer_norm32(float v, int lane, float wo, float bo) {
    // (round 6: the 32 values live in two rows of 16 lanes -> four DPP steps + two row picks per sum instead of a full-wave reduction; v_rsq_f32 (1 ulp)
    //  instead of a correctly rounded division by a square root: ~25 instructions less per LayerNorm on a wave whose instruction count is its time)
    const float mean = half_sum_f32_dpp(v) * (1.0f / kD);       // lanes >= 32 hold a duplicate of lanes 0..31 or garbage: only rows 0, 1 are picked
    const float dlt = v - mean;
    const float var = half_sum_f32_dpp(dlt * dlt) * (1.0f / kD);
    const float inv = __builtin_amdgcn_rsqf(var + 1e-5f);
    return dlt * inv * wo + bo;
}

// K consecutive inputs (from k0) of output row o of a weight matrix, into registers.  IMG: from the packed image (internal.h:
// [k/4][O][4] at float offset off: consecutive lanes -> consecutive 16 bytes); else from the row-major matrix rm (leading dim ld).
template <int K, bool IMG>
__device__ __forceinline__ RowRegs<K> wrow(const float* __restrict__ img, int off, int O, const float* __restrict__ rm, int ld, int o, int k0) {
    RowRegs<K> r;
    if (IMG) {
        const float4* p = reinterpret_cast<const float4*>(img + off) + (size_t)(k0 >> 2) * O + o;
#pragma unroll
        for (int k4 = 0; k4 < K / 4; ++k4) r.q[k4] = p[(size_t)k4 * O];
    } else {
        const float4* w4 = reinterpret_cast<const float4*>(rm + (size_t)o * ld + k0);
#pragma unroll
        for (int k4 = 0; k4 < K / 4; ++k4) r.q[k4] = w4[k4];
    }
    return r;
}

#ifdef CIRS_TRK_PROF
// stage timestamps of workgroup 0 / wave 0 (probe builds only: tools/probes/trk_prof.py)
__device__ unsigned long long g_trk_prof[64];
#define CIRS_STAMP(K) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_trk_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CIRS_STAMP(K) do { } while (0)
#endif

// ZS: the fused tail's pick reads the sampler's logit store (small env counts) -- a compile-time variant, policy_kernels.h: actor_pick_wave
template <int NHEAD, bool DROP, bool IMG, bool ZS>
__global__ __launch_bounds__(256) void tracker_step_kernel(cirs_tracker_cfg cfg, cirs_tracker_weights w,
                                                           cirs_tracker_state st, const int32_t* __restrict__ users,
                                                           const int64_t* __restrict__ items,
                                                           const double* __restrict__ rew,
                                                           const int32_t* __restrict__ env_ids,
                                                           const uint8_t* __restrict__ skip, int n,
                                                           float* __restrict__ state_out, long state_stride,
                                                           int lpad, TrunkFuse tf, TailFuse tl, const float* __restrict__ img,
                                                           TrkImg IL) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HD = kD / NHEAD;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nwv = blockDim.x >> 6;      // waves (= env rows) per workgroup: 4, or fewer when there are fewer rows than CUs (launch_tracker)
    const int j = blockIdx.x * nwv + wv;
    CIRS_STAMP(0);
    // ~1.1 KB of by-value arguments: five to eight dependent scalar-cache misses before the first vector load otherwise (common.h)
    kernarg_warm<sizeof(cirs_tracker_cfg) + sizeof(cirs_tracker_weights) + sizeof(cirs_tracker_state) + sizeof(TrunkFuse) + sizeof(TailFuse) +
                 sizeof(TrkImg) + 64>();   // <= the explicit argument bytes (the eight pointer / integer arguments add 72)
    CIRS_STAMP(1);
    if (j >= n) return;
// rows that do not step still owe the fused trunk its "skipped row" outputs
#define CIRS_TRUNK_ZERO()                                                  \
    do {                                                                   \
        if (tf.on) {                                                       \
            tf.h2[(size_t)j * kH + lane] = 0.f;                            \
            if (lane == 0 && tf.value) tf.value[j] = 0.f;                  \
        }                                                                  \
    } while (0)
    const bool is_init = users != nullptr;
    const int o32 = lane & (kD - 1);
    const int e = env_ids ? env_ids[j] : (tl.on ? tl.env_base + j : j);
    // the gate row of this lane's feature is needed right after the tail of the vector step: requested at kernel entry, it arrives
    // underneath the pick / env step -- but BEHIND the few loads the pick waits for first (memory returns in order).  Row-major
    // (no image) it is 33 floats with an odd row stride: one dword per lane and load, 64 cache lines per instruction.
    RowRegs<kD> gp;
    float g_r = 0.f, gbias = 0.f;
#define CIRS_GATE_PREFETCH()                                                                   \
    do {                                                                                       \
        if (!is_init) {                                                                        \
            if (IMG) {                                                                         \
                g_r = img[IL.gate_r + o32]; gbias = img[IL.gate_b + o32];                      \
                gp = wrow<kD, true>(img, IL.gate_p, kD, nullptr, 0, o32, 0);                   \
            } else {                                                                           \
                const float* gw = w.gate_w + (size_t)o32 * (kD + 1);   /* input order [r, a_0..a_31] */ \
                g_r = gw[0]; gbias = w.gate_b[o32];                                            \
                _Pragma("unroll") for (int k4 = 0; k4 < kD / 4; ++k4)                          \
                    gp.q[k4] = make_float4(gw[1 + 4 * k4], gw[2 + 4 * k4], gw[3 + 4 * k4], gw[4 + 4 * k4]); \
            }                                                                                  \
        }                                                                                      \
    } while (0)
    const int B = cfg.n_env, L = cfg.max_len;
    // per-wave scratch
    float* base = smem + (size_t)wv * (6 * kD + kHid + NHEAD * lpad);
    float* xs = base;            // [32] current vector (mat-vec input)
    float* qs = base + kD;       // [32] scaled query
    float* kcur = base + 2 * kD; // [32]
    float* vcur = base + 3 * kD; // [32]
    float* att = base + 4 * kD;  // [32]
    float* tmp = base + 5 * kD;  // [32]
    float* ffs = base + 6 * kD;  // [128]
    float* ps = ffs + kHid;      // [NHEAD][lpad] attention probabilities
    // ---- what depends on (env, position) only: requested as early as the position is known -------------------------------------
    //  * K/V cache rows of the earlier positions: the first batch of a layer (positions < 64 for K, < 32 for V) is requested one
    //    layer AHEAD -- layer 0 here, layer l + 1 right after layer l's attention -- and arrives under the stages in between
    //    (one register set, reused)
    //  * the positional-encoding element, layer 0's in_proj rows
    float4 kpre[kD / 4];
    float vpre[16];
    float pe_pre = 0.f;
    RowRegs<kD> pq, pv;
    float bq = 0.f, bv = 0.f;
    int pos = 0;
#define CIRS_KV_PREFETCH(LAYER)                                                                                  \
    do {                                                                                                         \
        const float* kc0_ = st.kcache + (((size_t)(LAYER) * B + e) * L) * kD;                                    \
        const float* vc0_ = st.vcache + (((size_t)(LAYER) * B + e) * L) * kD;                                    \
        if constexpr (NHEAD == 4) {   /* lane = (head lane >> 4, slot lane & 15): this head's 8 key components of positions slot and slot + 16 */ \
            /* UNCONDITIONAL loads of clamped positions (a conditional load is a branch + exec juggling per instruction: the 20 loads of this prefetch cost ~2 k cycles); */ \
            /* positions >= pos are never used (the scores select on jp <= pos, the current position comes from LDS) */ \
            const int pl_ = pos > 0 ? pos - 1 : 0;                                                               \
            const float4* kh_ = reinterpret_cast<const float4*>(kc0_) + (size_t)(2 * (lane >> 4)) * L;           \
            const int pa_ = min(lane & 15, pl_), pb_ = min((lane & 15) + 16, pl_);                               \
            kpre[0] = kh_[pa_]; kpre[1] = kh_[L + pa_]; kpre[2] = kh_[pb_]; kpre[3] = kh_[L + pb_];               \
        } else {                                                                                                 \
        const float4* k4_ = reinterpret_cast<const float4*>(kc0_) + lane;   /* K cache: [d/4][L][4], see below */ \
        _Pragma("unroll") for (int q4 = 0; q4 < kD / 4; ++q4)                                                    \
            kpre[q4] = lane < pos ? k4_[(size_t)q4 * L] : make_float4(0.f, 0.f, 0.f, 0.f);                       \
        }                                                                                                        \
        _Pragma("unroll") for (int u8 = 0; u8 < 16; ++u8) {                                                      \
            const int jp_ = (lane >> 5) + 2 * u8;                                                                \
            if constexpr (NHEAD == 4) vpre[u8] = vc0_[(size_t)min(jp_, pos > 0 ? pos - 1 : 0) * kD + o32];      /* unconditional, clamped: selected at use */ \
            else vpre[u8] = jp_ < pos ? vc0_[(size_t)jp_ * kD + o32] : 0.f;                                      \
        }                                                                                                        \
    } while (0)
#define CIRS_IN_PROJ_PREFETCH(LAYER)                                                                             \
    do {                                                                                                         \
        pq = wrow<kD, IMG>(img, IL.in_p[LAYER], 96, w.layer[LAYER].in_proj_w, kD, lane, 0);          /* rows 0..63 (q, k) */ \
        pv = wrow<kD, IMG>(img, IL.in_p[LAYER], 96, w.layer[LAYER].in_proj_w, kD, 2 * kD + o32, 0);  /* rows 64..95 (v) */   \
        bq = IMG ? img[IL.in_b[LAYER] + lane] : w.layer[LAYER].in_proj_b[lane];                                  \
        bv = IMG ? img[IL.in_b[LAYER] + 2 * kD + o32] : w.layer[LAYER].in_proj_b[2 * kD + o32];                  \
    } while (0)
    auto step_prefetch = [&]() {
        if (pos < L) {
            CIRS_KV_PREFETCH(0);
            pe_pre = w.pe[(size_t)pos * kD + o32];
        }
        CIRS_IN_PROJ_PREFETCH(0);
    };
    if (!tl.on) {
        CIRS_GATE_PREFETCH();
        if (!is_init) pos = st.len[e];
        step_prefetch();
    }
    // fused rollout: the tail of the vector step for this env row first (action, env step); its results stay in registers
    long it_f = -1;
    float r_f = 0.f, a_pre = 0.f;
    int fin_f = 0;
    if (tl.on) {
        int64_t act = -1;
        // the env's own state (user, turn, history entry of this lane, running reward) does not depend on the action: its loads
        // are issued BEFORE the merge of the sampler partials and complete underneath it
        const int et = tl.env_base + j;    // this row's env (env groups: row 0 of the launch is env env_base)
        CIRS_STAMP(32);
        const EnvPrefetch epf = env_prefetch(tl.cfg, tl.st, et, lane);
        CIRS_STAMP(33);
        // sampler inputs of this row: same round trip as the env state
        MergePre mpre{};
        PickPre ppre{};
        if (tl.pick_on) ppre = actor_pick_prefetch(j, lane, tl.pick.n_pad, tl.pick.n_chunks, tl.pick.lmass, tl.pick.h2);
        else mpre = actor_merge_prefetch(j, lane, tl.n_pad, tl.n_chunks, tl.pv);
        pos = st.len[e];
        CIRS_STAMP(34);
        CIRS_GATE_PREFETCH();
        CIRS_STAMP(35);
        if (epf.done) {  // finished env: the policy skipped it
            if (lane == 0) { tl.act_out[j] = -1; tl.logp_out[j] = 0.f; }
        } else if (tl.pick_on) {   // two-level draw: chunk, then item (per-wave LDS scratch: the feed-forward buffer, free here)
            float* stage = smem + (size_t)nwv * (6 * kD + kHid + NHEAD * lpad) + (size_t)wv * kPickStage;   // after the waves' scratch
            // the (env, position) prefetch goes out when the pick's own rows have arrived, under the item draw's second half
            const Cand r = actor_pick_wave<ZS>(tl.pick, j, et, lane, ffs, stage, &ppre, step_prefetch);
            act = r.bi == 0x7FFFFFFF ? -1 : (int64_t)r.bi;
            if (lane == 0) { tl.act_out[j] = act; tl.logp_out[j] = cand_logp(r); }
        } else {
            act = actor_merge_wave(j, lane, tl.n_pad, tl.n_chunks, tl.pv, tl.act_out, tl.logp_out, &mpre);
            step_prefetch();
        }
        if (tl.visited && act >= 0 && lane == 0) {
            const int words = (tl.cfg.n_items + 31) / 32;
            // this env's own row; a result-less atomic so that the wave does not wait for a load-modify-store round trip
            atomicOr(&tl.visited[(size_t)et * words + (act >> 5)], 1u << (act & 31));
        }
        CIRS_STAMP(2);
        // the tracker's input slot needs Emb_item[action]: requested now, it arrives underneath the env step
        if (act >= 0 && lane < kD) a_pre = w.emb_item[(size_t)act * kD + lane];
        EnvStepResult er;
        env_step_wave(tl.cfg, tl.tab, tl.st, et, j, act, lane, nullptr, tl.rew_out, tl.done_out, tl.ctr_out, nullptr, &er, &epf);
        const unsigned long long rb = __builtin_bit_cast(unsigned long long, er.reward);
        const unsigned rlo = __shfl((unsigned)rb, 0, CIRS_WAVE), rhi = __shfl((unsigned)(rb >> 32), 0, CIRS_WAVE);
        const double reward = __builtin_bit_cast(double, ((unsigned long long)rhi << 32) | rlo);
        fin_f = __shfl(er.done, 0, CIRS_WAVE);
        if (tl.force_length > 0 && act >= 0) {  // collector.py:253-258
            fin_f = tl.force_done;
            if (lane == 0) { tl.st.done[et] = (uint8_t)tl.force_done; tl.done_out[j] = (uint8_t)tl.force_done; }
        }
        it_f = act;
        r_f = (float)reward;
    }
    CIRS_STAMP(3);
    if (skip && skip[j]) { CIRS_TRUNK_ZERO(); return; }
    if (!is_init && (tl.on ? it_f : items[j]) < 0) { CIRS_TRUNK_ZERO(); return; }  // act = -1: env finished earlier in this rollout
    if (pos >= L) { CIRS_TRUNK_ZERO(); return; }  // history full: the caller never steps past max_turn

    // ---- 1. new input slot --------------------------------------------------------------------------------
    float x;
    if (is_init) {
        const int u = users[j];
        if (lane < kD) xs[lane] = w.emb_user[(size_t)u * kD + lane];
        __builtin_amdgcn_wave_barrier();
        x = dot_row<kD>(w.ffn_user_w + (size_t)o32 * kD, xs, w.ffn_user_b[o32]);
    } else {
        const long it = tl.on ? it_f : items[j];
        const float r = tl.on ? r_f : (float)rew[j];
        float a = 0.f;
        if (lane < kD) {
            a = tl.on ? a_pre : w.emb_item[(size_t)it * kD + lane];
            xs[lane] = a;
        }
        __builtin_amdgcn_wave_barrier();
        const float acc = dot_pre<kD>(gp, xs, __builtin_fmaf(g_r, r, gbias));   // input order [r, a_0..a_31]
        const float g = __builtin_amdgcn_rcpf(1.0f + __expf(-acc));      // hardware exp2 / rcp (1 ulp each)
        x = g * a;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < kD) st.x_hist[((size_t)e * L + pos) * kD + lane] = x;
    if (tl.on && tl.slot_only) {      // exact-redraw rollout: the state of every call comes from cirs_tracker_prefix_states over the slots; no cached decode
        if (lane == 0) st.len[e] = pos + 1;
        return;
    }
    // ---- 2. scale + positional encoding ---------------------------------------------------------------------
    float h = x * 5.656854249492381f + pe_pre;  // sqrt(32); pe_pre = pe[pos][o32]
    // dropout (DROP): counter-based masks keyed by (seed, global env, position, layer, site, element), csrc/rng.h
    const uint32_t d_thr = DROP ? dropout_threshold(cfg.dropout_p) : 0u;
    const float d_inv = DROP ? 1.0f / (1.0f - cfg.dropout_p) : 1.0f;
    const uint32_t d_env = (uint32_t)(cfg.drop_env_base + e);
#define CIRS_DROP(V, LAYER, SITE, ELEM) \
    (dropout_keep(cfg.dropout_seed, d_env, (uint32_t)pos, (uint32_t)(LAYER), (uint32_t)(SITE), (uint32_t)(ELEM), d_thr) ? (V) * d_inv : 0.f)
    if (DROP) h = CIRS_DROP(h, 0, CIRS_DROP_POS, o32);

    // ---- 3. encoder layers ----------------------------------------------------------------------------------
    CIRS_STAMP(4);
    for (int l = 0; l < cfg.nlayers; ++l) {
        const cirs_tracker_layer& ly = w.layer[l];
        // this layer's LayerNorm parameters: requested before any of the weight rows below, so they are there when needed
        const float n1w = IMG ? img[IL.ln[l] + o32] : ly.norm1_w[o32], n1b = IMG ? img[IL.ln[l] + kD + o32] : ly.norm1_b[o32];
        const float n2w = IMG ? img[IL.ln[l] + 2 * kD + o32] : ly.norm2_w[o32], n2b = IMG ? img[IL.ln[l] + 3 * kD + o32] : ly.norm2_b[o32];
        if (lane < kD) xs[lane] = h;
        __builtin_amdgcn_wave_barrier();
        // in_proj: 96 outputs; lanes 0..63 -> rows 0..63 (q,k), lanes 0..31 -> rows 64..95 (v)
        {
            const float r0 = dot_pre<kD>(pq, xs, bq);
            if (lane < kD) qs[lane] = r0 * (1.0f / sqrtf((float)HD));  // torch scales q before QK^T
            else kcur[lane - kD] = r0;
            if (lane < kD) vcur[lane] = dot_pre<kD>(pv, xs, bv);
        }
        // next stage's weights: out_proj row + biases, LayerNorm-1 parameters (consumed after the attention below)
        const RowRegs<kD> po = wrow<kD, IMG>(img, IL.out_p[l], kD, ly.out_proj_w, kD, o32, 0);
        const float bo = IMG ? img[IL.out_b[l] + o32] : ly.out_proj_b[o32];
        __builtin_amdgcn_wave_barrier();
        float* kc = st.kcache + (((size_t)l * B + e) * L) * kD;
        float* vc = st.vcache + (((size_t)l * B + e) * L) * kD;
        // K cache of (layer, env): [d/4][L][4] -- the scores' lane jp reads the float4 (d..d+3) of ITS position: consecutive lanes ->
        // consecutive 16 bytes (position-major [L][32] rows cost one cache line per lane and load)
        if (lane < kD) kc[((size_t)(lane >> 2) * L + pos) * 4 + (lane & 3)] = kcur[lane];
        else vc[(size_t)pos * kD + (lane - kD)] = vcur[lane - kD];
        CIRS_STAMP(5 + 6 * l);
        if constexpr (NHEAD == 4) {
            // ---- attention, 4 heads x 8 (round 6): lane = (head hq = lane >> 4, slot pq = lane & 15) owns the positions pq, pq + 16, pq + 32, ...  A head is a ROW of
            // 16 lanes, so its soft-max maximum and sum are four DPP steps each with no cross-row traffic (the lane = position layout reduced four heads one
            // after the other over the whole wave: ~90 instructions of DPP + v_readlane), a lane forms two 8-term dots instead of four per position, and the
            // probabilities leave NORMALISED (the V sum needs no per-head factor).  Positions < 32 -- every episode of the benchmark -- live in registers
            // (keys prefetched a layer ahead); longer episodes park their scores in the probability buffer.  Hardware exp2 / rcp.
            const int hq = lane >> 4, pq = lane & 15;
            const float4 q0 = *reinterpret_cast<const float4*>(qs + 8 * hq), q1 = *reinterpret_cast<const float4*>(qs + 8 * hq + 4);
            const float4 c0 = *reinterpret_cast<const float4*>(kcur + 8 * hq), c1 = *reinterpret_cast<const float4*>(kcur + 8 * hq + 4);   // this step's key
            auto dot8 = [&](const float4& a, const float4& b) {
                float sc = q0.x * a.x;
                sc = __builtin_fmaf(q0.y, a.y, sc); sc = __builtin_fmaf(q0.z, a.z, sc); sc = __builtin_fmaf(q0.w, a.w, sc);
                sc = __builtin_fmaf(q1.x, b.x, sc); sc = __builtin_fmaf(q1.y, b.y, sc); sc = __builtin_fmaf(q1.z, b.z, sc); sc = __builtin_fmaf(q1.w, b.w, sc);
                return sc;
            };
            const float4* kh = reinterpret_cast<const float4*>(kc) + (size_t)(2 * hq) * L;
            float sA, sB;
            {
                const bool curA = pq == pos, curB = pq + 16 == pos;
                const float4 a0 = curA ? c0 : kpre[0], a1 = curA ? c1 : kpre[1], b0 = curB ? c0 : kpre[2], b1 = curB ? c1 : kpre[3];
                sA = pq <= pos ? dot8(a0, a1) : -INFINITY;
                sB = pq + 16 <= pos ? dot8(b0, b1) : -INFINITY;
            }
            float mxh = fmaxf(sA, sB);
            for (int jp = pq + 32; jp <= pos; jp += 16) {      // (episodes longer than 31 steps only)
                const float4 a0 = jp == pos ? c0 : kh[jp], a1 = jp == pos ? c1 : kh[L + jp];
                const float sc = dot8(a0, a1);
                ps[hq * lpad + jp] = sc;
                mxh = fmaxf(mxh, sc);
            }
            mxh = row16_max_f32(mxh);
            const float eA = __expf(sA - mxh), eB = __expf(sB - mxh);      // exp(-inf) = 0 beyond the prefix
            float smh = eA + eB;
            for (int jp = pq + 32; jp <= pos; jp += 16) {
                const float ex = __expf(ps[hq * lpad + jp] - mxh);
                ps[hq * lpad + jp] = ex;
                smh += ex;
            }
            const float invh = __builtin_amdgcn_rcpf(row16_sum_f32(smh));
            // attention-probability dropout acts AFTER the softmax: the normaliser sums the unmasked terms
            if (pq <= pos) ps[hq * lpad + pq] = (DROP ? CIRS_DROP(eA, l, CIRS_DROP_ATTN, pq * NHEAD + hq) : eA) * invh;
            if (pq + 16 <= pos) ps[hq * lpad + pq + 16] = (DROP ? CIRS_DROP(eB, l, CIRS_DROP_ATTN, (pq + 16) * NHEAD + hq) : eB) * invh;
            for (int jp = pq + 32; jp <= pos; jp += 16) {
                const float ex = ps[hq * lpad + jp];
                ps[hq * lpad + jp] = (DROP ? CIRS_DROP(ex, l, CIRS_DROP_ATTN, jp * NHEAD + hq) : ex) * invh;
            }
            __builtin_amdgcn_wave_barrier();
            CIRS_STAMP(6 + 6 * l);
            // weighted sum of V: lane (half, d) walks the positions half, half + 2, ...; the first 32 positions from the prefetched registers, branch-free
            {
                const int half = lane >> 5, d = o32;
                const float* pr = ps + (d >> 3) * lpad;
                const float vnow = vcur[d];
                float acc = 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int jp = half + 2 * u;
                    const float pv_ = pr[jp <= pos ? jp : pos];            // (clamped: a valid address whatever max_len is)
                    acc = __builtin_fmaf(jp <= pos ? pv_ : 0.f, jp < pos ? vpre[u] : (jp == pos ? vnow : 0.f), acc);      // (vpre beyond the prefix: a clamped re-read, never used)
                }
                for (int j0 = 32 + half; j0 <= pos; j0 += 16) {            // (episodes longer than 31 steps only) 8 cached rows per batch in flight together
                    float v8[8];
#pragma unroll
                    for (int u8 = 0; u8 < 8; ++u8) {
                        const int jp = j0 + 2 * u8;
                        v8[u8] = jp < pos ? vc[(size_t)jp * kD + d] : 0.f;
                    }
#pragma unroll
                    for (int u8 = 0; u8 < 8; ++u8) {
                        const int jp = j0 + 2 * u8;
                        if (jp <= pos) acc = __builtin_fmaf(pr[jp], jp == pos ? vnow : v8[u8], acc);
                    }
                }
                acc += __shfl_xor(acc, 32, CIRS_WAVE);
                if (lane < kD) att[d] = acc;
            }
        } else {
        // scores: lanes stride over positions 0..pos
        float mx[NHEAD];
#pragma unroll
        for (int hh = 0; hh < NHEAD; ++hh) mx[hh] = -INFINITY;
        for (int jp = lane; jp <= pos; jp += CIRS_WAVE) {
            float kv[kD];
            if (jp == pos) {
#pragma unroll
                for (int d = 0; d < kD; ++d) kv[d] = kcur[d];
            } else {
                const float4* k4 = reinterpret_cast<const float4*>(kc) + jp;
#pragma unroll
                for (int q4 = 0; q4 < kD / 4; ++q4) {
                    const float4 t4 = jp == lane ? kpre[q4] : k4[(size_t)q4 * L];   // first batch (jp == lane): prefetched
                    kv[4 * q4] = t4.x; kv[4 * q4 + 1] = t4.y; kv[4 * q4 + 2] = t4.z; kv[4 * q4 + 3] = t4.w;
                }
            }
#pragma unroll
            for (int hh = 0; hh < NHEAD; ++hh) {
                float sc = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) sc = __builtin_fmaf(qs[hh * HD + d], kv[hh * HD + d], sc);
                ps[hh * lpad + jp] = sc;
                mx[hh] = fmaxf(mx[hh], sc);
            }
        }
        float sm[NHEAD];
#pragma unroll
        for (int hh = 0; hh < NHEAD; ++hh) {
            mx[hh] = wave_max_f32_dpp(mx[hh]);
            sm[hh] = 0.f;
        }
        for (int jp = lane; jp <= pos; jp += CIRS_WAVE) {
#pragma unroll
            for (int hh = 0; hh < NHEAD; ++hh) {
                const float pexp = expf(ps[hh * lpad + jp] - mx[hh]);
                // attention-probability dropout acts AFTER the softmax: the normaliser sums the unmasked terms
                ps[hh * lpad + jp] = DROP ? CIRS_DROP(pexp, l, CIRS_DROP_ATTN, jp * NHEAD + hh) : pexp;
                sm[hh] += pexp;
            }
        }
#pragma unroll
        for (int hh = 0; hh < NHEAD; ++hh) sm[hh] = 1.0f / wave_sum_f32_dpp(sm[hh]);
        __builtin_amdgcn_wave_barrier();
        CIRS_STAMP(6 + 6 * l);
        // weighted sum of V: lane (half, d): positions jp = half, half+2, ...
        {
            const int half = lane >> 5, d = o32, hh = d / HD;
            float acc = 0.f;
            // 8 cached rows per batch: the loads of a batch are in flight together (one L2 round trip per 16 positions instead of
            // one per position); the fma chain keeps its order (jp ascending per half-wave): same bits
            for (int j0 = half; j0 <= pos; j0 += 16) {
                float v8[8];
#pragma unroll
                for (int u8 = 0; u8 < 8; ++u8) {
                    const int jp = j0 + 2 * u8;
                    v8[u8] = j0 == half ? vpre[u8] : (j0 == half + 16 ? vpre[8 + u8] : (jp < pos ? vc[(size_t)jp * kD + d] : 0.f));   // first two batches: prefetched
                }
#pragma unroll
                for (int u8 = 0; u8 < 8; ++u8) {
                    const int jp = j0 + 2 * u8;
                    if (jp <= pos) acc = __builtin_fmaf(ps[hh * lpad + jp], jp == pos ? vcur[d] : v8[u8], acc);
                }
            }
            acc += __shfl_xor(acc, 32, CIRS_WAVE);
            float norm = sm[0];
#pragma unroll
            for (int q = 1; q < NHEAD; ++q) norm = hh == q ? sm[q] : norm;
            if (lane < kD) att[d] = acc * norm;
        }
        }
        // (unconditional, the layer clamped: hipcc follows a prefetch issued under a condition -- even a uniform one -- with its own wait where the paths merge)
        { const int ln_ = l + 1 < cfg.nlayers ? l + 1 : l; CIRS_KV_PREFETCH(ln_); }
        CIRS_STAMP(7 + 6 * l);
        // prefetch the feed-forward's first layer (two rows per lane) while out_proj + LayerNorm run
        const RowRegs<kD> pf0 = wrow<kD, IMG>(img, IL.l1_p[l], kHid, ly.lin1_w, kD, lane, 0);
        const RowRegs<kD> pf1 = wrow<kD, IMG>(img, IL.l1_p[l], kHid, ly.lin1_w, kD, 64 + lane, 0);
        const float bf0 = IMG ? img[IL.l1_b[l] + lane] : ly.lin1_b[lane], bf1 = IMG ? img[IL.l1_b[l] + 64 + lane] : ly.lin1_b[64 + lane];
        __builtin_amdgcn_wave_barrier();
        // out_proj + residual + LN1
        float sa = dot_pre<kD>(po, att, bo);
        if (DROP) sa = CIRS_DROP(sa, l, CIRS_DROP_RES1, o32);
        const float h1 = layer_norm32(h + sa, lane, n1w, n1b);
        __builtin_amdgcn_wave_barrier();
        CIRS_STAMP(8 + 6 * l);
        if (lane < kD) tmp[lane] = h1;
        // prefetch lin2's half row (64 inputs per half-wave)
        const int half2 = lane >> 5;
        const RowRegs<64> pl2 = wrow<64, IMG>(img, IL.l2_p[l], kD, ly.lin2_w, kHid, o32, half2 * 64);
        const float bl2 = half2 == 0 ? (IMG ? img[IL.l2_b[l] + o32] : ly.lin2_b[o32]) : 0.f;
        __builtin_amdgcn_wave_barrier();
        if (l == 0) CIRS_STAMP(27);
        // FF: 128 hidden = 2 rows per lane
        {
            float f0 = fmaxf(dot_pre<kD>(pf0, tmp, bf0), 0.f), f1 = fmaxf(dot_pre<kD>(pf1, tmp, bf1), 0.f);
            if (l == 0) CIRS_STAMP(28);
            if (DROP) { f0 = CIRS_DROP(f0, l, CIRS_DROP_FF, lane); f1 = CIRS_DROP(f1, l, CIRS_DROP_FF, 64 + lane); }
            ffs[lane] = f0;
            ffs[64 + lane] = f1;
        }
        // prefetch the next layer's in_proj rows (or nothing after the last layer)
        if (l + 1 < cfg.nlayers) CIRS_IN_PROJ_PREFETCH(l + 1);
        __builtin_amdgcn_wave_barrier();
        CIRS_STAMP(9 + 6 * l);
        // lin2: 32 outputs x 128 inputs, split k in two halves across the half-waves
        {
            float acc = dot_pre<64>(pl2, ffs + half2 * 64, bl2);
            if (l == 0) CIRS_STAMP(29);
            acc += __shfl_xor(acc, 32, CIRS_WAVE);
            if (DROP) acc = CIRS_DROP(acc, l, CIRS_DROP_RES2, o32);
            h = layer_norm32(h1 + acc, lane, n2w, n2b);
            if (l == 0) CIRS_STAMP(30);
        }
        __builtin_amdgcn_wave_barrier();
    }
    CIRS_STAMP(17);
    // ---- 4. decoder ------------------------------------------------------------------------------------------
    // the decoder's row and (image path) the fused trunk's weights in one batch (requesting them inside the layer loop would carry
    // ~130 registers around it)
    const int od = lane < cfg.dim_state ? lane : cfg.dim_state - 1;
    const RowRegs<kD> pdec = wrow<kD, IMG>(img, IL.dec_p, kD, w.dec_w, kD, IMG ? o32 : od, 0);
    const float bdec = IMG ? img[IL.dec_b + o32] : w.dec_b[od];
    RowRegs<kH> tw2;                  // trunk layer-2 row, layer-1 column, biases, critic weight of this lane
    float tw1[kD], tb1 = 0.f, tb2 = 0.f, twc = 0.f, tbc = 0.f;
    if (IMG && tf.on) {
        tb1 = img[IL.b1 + lane]; tb2 = img[IL.b2 + lane]; twc = img[IL.wc + lane]; tbc = img[IL.bc];
#pragma unroll
        for (int k = 0; k < kD; ++k) tw1[k] = k < tf.cfg.dim_state ? img[IL.w1_t + k * kH + lane] : 0.f;
        tw2 = wrow<kH, true>(img, IL.w2_p, kH, nullptr, 0, lane, 0);
    }
    if (lane < kD) xs[lane] = h;
    __builtin_amdgcn_wave_barrier();
    float sval = 0.f;
    if (lane < cfg.dim_state) {
        sval = dot_pre<kD>(pdec, xs, bdec);
        state_out[(size_t)j * state_stride + lane] = sval;
    }
    if (lane == 0) st.len[e] = pos + 1;
    CIRS_STAMP(18);
    if (tf.on) {  // policy trunk of the next vector step on this state
        if (tl.on ? fin_f != 0 : (tf.skip && tf.skip[j])) { CIRS_TRUNK_ZERO(); return; }
        float* txs = ffs;        // [64] input, then h2 (critic)
        float* ths = ffs + 64;   // [64] h1
        __builtin_amdgcn_wave_barrier();
        if (lane < cfg.dim_state) txs[lane] = sval;
        if (IMG) {   // trunk_compute (policy_kernels.h) on the prefetched image rows: the same fma chains
            const int S = tf.cfg.dim_state;
            __builtin_amdgcn_wave_barrier();
            float acc = tb1;
#pragma unroll
            for (int k = 0; k < kD; ++k)
                if (k < S) acc = __builtin_fmaf(tw1[k], txs[k], acc);
            ths[lane] = fmaxf(acc, 0.f);
            __builtin_amdgcn_wave_barrier();
            const float h2v = fmaxf(dot_pre<kH>(tw2, ths, tb2), 0.f);
            tf.h2[(size_t)j * kH + lane] = h2v;
            __builtin_amdgcn_wave_barrier();
            txs[lane] = h2v;
            ths[lane] = twc;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0 && tf.value) {  // critic: sequential chain (bit-reproducible), 64 fma
                float v = tbc;
#pragma unroll
                for (int k = 0; k < kH; ++k) v = __builtin_fmaf(ths[k], txs[k], v);
                tf.value[j] = v;
            }
        } else {
            trunk_compute(tf.cfg, tf.w, txs, ths, lane, j, tf.h2, tf.value, nullptr);
        }
    }
    CIRS_STAMP(19);
#undef CIRS_TRUNK_ZERO
#undef CIRS_KV_PREFETCH
#undef CIRS_DROP
}

// ---- packed weight image (internal.h: TrkImg) -------------------------------------------------------------------------------
// What a collect sets up before its first vector step, as ONE launch (three independent jobs on disjoint workgroups): the step kernel's weight image
// (the first n_img workgroups), the fp16 planes of the actor head for the chunk-mass kernels (wa_rplanes_kernel's workgroup per item tile; n_rp of them,
// 0: none) and env.reset (the rest; 0: none).  Each job is the stand-alone kernel's code on the same data: same bits, two launches fewer per collect.
struct SetupExtra {
    int n_img, n_rp, n_reset;
    const float* wa; int n_items; uint4* planes;
    cirs_env_cfg ecfg; cirs_env_state est; const int32_t* users; int n_env; int64_t* obs_scratch;
};
__global__ __launch_bounds__(256) void pack_tracker_image_kernel(cirs_tracker_cfg cfg, cirs_tracker_weights w, cirs_policy_weights pol, int S,
                                                                 TrkImg L, float* __restrict__ img, SetupExtra x) {
    if ((int)blockIdx.x >= x.n_img) {
        const int b = (int)blockIdx.x - x.n_img;
        if (b < x.n_rp) {      // wa_rplanes_kernel (policy_kernels.h), tile b
            const int tid = threadIdx.x, item = b * kTileN + (tid >> 3), col = 8 * (tid & 7);
            typedef float rp_v4 __attribute__((ext_vector_type(4)));
            rp_v4 a = {0.f, 0.f, 0.f, 0.f}, bb = a;
            if (item < x.n_items) {
                const rp_v4* src = reinterpret_cast<const rp_v4*>(x.wa + (size_t)item * kH + col);
                a = src[0]; bb = src[1];
            }
            a *= kMassScWa; bb *= kMassScWa;
            const Planes2 pl = split8h(a.x, a.y, a.z, a.w, bb.x, bb.y, bb.z, bb.w);
            uint4* out = x.planes + (size_t)b * kMassTileU4;
            out[tid] = __builtin_bit_cast(uint4, pl.h); out[256 + tid] = __builtin_bit_cast(uint4, pl.l);
        } else {
            env_reset_body(x.ecfg, x.est, x.users, nullptr, x.n_env, x.obs_scratch, (long)(b - x.n_rp) * 256 + threadIdx.x, (long)x.n_reset * 256);
        }
        return;
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = x.n_img * blockDim.x;
    // The ~40 regions of the image are dealt to the threads one after the other (region r starts at the thread where region r - 1 ended, modulo the thread
    // count): with every region starting at thread 0 the first 32 threads walked ALL of them -- ~40 dependent load -> store round trips, 9 us of a 13 us launch.
    int rot = 0;
    auto first = [&](int n) { int i = t - rot; if (i < 0) i += nt; rot += n; if (rot >= nt) rot -= nt * (rot / nt); return i; };
    // img[dst + ((k / 4) * O + o) * 4 + k % 4] = src[o * ld + k0 + k]  (rows >= o_src: zero)
    auto pk = [&](int dst, const float* src, int O, int o_src, int K, int ld, int k0) {
        for (int i = first(O * K); i < O * K; i += nt) {
            const int o = i / K, k = i % K;
            img[dst + ((k >> 2) * O + o) * 4 + (k & 3)] = o < o_src ? src[(size_t)o * ld + k0 + k] : 0.f;
        }
    };
    auto cp = [&](int dst, const float* src, int n) { for (int i = first(n); i < n; i += nt) img[dst + i] = src[i]; };
    for (int i = first(kD); i < kD; i += nt) img[L.gate_r + i] = w.gate_w[(size_t)i * (kD + 1)];   // input order [r, a_0..a_31]
    pk(L.gate_p, w.gate_w, kD, kD, kD, kD + 1, 1); cp(L.gate_b, w.gate_b, kD);
    for (int l = 0; l < cfg.nlayers; ++l) {
        const cirs_tracker_layer& y = w.layer[l];
        pk(L.in_p[l], y.in_proj_w, 96, 96, kD, kD, 0); cp(L.in_b[l], y.in_proj_b, 96);
        pk(L.out_p[l], y.out_proj_w, kD, kD, kD, kD, 0); cp(L.out_b[l], y.out_proj_b, kD);
        pk(L.l1_p[l], y.lin1_w, kHid, kHid, kD, kD, 0); cp(L.l1_b[l], y.lin1_b, kHid);
        pk(L.l2_p[l], y.lin2_w, kD, kD, kHid, kHid, 0); cp(L.l2_b[l], y.lin2_b, kD);
        cp(L.ln[l], y.norm1_w, kD); cp(L.ln[l] + kD, y.norm1_b, kD); cp(L.ln[l] + 2 * kD, y.norm2_w, kD); cp(L.ln[l] + 3 * kD, y.norm2_b, kD);
    }
    pk(L.dec_p, w.dec_w, kD, cfg.dim_state, kD, kD, 0);
    for (int i = first(kD); i < kD; i += nt) img[L.dec_b + i] = i < cfg.dim_state ? w.dec_b[i] : 0.f;
    if (pol.w1) {
        for (int i = first(S * kH); i < S * kH; i += nt) { const int o = i / S, k = i % S; img[L.w1_t + k * kH + o] = pol.w1[i]; }
        cp(L.b1, pol.b1, kH);
        pk(L.w2_p, pol.w2, kH, kH, kH, kH, 0); cp(L.b2, pol.b2, kH);
        cp(L.wc, pol.wc, kH);
        if (t == 0) img[L.bc] = pol.bc[0];
    }
}

int pack_tracker_image(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_policy_weights* pol, int S, float* img,
                       hipStream_t s, const float* wa, int n_items, uint4* planes, const cirs_env_cfg* ecfg, const cirs_env_state* est,
                       const int32_t* users, int n_env, int64_t* obs_scratch) {
    const TrkImg L = trk_img_layout(cfg->nlayers, S);
    if ((int64_t)L.total * 4 > kTrkImgBytes) return fail(CIRS_E_UNSUPPORTED, "tracker weight image exceeds its reserved workspace");
    cirs_policy_weights pw{};
    if (pol) pw = *pol;
    SetupExtra x{};
    x.n_img = 192;      // (~48 k image floats: one or two per thread; 64 workgroups took 9 us, the launch's longest role)
    if (planes) { x.n_rp = n_chunks_of(n_items) * kTilesPerChunk; x.wa = wa; x.n_items = n_items; x.planes = planes; }
    if (ecfg) {
        CIRS_REQUIRE(est && users && n_env > 0, "collect setup: env reset arguments");
        const long total = (long)n_env * ecfg->max_turn;
        x.n_reset = (int)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
        x.ecfg = *ecfg; x.est = *est; x.users = users; x.n_env = n_env; x.obs_scratch = obs_scratch;
    }
    hipLaunchKernelGGL(pack_tracker_image_kernel, dim3(x.n_img + x.n_rp + x.n_reset), dim3(256), 0, s, *cfg, *w, pw, S, L, img, x);
    CIRS_CHECK_LAUNCH("pack_tracker_image_kernel");
    return CIRS_OK;
}

static int validate_tracker(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st) {
    CIRS_REQUIRE(cfg && w && st, "tracker cfg/weights/state null");
    if (cfg->dim_model != kD || cfg->d_hid != kHid) return fail(CIRS_E_UNSUPPORTED, "this build supports dim_model == 32 and d_hid == 128");
    if (!(cfg->nhead == 1 || cfg->nhead == 2 || cfg->nhead == 4 || cfg->nhead == 8)) return fail(CIRS_E_UNSUPPORTED, "nhead must be 1, 2, 4 or 8");
    CIRS_REQUIRE(cfg->nlayers >= 1 && cfg->nlayers <= CIRS_MAX_TRACKER_LAYERS, "nlayers out of range");
    CIRS_REQUIRE(cfg->dim_state >= 1 && cfg->dim_state <= 32, "dim_state must be in 1..32");
    CIRS_REQUIRE(cfg->max_len >= 1 && cfg->max_len <= 4096, "max_len out of range");
    CIRS_REQUIRE(cfg->n_env >= 1, "n_env must be positive");
    CIRS_REQUIRE(w->emb_user && w->emb_item && w->ffn_user_w && w->ffn_user_b && w->gate_w && w->gate_b && w->pe && w->dec_w && w->dec_b, "tracker weight pointer null");
    for (int l = 0; l < cfg->nlayers; ++l) {
        const cirs_tracker_layer& y = w->layer[l];
        CIRS_REQUIRE(y.in_proj_w && y.in_proj_b && y.out_proj_w && y.out_proj_b && y.lin1_w && y.lin1_b && y.lin2_w && y.lin2_b && y.norm1_w && y.norm1_b && y.norm2_w && y.norm2_b, "tracker layer weight pointer null");
    }
    CIRS_REQUIRE(st->x_hist && st->kcache && st->vcache && st->len, "tracker state pointer null");
    return CIRS_OK;
}

static int launch_tracker(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, cirs_tracker_state* st,
                          const int32_t* users, const int64_t* items, const double* rew, const int32_t* env_ids,
                          const uint8_t* skip, int n, float* state_out, long state_stride, hipStream_t s,
                          const TrunkFuse* fuse = nullptr, const TailFuse* tail = nullptr, const float* img = nullptr) {
    TailFuse tl{};
    if (tail) tl = *tail;
    TrunkFuse tf{};
    if (fuse) {
        tf = *fuse;
        if (tf.on && (tf.cfg.hidden != kH || tf.cfg.dim_state != cfg->dim_state || !tf.h2)) return fail(CIRS_E_INVALID, "fused trunk: bad policy configuration");
    }
    const int lpad = (cfg->max_len + 3) & ~3;
    // + the row stage of the two-level sampler's item draw when the tail of the vector step is fused in (69.6 KB per workgroup)
    // Few rows (the 64-env shape): one wavefront per workgroup and an LDS request that keeps workgroups on DIFFERENT CUs -- four env rows on one CU share its
    // ~11 B/cycle load path (the pick's rows, the K/V caches, the weight image) while 240 CUs idle.  Same per-wavefront code: same bits.
    int wpw = 4;
    {
        const int cus = device_cu_count();
        const char* ev = getenv("CIRS_STEP_WAVES");
        if (ev && (atoi(ev) == 1 || atoi(ev) == 2 || atoi(ev) == 4)) wpw = atoi(ev);
        else if (n <= cus) wpw = 1;
        else if (n <= 2 * cus) wpw = 2;
    }
    size_t shmem = wpw * sizeof(float) * (6 * kD + kHid + (size_t)cfg->nhead * lpad) + (tl.on && tl.pick_on ? wpw * sizeof(float) * kPickStage : 0);
    if (wpw < 4 && shmem <= 81 * 1024) shmem = 81 * 1024;      // (more than half a CU's 160 KB: one workgroup per CU)
    if (shmem > 160 * 1024) return fail(CIRS_E_UNSUPPORTED, "max_len too large for the LDS score buffer");
    const dim3 grid(cdiv(n, wpw)), block(64 * wpw);
    const bool drop = cfg->dropout_p > 0.f;
    const TrkImg IL = img ? trk_img_layout(cfg->nlayers, tf.on ? tf.cfg.dim_state : cfg->dim_state) : TrkImg{};
    if (drop && !(cfg->dropout_p < 1.f)) return fail(CIRS_E_INVALID, "dropout_p must be in [0, 1)");
#define CIRS_TRK_LAUNCH(NH, DR, IM, ZS)                                                                               \
    do {                                                                                                              \
        if (shmem > 48 * 1024) {   /* more dynamic LDS than the default window: opt in once per instantiation */        \
            static bool optin = false;                                                                                \
            if (!optin) {                                                                                             \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tracker_step_kernel<NH, DR, IM, ZS>),          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)        \
                    return fail(CIRS_E_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");            \
                optin = true;                                                                                         \
            }                                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((tracker_step_kernel<NH, DR, IM, ZS>), grid, block, shmem, s, *cfg, *w, *st, users, items, rew, env_ids, skip, \
                           n, state_out, state_stride, lpad, tf, tl, img, IL);                                        \
    } while (0)
    // the logit-store variant exists for the image path only (the fused rollout always packs the image)
    const bool zs = tl.on && tl.pick_on && tl.pick.zstore != nullptr;
    if (zs && !img) return fail(CIRS_E_INVALID, "the sampler's logit store needs the packed weight image");
#define CIRS_TRK(NH)                                                                                                  \
    do {                                                                                                              \
        if (drop) { if (zs) CIRS_TRK_LAUNCH(NH, true, true, true); else if (img) CIRS_TRK_LAUNCH(NH, true, true, false); else CIRS_TRK_LAUNCH(NH, true, false, false); }    \
        else { if (zs) CIRS_TRK_LAUNCH(NH, false, true, true); else if (img) CIRS_TRK_LAUNCH(NH, false, true, false); else CIRS_TRK_LAUNCH(NH, false, false, false); }      \
    } while (0)
    switch (cfg->nhead) {
        case 1: CIRS_TRK(1); break;
        case 2: CIRS_TRK(2); break;
        case 4: CIRS_TRK(4); break;
        default: CIRS_TRK(8); break;
    }
#undef CIRS_TRK
#undef CIRS_TRK_LAUNCH
    CIRS_CHECK_LAUNCH("tracker_step_kernel");
    return CIRS_OK;
}

int tracker_step_internal(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, cirs_tracker_state* st, const int32_t* users,
                          const int64_t* items, const double* rew, const int32_t* env_ids, const uint8_t* skip, int n, float* state_out,
                          long state_stride, const TrunkFuse* tf, hipStream_t s, const TailFuse* tail, const float* img) {
    return launch_tracker(cfg, w, st, users, items, rew, env_ids, skip, n, state_out, state_stride, s, tf, tail, img);
}

}  // namespace cirs

#ifdef CIRS_TRK_PROF
extern "C" int cirs_debug_trk_prof(unsigned long long* out_host64) {
    return hipMemcpyFromSymbol(out_host64, HIP_SYMBOL(cirs::g_trk_prof), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int cirs_tracker_init(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, cirs_tracker_state* st,
                                 const int32_t* users, const int32_t* env_ids, int32_t n, float* state_out,
                                 int64_t state_stride, void* stream) {
    using namespace cirs;
    if (int rc = validate_tracker(cfg, w, st)) return rc;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(users && state_out, "users/state_out null");
    CIRS_REQUIRE(state_stride >= cfg->dim_state, "state_stride < dim_state");
    if (n <= 0) return CIRS_OK;
    return launch_tracker(cfg, w, st, users, nullptr, nullptr, env_ids, nullptr, n, state_out, (long)state_stride, (hipStream_t)stream);
}

extern "C" int cirs_tracker_step(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, cirs_tracker_state* st,
                                 const int64_t* items, const double* rew, const int32_t* env_ids, const uint8_t* skip,
                                 int32_t n, float* state_out, int64_t state_stride, void* stream) {
    using namespace cirs;
    if (int rc = validate_tracker(cfg, w, st)) return rc;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(items && rew && state_out, "items/rew/state_out null");
    CIRS_REQUIRE(state_stride >= cfg->dim_state, "state_stride < dim_state");
    if (n <= 0) return CIRS_OK;
    return launch_tracker(cfg, w, st, nullptr, items, rew, env_ids, skip, n, state_out, (long)state_stride, (hipStream_t)stream);
}
