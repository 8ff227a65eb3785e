// deepfm.hip -- DeepFM user model (UserModel_Pairwise) for gfx950: generic pair scorer + full-catalogue sweep.
//
// deepfm_forward_kernel   one wavefront per (user,item) row, literal op order of the reference
//                         (core/user_model_pairwise.py:98-129): gather 6 embeddings -> concat(6E+1) -> DNN 64-64 -> last;
//                         linear logit; FM bi-interaction.  Used for recommend_k_item-style calls and as the check of
//                         the sweep.
// sweep (compute_normed_reward, kuaishouEnv.py:113-145; U x I pairs):
//   prep_items / prep_users  the first DNN layer is linear in the concatenated input, so W1 x = W1[:, :E] v_user +
//                            (W1[:, E:] [v_item, v_f0..3, dur] + b1): one 64-vector per user and per item, computed once.
//                            FM: 0.5*sum_e[(v_u + S_i)^2 - v_u^2 - Q_i] = v_u . S_i + 0.5(|S_i|^2 - Q_i) with
//                            S_i = sum of the item-side field vectors -> one E-long dot per pair + a per-item constant.
//   sweep_kernel             per wave: a tile of 32 items held in registers, loop over a chunk of users staged in LDS;
//                            per (user, 32 items): h1 = relu(A_u + A_i) formed directly in the MFMA B-operand layout,
//                            H2^T[64 x 32 pairs] = W2[64 x 64] * h1^T on the fp32 matrix cores (64 v_mfma_f32_32x32x2),
//                            a lane owns one pair so relu/last-dot/FM/bias are lane-local; min/max per wavefront slot + one reduce launch.
// Roofline: 2*(64*64+64+E) = 8.4 kFLOP executed per pair (E = 16) vs 20.7 kFLOP of the unfactored algorithm
// (SURVEY §8(d) F_sweep); MFMA-bound: 64 MFMA x 64 cycles per 32 pairs -> 19.2 G pairs/s at the fp32 MFMA peak.
// HBM traffic per launch ~ item/user rows once + 4 B per pair of output: far below the MFMA time.
#include "common.h"
#include "bf16x6.h"

namespace cirs {

constexpr int fH = 64;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------- generic pair scorer ----------------
__global__ __launch_bounds__(256) void deepfm_forward_kernel(cirs_deepfm_cfg cfg, cirs_deepfm_weights w,
                                                             const int64_t* __restrict__ uid, const int64_t* __restrict__ pid,
                                                             const int32_t* __restrict__ feats, const float* __restrict__ dur,
                                                             int n, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int E = cfg.emb_dim, K = 6 * E + 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= n) return;
    float* x = smem + (size_t)wv * (K + fH + 3);
    float* h1 = x + K + 1;
    const long u = uid[r], p = pid[r];
    int f[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = feats[(size_t)r * 4 + q];
    for (int k = lane; k < 6 * E; k += CIRS_WAVE) {
        const int fld = k / E, e = k % E;
        float v;
        if (fld == 0) v = w.emb_user[(size_t)u * E + e];
        else if (fld == 1) v = w.emb_item[(size_t)p * E + e];
        else v = w.emb_feat[(size_t)f[fld - 2] * E + e];
        x[k] = v;
    }
    const float d = dur[r];
    if (lane == 0) x[6 * E] = d;
    __builtin_amdgcn_wave_barrier();
    // FM: lanes over e
    float cross = 0.f;
    for (int e = lane; e < E; e += CIRS_WAVE) {
        float s = 0.f, q = 0.f;
        for (int fl = 0; fl < 6; ++fl) { const float v = x[fl * E + e]; s += v; q += v * v; }
        cross += s * s - q;
    }
    cross = wave_sum_f32(cross);
    float logit = w.lin_user[u] + w.lin_item[p];
#pragma unroll
    for (int q = 0; q < 4; ++q) logit += w.lin_feat[f[q]];
    logit += d * w.lin_dense[0];
    logit += 0.5f * cross;
    // DNN: lane o
    float acc = w.b1[lane];
    const float* w1r = w.w1 + (size_t)lane * K;
    for (int k = 0; k < K; ++k) acc = __builtin_fmaf(w1r[k], x[k], acc);
    h1[lane] = fmaxf(acc, 0.f);
    __builtin_amdgcn_wave_barrier();
    acc = w.b2[lane];
    const float* w2r = w.w2 + (size_t)lane * fH;
    for (int k = 0; k < fH; ++k) acc = __builtin_fmaf(w2r[k], h1[k], acc);
    const float dnn = wave_sum_f32(w.last[lane] * fmaxf(acc, 0.f));
    if (lane == 0) out[r] = logit + (dnn + w.out_bias[0]);
}

// ---------------- sweep: per-item / per-user precompute ----------------
// item side: AI[i][64], SI[i][E], CI[i] = lin_i + 0.5(|S|^2 - Q) + out_bias
__global__ __launch_bounds__(256) void prep_items_kernel(cirs_deepfm_cfg cfg, cirs_deepfm_weights w, const int64_t* __restrict__ item_ids,
                                                         const int32_t* __restrict__ item_feats, const float* __restrict__ item_dur,
                                                         int ni, float* __restrict__ AI, float* __restrict__ SI, float* __restrict__ CI) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int E = cfg.emb_dim, K = 6 * E + 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= ni) return;
    float* x = smem + (size_t)wv * (5 * E + 4);  // [v_item, v_f0..3]
    const long p = item_ids[i];
    int f[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = item_feats[(size_t)i * 4 + q];
    for (int k = lane; k < 5 * E; k += CIRS_WAVE) {
        const int fld = k / E, e = k % E;
        x[k] = fld == 0 ? w.emb_item[(size_t)p * E + e] : w.emb_feat[(size_t)f[fld - 1] * E + e];
    }
    __builtin_amdgcn_wave_barrier();
    const float d = item_dur[i];
    float acc = w.b1[lane];
    const float* w1r = w.w1 + (size_t)lane * K + E;  // columns E .. 6E (item fields), then 6E (duration)
    for (int k = 0; k < 5 * E; ++k) acc = __builtin_fmaf(w1r[k], x[k], acc);
    acc = __builtin_fmaf(w1r[5 * E], d, acc);
    AI[(size_t)i * fH + lane] = acc;
    float s2 = 0.f, q = 0.f;
    for (int e = lane; e < E; e += CIRS_WAVE) {
        float s = 0.f;
        for (int fl = 0; fl < 5; ++fl) { const float v = x[fl * E + e]; s += v; q += v * v; }
        SI[(size_t)i * E + e] = s;
        s2 += s * s;
    }
    s2 = wave_sum_f32(s2); q = wave_sum_f32(q);
    if (lane == 0) {
        float lin = w.lin_item[p];
#pragma unroll
        for (int k = 0; k < 4; ++k) lin += w.lin_feat[f[k]];
        lin += d * w.lin_dense[0];
        CI[i] = lin + 0.5f * (s2 - q) + w.out_bias[0];
    }
}
// user side: AU[u][64] = W1[:, :E] v_u, VU[u][E] = v_u, LU[u] = lin_user
__global__ __launch_bounds__(256) void prep_users_kernel(cirs_deepfm_cfg cfg, cirs_deepfm_weights w, const int64_t* __restrict__ user_ids,
                                                         int nu, float* __restrict__ AU, float* __restrict__ VU, float* __restrict__ LU) {
    const int E = cfg.emb_dim, K = 6 * E + 1;
    const int lane = threadIdx.x & 63;
    const int ui = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ui >= nu) return;
    const long u = user_ids[ui];
    const float* v = w.emb_user + (size_t)u * E;
    float acc = 0.f;
    const float* w1r = w.w1 + (size_t)lane * K;
    for (int k = 0; k < E; ++k) acc = __builtin_fmaf(w1r[k], v[k], acc);
    AU[(size_t)ui * fH + lane] = acc;
    if (lane < E) VU[(size_t)ui * E + lane] = v[lane];
    if (lane == 0) LU[ui] = w.lin_user[u];
}

// order-preserving float <-> uint key so min/max can use integer atomics for any sign mix
__device__ __forceinline__ unsigned int f32_key(float v) {
    const unsigned int b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

constexpr int kUserChunk = 64;

// grid = (ceil(n_item_tiles/4), ceil(nu/kUserChunk)); block = 4 waves, wave = one tile of 32 items
template <int E>
__global__ __launch_bounds__(256, 1) void sweep_kernel(const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ last,
                                                       const float* __restrict__ AI, const float* __restrict__ SI, const float* __restrict__ CI,
                                                       const float* __restrict__ AU, const float* __restrict__ VU, const float* __restrict__ LU,
                                                       int nu, int ni, float* __restrict__ pred, float* __restrict__ mmpart) {
    __shared__ float sAU[kUserChunk][fH];
    __shared__ float sVU[kUserChunk][E];
    __shared__ float sLU[kUserChunk];
    __shared__ float sB2[fH], sLast[fH];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int u0 = blockIdx.y * kUserChunk;
    const int nuc = min(kUserChunk, nu - u0);
    for (int i = threadIdx.x; i < nuc * fH; i += blockDim.x) sAU[i / fH][i % fH] = AU[(size_t)u0 * fH + i];
    for (int i = threadIdx.x; i < nuc * E; i += blockDim.x) sVU[i / E][i % E] = VU[(size_t)u0 * E + i];
    for (int i = threadIdx.x; i < nuc; i += blockDim.x) sLU[i] = LU[u0 + i];
    if (threadIdx.x < fH) { sB2[threadIdx.x] = b2[threadIdx.x]; sLast[threadIdx.x] = last[threadIdx.x]; }
    __syncthreads();
    const int tile0 = (blockIdx.x * 4 + wv) * 32;
    if (tile0 >= ni) {
        if (lane == 0) {
            const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv;
            mmpart[2 * slot] = INFINITY; mmpart[2 * slot + 1] = -INFINITY;
        }
        return;
    }
    const int item = tile0 + lo;
    const bool ok = item < ni;
    // A operand (persistent): W2 rows as bf16 planes (bf16x6.h: fp32 products on the bf16 matrix pipe); lane (out row = lo
    // [+32], hi), k-step s4: w2[out][16 s4 + 8 hi + j]
    Planes wa0[4], wa1[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const float4* a = reinterpret_cast<const float4*>(w2 + (size_t)lo * fH + 16 * s4 + 8 * hi);
        const float4* b = reinterpret_cast<const float4*>(w2 + (size_t)(32 + lo) * fH + 16 * s4 + 8 * hi);
        const float4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
        wa0[s4] = split8(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w);
        wa1[s4] = split8(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w);
    }
    // item side of this lane's pair column: ai[8 s4 + j] = feature 16 s4 + 8 hi + j
    float ai[32], si[E / 2];
#pragma unroll
    for (int q = 0; q < 32; ++q) ai[q] = ok ? AI[(size_t)item * fH + 16 * (q >> 3) + 8 * hi + (q & 7)] : 0.f;
#pragma unroll
    for (int q = 0; q < E / 2; ++q) si[q] = ok ? SI[(size_t)item * E + hi * (E / 2) + q] : 0.f;
    const float ci = ok ? CI[item] : 0.f;
    float vmin = INFINITY, vmax = -INFINITY;
    for (int uu = 0; uu < nuc; ++uu) {
        float h1[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) h1[q] = fmaxf(ai[q] + sAU[uu][16 * (q >> 3) + 8 * hi + (q & 7)], 0.f);
        f32x16 acc0, acc1;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int o = (s & 3) + 8 * (s >> 2) + 4 * hi;
            acc0[s] = sB2[o];
            acc1[s] = sB2[32 + o];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const Planes hb = split8(h1[8 * s4], h1[8 * s4 + 1], h1[8 * s4 + 2], h1[8 * s4 + 3], h1[8 * s4 + 4], h1[8 * s4 + 5],
                                     h1[8 * s4 + 6], h1[8 * s4 + 7]);
            mfma_bf16x6_pair_b(wa0[s4], wa1[s4], hb, acc0, acc1);
        }
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int o = (s & 3) + 8 * (s >> 2) + 4 * hi;
            part = __builtin_fmaf(sLast[o], fmaxf(acc0[s], 0.f), part);
            part = __builtin_fmaf(sLast[32 + o], fmaxf(acc1[s], 0.f), part);
        }
#pragma unroll
        for (int q = 0; q < E / 2; ++q) part = __builtin_fmaf(sVU[uu][hi * (E / 2) + q], si[q], part);  // FM cross term
        part += __shfl_xor(part, 32, CIRS_WAVE);
        const float y = part + ci + sLU[uu];
        if (ok && hi == 0) {
            if (pred) pred[(size_t)(u0 + uu) * ni + item] = y;
            vmin = fminf(vmin, y); vmax = fmaxf(vmax, y);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        vmin = fminf(vmin, __shfl_xor(vmin, off, CIRS_WAVE));
        vmax = fmaxf(vmax, __shfl_xor(vmax, off, CIRS_WAVE));
    }
    // one (min, max) slot per wavefront, reduced by minmax_end_kernel: no atomics (tens of thousands of device-scope
    // atomics on one address serialise at ~60 ns each and used to dominate the launch)
    if (lane == 0) {
        const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv;
        mmpart[2 * slot] = vmin;
        mmpart[2 * slot + 1] = vmax;
    }
}

// (min, max) over the per-wavefront slots, optionally merged with the running pair in mm (init == 0)
__global__ __launch_bounds__(1024) void minmax_end_kernel(const float* __restrict__ part, long n_slots, float* __restrict__ mm, int init) {
    __shared__ float s_lo[1024], s_hi[1024];
    const int tid = threadIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    for (long i = tid; i < n_slots; i += 1024) { lo = fminf(lo, part[2 * i]); hi = fmaxf(hi, part[2 * i + 1]); }
    s_lo[tid] = lo; s_hi[tid] = hi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) { s_lo[tid] = fminf(s_lo[tid], s_lo[tid + s]); s_hi[tid] = fmaxf(s_hi[tid], s_hi[tid + s]); }
        __syncthreads();
    }
    if (tid == 0) {
        mm[0] = init ? s_lo[0] : fminf(mm[0], s_lo[0]);
        mm[1] = init ? s_hi[0] : fmaxf(mm[1], s_hi[0]);
    }
}

__global__ __launch_bounds__(256) void normed_kernel(const float* __restrict__ pred, long n, const float* __restrict__ mm, double* __restrict__ out) {
    const double lo = (double)mm[0], hi = (double)mm[1];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = ((double)pred[i] - lo) / (hi - lo);
}

static int validate_deepfm(const cirs_deepfm_cfg* cfg, const cirs_deepfm_weights* w) {
    CIRS_REQUIRE(cfg && w, "deepfm cfg/weights null");
    if (cfg->hidden != fH) return fail(CIRS_E_UNSUPPORTED, "deepfm: hidden == 64 only");
    if (!(cfg->emb_dim == 8 || cfg->emb_dim == 16 || cfg->emb_dim == 32 || cfg->emb_dim == 64)) return fail(CIRS_E_UNSUPPORTED, "deepfm: emb_dim must be 8, 16, 32 or 64");
    CIRS_REQUIRE(w->emb_user && w->emb_item && w->emb_feat && w->lin_user && w->lin_item && w->lin_feat && w->lin_dense && w->w1 && w->b1 && w->w2 && w->b2 && w->last && w->out_bias, "deepfm weight pointer null");
    return CIRS_OK;
}

}  // namespace cirs

extern "C" int cirs_deepfm_forward(const cirs_deepfm_cfg* cfg, const cirs_deepfm_weights* w, const int64_t* uid,
                                   const int64_t* pid, const int32_t* feats, const float* dur, int32_t n, float* out, void* stream) {
    using namespace cirs;
    if (int rc = validate_deepfm(cfg, w)) return rc;
    CIRS_REQUIRE(uid && pid && feats && dur && out, "null argument");
    if (n <= 0) return CIRS_OK;
    const size_t shmem = 4 * sizeof(float) * (6 * cfg->emb_dim + 1 + fH + 3);
    hipLaunchKernelGGL(deepfm_forward_kernel, dim3(cdiv(n, 4)), dim3(256), shmem, (hipStream_t)stream, *cfg, *w, uid, pid, feats, dur, n, out);
    CIRS_CHECK_LAUNCH("deepfm_forward_kernel");
    return CIRS_OK;
}

extern "C" int64_t cirs_deepfm_sweep_workspace_bytes(const cirs_deepfm_cfg* cfg, int32_t n_users, int32_t n_items) {
    if (!cfg) return 0;
    const int64_t E = cfg->emb_dim;
    const int64_t n_wg = (int64_t)cirs::cdiv(cirs::cdiv(n_items, 32), 4) * cirs::cdiv(n_users, cirs::kUserChunk);
    return 4 * ((int64_t)n_items * (cirs::fH + E + 1) + (int64_t)n_users * (cirs::fH + E + 1) + 8 * n_wg + 64);
}

extern "C" int cirs_deepfm_sweep(const cirs_deepfm_cfg* cfg, const cirs_deepfm_weights* w, const int64_t* user_ids, int32_t nu,
                                 const int64_t* item_ids, const int32_t* item_feats, const float* item_dur, int32_t ni,
                                 float* pred_out, float* minmax, int32_t init_minmax, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
    using namespace cirs;
    if (int rc = validate_deepfm(cfg, w)) return rc;
    CIRS_REQUIRE(user_ids && item_ids && item_feats && item_dur && minmax && workspace, "null argument");
    CIRS_REQUIRE(nu > 0 && ni > 0, "empty sweep");
    CIRS_REQUIRE(workspace_bytes >= cirs_deepfm_sweep_workspace_bytes(cfg, nu, ni), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int E = cfg->emb_dim;
    float* AI = (float*)workspace;
    float* SI = AI + (size_t)ni * fH;
    float* CI = SI + (size_t)ni * E;
    float* AU = CI + ni;
    float* VU = AU + (size_t)nu * fH;
    float* LU = VU + (size_t)nu * E;
    float* mmpart = LU + nu + 2;
    hipLaunchKernelGGL(prep_items_kernel, dim3(cdiv(ni, 4)), dim3(256), 4 * sizeof(float) * (5 * E + 4), s, *cfg, *w, item_ids, item_feats,
                       item_dur, ni, AI, SI, CI);
    hipLaunchKernelGGL(prep_users_kernel, dim3(cdiv(nu, 4)), dim3(256), 0, s, *cfg, *w, user_ids, nu, AU, VU, LU);
    CIRS_CHECK_LAUNCH("deepfm prep");
    const dim3 grid(cdiv(cdiv(ni, 32), 4), cdiv(nu, kUserChunk));
#define SWEEP(EE) hipLaunchKernelGGL(sweep_kernel<EE>, grid, dim3(256), 0, s, w->w2, w->b2, w->last, AI, SI, CI, AU, VU, LU, nu, ni, pred_out, mmpart)
    switch (E) {
        case 8: SWEEP(8); break;
        case 16: SWEEP(16); break;
        case 32: SWEEP(32); break;
        default: SWEEP(64); break;
    }
#undef SWEEP
    hipLaunchKernelGGL(minmax_end_kernel, dim3(1), dim3(1024), 0, s, mmpart, (long)grid.x * grid.y * 4, minmax, init_minmax);
    CIRS_CHECK_LAUNCH("sweep_kernel");
    return CIRS_OK;
}

extern "C" int cirs_normed_reward(const float* pred, int64_t n, const float* minmax, double* normed_out, void* stream) {
    using namespace cirs;
    CIRS_REQUIRE(pred && minmax && normed_out && n > 0, "bad arguments");
    const int grid = cdiv(n, 256) < 4096 ? cdiv(n, 256) : 4096;
    hipLaunchKernelGGL(normed_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, (long)n, minmax, normed_out);
    CIRS_CHECK_LAUNCH("normed_kernel");
    return CIRS_OK;
}
