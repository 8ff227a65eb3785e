// deepfm_train.hip -- one optimiser step of the pairwise DeepFM user model:
//   UserModel_Pairwise.get_loss (reference core/user_model_pairwise.py:134-151), loss_kuaishou_pairwise
//   (CIRS-UserModel-kuaishou.py:262-278), get_regularization_loss (core/user_model.py:401-417), total_loss.backward() and
//   optim.step() (Adam) of fit_data's inner loop (core/user_model.py:150-170).
//
//   train_rows_kernel   one wavefront per sample: DeepFM forward of the positive and the negative pair (activations in LDS),
//                       the sample's loss terms and d loss / d y, then the backward of both pairs down to the embedding rows:
//                       per-row DNN pre-activation gradients / inputs go to global memory for the weight-gradient GEMMs,
//                       per-row embedding contributions (FM + DNN input gradient + linear term + alpha/beta) to the
//                       contribution tables.
//   dw_gemm (small_gemm.h)  dW1|db1, dW2|db2, dlast|dout_bias, dlin_dense: fp32 MFMA row-slab partials, fixed-order sums.
//   scatter             stable radix sort of (key, row) + ordered segment sums (no float atomics), one pass per key space
//                       (user rows: [d emb_user | d lin_user | d alpha], item rows: [d emb_item | d lin_item | d beta],
//                       feature rows: [d emb_feat | d lin_feat]; padding row 0 of emb_feat gets no gradient).
//   adam_l2_kernel      g += 2 * l2_c * p (the regulariser is dense: every row of every table decays), Adam, and the
//                       regulariser's value as per-workgroup partials.
// All reductions have a fixed order: two runs give identical bits.
#include <hipcub/hipcub.hpp>
#include "small_gemm.h"

namespace cirs {

constexpr int tH = 64;

struct TrainLayout {  // offsets (floats) into the flat parameter / gradient / moment buffers
    long emb_user, emb_item, emb_feat, lin_user, lin_item, lin_feat, lin_dense, w1, b1, w2, b2, last, out_bias, alpha_u, beta_i,
        lm_user, lm_item, lm_feat, lm_dense, total;
};
__host__ __device__ inline TrainLayout train_layout(const cirs_deepfm_cfg& c) {
    const long U = c.n_user_vocab, I = c.n_item_vocab, F = c.n_feat_vocab, E = c.emb_dim, K = 6 * E + 1;
    TrainLayout L;
    long o = 0;
    L.emb_user = o; o += U * E; L.emb_item = o; o += I * E; L.emb_feat = o; o += F * E;
    L.lin_user = o; o += U; L.lin_item = o; o += I; L.lin_feat = o; o += F; L.lin_dense = o; o += 1;
    L.w1 = o; o += tH * K; L.b1 = o; o += tH; L.w2 = o; o += tH * tH; L.b2 = o; o += tH; L.last = o; o += tH; L.out_bias = o; o += 1;
    L.alpha_u = o; o += U; L.beta_i = o; o += I;
    L.lm_user = o; o += U; L.lm_item = o; o += I; L.lm_feat = o; o += F; L.lm_dense = o; o += 1;
    L.total = o;
    return L;
}

struct TrainRows {  // per-pair-row outputs of train_rows_kernel, R = 2n rows (positives first)
    float *X, *H1, *H2, *DA1, *DA2, *DY, *DUR;  // [R,K] [R,64] [R,64] [R,64] [R,64] [R] [R]
    float *CU, *CI, *CF;                        // contributions [R,E+2] [R,E+2] [4R,E+1]
    int32_t *KU, *KI, *KF;                      // keys [R] [R] [4R]
    float* LP;                                  // [n,4] per-sample loss terms {sq err, bpr, (alpha-1)^2, (beta-1)^2}
};

// forward of one (user, item) pair by one wavefront; x / S / a1 / a2 stay in LDS for the backward
__device__ __forceinline__ float pair_forward(const float* __restrict__ P, const TrainLayout& L, int E, int K, long u, long p,
                                              const int32_t* f4, float dur, int lane, float* x, float* S, float* a1, float* a2) {
    for (int k = lane; k < 6 * E; k += CIRS_WAVE) {
        const int fld = k / E, e = k % E;
        float v;
        if (fld == 0) v = P[L.emb_user + u * E + e];
        else if (fld == 1) v = P[L.emb_item + p * E + e];
        else v = P[L.emb_feat + (long)f4[fld - 2] * E + e];
        x[k] = v;
    }
    if (lane == 0) x[6 * E] = dur;
    __builtin_amdgcn_wave_barrier();
    float cross = 0.f;
    for (int e = lane; e < E; e += CIRS_WAVE) {
        float s = 0.f, q = 0.f;
        for (int fl = 0; fl < 6; ++fl) { const float v = x[fl * E + e]; s += v; q += v * v; }
        S[e] = s;
        cross += s * s - q;
    }
    cross = wave_sum_f32(cross);
    float logit = P[L.lin_user + u] + P[L.lin_item + p];
#pragma unroll
    for (int q = 0; q < 4; ++q) logit += P[L.lin_feat + f4[q]];
    logit += dur * P[L.lin_dense];
    logit += 0.5f * cross;
    float acc = P[L.b1 + lane];
    const float* w1r = P + L.w1 + (size_t)lane * K;
    for (int k = 0; k < K; ++k) acc = __builtin_fmaf(w1r[k], x[k], acc);
    a1[lane] = acc;
    __builtin_amdgcn_wave_barrier();
    acc = P[L.b2 + lane];
    const float* w2r = P + L.w2 + (size_t)lane * tH;
    for (int k = 0; k < tH; ++k) acc = __builtin_fmaf(w2r[k], fmaxf(a1[k], 0.f), acc);
    a2[lane] = acc;
    const float dnn = wave_sum_f32(P[L.last + lane] * fmaxf(acc, 0.f));
    __builtin_amdgcn_wave_barrier();
    return logit + (dnn + P[L.out_bias]);
}

// backward of one pair row r given dy; writes the row's GEMM operands and embedding contributions
__device__ __forceinline__ void pair_backward(const float* __restrict__ P, const TrainLayout& L, int E, int K, long u, long p,
                                              const int32_t* f4, float dur, float dy, float dalpha, float dbeta, int lane, int r,
                                              const float* x, const float* S, const float* a1, const float* a2, float* t64,
                                              float* dxs, const TrainRows& o) {
    // DNN: da2 = dy * last * relu'(a2); dh1 = W2^T da2; da1 = dh1 * relu'(a1); dx = W1^T da1
    const float da2 = a2[lane] > 0.f ? dy * P[L.last + lane] : 0.f;
    o.DA2[(size_t)r * tH + lane] = da2;
    o.H2[(size_t)r * tH + lane] = fmaxf(a2[lane], 0.f);
    o.H1[(size_t)r * tH + lane] = fmaxf(a1[lane], 0.f);
    t64[lane] = da2;
    __builtin_amdgcn_wave_barrier();
    float dh1 = 0.f;
    for (int q = 0; q < tH; ++q) dh1 = __builtin_fmaf(P[L.w2 + (size_t)q * tH + lane], t64[q], dh1);
    const float da1 = a1[lane] > 0.f ? dh1 : 0.f;
    o.DA1[(size_t)r * tH + lane] = da1;
    __builtin_amdgcn_wave_barrier();
    t64[lane] = da1;
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < K; k += CIRS_WAVE) {
        o.X[(size_t)r * K + k] = x[k];
        if (k < 6 * E) {
            float dx = 0.f;
            for (int q = 0; q < tH; ++q) dx = __builtin_fmaf(P[L.w1 + (size_t)q * K + k], t64[q], dx);
            // FM: d/dv_f,e of 0.5 * sum_e (S_e^2 - Q_e) = S_e - v_f,e
            dxs[k] = __builtin_fmaf(dy, S[k % E] - x[k], dx);
        }
    }
    if (lane == 0) { o.DY[r] = dy; o.DUR[r] = dur; o.KU[r] = (int32_t)u; o.KI[r] = (int32_t)p; }
    __builtin_amdgcn_wave_barrier();
    const int WU = E + 2, WF = E + 1;
    for (int e = lane; e < WU; e += CIRS_WAVE) {
        o.CU[(size_t)r * WU + e] = e < E ? dxs[e] : (e == E ? dy : dalpha);
        o.CI[(size_t)r * WU + e] = e < E ? dxs[E + e] : (e == E ? dy : dbeta);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int fid = f4[q];
        for (int e = lane; e < WF; e += CIRS_WAVE)  // padding_idx = 0: the embedding row gets no gradient, the 1-d weight does
            o.CF[((size_t)r * 4 + q) * WF + e] = e < E ? (fid == 0 ? 0.f : dxs[(2 + q) * E + e]) : dy;
        if (lane == 0) o.KF[(size_t)r * 4 + q] = fid;
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void train_rows_kernel(cirs_deepfm_cfg cfg, const float* __restrict__ P, const int64_t* __restrict__ uid_pos,
                                                         const int64_t* __restrict__ pid_pos, const int32_t* __restrict__ feats_pos,
                                                         const float* __restrict__ dur_pos, const int64_t* __restrict__ uid_neg,
                                                         const int64_t* __restrict__ pid_neg, const int32_t* __restrict__ feats_neg,
                                                         const float* __restrict__ dur_neg, const float* __restrict__ y,
                                                         const float* __restrict__ exposure, int n, int use_ab, float lambda_ab, TrainRows o) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int E = cfg.emb_dim, K = 6 * E + 1;
    const TrainLayout L = train_layout(cfg);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= n) return;
    const int per_pair = K + 1 + E + 2 * tH;
    float* base = smem + (size_t)wv * (2 * per_pair + tH + 6 * E + 8);
    float *xp = base, *Sp = xp + K + 1, *a1p = Sp + E, *a2p = a1p + tH;
    float *xn = base + per_pair, *Sn = xn + K + 1, *a1n = Sn + E, *a2n = a1n + tH;
    float* t64 = base + 2 * per_pair;
    float* dxs = t64 + tH;
    const long up = uid_pos[i], pp = pid_pos[i], un = uid_neg[i], pn = pid_neg[i];
    int32_t fp[4], fn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { fp[q] = feats_pos[(size_t)i * 4 + q]; fn[q] = feats_neg[(size_t)i * 4 + q]; }
    const float dp = dur_pos[i], dn = dur_neg[i];
    const float yp = pair_forward(P, L, E, K, up, pp, fp, dp, lane, xp, Sp, a1p, a2p);
    const float yn = pair_forward(P, L, E, K, un, pn, fn, dn, lane, xn, Sn, a1n, a2n);
    // ---- loss_kuaishou_pairwise on this sample; every mean is over the n samples of the batch ---------------------
    const float inv_n = 1.0f / (float)n;
    const float ex = exposure[i];
    float alpha = 1.f, beta = 1.f;
    if (use_ab) { alpha = P[L.alpha_u + up]; beta = P[L.beta_i + pp]; }
    const float ex_new = use_ab ? ex * alpha * beta : ex;
    const float inv1 = 1.0f / (1.0f + ex_new);
    const float y_exp = inv1 * yp;
    const float err = y_exp - y[i];
    const float sg = 1.0f / (1.0f + expf(-(yp - yn)));   // sigmoid(y_pos - y_neg)
    const float dyp = 2.0f * inv_n * err * inv1 - inv_n * (1.0f - sg);
    const float dyn = inv_n * (1.0f - sg);
    float dalpha = 0.f, dbeta = 0.f;
    if (use_ab) {
        const float dex = 2.0f * inv_n * err * (-yp * inv1 * inv1);   // d loss_y / d exposure_new
        dalpha = dex * ex * beta + lambda_ab * 2.0f * inv_n * (alpha - 1.0f);
        dbeta = dex * ex * alpha + lambda_ab * 2.0f * inv_n * (beta - 1.0f);
    }
    if (lane == 0) {
        float* lp = o.LP + (size_t)i * 4;
        lp[0] = err * err;
        lp[1] = -logf(sg);
        lp[2] = use_ab ? (alpha - 1.0f) * (alpha - 1.0f) : 0.f;
        lp[3] = use_ab ? (beta - 1.0f) * (beta - 1.0f) : 0.f;
    }
    pair_backward(P, L, E, K, up, pp, fp, dp, dyp, dalpha, dbeta, lane, i, xp, Sp, a1p, a2p, t64, dxs, o);
    pair_backward(P, L, E, K, un, pn, fn, dn, dyn, 0.f, 0.f, lane, n + i, xn, Sn, a1n, a2n, t64, dxs, o);
}

// batch loss terms: fixed-order sums of the per-sample terms -> {loss, loss_y, bpr, loss_ab}
__global__ __launch_bounds__(256) void train_loss_kernel(const float* __restrict__ LP, int n, float lambda_ab, float* __restrict__ loss_out) {
    __shared__ float sh[4][256];
    const int tid = threadIdx.x;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < n; i += 256)
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += LP[(size_t)i * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; ++q) sh[q][tid] = a[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int q = 0; q < 4; ++q) sh[q][tid] += sh[q][tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        const float inv = 1.0f / (float)n;
        const float ly = sh[0][0] * inv, bpr = sh[1][0] * inv, lab = sh[2][0] * inv + sh[3][0] * inv;
        loss_out[0] = ly + bpr + lambda_ab * lab;
        loss_out[1] = ly; loss_out[2] = bpr; loss_out[3] = lab;
    }
}

// ---- sorted scatter: contributions [R, W] keyed by row id -> up to three destination tables -----------------------
__global__ __launch_bounds__(256) void train_keys_kernel(const int32_t* __restrict__ keys, int R, int n_table, uint32_t* __restrict__ k_out,
                                                         int32_t* __restrict__ rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int32_t k = keys[r];
    k_out[r] = (k < 0 || k >= n_table) ? (uint32_t)n_table : (uint32_t)k;
    rows[r] = r;
}

struct ScatterDst { float* p[3]; int w[3]; };  // column ranges of the contribution row -> table [n_table, w] each

__global__ __launch_bounds__(256) void train_segment_sum_kernel(const uint32_t* __restrict__ ks, const int32_t* __restrict__ rs,
                                                                const float* __restrict__ contrib, int R, int W, int n_table, ScatterDst dst) {
    const int l = threadIdx.x & 31;
    const int pth = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (pth >= R) return;
    const uint32_t key = ks[pth];
    if (key >= (uint32_t)n_table || (pth > 0 && ks[pth - 1] == key)) return;  // not a segment head
    // segment length once (the 32 lanes scan cooperatively), then every lane streams its columns with 16 loads in flight
    int len = 0;
    for (int base = pth; base < R; base += 32) {
        const bool same = base + l < R && ks[base + l] == key;
        const unsigned long long m = __ballot(same) >> ((threadIdx.x & 32) ? 32 : 0) & 0xFFFFFFFFull;
        const int run = m == 0xFFFFFFFFull ? 32 : __builtin_ctzll(~m);
        len += run;
        if (run < 32) break;
    }
    for (int d = l; d < W; d += 32) {
        float acc = 0.f;
        for (int q0 = 0; q0 < len; q0 += 16) {
            float t16[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t16[u] = (q0 + u < len) ? contrib[(size_t)rs[pth + q0 + u] * W + d] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += t16[u];      // rows ascend inside a key: fixed order
        }
        int c = d;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (dst.p[t] && c < dst.w[t]) { dst.p[t][(size_t)key * dst.w[t] + c] = acc; break; }
            c -= dst.w[t];
        }
    }
}

static size_t train_sort_bytes(long R) { return (size_t)R * 64 + (1u << 20); }

static int train_scatter(const int32_t* keys, const float* contrib, int R, int W, int n_table, const ScatterDst& dst, void* scratch,
                         size_t scratch_bytes, hipStream_t s) {
    uint32_t* k_in = (uint32_t*)scratch;
    uint32_t* k_out = k_in + R;
    int32_t* r_in = (int32_t*)(k_out + R);
    int32_t* r_out = r_in + R;
    char* temp = (char*)(((uintptr_t)(r_out + R) + 255) & ~(uintptr_t)255);
    const size_t avail = scratch_bytes - (size_t)(temp - (char*)scratch);
    int end_bit = 1;
    while ((1u << end_bit) <= (uint32_t)n_table && end_bit < 32) ++end_bit;
    size_t need = 0;
    CIRS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k_in, k_out, r_in, r_out, R, 0, end_bit, s));
    CIRS_REQUIRE(need <= avail, "deepfm train: sort scratch too small");
    hipLaunchKernelGGL(train_keys_kernel, dim3(cdiv(R, 256)), dim3(256), 0, s, keys, R, n_table, k_in, r_in);
    CIRS_HIP(hipcub::DeviceRadixSort::SortPairs(temp, need, k_in, k_out, r_in, r_out, R, 0, end_bit, s));
    hipLaunchKernelGGL(train_segment_sum_kernel, dim3(cdiv(R, 8)), dim3(256), 0, s, k_out, r_out, contrib, R, W, n_table, dst);
    CIRS_CHECK_LAUNCH("train_segment_sum_kernel");
    return CIRS_OK;
}

// ---- regulariser + Adam over the whole flat buffer ----------------------------------------------------------------
struct L2Segs { long end[6]; float c[6]; int n; };  // element i belongs to the first segment with i < end

constexpr int kRegBlocks = 1024;
__global__ __launch_bounds__(256) void adam_l2_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                      long n, L2Segs segs, float beta1, float beta2, float eps, float step_size, float bc2s,
                                                      float* __restrict__ reg_partial) {
    __shared__ float sh[256];
    const int tid = threadIdx.x;
    float reg = 0.f;
    for (long i = blockIdx.x * 256L + tid; i < n; i += (long)kRegBlocks * 256) {
        float c = segs.c[segs.n - 1];
#pragma unroll
        for (int q = 5; q >= 0; --q)
            if (q < segs.n && i < segs.end[q]) c = segs.c[q];
        const float pi = p[i];
        reg = __builtin_fmaf(c * pi, pi, reg);
        const float gi = __builtin_fmaf(2.0f * c, pi, g[i]);   // d/dp of c * p^2 joins the data gradient
        g[i] = gi;
        const float mi = m[i] + (1.0f - beta1) * (gi - m[i]);
        const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2s + eps));
    }
    sh[tid] = reg;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) reg_partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void reg_final_kernel(const float* __restrict__ part, float* __restrict__ loss_out) {
    __shared__ float sh[256];
    const int tid = threadIdx.x;
    float t = 0.f;
    for (int q = tid; q < kRegBlocks; q += 256) t += part[q];
    sh[tid] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) loss_out[4] = sh[0];
}

static size_t train_ws_floats(const cirs_deepfm_cfg* cfg, long n) {
    const long E = cfg->emb_dim, K = 6 * E + 1, R = 2 * n;
    size_t f = 0;
    f += (size_t)R * K + 4 * (size_t)R * tH + 2 * (size_t)R;          // X, H1, H2, DA1, DA2, DY, DUR
    f += 2 * (size_t)R * (E + 2) + 4 * (size_t)R * (E + 1);           // CU, CI, CF
    f += 2 * (size_t)R + 4 * (size_t)R;                               // keys
    f += 4 * (size_t)n + 64;                                          // LP
    f += dwg_partial_floats(R, tH, (int)K) + 64;                      // dW slab partials (largest job: 64 x K)
    f += kRegBlocks + 64;
    f += train_sort_bytes(4 * R) / 4 + 64;
    return f + 64 * 16;
}

}  // namespace cirs

extern "C" int64_t cirs_deepfm_train_param_count(const cirs_deepfm_cfg* cfg) {
    if (!cfg) return 0;
    return cirs::train_layout(*cfg).total;
}

extern "C" int64_t cirs_deepfm_train_workspace_bytes(const cirs_deepfm_cfg* cfg, int32_t n) {
    if (!cfg || n <= 0) return 0;
    return (int64_t)cirs::train_ws_floats(cfg, n) * 4;
}

extern "C" int cirs_deepfm_train_step(const cirs_deepfm_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v,
                                      int64_t step_before, const int64_t* uid_pos, const int64_t* pid_pos, const int32_t* feats_pos,
                                      const float* dur_pos, const int64_t* uid_neg, const int64_t* pid_neg, const int32_t* feats_neg,
                                      const float* dur_neg, const float* y, const float* exposure, int32_t n, int32_t use_ab,
                                      float lambda_ab, float l2_embedding, float l2_linear, float l2_all, float lr, float beta1,
                                      float beta2, float eps, float* loss_out, void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace cirs;
    CIRS_REQUIRE(cfg && params && grads && adam_m && adam_v && loss_out && workspace, "null argument");
    if (cfg->hidden != tH) return fail(CIRS_E_UNSUPPORTED, "deepfm train: hidden == 64 only");
    CIRS_REQUIRE(cfg->emb_dim >= 1 && cfg->emb_dim <= 64, "emb_dim out of range");
    CIRS_REQUIRE(n >= 1, "empty batch");
    CIRS_REQUIRE(uid_pos && pid_pos && feats_pos && dur_pos && uid_neg && pid_neg && feats_neg && dur_neg && y && exposure, "null batch column");
    CIRS_REQUIRE(workspace_bytes >= cirs_deepfm_train_workspace_bytes(cfg, n), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int E = cfg->emb_dim, K = 6 * E + 1, R = 2 * n;
    const TrainLayout L = train_layout(*cfg);
    float* p = (float*)workspace;
    auto take = [&](size_t cnt) { float* r = p; p += (cnt + 3) & ~(size_t)3; return r; };
    TrainRows o;
    o.X = take((size_t)R * K); o.H1 = take((size_t)R * tH); o.H2 = take((size_t)R * tH); o.DA1 = take((size_t)R * tH); o.DA2 = take((size_t)R * tH);
    o.DY = take(R); o.DUR = take(R);
    o.CU = take((size_t)R * (E + 2)); o.CI = take((size_t)R * (E + 2)); o.CF = take((size_t)4 * R * (E + 1));
    o.KU = (int32_t*)take(R); o.KI = (int32_t*)take(R); o.KF = (int32_t*)take((size_t)4 * R);
    o.LP = take((size_t)4 * n + 8);
    float* partial = take(dwg_partial_floats(R, tH, K) + 64);
    float* regp = take(kRegBlocks + 8);
    void* sort_ws = (void*)take(train_sort_bytes(4L * R) / 4 + 64);
    // the data gradient is written sparsely (touched table rows, dense layers): start from zero
    CIRS_HIP(hipMemsetAsync(grads, 0, sizeof(float) * (size_t)L.total, s));
    const size_t shmem = sizeof(float) * 4 * (2 * (size_t)(K + 1 + E + 2 * tH) + tH + 6 * E + 8);
    hipLaunchKernelGGL(train_rows_kernel, dim3(cdiv(n, 4)), dim3(256), shmem, s, *cfg, (const float*)params, uid_pos, pid_pos, feats_pos, dur_pos,
                       uid_neg, pid_neg, feats_neg, dur_neg, y, exposure, (int)n, (int)use_ab, lambda_ab, o);
    CIRS_CHECK_LAUNCH("train_rows_kernel");
    hipLaunchKernelGGL(train_loss_kernel, dim3(1), dim3(256), 0, s, (const float*)o.LP, (int)n, lambda_ab, loss_out);
    // dense layers: dW = dY^T X over the 2n pair rows
    launch_dw_gemm(o.DA1, tH, o.X, K, R, tH, K, grads + L.w1, grads + L.b1, partial, s);
    launch_dw_gemm(o.DA2, tH, o.H1, tH, R, tH, tH, grads + L.w2, grads + L.b2, partial, s);
    launch_dw_gemm(o.DY, 1, o.H2, tH, R, 1, tH, grads + L.last, grads + L.out_bias, partial, s);
    launch_dw_gemm(o.DY, 1, o.DUR, 1, R, 1, 1, grads + L.lin_dense, nullptr, partial, s);
    CIRS_CHECK_LAUNCH("deepfm train dW");
    // table rows
    ScatterDst du{{grads + L.emb_user, grads + L.lin_user, use_ab ? grads + L.alpha_u : nullptr}, {E, 1, 1}};
    ScatterDst di{{grads + L.emb_item, grads + L.lin_item, use_ab ? grads + L.beta_i : nullptr}, {E, 1, 1}};
    ScatterDst df{{grads + L.emb_feat, grads + L.lin_feat, nullptr}, {E, 1, 0}};
    if (int rc = train_scatter(o.KU, o.CU, R, E + 2, cfg->n_user_vocab, du, sort_ws, train_sort_bytes(4L * R), s)) return rc;
    if (int rc = train_scatter(o.KI, o.CI, R, E + 2, cfg->n_item_vocab, di, sort_ws, train_sort_bytes(4L * R), s)) return rc;
    if (int rc = train_scatter(o.KF, o.CF, 4 * R, E + 1, cfg->n_feat_vocab, df, sort_ws, train_sort_bytes(4L * R), s)) return rc;
    // regulariser + Adam (torch.optim.Adam, bias corrections from the step count)
    L2Segs segs;
    segs.n = 4;
    segs.end[0] = L.lin_user;  segs.c[0] = l2_embedding + l2_all;   // embedding_dict.*            (core/user_model.py:60-63)
    segs.end[1] = L.lm_user;   segs.c[1] = l2_all;                  // linear.*, dnn, last, out.bias, alpha_u, beta_i
    segs.end[2] = L.total;     segs.c[2] = l2_linear + l2_all;      // linear_model.* (unused in forward, still decays)
    segs.end[3] = L.total;     segs.c[3] = l2_linear + l2_all;
    const double t = (double)(step_before + 1);
    const float step_size = (float)((double)lr / (1.0 - pow((double)beta1, t)));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, t));
    hipLaunchKernelGGL(adam_l2_kernel, dim3(kRegBlocks), dim3(256), 0, s, params, grads, adam_m, adam_v, L.total, segs, beta1, beta2, eps,
                       step_size, bc2s, regp);
    hipLaunchKernelGGL(reg_final_kernel, dim3(1), dim3(256), 0, s, (const float*)regp, loss_out);
    CIRS_CHECK_LAUNCH("adam_l2_kernel");
    return CIRS_OK;
}
