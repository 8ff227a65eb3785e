// gather_fm.hip -- the HBM/cache-bound front of the DeepFM user model on gfx950: embedding gather (K1) + linear logit and
// FM bi-interaction (K2), WITHOUT the DNN (the stage the north-star's "achieved HBM GB/s vs peak"
// is about).  Reference: core/user_model.py:419-447 (input_from_feature_columns: X[:, f].long() -> nn.Embedding rows),
// core/layers.py:59-70 (Linear: sum of 1-d embeddings + dense * w), DeepCTR-Torch layers/interaction.py:26-34 (FM).
//
//   X [n, 7] float32 rows = [user_id, photo_id, feat0..feat3, photo_duration]  -- exactly the tensor the reference feeds to
//   UserModel_Pairwise.forward (ids travel as float32; exact below 2^24)
//   out[r] = sum_f lin_f[x_f] + dur * w_d + 0.5 * sum_e [ (sum_f v_f[e])^2 - sum_f v_f[e]^2 ]
//
// Layout / mapping.  An embedding row is E fp32 = E/4 float4; LPP = E/4 lanes own one pair (16 B per lane per table row: every
// gather is a full-width 16 B/lane request, a row is one or two 128 B lines), a wavefront scores 64/LPP pairs per pass and
// kUnroll passes are in flight per lane (2*kUnroll independent 16 B gathers + the X row) before the first use.  The shared
// `feat` table (32 x E for KuaiRec) is staged once per workgroup in LDS; the 1-d linear tables are gathered by the pair's
// lane 0.  Reductions: 4 components in the lane, then a butterfly over the LPP lanes (xor 1, 2, ...): fixed order, restated
// by oracle_gather_fm.  ALGORITHMIC bytes per pair: 28 (X row) + 2*(4E+4) (user + item rows with their linear weights) + 4
// (out) = 8E + 40  (E=32: 296 B; E=16: 168 B; E=64: 552 B); the feat table is cache/LDS resident and not counted (SURVEY
// §8(d)).  ~70 VALU ops per lane per pair: far from compute-bound; the kernel is bound by L2 (tables that fit a 4 MiB L2, C3)
// or HBM (C5: 2^20-row tables) gather bandwidth.
#include "common.h"

namespace cirs {

constexpr int kGfmUnroll = 4;
constexpr int kGfmMaxLdsFloats = 8192;   // feat table up to 32 KiB in LDS, otherwise gathered from global (L1/L2)

template <int E, bool kFeatLds>
__global__ __launch_bounds__(256) void gather_fm_kernel(cirs_deepfm_weights w, int n_feat_vocab, const float* __restrict__ X, long n,
                                                        float* __restrict__ out) {
    constexpr int LPP = E / 4;            // lanes per pair
    constexpr int PPW = 64 / LPP;         // pairs per wavefront pass
    extern __shared__ __attribute__((aligned(16))) float feat_lds[];
    if (kFeatLds) {
        const float4* src = reinterpret_cast<const float4*>(w.emb_feat);
        float4* dst = reinterpret_cast<float4*>(feat_lds);
        for (int k = threadIdx.x; k < n_feat_vocab * LPP; k += blockDim.x) dst[k] = src[k];
        __syncthreads();
    }
    const float4* featp = kFeatLds ? reinterpret_cast<const float4*>(feat_lds) : reinterpret_cast<const float4*>(w.emb_feat);
    const float4* eu = reinterpret_cast<const float4*>(w.emb_user);
    const float4* ei = reinterpret_cast<const float4*>(w.emb_item);
    const int lane = threadIdx.x & 63;
    const int g = lane % LPP, sub = lane / LPP;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long n_waves = (long)gridDim.x * (blockDim.x >> 6);
    const long step = (long)PPW * kGfmUnroll;
    const float wd = w.lin_dense[0];
    for (long base = wave * step; base < n; base += n_waves * step) {
        float4 vu[kGfmUnroll], vi[kGfmUnroll];
        float xr[kGfmUnroll][7];
        long r[kGfmUnroll];
#pragma unroll
        for (int k = 0; k < kGfmUnroll; ++k) {      // issue every independent load of the kUnroll pairs first
            r[k] = base + (long)k * PPW + sub;
            const long rr = r[k] < n ? r[k] : n - 1;
            const float* xp = X + rr * 7;
#pragma unroll
            for (int c = 0; c < 7; ++c) xr[k][c] = xp[c];
        }
#pragma unroll
        for (int k = 0; k < kGfmUnroll; ++k) {
            vu[k] = eu[(long)xr[k][0] * LPP + g];
            vi[k] = ei[(long)xr[k][1] * LPP + g];
        }
#pragma unroll
        for (int k = 0; k < kGfmUnroll; ++k) {
            const int f0 = (int)xr[k][2], f1 = (int)xr[k][3], f2 = (int)xr[k][4], f3 = (int)xr[k][5];
            const float4 a = featp[f0 * LPP + g], b = featp[f1 * LPP + g], c = featp[f2 * LPP + g], d = featp[f3 * LPP + g];
            const float4 u = vu[k], it = vi[k];
            float part = 0.f;
            {
#define CIRS_GFM_COMP(m)                                                                                   \
    {                                                                                                      \
        const float s = ((((u.m + it.m) + a.m) + b.m) + c.m) + d.m;                                        \
        float q = u.m * u.m;                                                                               \
        q = __builtin_fmaf(it.m, it.m, q); q = __builtin_fmaf(a.m, a.m, q); q = __builtin_fmaf(b.m, b.m, q); \
        q = __builtin_fmaf(c.m, c.m, q); q = __builtin_fmaf(d.m, d.m, q);                                  \
        part += s * s - q;                                                                                 \
    }
                CIRS_GFM_COMP(x) CIRS_GFM_COMP(y) CIRS_GFM_COMP(z) CIRS_GFM_COMP(w)
#undef CIRS_GFM_COMP
            }
#pragma unroll
            for (int m = 1; m < LPP; m <<= 1) part += __shfl_xor(part, m, CIRS_WAVE);
            if (g == 0 && r[k] < n) {
                float lin = w.lin_user[(long)xr[k][0]] + w.lin_item[(long)xr[k][1]];
                lin += w.lin_feat[f0]; lin += w.lin_feat[f1]; lin += w.lin_feat[f2]; lin += w.lin_feat[f3];
                lin += xr[k][6] * wd;
                out[r[k]] = lin + 0.5f * part;
            }
        }
    }
}

template <int E>
static int launch_gather_fm(const cirs_deepfm_cfg* cfg, const cirs_deepfm_weights* w, const float* X, long n, float* out, hipStream_t s) {
    constexpr int PPW = 64 / (E / 4);
    const long per_wg = (long)PPW * kGfmUnroll * 4;
    // enough workgroups to fill 256 CUs x 8 waves/SIMD, grid-stride beyond that (the LDS copy of the feat table is amortised)
    const int grid = (int)std::min<long>((n + per_wg - 1) / per_wg, 256L * 8);
    const bool lds = (long)cfg->n_feat_vocab * E <= kGfmMaxLdsFloats;
    if (lds)
        hipLaunchKernelGGL((gather_fm_kernel<E, true>), dim3(grid), dim3(256), (size_t)cfg->n_feat_vocab * E * sizeof(float), s, *w,
                           cfg->n_feat_vocab, X, n, out);
    else
        hipLaunchKernelGGL((gather_fm_kernel<E, false>), dim3(grid), dim3(256), 0, s, *w, cfg->n_feat_vocab, X, n, out);
    CIRS_CHECK_LAUNCH("gather_fm_kernel");
    return CIRS_OK;
}

}  // namespace cirs

extern "C" int cirs_gather_fm(const cirs_deepfm_cfg* cfg, const cirs_deepfm_weights* w, const float* X, int64_t n, float* out,
                              void* stream) {
    using namespace cirs;
    CIRS_REQUIRE(cfg && w, "deepfm cfg/weights null");
    CIRS_REQUIRE(w->emb_user && w->emb_item && w->emb_feat && w->lin_user && w->lin_item && w->lin_feat && w->lin_dense,
                 "gather_fm: embedding / linear table pointer null");
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(X && out, "null argument");
    hipStream_t s = (hipStream_t)stream;
    switch (cfg->emb_dim) {
        case 8: return launch_gather_fm<8>(cfg, w, X, n, out, s);
        case 16: return launch_gather_fm<16>(cfg, w, X, n, out, s);
        case 32: return launch_gather_fm<32>(cfg, w, X, n, out, s);
        case 64: return launch_gather_fm<64>(cfg, w, X, n, out, s);
        default: return fail(CIRS_E_UNSUPPORTED, "gather_fm: emb_dim must be 8, 16, 32 or 64");
    }
}
