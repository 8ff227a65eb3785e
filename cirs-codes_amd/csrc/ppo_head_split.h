// ppo_head_split.h -- the actor head of a PPO minibatch step as TWO kernels without gradient slabs per row block (round 4).
// Included by ppo.hip (uses its MbView / RowTerms / staging constants).  Reference: core/policy/ppo.py:181-233 (the loss and its
// backward through Actor.last, tianshou/utils/net/discrete.py:56-67).
//
// Round 3's fused backward kernel owned (row block x item chunk) tiles, so BOTH of its outputs were split-K partials: 8 row-block
// slabs of dWa (22 MB) and 31 chunk slabs of dH2 (8 MB) per launch, re-read by two more launches.  The two reductions have
// different shapes -- dWa sums over 1024 rows into 10728 x 64 outputs, dH2 sums over 10728 items into 1024 x 64 outputs -- and here
// each gets the ownership that suits it:
//
//   head_fwd_kernel   (row block x item chunk, H2 rows stationary in registers, Wa tiles stream through LDS)
//       logits tile Z^T = Wa H2^T once; P = exp(z - m_ref) against a per-(row, chunk) reference maximum fixed at the chunk's first
//       tile (re-based only if a later logit exceeds it by 2^64: floating point keeps the relative precision, so no running-max
//       rescale per tile); per-chunk partials (m_ref, s' = sum P, t' = sum P z) with the row's TAKEN action excluded, and
//       O' = P Wa (rows x 64, action excluded) from the accumulator registers -- the soft-max-weighted head row, i.e. the d h2 product
//       moved into the forward pass:   d h2 = c_logp (Wa[a] (1 - p_a) - sum_{i != a} p_i Wa[i]),   1 - p_a = s'/S  (no cancellation
//       when the policy is sharp: the action's own term never enters the sums).  One O' slab per chunk (31 x 256 KB), as before.
//   head_dwa_kernel   (4 item tiles x row range, Wa tiles stationary in registers, H2 tiles stream through LDS)
//       prologue: the chunk partials of the workgroup's rows -> lse, loss terms, backward coefficients (as round 3's merge prologue);
//       side job: a slice of the rows' O' slabs folded into d h2 (replaces dh2_sum_kernel);
//       main loop: Z = H2 Wa^T (lane owns an ITEM, registers are rows), dZ = c_logp (delta - p) in place, and the SAME registers are
//       the B operand of dWa^T += H2^T dZ (the H2 planes are pre-laid-out in the accumulators' row order): no transpose through LDS,
//       no cross-wave reduction, the dWa tile of a wave stays in its accumulators over the whole row range and is written once.
//       n_ranges (3 at C3) slabs of dWa instead of 8; the entropy's clamp correction is one scalar per workgroup.
// ent_coef != 0 (the entropy term in dZ) keeps the round-3 kernels.
#pragma once

namespace cirs {

// probe builds (-DCIRS_HEAD_PROF, tools/probes/head_split_prof.py): s_memtime stamps of workgroup (0, 0), thread 0
#ifdef CIRS_HEAD_PROF
#define CIRS_XSTAMP(K) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_head_prof[K] = __builtin_amdgcn_s_memtime(); } while (0)
#define CIRS_XTSTAMP(IT, K) do { if ((IT) == 2) CIRS_XSTAMP(K); } while (0)
#else
#define CIRS_XSTAMP(K) do { } while (0)
#define CIRS_XTSTAMP(IT, K) do { } while (0)
#endif

constexpr int kDwaMaxSlabs = 8;      // row ranges of head_dwa_kernel = dWa slabs (wa_slab_sum_block keeps 8 in flight)
constexpr int kDwaMaxRows = 2048;    // rows of one row range (LDS arrays of the row scalars)
constexpr int kH2TileU4 = 1536;      // uint4 per 32-row tile of the H2 planes: h2z (768) + h2b (768)

struct HeadSplitGeom { int n_groups, n_ranges, tiles_per_range; };
inline HeadSplitGeom head_split_geom(int I, int n_pad) {
    const int n_item_tiles = cdiv(I, kTileN), n_groups = cdiv(n_item_tiles, 4), n_row_tiles = n_pad / kTileM;
    int R = (device_cu_count() + n_groups / 2) / n_groups;     // ~ one workgroup per CU
    R = R < 1 ? 1 : R;
    R = R > kDwaMaxSlabs ? kDwaMaxSlabs : R;
    R = R > n_row_tiles ? n_row_tiles : R;
    const int tpr = cdiv(n_row_tiles, R);
    return HeadSplitGeom{n_groups, cdiv(n_row_tiles, tpr), tpr};
}

// ---- forward: statistics partials + O' = P Wa ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void head_fwd_kernel(int I, int mb, int n_pad, int tiles_per_chunk, const uint4* __restrict__ planes,
                                                          const float* __restrict__ ba, const uint4* __restrict__ h2z,
                                                          const int32_t* __restrict__ act_rows, ActorPartialView pv,
                                                          float* __restrict__ oslab, float* __restrict__ za_out, float* __restrict__ ea_out) {
    __shared__ __attribute__((aligned(16))) unsigned char sW[2][kWBufB];
    __shared__ __attribute__((aligned(16))) float sB[2][kTileN];
    const int tid = threadIdx.x;
    CIRS_XSTAMP(40);
    const int lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int row0 = (blockIdx.y * 4 + wv) * kTileM;
    const bool wave_ok = row0 < n_pad;          // idle row tiles (padding of the last row block) still stage and meet the barriers
    const int chunk = blockIdx.x;
    // gridDim.x is padded to a multiple of 8: workgroups are dealt round-robin to the 8 XCDs, so the row blocks of one chunk share an L2
    if (chunk * tiles_per_chunk * kTileN >= I) return;
    const int jr = wave_ok ? row0 + lo : 0;
    const float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    Planes hz[4];      // B operand of Z^T: this lane's row of H2 (pre-split by trunk_adv_kernel in register order)
    {
        const uint4* zp = h2z + (size_t)((wave_ok ? row0 : 0) >> 5) * 12 * 64 + lane;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            hz[s4].h = __builtin_bit_cast(bf16x8, zp[(3 * s4) * 64]);
            hz[s4].m = __builtin_bit_cast(bf16x8, zp[(3 * s4 + 1) * 64]);
            hz[s4].l = __builtin_bit_cast(bf16x8, zp[(3 * s4 + 2) * 64]);
        }
    }
    const int act_r = (wave_ok && jr < mb) ? act_rows[jr] : -1;
    f32x16 dh0, dh1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dh0[r] = 0.f; dh1[r] = 0.f; }
    float s_acc = 0.f, t_acc = 0.f;      // sum P, sum P z over this half-wave's items (the row's action excluded)
    float nm2 = 0.f;                     // -(m_ref log2 e): P = exp2(z log2 e + nm2)
    float za_val = 0.f, ea_val = 0.f;
    bool za_have = false;

    const int first_tile = chunk * tiles_per_chunk * kTileN;
    const int n_tiles = max(0, min(tiles_per_chunk, (I - first_tile + kTileN - 1) / kTileN));
    const int dst_r = (tid >> 3) * kRowB + (tid & 7) * 16;
    const int dst_c = 3 * kRPlaneB + (tid >> 2) * kColB + (tid & 3) * 16;
    uint4 gr0, gr1, gr2, gc0, gc1, gc2;
    float gb = 0.f;
    // items beyond the catalogue (last tile): zero weight rows and a bias of -1e30 -> z = -1e30, P = 0 exactly, P z = -0: no masks
#define CIRS_ISSUE(TILE0)                                                                                  \
    do {                                                                                                   \
        const uint4* src_ = planes + (size_t)((TILE0) / kTileN) * kPlaneTileU4 + tid;                      \
        gr0 = src_[0]; gr1 = src_[256]; gr2 = src_[512]; gc0 = src_[768]; gc1 = src_[1024]; gc2 = src_[1280]; \
        if (tid < kTileN) gb = ((TILE0) + tid) < I ? ba[(TILE0) + tid] : -1e30f;                           \
    } while (0)
#define CIRS_COMMIT(BUF)                                                                                   \
    do {                                                                                                   \
        unsigned char* base_ = sW[BUF];                                                                    \
        *reinterpret_cast<uint4*>(base_ + dst_r) = gr0;                                                    \
        *reinterpret_cast<uint4*>(base_ + kRPlaneB + dst_r) = gr1;                                         \
        *reinterpret_cast<uint4*>(base_ + 2 * kRPlaneB + dst_r) = gr2;                                     \
        *reinterpret_cast<uint4*>(base_ + dst_c) = gc0;                                                    \
        *reinterpret_cast<uint4*>(base_ + kCPlaneB + dst_c) = gc1;                                         \
        *reinterpret_cast<uint4*>(base_ + 2 * kCPlaneB + dst_c) = gc2;                                     \
        if (tid < kTileN) sB[BUF][tid] = gb;                                                               \
    } while (0)
    if (n_tiles > 0) { CIRS_ISSUE(first_tile); CIRS_COMMIT(0); }
    __syncthreads();
    CIRS_XSTAMP(41);
    for (int it = 0; it < n_tiles; ++it) {
        const int buf = it & 1;
        const int tile0 = first_tile + it * kTileN;
        CIRS_XTSTAMP(it, 44);
        if (it + 1 < n_tiles) CIRS_ISSUE(tile0 + kTileN);
        if (wave_ok) {
            const unsigned char* tw = sW[buf];
            Planes za[4], cb[2][2];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const unsigned char* ap = tw + lo * kRowB + (16 * s4 + 8 * hi) * 2;
                za[s4].h = *reinterpret_cast<const bf16x8*>(ap);
                za[s4].m = *reinterpret_cast<const bf16x8*>(ap + kRPlaneB);
                za[s4].l = *reinterpret_cast<const bf16x8*>(ap + 2 * kRPlaneB);
            }
            f32x16 acc, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = sB[buf][acc_row(r, hi)]; acc1[r] = 0.f; }
            CIRS_XTSTAMP(it, 45);
            mfma_bf16x6_two(za[0], hz[0], acc, za[2], hz[2], acc1);
            mfma_bf16x6_two(za[1], hz[1], acc, za[3], hz[3], acc1);
            CIRS_XTSTAMP(it, 46);
            // the B planes of the O' product are requested now (the A planes of the logits are dead); the exponentials cover their latency
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const unsigned char* bp = tw + 3 * kRPlaneB + (32 * c + lo) * kColB + (16 * t + 8 * hi) * 2;
                    cb[c][t].h = *reinterpret_cast<const bf16x8*>(bp);
                    cb[c][t].m = *reinterpret_cast<const bf16x8*>(bp + kCPlaneB);
                    cb[c][t].l = *reinterpret_cast<const bf16x8*>(bp + 2 * kCPlaneB);
                }
            f32x16 zs, tk;
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                zs[r] = acc[r] + acc1[r]; zs[r + 1] = acc[r + 1] + acc1[r + 1];
                tk[r] = __builtin_fmaf(zs[r], kLog2e, nm2); tk[r + 1] = __builtin_fmaf(zs[r + 1], kLog2e, nm2);
                tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, tk[r]), tk[r + 1]);
            }
            // Reference maximum of the (row, chunk): the row's maximum over the chunk's FIRST tile (both half-waves agree: they feed one
            // MFMA row).  Later tiles re-base only when a logit exceeds the reference by 2^64 (never in practice; any lane -> the wave).
            if (it == 0 || __any(tmax > 64.0f)) {
                const float mrow = __builtin_fmaxf(tmax, __shfl_xor(tmax, 32, CIRS_WAVE));
                const float sh = it == 0 ? mrow : __builtin_fmaxf(mrow, 0.f);      // shift of the reference in log2 units (it only grows)
                const float fac = it == 0 ? 0.f : __builtin_amdgcn_exp2f(-sh);      // (nothing accumulated yet at it == 0)
                s_acc *= fac; t_acc *= fac; ea_val *= fac;
                if (it != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {      // the O' accumulators hold 16 different ROWS of one column: that row's factor
                        const float fr = __shfl(fac, acc_row(r, hi), CIRS_WAVE);
                        dh0[r] *= fr; dh1[r] *= fr;
                    }
                }
                nm2 -= sh;
#pragma unroll
                for (int r = 0; r < 16; ++r) tk[r] -= sh;
            }
            CIRS_XTSTAMP(it, 47);
            f32x16 p;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(tk[r]);
            {   // the row's taken action: its logit and its P leave through za / ea, and it is excluded from every sum of this kernel
                const int arel = act_r - tile0;
                const bool mine = act_r >= 0 && arel >= 0 && arel < kTileN && ((arel >> 2) & 1) == hi;
                if (__any(mine)) {       // (wave-uniform: ~9 % of the tiles hold an action of the wave's rows)
                    const int rsel = (arel & 3) + 4 * (arel >> 3);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool hit = mine && rsel == r;
                        za_val = hit ? zs[r] : za_val;
                        ea_val = hit ? p[r] : ea_val;
                        p[r] = hit ? 0.f : p[r];
                    }
                    za_have = za_have || mine;
                }
            }
            float ss = 0.f, tt = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ss += p[r]; tt = __builtin_fmaf(p[r], zs[r], tt); }
            s_acc += ss; t_acc += tt;
            CIRS_XTSTAMP(it, 48);
            if (it + 1 < n_tiles) CIRS_COMMIT(buf ^ 1);     // the other buffer was last read one barrier ago
            CIRS_XTSTAMP(it, 49);
            {
                const Planes a0 = split8(p, 0), a1 = split8(p, 8);   // element j: P[row lo][item acc_row(8 t + j, hi)]
                mfma_bf16x6_pair(a0, cb[0][0], cb[1][0], dh0, dh1);
                mfma_bf16x6_pair(a1, cb[0][1], cb[1][1], dh0, dh1);
            }
            CIRS_XTSTAMP(it, 50);
        } else {
            if (it + 1 < n_tiles) CIRS_COMMIT(buf ^ 1);
        }
        lds_barrier();
        CIRS_XTSTAMP(it, 51);
    }
    CIRS_XSTAMP(42);
#undef CIRS_ISSUE
#undef CIRS_COMMIT
    if (!wave_ok) return;
    float* hslab = oslab + (size_t)chunk * n_pad * kH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + acc_row(r, hi);
        hslab[(size_t)row * kH + lo] = dh0[r];
        hslab[(size_t)row * kH + 32 + lo] = dh1[r];
    }
    s_acc += __shfl_xor(s_acc, 32, CIRS_WAVE);      // the two half-waves: same row, same reference, disjoint items
    t_acc += __shfl_xor(t_acc, 32, CIRS_WAVE);
    if (hi == 0) {
        const size_t po = (size_t)chunk * n_pad + jr;
        pv.m[po] = -nm2 * kLn2; pv.s[po] = s_acc; pv.score[po] = t_acc;
    }
    if (za_have) { za_out[jr] = za_val; ea_out[jr] = ea_val; }
    CIRS_XSTAMP(43);
}

// ---- backward: row merge + d h2 fold + dWa / dba -------------------------------------------------------------------------------
struct HeadDwaArgs {
    cirs_ppo_cfg cfg;
    int mb_norm;          // rows of the (global) minibatch every mean is taken over
    int n_schunks;        // chunks of head_fwd_kernel
    int tiles_per_chunk;  // its tiles per chunk (locates the chunk that saw a row's action)
    int n_groups, n_ranges, tiles_per_range;
    ActorPartialView pv;
    float* oslab;         // [n_schunks][n_pad][64] O' slabs; slab 0 receives d h2
    const float* wa;      // fp32 head weights (the action's row in the d h2 fold)
    float* entw;          // [n_groups * n_ranges] clamp correction of the entropy, one scalar per workgroup
};

__global__ __launch_bounds__(256, 1) void head_dwa_kernel(int I, int mb, int n_pad, const uint4* __restrict__ planes,
                                                          const float* __restrict__ ba, MbView v, float* __restrict__ dwap, HeadDwaArgs a) {
    __shared__ __attribute__((aligned(16))) uint4 sH[2][kH2TileU4];
    __shared__ __attribute__((aligned(16))) float sNl[kDwaMaxRows];     // -(lse log2 e)   (rows beyond the minibatch: -1e30 log2 e -> p = 0)
    __shared__ __attribute__((aligned(16))) float sNc[kDwaMaxRows];     // -c_logp
    __shared__ __attribute__((aligned(16))) int sAct[kDwaMaxRows];      // taken action (-1: none)
    __shared__ float sM[kDwaMaxRows], sInvS[kDwaMaxRows], sQa[kDwaMaxRows];   // row maximum, 1 / sum exp, 1 - p_a (d h2 fold)
    __shared__ unsigned int sMask[4][kDwaMaxRows / kTileM / 32];        // per wave: row tiles that hold an action inside the wave's item tile
    __shared__ float sEnt[4];
    const int tid = threadIdx.x;
    CIRS_XSTAMP(30);
    const int lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int g = blockIdx.x, rr = blockIdx.y;
    const int rt0 = rr * a.tiles_per_range, rt1 = min(n_pad / kTileM, rt0 + a.tiles_per_range);
    const int n_rt = rt1 - rt0, r_begin = rt0 * kTileM, n_rows = n_rt * kTileM;
    const int tile0 = (g * 4 + wv) * kTileN;
    const bool wave_ok = tile0 < I;
    const float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    // ---- this wave's item tile: the Wa planes (B operand of Z = H2 Wa^T: lane = item, 8 consecutive columns) and the bias ----
    Planes zb[4];
    {
        const uint4* rp = planes + (size_t)(wave_ok ? g * 4 + wv : 0) * kPlaneTileU4 + lo * 8 + hi;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            zb[s4].h = __builtin_bit_cast(bf16x8, rp[2 * s4]);
            zb[s4].m = __builtin_bit_cast(bf16x8, rp[256 + 2 * s4]);
            zb[s4].l = __builtin_bit_cast(bf16x8, rp[512 + 2 * s4]);
        }
    }
    const int my_item = tile0 + lo;
    const float bias = (wave_ok && my_item < I) ? ba[my_item] : -1e30f;      // beyond the catalogue: z = -1e30 -> p = 0, dZ = 0, no masks
    // first H2 tile on its way while the rows are merged
    uint4 st0, st1, st2, st3, st4, st5;
#define CIRS_ISSUE(RT)                                                                                     \
    do {                                                                                                   \
        const uint4* z_ = v.h2z + (size_t)(RT) * 768 + tid;                                                \
        const uint4* b_ = v.h2b + (size_t)(RT) * 768 + tid;                                                \
        st0 = z_[0]; st1 = z_[256]; st2 = z_[512]; st3 = b_[0]; st4 = b_[256]; st5 = b_[512];              \
    } while (0)
#define CIRS_COMMIT(BUF)                                                                                   \
    do {                                                                                                   \
        uint4* d_ = sH[BUF] + tid;                                                                         \
        d_[0] = st0; d_[256] = st1; d_[512] = st2; d_[768] = st3; d_[1024] = st4; d_[1280] = st5;          \
    } while (0)
    if (n_rt > 0) CIRS_ISSUE(rt0);
    if (tid < 4 * (kDwaMaxRows / kTileM / 32)) (&sMask[0][0])[tid] = 0u;
    __syncthreads();
    CIRS_XSTAMP(31);
    // ---- row merge: chunk partials of the range's rows -> lse, loss terms, backward coefficients (one thread per row) ----
    {
        const int nsc = a.n_schunks;
        const float red0 = v.red[0], red1 = v.red[1];
        const float inv_mb = 1.0f / (float)a.mb_norm;
        for (int rb = 0; rb < n_rows; rb += 256) {
            const int rl = rb + tid;
            const bool in = rl < n_rows;
            const int jr = r_begin + (in ? rl : 0);
            const bool real = in && jr < mb;
            const int jc = real ? jr : 0;
            const int act = v.act[jc];
            const float za = v.za[jc], ea = v.ez[jc], adv = v.adv[jc], lpo = v.adv[n_pad + jc], ret = v.adv[2 * (size_t)n_pad + jc],
                        vs = v.adv[3 * (size_t)n_pad + jc], val = v.value[jc];
            const int ca = (act / kTileN) / a.tiles_per_chunk;
            const float* __restrict__ pm_ = a.pv.m + jc;
            const float* __restrict__ ps_ = a.pv.s + jc;
            const float* __restrict__ pt_ = a.pv.score + jc;
            const float m_ca = pm_[(size_t)ca * n_pad];
            float M = -INFINITY, ssum = 0.f, tsum = 0.f;
            for (int cb = 0; cb < nsc; cb += 32) {      // all 3 x 32 partials of a batch requested before the first is used
                float m32[32], s32[32], u32[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const int c = cb + q;
                    const size_t o = (size_t)(c < nsc ? c : nsc - 1) * n_pad;
                    m32[q] = pm_[o]; s32[q] = ps_[o]; u32[q] = pt_[o];
                }
                float mb_ = -INFINITY;
#pragma unroll
                for (int q = 0; q < 32; ++q) mb_ = fmaxf(mb_, m32[q]);
                const float Mn = fmaxf(M, mb_);
                const float keep = __expf(M - Mn);      // first batch: exp(-inf) = 0
                ssum *= keep; tsum *= keep;
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const float f = cb + q < nsc ? __expf(m32[q] - Mn) : 0.f;
                    ssum = __builtin_fmaf(s32[q], f, ssum);
                    tsum = __builtin_fmaf(u32[q], f, tsum);
                }
                M = Mn;
            }
            const float ea_s = ea * __expf(m_ca - M);
            const float S = ssum + ea_s;
            const float lse_r = M + __logf(S);
            const RowTerms rt = ppo_row_terms(a.cfg, za, lse_r, __builtin_fmaf(ea_s, za, tsum) / S, lpo, adv, red0, red1, val, vs, ret, inv_mb);
            if (in) {
                sNl[rl] = real ? -(lse_r * kLog2e) : -1.4426950408889634e30f;
                sNc[rl] = real ? -rt.c_logp : 0.f;
                sAct[rl] = real ? act : -1;
                sM[rl] = M; sInvS[rl] = 1.0f / S; sQa[rl] = ssum / S;
                if (real) {
                    const int ita = act / kTileN;
                    if ((ita >> 2) == g) atomicOr(&sMask[ita & 3][(rl >> 5) >> 5], 1u << ((rl >> 5) & 31));
                }
                if (g == 0) {      // one writer per row: what trunk_bwd_kernel and the loss sums read
                    v.dvalue[jr] = real ? rt.dvalue : 0.f;
                    v.clip_row[jr] = real ? rt.clip_row : 0.f;
                    v.vf_row[jr] = real ? rt.vf_row : 0.f;
                    v.ent_row[jr] = real ? rt.h_ent : 0.f;       // un-clamped entropy lse - E_p[z]; the clamp correction travels per workgroup (entw)
                }
            }
        }
    }
    CIRS_XSTAMP(32);
    if (n_rt > 0) CIRS_COMMIT(0);
    __syncthreads();
    CIRS_XSTAMP(33);
    // ---- side job: this workgroup's slice of the range's rows, O' slabs -> d h2 (slab 0) ----
    {
        const int nsc = a.n_schunks;
        const int k = (n_rows + a.n_groups - 1) / a.n_groups;
        const int s0 = g * k;
        for (int w = tid; w < k * 16; w += 256) {
            const int rl = s0 + (w >> 4), c4 = (w & 15) * 4;
            if (rl >= n_rows) break;
            const int jr = r_begin + rl;
            const float M = sM[rl];
            const float* __restrict__ pm_ = a.pv.m + jr;
            float* __restrict__ op_ = a.oslab + (size_t)jr * kH + c4;
            const size_t cstride = (size_t)n_pad * kH;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int cb = 0; cb < nsc; cb += 32) {
                f32x4 o32[32];
                float m32[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const int c = cb + q < nsc ? cb + q : nsc - 1;
                    m32[q] = pm_[(size_t)c * n_pad];
                    o32[q] = *reinterpret_cast<const f32x4*>(op_ + (size_t)c * cstride);
                }
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const float f = cb + q < nsc ? __expf(m32[q] - M) : 0.f;
                    acc += o32[q] * f;
                }
            }
            const int act = sAct[rl];
            f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
            if (act >= 0) w4 = *reinterpret_cast<const f32x4*>(a.wa + (size_t)act * kH + c4);
            const float cl = -sNc[rl], qa = sQa[rl], is = sInvS[rl];
            f32x4 out = (w4 * qa - acc * is) * cl;
            if (act < 0) out = f32x4{0.f, 0.f, 0.f, 0.f};       // rows beyond the minibatch
            *reinterpret_cast<f32x4*>(op_) = out;
        }
    }
    CIRS_XSTAMP(34);
    // ---- main loop over the row tiles of the range ----
    static_assert(kDwaMaxRows / kTileM / 32 == 2, "two mask words per wave");
    const unsigned long long amask = (unsigned long long)__builtin_amdgcn_readfirstlane(sMask[wv][0]) |
                                     ((unsigned long long)__builtin_amdgcn_readfirstlane(sMask[wv][1]) << 32);
    f32x16 dw0, dw1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dw0[r] = 0.f; dw1[r] = 0.f; }
    float db = 0.f, ent = 0.f;
    f32x2_b elo2 = {0.f, 0.f};
    const float eps = 1.1920928955078125e-7f, kLog1mEps = -1.1920929665620834e-7f, kTEps = -23.0f;
    for (int kt = 0; kt < n_rt; ++kt) {
        const int buf = kt & 1;
        CIRS_XTSTAMP(kt, 0);
        if (kt + 1 < n_rt) CIRS_ISSUE(rt0 + kt + 1);
        if (wave_ok) {
            const uint4* th = sH[buf] + lane;
            Planes hz[4], hb[2][2];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                hz[s4].h = __builtin_bit_cast(bf16x8, th[(3 * s4) * 64]);
                hz[s4].m = __builtin_bit_cast(bf16x8, th[(3 * s4 + 1) * 64]);
                hz[s4].l = __builtin_bit_cast(bf16x8, th[(3 * s4 + 2) * 64]);
            }
            f32x16 acc, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = bias; acc1[r] = 0.f; }
            CIRS_XTSTAMP(kt, 1);
            // Z[row][item]: A = the H2 rows (lane = row), B = this wave's Wa tile (lane = item) -> the lane owns item lo, registers are rows
            mfma_bf16x6_two(hz[0], zb[0], acc, hz[2], zb[2], acc1);
            mfma_bf16x6_two(hz[1], zb[1], acc, hz[3], zb[3], acc1);
            CIRS_XTSTAMP(kt, 2);
            // A operand of the dWa^T product (H2^T: lane = column, 8 rows in accumulator order) + the rows' scalars
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    hb[c][t].h = __builtin_bit_cast(bf16x8, th[(12 + 3 * (2 * c + t)) * 64]);
                    hb[c][t].m = __builtin_bit_cast(bf16x8, th[(12 + 3 * (2 * c + t) + 1) * 64]);
                    hb[c][t].l = __builtin_bit_cast(bf16x8, th[(12 + 3 * (2 * c + t) + 2) * 64]);
                }
            const int base = kt * kTileM + 4 * hi;       // register r of the lane is row base + 8 (r >> 2) + (r & 3) of the range
            f32x4 nl4[4], nc4[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                nl4[g4] = *reinterpret_cast<const f32x4*>(&sNl[base + 8 * g4]);
                nc4[g4] = *reinterpret_cast<const f32x4*>(&sNc[base + 8 * g4]);
            }
            // dZ in place: t = (z - lse) log2 e as one fma, p = exp2(t), dZ = -c_logp p (+ c_logp on the row's action, below).  Entropy clamp
            // correction, low side (p < eps = 2^-23 <=> t < -23): ln2 p (-23 - t), branch-free; high side (p > 1 - eps) a rare branch.
            CIRS_XTSTAMP(kt, 3);
            f32x16 tk;
            float pmax = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float t0 = __builtin_fmaf(acc[r] + acc1[r], kLog2e, nl4[r >> 2][r & 3]);
                const float t1 = __builtin_fmaf(acc[r + 1] + acc1[r + 1], kLog2e, nl4[r >> 2][(r & 3) + 1]);
                const float p0 = __builtin_amdgcn_exp2f(t0), p1 = __builtin_amdgcn_exp2f(t1);
                tk[r] = t0; tk[r + 1] = t1;
                pmax = __builtin_fmaxf(__builtin_fmaxf(pmax, p0), p1);
                const f32x2_b d2 = f32x2_b{kTEps, kTEps} - f32x2_b{t0, t1};
                const f32x2_b w2 = {__builtin_fmaxf(d2.x, 0.f), __builtin_fmaxf(d2.y, 0.f)};
                elo2 = f32x2_b{p0, p1} * w2 + elo2;
                acc[r] = nc4[r >> 2][r & 3] * p0; acc[r + 1] = nc4[r >> 2][(r & 3) + 1] * p1;
            }
            CIRS_XTSTAMP(kt, 4);
            if ((amask >> kt) & 1ull) {      // scalar test: some row of this tile took an item of this wave's tile
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int4 a4 = *reinterpret_cast<const int4*>(&sAct[base + 8 * g4]);
                    acc[4 * g4] -= a4.x == my_item ? nc4[g4][0] : 0.f;
                    acc[4 * g4 + 1] -= a4.y == my_item ? nc4[g4][1] : 0.f;
                    acc[4 * g4 + 2] -= a4.z == my_item ? nc4[g4][2] : 0.f;
                    acc[4 * g4 + 3] -= a4.w == my_item ? nc4[g4][3] : 0.f;
                }
            }
            if (__any(pmax > 1.0f - eps)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(tk[r]);
                    ent -= p > 1.0f - eps ? p * (kLog1mEps - tk[r] * kLn2) : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) db += acc[r];
            CIRS_XTSTAMP(kt, 5);
            if (kt + 1 < n_rt) CIRS_COMMIT(buf ^ 1);
            CIRS_XTSTAMP(kt, 6);
            {
                const Planes b0 = split8(acc, 0), b1 = split8(acc, 8);   // element j of k-step t: dZ[row acc_row(8 t + j, hi)][item lo]
                mfma_bf16x6_pair_b(hb[0][0], hb[1][0], b0, dw0, dw1);
                mfma_bf16x6_pair_b(hb[0][1], hb[1][1], b1, dw0, dw1);
            }
            CIRS_XTSTAMP(kt, 7);
        } else {
            if (kt + 1 < n_rt) CIRS_COMMIT(buf ^ 1);
        }
        lds_barrier();
        CIRS_XTSTAMP(kt, 8);
    }
    CIRS_XSTAMP(35);
#undef CIRS_ISSUE
#undef CIRS_COMMIT
    // ---- results: the wave's dWa tile (lane = item, registers = 4 consecutive columns x 8), dba, the workgroup's entropy correction ----
    ent -= kLn2 * (elo2.x + elo2.y);
    if (!wave_ok) ent = 0.f;
    ent = wave_sum_f32_dpp(ent);
    if (lane == 0) sEnt[wv] = ent;
    if (wave_ok) {
        float* slab = dwap + (size_t)rr * dwa_slab_stride(I);
        if (my_item < I) {
            float* orow = slab + (size_t)my_item * kH + 4 * hi;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                *reinterpret_cast<f32x4*>(orow + 8 * g4) = f32x4{dw0[4 * g4], dw0[4 * g4 + 1], dw0[4 * g4 + 2], dw0[4 * g4 + 3]};
                *reinterpret_cast<f32x4*>(orow + 32 + 8 * g4) = f32x4{dw1[4 * g4], dw1[4 * g4 + 1], dw1[4 * g4 + 2], dw1[4 * g4 + 3]};
            }
        }
        db += __shfl_xor(db, 32, CIRS_WAVE);
        if (hi == 0 && my_item < I) slab[(size_t)I * kH + my_item] = db;
    }
    __syncthreads();
    if (tid == 0) a.entw[rr * a.n_groups + g] = (sEnt[0] + sEnt[1]) + (sEnt[2] + sEnt[3]);
    CIRS_XSTAMP(36);
}

}  // namespace cirs
