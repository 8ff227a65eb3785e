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

// global -> LDS copy of 16 bytes per lane without staging registers (global_load_lds_dwordx4): lane l of the wave writes lds + 16 l
typedef __attribute__((address_space(1))) const void* glds_src_t;
typedef __attribute__((address_space(3))) void* glds_dst_t;
__device__ __forceinline__ void glds16(const void* g_lane, void* lds_wave) {
    __builtin_amdgcn_global_load_lds((glds_src_t)g_lane, (glds_dst_t)lds_wave, 16, 0, 0);
}

constexpr int kDwaMaxSlabs = 8;      // row ranges of head_dwa_kernel = dWa slabs (wa_slab_sum_block keeps 8 in flight)
constexpr int kDwaMaxRows = 1024;    // rows of one row range (LDS arrays of the row scalars; one 32-bit mask of row tiles per item tile)
constexpr int kH2TileU4 = 1536;      // uint4 per 32-row tile of the H2 planes: h2z (768) + h2b (768)

struct HeadSplitGeom { int n_groups, n_ranges, tiles_per_range; };
inline HeadSplitGeom head_split_geom(int I, int n_pad) {
    const int n_item_tiles = cdiv(I, kTileN), n_groups = cdiv(n_item_tiles, 4), n_row_tiles = n_pad / kTileM;
    int R = (device_cu_count() + n_groups / 2) / n_groups;     // ~ one workgroup per CU
    R = R < 1 ? 1 : R;
    R = R > kDwaMaxSlabs ? kDwaMaxSlabs : R;
    R = R > n_row_tiles ? n_row_tiles : R;
    const int r_min = cdiv(n_row_tiles, kDwaMaxRows / kTileM);    // rows of a range must fit the LDS arrays
    R = R < r_min ? r_min : R;
    const int tpr = cdiv(n_row_tiles, R);
    return HeadSplitGeom{n_groups, cdiv(n_row_tiles, tpr), tpr};
}

// ---- forward: statistics partials + O' = P Wa ---------------------------------------------------------------------------------
// Row stage of the head (what round 3 ran in the prologue of its backward kernel, and the first version of this file in the prologue of
// EVERY head_dwa_kernel workgroup -- 84 redundant merges of each row, 175 KB per workgroup pulled at the ~11 B/cycle a CU gets when
// all 256 start at once: 20 k of that kernel's 70 k cycles).  Here the LAST workgroup of a row block to finish (arrival counter) folds the
// chunk partials of the block's 128 rows once: lse, loss terms, backward coefficients.  The partials cross workgroups inside one launch, so
// they are written through (sc1 stores, completed before the arrival) and read around L1 (sc1 loads): the write-through form of
// MI355X_MICROARCH.md's inter-workgroup visibility rules -- no L2 write-back fence (that cost 6 us per workgroup in round 1).
struct HeadRowArgs {
    cirs_ppo_cfg cfg;
    int mb_norm;            // rows of the (global) minibatch every mean is taken over
    int n_chunks;           // chunks that arrive per row block
    int* counters;          // [n_row_blocks] arrival counters, zeroed by trunk_adv_kernel
    MbView v;
    float* row_m;           // [n_pad] out: the row's reference maximum M (d h2 fold); lse -> v.lse, c_logp -> v.c_logp, 1/S -> v.h_ent, 1 - p_a -> v.c_ent
};
__device__ __forceinline__ void st_sc1(float2* p, float a, float b) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float* p, float a) { __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float2 ld_sc1(const float2* p) {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return float2{__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))};
}
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256, 1) void head_fwd_kernel(int I, int mb, int n_pad, int tiles_per_chunk, const uint4* __restrict__ planes,
                                                          const float* __restrict__ ba, const uint4* __restrict__ h2z,
                                                          const int32_t* __restrict__ act_rows, float2* __restrict__ part_ms, float2* __restrict__ part_tz,
                                                          float* __restrict__ oslab, float* __restrict__ za_out, float* __restrict__ ea_out, HeadRowArgs ra) {
    __shared__ __attribute__((aligned(16))) unsigned char sW[3][kWBufB];      // the planes of three consecutive item tiles
    __shared__ __attribute__((aligned(16))) float sB[3][kTileN];
    const int tid = threadIdx.x;
    CIRS_XSTAMP(40);
    const int lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    const int row0 = (blockIdx.y * 4 + wv) * kTileM;
    const bool wave_ok = row0 < n_pad;          // (idle row tiles -- padding of the last row block -- compute on row tile 0 and store nothing)
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
    float zmx = -INFINITY;               // largest (z - m_ref) log2 e seen by this half-wave (the action's own logit included)
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
#define CIRS_READ_ZA(BUF)                                                                                  \
    do {                                                                                                   \
        const unsigned char* tw_ = sW[BUF];                                                                \
        _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                                 \
            const unsigned char* ap = tw_ + lo * kRowB + (16 * s4 + 8 * hi) * 2;                           \
            za[s4].h = *reinterpret_cast<const bf16x8*>(ap);                                               \
            za[s4].m = *reinterpret_cast<const bf16x8*>(ap + kRPlaneB);                                    \
            za[s4].l = *reinterpret_cast<const bf16x8*>(ap + 2 * kRPlaneB);                                \
        }                                                                                                  \
    } while (0)
    if (n_tiles > 0) { CIRS_ISSUE(first_tile); CIRS_COMMIT(0); }
    if (n_tiles > 1) { CIRS_ISSUE(first_tile + kTileN); CIRS_COMMIT(1); }
    __syncthreads();
    CIRS_XSTAMP(41);
    // The loop is software-pipelined inside the wave like head_dwa_kernel's: iteration k holds  [O'_k-1 beside the exponentials of tile k]  and
    // [the logits of tile k + 1 beside the sums and splits of tile k]  -- a dependent chain L -> P -> O' per tile left the matrix pipe idle 60 %
    // of the time (3.9 k cycles per tile for 1.5 k cycles of MFMAs).
    Planes za[4];
    f32x16 acc, acc1;      // logits of the CURRENT tile: bias + the h*h terms | the cross terms
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = n_tiles > 0 ? sB[0][acc_row(r, hi)] : 0.f; acc1[r] = 0.f; }
    CIRS_READ_ZA(0);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) mfma_bf16x6_split(za[s4], hz[s4], acc, acc1);
    Planes cbP[2][2], plP[2];      // operands of the O' product of the PREVIOUS tile (zeros before the first)
    {
        const pk4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) { cbP[c][t].h = __builtin_bit_cast(bf16x8, z4); cbP[c][t].m = cbP[c][t].h; cbP[c][t].l = cbP[c][t].h; }
        plP[0].h = plP[0].m = plP[0].l = plP[1].h = plP[1].m = plP[1].l = __builtin_bit_cast(bf16x8, z4);
    }
    int buf = 0;
    for (int it = 0; it < n_tiles; ++it) {
        const int bn = buf == 2 ? 0 : buf + 1, b2 = bn == 2 ? 0 : bn + 1;
        const bool more = it + 1 < n_tiles;
        const int tile0 = first_tile + it * kTileN;
        CIRS_XTSTAMP(it, 44);
        if (it + 2 < n_tiles) CIRS_ISSUE(tile0 + 2 * kTileN);
        // ---- block 1: O'_it-1 (operands in registers since the previous iteration) beside z, t, max, exp of tile it ----
        CIRS_READ_ZA(more ? bn : buf);        // (the next tile's A planes; after the last tile a harmless re-read)
        f32x16 an, an1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { an[r] = sB[more ? bn : buf][acc_row(r, hi)]; an1[r] = 0.f; }
        mfma_bf16x6_pair(plP[0], cbP[0][0], cbP[1][0], dh0, dh1);
        mfma_bf16x6_pair(plP[1], cbP[0][1], cbP[1][1], dh0, dh1);
        f32x16 zs, tk;
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            zs[r] = acc[r] + acc1[r]; zs[r + 1] = acc[r + 1] + acc1[r + 1];
            tk[r] = __builtin_fmaf(zs[r], kLog2e, nm2); tk[r + 1] = __builtin_fmaf(zs[r + 1], kLog2e, nm2);
            tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, tk[r]), tk[r + 1]);
        }
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x402, 3, 0);
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
            nm2 -= sh; zmx -= sh; tmax -= sh;
#pragma unroll
            for (int r = 0; r < 16; ++r) tk[r] -= sh;
        }
        zmx = __builtin_fmaxf(zmx, tmax);
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
        // ---- block 2: the logits of tile it + 1 beside the sums and the splits of tile it; this tile's C planes for the next iteration ----
        CIRS_XTSTAMP(it, 45);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) mfma_bf16x6_split(za[s4], hz[s4], an, an1);
        float ss = 0.f, tt = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ss += p[r]; tt = __builtin_fmaf(p[r], zs[r], tt); }
        s_acc += ss; t_acc += tt;
        plP[0] = split8(p, 0); plP[1] = split8(p, 8);   // element j: P[row lo][item acc_row(8 t + j, hi)]
        {
            const unsigned char* tw = sW[buf];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const unsigned char* bp = tw + 3 * kRPlaneB + (32 * c + lo) * kColB + (16 * t + 8 * hi) * 2;
                    cbP[c][t].h = *reinterpret_cast<const bf16x8*>(bp);
                    cbP[c][t].m = *reinterpret_cast<const bf16x8*>(bp + kCPlaneB);
                    cbP[c][t].l = *reinterpret_cast<const bf16x8*>(bp + 2 * kCPlaneB);
                }
        }
        if (it + 2 < n_tiles) CIRS_COMMIT(b2);     // (that buffer held tile it - 1: last read one barrier ago)
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
        }
        acc = an; acc1 = an1;
        CIRS_XTSTAMP(it, 46);
        lds_barrier();
        CIRS_XTSTAMP(it, 47);
        buf = bn;
    }
    // O' of the last tile
    mfma_bf16x6_pair(plP[0], cbP[0][0], cbP[1][0], dh0, dh1);
    mfma_bf16x6_pair(plP[1], cbP[0][1], cbP[1][1], dh0, dh1);
#undef CIRS_READ_ZA
    CIRS_XSTAMP(42);
#undef CIRS_ISSUE
#undef CIRS_COMMIT
    if (wave_ok) {
        s_acc += __shfl_xor(s_acc, 32, CIRS_WAVE);      // the two half-waves: same row, same reference, disjoint items
        t_acc += __shfl_xor(t_acc, 32, CIRS_WAVE);
        zmx = __builtin_fmaxf(zmx, __shfl_xor(zmx, 32, CIRS_WAVE));
        if (hi == 0) {
            const size_t po = (size_t)chunk * n_pad + jr;
            st_sc1(&part_ms[po], -nm2 * kLn2, s_acc);               // {m_ref, s'}
            st_sc1(&part_tz[po], t_acc, (zmx - nm2) * kLn2);        // {t', the chunk's largest logit} (the reported entropy)
        }
        if (za_have) { st_sc1(&za_out[jr], za_val); st_sc1(&ea_out[jr], ea_val); }
    }
    // ---- arrival: this workgroup's partials are complete in memory (the O' slab, which the row stage does not read, is stored AFTER the
    // arrival: its 128 KB would otherwise sit in front of the wait); the last of the row block's chunks merges the rows ----
    __shared__ int sLast;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) sLast = __hip_atomic_fetch_add(&ra.counters[blockIdx.y], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ra.n_chunks - 1;
    __syncthreads();
    auto store_slab = [&]() {
        if (!wave_ok) return;
        float* hslab = oslab + (size_t)chunk * n_pad * kH;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + acc_row(r, hi);
            hslab[(size_t)row * kH + lo] = dh0[r];
            hslab[(size_t)row * kH + 32 + lo] = dh1[r];
        }
    };
    CIRS_XSTAMP(43);
    if (!sLast) { store_slab(); return; }
    {
        const MbView& v = ra.v;
        const int nsc = ra.n_chunks;
        const float red0 = v.red[0], red1 = v.red[1];
        const float inv_mb = 1.0f / (float)ra.mb_norm;
        const int jr2 = blockIdx.y * 128 + (tid & 127);           // two threads per row: the chunks split in halves, combined through LDS
        const int half = tid >> 7;
        __shared__ float sMrg[128][4];
        const bool in = jr2 < n_pad;
        const bool real = in && jr2 < mb;
        const int jc = real ? jr2 : 0;
        const int act = act_rows[jc];
        const float za = ld_sc1(&za_out[jc]), ea = ld_sc1(&ea_out[jc]);
        const float adv = v.adv[jc], lpo = v.adv[n_pad + jc], ret = v.adv[2 * (size_t)n_pad + jc], vs = v.adv[3 * (size_t)n_pad + jc], val = v.value[jc];
        const int ca = (act / kTileN) / tiles_per_chunk;
        float m_ca = -INFINITY;       // reference maximum of the chunk that saw the action (picked from the batch below: no dependent load)
        float M = -INFINITY, ssum = 0.f, tsum = 0.f, ztop = -INFINITY;
        const int c_lo = half * ((nsc + 1) >> 1), c_hi = half ? nsc : ((nsc + 1) >> 1);
        for (int cb = c_lo; cb < c_hi; cb += 16) {      // 2 x 16 partials of a batch requested before the first is used
            float2 p16[16], q16[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int c = cb + q < c_hi ? cb + q : c_hi - 1;
                p16[q] = ld_sc1(&part_ms[(size_t)c * n_pad + jc]);
                q16[q] = ld_sc1(&part_tz[(size_t)c * n_pad + jc]);
            }
            float mb_ = -INFINITY;
#pragma unroll
            for (int q = 0; q < 16; ++q) { mb_ = fmaxf(mb_, p16[q].x); m_ca = (cb + q == ca && cb + q < c_hi) ? p16[q].x : m_ca; ztop = fmaxf(ztop, q16[q].y); }
            const float Mn = fmaxf(M, mb_);
            const float keep = __expf(M - Mn);      // first batch: exp(-inf) = 0
            ssum *= keep; tsum *= keep;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float f = cb + q < c_hi ? __expf(p16[q].x - Mn) : 0.f;
                ssum = __builtin_fmaf(p16[q].y, f, ssum);
                tsum = __builtin_fmaf(q16[q].x, f, tsum);
            }
            M = Mn;
        }
        if (half == 1) { sMrg[tid & 127][0] = M; sMrg[tid & 127][1] = ssum; sMrg[tid & 127][2] = tsum; sMrg[tid & 127][3] = ztop; }
        __shared__ float sMca[128];
        if (half == 1) sMca[tid & 127] = m_ca;
        __syncthreads();
        if (half == 0 && in) {
            const float M1 = sMrg[tid][0], s1 = sMrg[tid][1], t1 = sMrg[tid][2];
            ztop = fmaxf(ztop, sMrg[tid][3]);
            m_ca = fmaxf(m_ca, sMca[tid]);                       // exactly one half saw the action's chunk (the other holds -inf)
            const float Mn = fmaxf(M, M1);
            const float f0 = __expf(M - Mn), f1 = __expf(M1 - Mn);   // (lower half first: fixed order)
            ssum = ssum * f0 + s1 * f1; tsum = tsum * f0 + t1 * f1; M = Mn;
            const float ea_s = ea * __expf(m_ca - M);
            const float S = ssum + ea_s;
            const float lse_r = M + __logf(S);
            const RowTerms rt = ppo_row_terms(ra.cfg, za, lse_r, __builtin_fmaf(ea_s, za, tsum) / S, lpo, adv, red0, red1, val, vs, ret, inv_mb);
            // Categorical.entropy takes log(clamp(p, eps, 1 - eps)).  High side (one item holds the whole row, p > 1 - eps): at most one item per row,
            // the row's largest logit; its term p (log(1 - eps) - log p) equals -(p - (1 - eps)) for the two floats above 1 - eps.  (The low
            // side is a sum over the whole catalogue: head_dwa_kernel's loop, per workgroup.)
            const float p_top = __expf(ztop - lse_r);
            const float ent_hi = -fmaxf(p_top - (1.0f - 1.1920928955078125e-7f), 0.f);
            v.lse[jr2] = real ? lse_r : 1e30f;            // rows beyond the minibatch: p = exp(z - lse) = 0
            v.c_logp[jr2] = real ? rt.c_logp : 0.f;
            v.h_ent[jr2] = 1.0f / S; v.c_ent[jr2] = ssum / S; ra.row_m[jr2] = M;
            v.dvalue[jr2] = real ? rt.dvalue : 0.f;
            v.clip_row[jr2] = real ? rt.clip_row : 0.f;
            v.vf_row[jr2] = real ? rt.vf_row : 0.f;
            v.ent_row[jr2] = real ? rt.h_ent + ent_hi : 0.f;       // lse - E_p[z] + the high-side clamp term; the low side travels per workgroup (entw)
        }
    }
    store_slab();      // (the last workgroup's own O' slab goes out behind the row stage: its 128 KB of stores would sit in front of the partial loads)
}

// ---- backward: row merge + d h2 fold + dWa / dba -------------------------------------------------------------------------------
struct HeadDwaArgs {
    int n_schunks;        // chunks of head_fwd_kernel
    int tiles_per_chunk;  // its tiles per chunk (locates the chunk that saw a row's action)
    int n_groups, n_ranges, tiles_per_range;
    const float2* part_ms;  // [n_schunks][n_pad] {m_ref, s'} partials of head_fwd_kernel (the chunk references of the d h2 fold)
    const float* row_m;     // [n_pad] the rows' reference maxima (head_fwd_kernel's row stage; lse, c_logp, 1/S, 1 - p_a: v.lse, v.c_logp, v.h_ent, v.c_ent)
    float* oslab;         // [n_schunks][n_pad][64] O' slabs; slab 0 receives d h2
    const float* wa;      // fp32 head weights (the action's row in the d h2 fold)
    float* entw;          // [n_groups * n_ranges] clamp correction of the entropy, one scalar per workgroup
};

constexpr int kDwaThreads = 256;       // 4 waves = the 4 item tiles of the group, ONE wave per SIMD (512 registers: two row tiles in flight per wave)
constexpr int kXStride = 68;           // floats per item row of the exchange tile (conflict-free b128 accesses)

// d h2 of one (row, 4 columns): sum over the chunks of f_c O'_c with f_c = exp(m_c - M), then c_logp (Wa[a] (1 - p_a) - O' / S)
__device__ __forceinline__ void fold_store(float* __restrict__ op, f32x4 acc, int act, const float* __restrict__ wa, int c4, float cl, float qa, float is) {
    f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
    if (act >= 0) w4 = *reinterpret_cast<const f32x4*>(wa + (size_t)act * kH + c4);
    f32x4 out = (w4 * qa - acc * is) * cl;
    if (act < 0) out = f32x4{0.f, 0.f, 0.f, 0.f};       // rows beyond the minibatch
    *reinterpret_cast<f32x4*>(op) = out;
}

// Main loop of head_dwa_kernel.  A wave's row tile is three segments: L (logits, 24 MFMAs) -> V (dZ, clamp correction, splits: ~250 VALU
// + 32 LDS reads) -> D (dWa^T, 24 MFMAs), each depending on the one before, and on this part a dependent chain leaves the matrix pipe idle
// while the vector pipe works (measured: 3.9 k cycles per tile for 1.5 k cycles of MFMAs; running two waves per SIMD in opposite phases
// did not help: a wave issuing VALU beside its partner's MFMA stream gets ~5 issue slots per MFMA, the same as inside one wave).  So the
// loop is software-pipelined inside the wave: iteration j holds L of tile j + 1 and V of tile j in ONE block -- independent, so the
// scheduler fills the MFMA gaps of one with the VALU of the other -- followed by D of tile j, with the LDS reads of the next iteration
// behind its MFMAs.  The H2 tiles arrive by LDS-DMA three buffers deep (tile j + 2 is requested at the top of iteration j).
__global__ __launch_bounds__(kDwaThreads, 1) void head_dwa_kernel(int I, int mb, int n_pad, const uint4* __restrict__ planes,
                                                                  const float* __restrict__ ba, MbView v, float* __restrict__ dwap, HeadDwaArgs a) {
    __shared__ __attribute__((aligned(16))) uint4 sH[3][kH2TileU4];     // the H2 planes of one 32-row tile (h2z | h2b), three tiles deep
    __shared__ __attribute__((aligned(16))) float sNl[kDwaMaxRows];     // lse   (rows beyond the minibatch: 1e30 -> p = 0)
    __shared__ __attribute__((aligned(16))) float sNc[kDwaMaxRows];     // -c_logp
    __shared__ __attribute__((aligned(16))) int sAct[kDwaMaxRows];      // taken action (-1: none)
    __shared__ unsigned int sMask[4];                                   // per item tile: row tiles of the range that hold an action inside it
    __shared__ float sEnt[4];
    static_assert(kDwaMaxRows / kTileM <= 32, "one mask word per item tile");
    static_assert(4 * kTileN * kXStride * 4 <= (int)sizeof(uint4) * 2 * kH2TileU4, "the exchange tiles reuse two staging buffers");
    const int tid = threadIdx.x;
    CIRS_XSTAMP(30);
    kernarg_warm<448>();      // (the two by-value views are ~0.5 KB of kernel arguments: one round trip instead of one per first use)
    const int lane = tid & 63, iw = tid >> 6;     // iw: item tile of the group
    const int hi = lane >> 5, lo = lane & 31;
    const int g = blockIdx.x, rr = blockIdx.y;
    const int rt0 = rr * a.tiles_per_range, rt1 = min(n_pad / kTileM, rt0 + a.tiles_per_range);
    const int n_rt = rt1 - rt0, r_begin = rt0 * kTileM, n_rows = n_rt * kTileM;
    const int tile0 = (g * 4 + iw) * kTileN;
    const bool wave_ok = tile0 < I;
    const float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    // the range's row scalars (head_fwd_kernel's row stage) are requested FIRST: memory returns in order, and they are what the loop waits for
    float rs_lse[kDwaMaxRows / kDwaThreads], rs_cl[kDwaMaxRows / kDwaThreads];
    int rs_act[kDwaMaxRows / kDwaThreads];
#pragma unroll
    for (int q = 0; q < kDwaMaxRows / kDwaThreads; ++q) {
        const int rl = tid + kDwaThreads * q;
        const int jr = r_begin + (rl < n_rows ? rl : 0);
        rs_lse[q] = v.lse[jr]; rs_cl[q] = v.c_logp[jr];
        rs_act[q] = (rl < n_rows && jr < mb) ? v.act[jr] : -1;
    }
    const int my_item = tile0 + lo;
    const float bias = (wave_ok && my_item < I) ? ba[my_item] : -1e30f;      // beyond the catalogue: z = -1e30 -> p = 0, dZ = 0, no masks
    // ---- this wave's item tile: the Wa planes (B operand of Z = H2 Wa^T: lane = item, 8 consecutive columns); they stay in registers ----
    Planes zb[4];
    {
        const uint4* rp = planes + (size_t)(wave_ok ? g * 4 + iw : 0) * kPlaneTileU4 + lo * 8 + hi;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            zb[s4].h = __builtin_bit_cast(bf16x8, rp[2 * s4]);
            zb[s4].m = __builtin_bit_cast(bf16x8, rp[256 + 2 * s4]);
            zb[s4].l = __builtin_bit_cast(bf16x8, rp[512 + 2 * s4]);
        }
    }
    // staging: straight from global memory into LDS (global_load_lds: the planes already are in register order, so the image is lane-linear:
    // no staging registers, no ds_write pass).  Wave iw copies units [64 iw, 64 iw + 64) of each of the six 256-unit plane blocks.
#define CIRS_STAGE(KT, BUF)                                                                                \
    do {                                                                                                   \
        const uint4* z_ = v.h2z + (size_t)(rt0 + (KT)) * 768 + 64 * iw + lane;                             \
        const uint4* b_ = v.h2b + (size_t)(rt0 + (KT)) * 768 + 64 * iw + lane;                             \
        uint4* d_ = sH[BUF] + 64 * iw;                                                                     \
        glds16(z_, d_); glds16(z_ + 256, d_ + 256); glds16(z_ + 512, d_ + 512);                            \
        glds16(b_, d_ + 768); glds16(b_ + 256, d_ + 1024); glds16(b_ + 512, d_ + 1280);                    \
    } while (0)
    if (n_rt > 0) CIRS_STAGE(0, 0);
    if (n_rt > 1) CIRS_STAGE(1, 1);
    if (tid < 4) sMask[tid] = 0u;
    const int nsc = a.n_schunks;
    // ---- d h2 fold, first half: this workgroup's slice of the range's rows.  One (row, 4 columns) item per thread and third of the chunks:
    // its O' slab pieces and chunk references are requested here and travel while the rows are merged ----
    const int fold_k = (n_rows + a.n_groups - 1) / a.n_groups, fold_s0 = g * fold_k;
    const int f_per = (nsc + 2) / 3;                     // chunks per third
    const bool fold_fast = fold_k * 16 * 3 <= kDwaThreads && f_per <= 12;
    const int f_item = tid % (fold_k * 16 > 0 ? fold_k * 16 : 1), f_third = tid / (fold_k * 16 > 0 ? fold_k * 16 : 1);
    const int f_rl = fold_s0 + (f_item >> 4), f_c4 = (f_item & 15) * 4;
    const bool f_on = fold_fast && f_third < 3 && f_rl < n_rows;
    f32x4 fo[12];
    float fm[12];
    if (f_on) {
        const int jr = r_begin + f_rl;
        const float* __restrict__ pm_ = reinterpret_cast<const float*>(a.part_ms + jr);
        const float* __restrict__ op_ = a.oslab + (size_t)jr * kH + f_c4;
        const size_t cstride = (size_t)n_pad * kH;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int c0 = f_third * f_per + q;
            const int c = c0 < nsc ? c0 : nsc - 1;
            fm[q] = pm_[(size_t)c * n_pad * 2];
            fo[q] = *reinterpret_cast<const f32x4*>(op_ + (size_t)c * cstride);
        }
    }
    lds_barrier();      // (the mask words are cleared; NOT __syncthreads: the requests above stay in flight)
    CIRS_XSTAMP(31);
    // ---- the range's row scalars (head_fwd_kernel's row stage): lse, -c_logp, action -> LDS; the row tiles that hold an action of each item tile ----
    float f_M = 0.f, f_is = 0.f, f_qa = 0.f;
    if (f_on) { const int jr = r_begin + f_rl; f_M = a.row_m[jr]; f_is = v.h_ent[jr]; f_qa = v.c_ent[jr]; }
#pragma unroll
    for (int q = 0; q < kDwaMaxRows / kDwaThreads; ++q) {
        const int rl = tid + kDwaThreads * q;
        if (rl < n_rows) {
            sNl[rl] = rs_lse[q];
            sNc[rl] = -rs_cl[q];
            sAct[rl] = rs_act[q];
            if (rs_act[q] >= 0 && ((rs_act[q] / kTileN) >> 2) == g) atomicOr(&sMask[(rs_act[q] / kTileN) & 3], 1u << (rl >> 5));
        }
    }
    CIRS_XSTAMP(32);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CIRS_XSTAMP(33);
    // ---- d h2 fold, second half: scale by f_c = exp(m_c - M), the three thirds meet in LDS (fixed order), c_logp (Wa[a] (1 - p_a) - O' / S) -> slab 0 ----
    float* xf = reinterpret_cast<float*>(&sH[2][0]);      // (tile 2 is staged at the top of iteration 0, after the barrier below)
    if (fold_fast) {
        if (f_on) {
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 12; ++q) acc4 += fo[q] * ((q < f_per && f_third * f_per + q < nsc) ? __expf(fm[q] - f_M) : 0.f);
            *reinterpret_cast<f32x4*>(xf + (size_t)(f_third * fold_k * 16 + f_item) * 4) = acc4;
        }
        __syncthreads();
        if (f_on && f_third == 0) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(xf + (size_t)f_item * 4), t1 = *reinterpret_cast<const f32x4*>(xf + (size_t)(fold_k * 16 + f_item) * 4),
                        t2 = *reinterpret_cast<const f32x4*>(xf + (size_t)(2 * fold_k * 16 + f_item) * 4);
            fold_store(a.oslab + (size_t)(r_begin + f_rl) * kH + f_c4, (t0 + t1) + t2, sAct[f_rl], a.wa, f_c4, -sNc[f_rl], f_qa, f_is);
        }
        __syncthreads();
    } else {        // (large slices / many chunks: plain loop)
        for (int w = tid; w < fold_k * 16; w += kDwaThreads) {
            const int rl = fold_s0 + (w >> 4), c4 = (w & 15) * 4;
            if (rl >= n_rows) break;
            const int jr = r_begin + rl;
            const float M = a.row_m[jr];
            const float* __restrict__ pm_ = reinterpret_cast<const float*>(a.part_ms + jr);
            float* __restrict__ op_ = a.oslab + (size_t)jr * kH + c4;
            const size_t cstride = (size_t)n_pad * kH;
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
            for (int cb = 0; cb < nsc; cb += 16) {      // 16 slabs in flight per batch (a load-use loop is one memory round trip per chunk)
                f32x4 o16[16];
                float m16[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = cb + q < nsc ? cb + q : nsc - 1;
                    m16[q] = pm_[(size_t)c * n_pad * 2];
                    o16[q] = *reinterpret_cast<const f32x4*>(op_ + (size_t)c * cstride);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) acc4 += o16[q] * (cb + q < nsc ? __expf(m16[q] - M) : 0.f);
            }
            fold_store(op_, acc4, sAct[rl], a.wa, c4, -sNc[rl], v.c_ent[jr], v.h_ent[jr]);
        }
    }
    CIRS_XSTAMP(34);
    const unsigned int amask = __builtin_amdgcn_readfirstlane(sMask[iw]);
    f32x16 dw0, dw1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dw0[r] = 0.f; dw1[r] = 0.f; }
    f32x16 acc, acs;       // logits of the CURRENT tile: bias + the h*h terms | the cross terms (mfma_bf16x6_split)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = bias; acs[r] = 0.f; }
    Planes hz[4];
    {   // L_0
        const uint4* th = sH[0] + lane;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            hz[s4].h = __builtin_bit_cast(bf16x8, th[(3 * s4) * 64]);
            hz[s4].m = __builtin_bit_cast(bf16x8, th[(3 * s4 + 1) * 64]);
            hz[s4].l = __builtin_bit_cast(bf16x8, th[(3 * s4 + 2) * 64]);
        }
        // Z[row][item]: A = the H2 rows (lane = row), B = this wave's Wa tile (lane = item) -> the lane owns item lo, registers are rows
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) mfma_bf16x6_split(hz[s4], zb[s4], acc, acs);
    }
    float db = 0.f;
    float elo = 0.f;
    const float kLogEps = -15.942385152878742f;
    // operands of the D product of the PREVIOUS tile (zeros before the first: D_-1 adds nothing)
    Planes hbP[2][2], plP[2];
    {
        const pk4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) { hbP[c][t].h = __builtin_bit_cast(bf16x8, z4); hbP[c][t].m = hbP[c][t].h; hbP[c][t].l = hbP[c][t].h; }
        plP[0].h = plP[0].m = plP[0].l = plP[1].h = plP[1].m = plP[1].l = __builtin_bit_cast(bf16x8, z4);
    }
    int buf = 0;       // buffer of tile j
    for (int j = 0; j < n_rt; ++j) {
        const int bn = buf == 2 ? 0 : buf + 1;       // buffer of tile j + 1; tile j + 2 goes where tile j - 1 was
        const bool more = j + 1 < n_rt;
        CIRS_XTSTAMP(j, 0);
        if (j + 2 < n_rt) CIRS_STAGE(j + 2, bn == 2 ? 0 : bn + 1);
        const uint4* th = sH[buf] + lane;
        const uint4* tn = sH[more ? bn : buf] + lane;     // (after the last tile: a harmless re-read, those logits are not used)
        const int base = j * kTileM + 4 * hi;       // register r of the lane is row base + 8 (r >> 2) + (r & 3) of the range
        // ---- one branch-free block per iteration:  [D_j-1 beside dZ_j]  then  [L_j+1 beside split_j] ----
        f32x4 nl4[4], nc4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            nl4[g4] = *reinterpret_cast<const f32x4*>(&sNl[base + 8 * g4]);
            nc4[g4] = *reinterpret_cast<const f32x4*>(&sNc[base + 8 * g4]);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            hz[s4].h = __builtin_bit_cast(bf16x8, tn[(3 * s4) * 64]);
            hz[s4].m = __builtin_bit_cast(bf16x8, tn[(3 * s4 + 1) * 64]);
            hz[s4].l = __builtin_bit_cast(bf16x8, tn[(3 * s4 + 2) * 64]);
        }
        // D_j-1: dWa^T += H2^T dZ of the previous tile (operands in registers since the previous iteration)
        mfma_bf16x6_pair_b(hbP[0][0], hbP[1][0], plP[0], dw0, dw1);
        mfma_bf16x6_pair_b(hbP[0][1], hbP[1][1], plP[1], dw0, dw1);
        // dZ_j in place: d = z - lse FIRST (exact where it matters: the items that carry the row's probability have z ~ lse; one fma against
        // -lse log2 e loses |lse| 2^-24 there, 3x the error of a float32 soft-max on a sharp policy), p = exp2(d log2 e), dZ = -c_logp p.
        // Entropy clamp correction, low side (p < eps <=> d < log eps): p (log eps - d), branch-free (the high side is one item per row: the
        // merge handles it).  Plain (not packed) operations: a packed fp32 op costs several issue slots beside MFMAs on this part.
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = (acc[r] + acs[r]) - nl4[r >> 2][r & 3];
            const float p = __builtin_amdgcn_exp2f(d * kLog2e);
            elo = __builtin_fmaf(p, __builtin_fmaxf(kLogEps - d, 0.f), elo);
            acc[r] = nc4[r >> 2][r & 3] * p;
        }
        // (the schedule of the block above: every MFMA of D with its share of the dZ work behind it)
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
        }
        // The taken action's own term (+ c_logp on item a_r of row r) is applied to dZ itself, before the split: c (1 - p) is formed in one fp32
        // subtraction per element (a rank-one fp32 update of the accumulators instead was measured: the accumulator then carries the large
        // c H2[r] until the -c p H2[r] terms cancel it, and a sharp policy loses a digit).  Scalar test per tile (~9 % of the tiles enter).
        if ((amask >> j) & 1u) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int4 a4 = *reinterpret_cast<const int4*>(&sAct[base + 8 * g4]);
                const float f0 = a4.x == my_item ? nc4[g4][0] : 0.f, f1 = a4.y == my_item ? nc4[g4][1] : 0.f,
                            f2 = a4.z == my_item ? nc4[g4][2] : 0.f, f3 = a4.w == my_item ? nc4[g4][3] : 0.f;
                acc[4 * g4] -= f0; acc[4 * g4 + 1] -= f1; acc[4 * g4 + 2] -= f2; acc[4 * g4 + 3] -= f3;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) db += acc[r];      // (after the action's term: the sum never carries the large -c p of a row's own item alone)
        // split_j -> the B operand of D_j; the A operand (H2^T: lane = column, 8 rows in accumulator order) for the next iteration
        plP[0] = split8(acc, 0); plP[1] = split8(acc, 8);   // element j of k-step t: dZ[row acc_row(8 t + j, hi)][item lo]
        // L_j+1 (needs the rows read above)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = bias; acs[r] = 0.f; }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) mfma_bf16x6_split(hz[s4], zb[s4], acc, acs);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                hbP[c][t].h = __builtin_bit_cast(bf16x8, th[(12 + 3 * (2 * c + t)) * 64]);
                hbP[c][t].m = __builtin_bit_cast(bf16x8, th[(12 + 3 * (2 * c + t) + 1) * 64]);
                hbP[c][t].l = __builtin_bit_cast(bf16x8, th[(12 + 3 * (2 * c + t) + 2) * 64]);
            }
        // the schedule of this block: every MFMA with its share of the vector work behind it
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);
        }
        CIRS_XTSTAMP(j, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of tile j + 2 has landed (requested a whole iteration ago)
        lds_barrier();
        CIRS_XTSTAMP(j, 2);
        buf = bn;
    }
    // D of the last tile
    mfma_bf16x6_pair_b(hbP[0][0], hbP[1][0], plP[0], dw0, dw1);
    mfma_bf16x6_pair_b(hbP[0][1], hbP[1][1], plP[1], dw0, dw1);
    float ent = -elo;
    CIRS_XSTAMP(35);
#undef CIRS_STAGE
    // ---- results: the wave's dWa tile leaves through LDS as whole 256-byte item rows; dba; the workgroup's entropy correction ----
    if (!wave_ok) ent = 0.f;
    ent = wave_sum_f32_dpp(ent);
    if (lane == 0) sEnt[iw] = ent;
    db += __shfl_xor(db, 32, CIRS_WAVE);
    float* xt = reinterpret_cast<float*>(&sH[0][0]) + iw * (kTileN * kXStride);     // (every wave is past the last barrier: the staging buffers are free)
    float* xrow = xt + lo * kXStride + 4 * hi;        // lane = item lo; its registers are columns 32 c + 8 g4 + 4 hi + (0..3)
    if (wave_ok) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            *reinterpret_cast<f32x4*>(xrow + 8 * g4) = f32x4{dw0[4 * g4], dw0[4 * g4 + 1], dw0[4 * g4 + 2], dw0[4 * g4 + 3]};
            *reinterpret_cast<f32x4*>(xrow + 32 + 8 * g4) = f32x4{dw1[4 * g4], dw1[4 * g4 + 1], dw1[4 * g4 + 2], dw1[4 * g4 + 3]};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* slab = dwap + (size_t)rr * dwa_slab_stride(I);
        const bool full = tile0 + kTileN <= I;
#pragma unroll
        for (int q = 0; q < 8; ++q) {      // 4 item rows (1 KB of consecutive memory) per store instruction
            const int il = 4 * q + (lane >> 4), c4 = (lane & 15) * 4;
            const f32x4 t = *reinterpret_cast<const f32x4*>(xt + il * kXStride + c4);
            if (full || tile0 + il < I) *reinterpret_cast<f32x4*>(slab + (size_t)(tile0 + il) * kH + c4) = t;
        }
        if (hi == 0 && my_item < I) slab[(size_t)I * kH + my_item] = db;
    }
    lds_barrier();
    if (tid == 0) a.entw[rr * a.n_groups + g] = (sEnt[0] + sEnt[1]) + (sEnt[2] + sEnt[3]);
    CIRS_XSTAMP(36);
}

}  // namespace cirs
