// internal.h -- entry points shared between translation units of libcirs_hip.so (not part of the C ABI).
#pragma once
#include "common.h"
#include "policy_kernels.h"

namespace cirs {

// Optional tail of the tracker step: the policy trunk of the NEXT vector step on the state the tracker just produced
// (same wavefront, same fma chains as trunk_kernel) -- saves one launch per rollout step.
struct TrunkFuse {
    int on;
    cirs_policy_cfg cfg;
    cirs_policy_weights w;
    const uint8_t* skip;  // [n] rows whose env has finished: h2 = 0, value = 0 (what trunk_kernel writes for skipped rows)
    float* h2;            // [n, 64]
    float* value;         // [n] or null
};

// CUs of the current device (cached)
inline int device_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 256;
    }
    return cus;
}

// Item tiles per workgroup of a head kernel so that ALL its workgroups are co-resident (`wgs_per_cu` per CU) with equal tile
// counts: one round, no tail.  Never fewer than the 4 tiles the partial arrays are sized for (more tiles = fewer chunks).
inline int head_tiles_per_chunk(int n_item_tiles, int n_row_blocks, int wgs_per_cu) {
    const long slots = (long)wgs_per_cu * device_cu_count();
    const int tpc = (int)(((long)n_item_tiles * n_row_blocks + slots - 1) / slots);
    return tpc < 4 ? 4 : tpc;
}

inline HeadGrid sampler_grid(int n_items, int n_pad) {
    HeadGrid g;
    const int tiles = (n_items + kTileN - 1) / kTileN;
    g.n_row_blocks = (n_pad / kTileM + 3) / 4;
    g.tiles_per_chunk = head_tiles_per_chunk(tiles, g.n_row_blocks, kSamplerWgsPerCu);
    g.n_chunks = (tiles + g.tiles_per_chunk - 1) / g.tiles_per_chunk;
    g.grid_x = (g.n_chunks + 7) & ~7;
    if (g.grid_x > n_chunks_of(n_items)) g.grid_x = g.n_chunks;   // the partial arrays hold n_chunks_of(n_items) chunks
    return g;
}

// Optional head of the tracker step (fused rollout): the tail of the vector step for the same env row -- merge of the
// actor-head partials into action / log-prob, visited bit, env step, forced episode length -- runs in the wavefront that
// then appends the new position to the tracker: one launch per vector step less.
struct TailFuse {
    int on;
    cirs_env_cfg cfg;
    cirs_env_tables tab;
    cirs_env_state st;
    int n_pad, n_chunks;
    ActorPartialView pv;       // harness-noise mode: partials of actor_head_kernel, merged here
    int env_base;              // env id of row 0 of this launch (env groups on separate streams): state / noise / visited use env_base + j
    int pick_on;               // counter-based mode: chunk masses of actor_mass_kernel, chunk + item drawn here (two-level sampler)
    PickArgs pick;
    uint32_t* visited;
    int force_length, force_done;
    int slot_only;             // exact-redraw rollout: the tracker part only appends the input slot (the states come from the per-call prefix passes)
    int64_t* act_out;
    float* logp_out;
    double* rew_out;
    uint8_t* done_out;
    double* ctr_out;
};

// per-kernel timing hook (api.hip): ids 1 = head_bwd_fused_kernel, 2 = head_stats_kernel, 3 = actor_head_kernel (sampler)
bool prof_before(int kernel_id, hipStream_t s);
void prof_after(hipStream_t s);
#define CIRS_PROF_LAUNCH(ID, STREAM, ...)                       \
    do {                                                        \
        const bool prof_ = ::cirs::prof_before(ID, STREAM);     \
        __VA_ARGS__;                                            \
        if (prof_) ::cirs::prof_after(STREAM);                  \
    } while (0)

// chunks of 128 items one workgroup of actor_mass_kernel walks: the smallest count that puts every workgroup on its own CU in ONE
// round (C3: 84 chunks x 8 row blocks on 256 CUs -> 3; one chunk each = 672 workgroups pays the ~6 k-cycle prologue -- kernel
// arguments, hidden rows, first staged tile -- per chunk and runs 2.6 rounds: 1.73 vs 1.67 ms per collect).  Results do not depend on it.
inline int mass_chunks_per_wg(int n_chunks, int n_row_blocks) {
    const long units = (long)n_chunks * n_row_blocks;
    const int cus = device_cu_count();
    const int cpw = (int)((units + cus - 1) / cus);
    return cpw < 1 ? 1 : cpw;
}

// ---- packed weight image of the fused rollout's step kernel ----------------------------------------------------------------
// tracker_step_kernel runs one wavefront per env and lane o owns output feature o of every mat-vec: with the row-major torch
// layout W[o][k] every load instruction touches 64 different cache lines (one 16-byte piece of 64 rows), which the texture
// addresser serialises -- ~190 such instructions per step, four wavefronts per CU.  The image stores every matrix as
// [k/4][O][4] (lane o reads the float4 (k..k+3) of its row: consecutive lanes -> consecutive 16 bytes, 1 KB per instruction) and
// the small vectors contiguously.  Built once per cirs_rollout_steps call by pack_tracker_image (weights change between calls);
// the fma order of every dot product is unchanged.  Offsets in floats, all multiples of 4.
struct TrkImg {
    int gate_r, gate_p, gate_b;
    int in_p[CIRS_MAX_TRACKER_LAYERS], in_b[CIRS_MAX_TRACKER_LAYERS], out_p[CIRS_MAX_TRACKER_LAYERS], out_b[CIRS_MAX_TRACKER_LAYERS];
    int l1_p[CIRS_MAX_TRACKER_LAYERS], l1_b[CIRS_MAX_TRACKER_LAYERS], l2_p[CIRS_MAX_TRACKER_LAYERS], l2_b[CIRS_MAX_TRACKER_LAYERS];
    int ln[CIRS_MAX_TRACKER_LAYERS];   // norm1_w | norm1_b | norm2_w | norm2_b, 32 floats each
    int dec_p, dec_b;
    int w1_t, b1, w2_p, b2, wc, bc;    // policy trunk (w1 transposed [S][64]: one dword per lane and k)
    int total;
};
constexpr int64_t kTrkImgBytes = 256 * 1024;   // reserved at the end of the policy workspace (cirs_policy_workspace_bytes)
inline TrkImg trk_img_layout(int nlayers, int S) {
    TrkImg L{};
    int o = 0;
    auto take = [&](int n) { const int r = o; o += (n + 3) & ~3; return r; };
    L.gate_r = take(32); L.gate_p = take(32 * 32); L.gate_b = take(32);
    for (int l = 0; l < nlayers && l < CIRS_MAX_TRACKER_LAYERS; ++l) {
        L.in_p[l] = take(96 * 32); L.in_b[l] = take(96);
        L.out_p[l] = take(32 * 32); L.out_b[l] = take(32);
        L.l1_p[l] = take(128 * 32); L.l1_b[l] = take(128);
        L.l2_p[l] = take(32 * 128); L.l2_b[l] = take(32);
        L.ln[l] = take(4 * 32);
    }
    L.dec_p = take(32 * 32); L.dec_b = take(32);
    L.w1_t = take(S * 64); L.b1 = take(64); L.w2_p = take(64 * 64); L.b2 = take(64); L.wc = take(64); L.bc = take(4);
    L.total = o;
    return L;
}
// bf16 planes of the actor head for the sampler's chunk-mass kernels (policy_kernels.h: wa_rplanes_kernel): in front of the weight image at the end of the
// policy workspace (cirs_policy_workspace_bytes reserves both), rebuilt by every call that samples (the weights change between calls, never inside one)
inline uint4* ws_rplanes(void* workspace, int64_t workspace_bytes, int n_items) {
    const int64_t img_off = (workspace_bytes - kTrkImgBytes) & ~(int64_t)255;
    return (uint4*)((char*)workspace + ((img_off - (int64_t)ws_rplanes_bytes(n_items)) & ~(int64_t)255));
}
inline int build_rplanes(const float* wa, int n_items, uint4* planes, hipStream_t s) {
    hipLaunchKernelGGL(wa_rplanes_kernel, dim3(n_chunks_of(n_items) * kTilesPerChunk), dim3(256), 0, s, wa, n_items, planes);
    CIRS_CHECK_LAUNCH("wa_rplanes_kernel");
    return CIRS_OK;
}

// img <- the image of (w, pol); pol may be null (no trunk fusion).  In the same launch, on workgroups of their own: the sampler's planes of the actor head
// (planes non-null: build_rplanes) and env.reset for envs 0 .. n_env-1 (ecfg non-null: cirs_env_reset) -- what a collect sets up before its first step.
int pack_tracker_image(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_policy_weights* pol, int S, float* img,
                       hipStream_t s, const float* wa = nullptr, int n_items = 0, uint4* planes = nullptr, const cirs_env_cfg* ecfg = nullptr,
                       const cirs_env_state* est = nullptr, const int32_t* users = nullptr, int n_env = 0, int64_t* obs_scratch = nullptr);

int tracker_step_internal(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, cirs_tracker_state* st, const int32_t* users,
                          const int64_t* items, const double* rew, const int32_t* env_ids, const uint8_t* skip, int n, float* state_out,
                          long state_stride, const TrunkFuse* tf, hipStream_t s, const TailFuse* tail = nullptr, const float* img = nullptr);

// cirs_tracker_prefix_states with the policy trunk of the produced states in the same launch (exact-redraw rollout: one launch per call instead of two).
// *fused = 1 when the one-launch pass ran and applied `tf`; 0: the states are written, the caller runs trunk_kernel itself.
int tracker_prefix_states_trunk(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st, const int32_t* row_env,
                                const int32_t* row_t, const int32_t* offsets, const int32_t* lens, int32_t n_rows, float* state_out, int64_t state_stride,
                                void* workspace, int64_t workspace_bytes, void* stream, const TrunkFuse* tf, int* fused);

}  // namespace cirs
