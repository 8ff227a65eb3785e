// rollout.hip -- the Collector hot loop as a stream of launches with no host synchronisation (reference:
// core/collector.py:219-317).  Per vector step: policy trunk + MFMA actor head + merge (policy.hip), env step
// (env.hip), tracker decode step (tracker.hip); finished envs propagate as act = -1 so no compaction is needed and
// every env keeps its row (RNG keyed by env id -> results independent of which other envs are still alive).
#include "env_kernels.h"
#include "internal.h"
#include "policy_kernels.h"

namespace cirs {

__global__ __launch_bounds__(256) void mark_visited_kernel(const int64_t* __restrict__ act, int n, int n_items,
                                                           uint32_t* __restrict__ visited) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long a = act[j];
    if (a < 0) return;
    const int words = (n_items + 31) / 32;
    atomicOr(&visited[(size_t)j * words + (a >> 5)], 1u << (a & 31));
}

// core/collector.py:253-258: with force_length every env's done flag is replaced by (cnt_loop >= force_length)
__global__ __launch_bounds__(256) void force_done_kernel(uint8_t* __restrict__ st_done, uint8_t* __restrict__ done_row,
                                                         const int64_t* __restrict__ act, int n, int force_done) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (act[j] < 0) return;
    st_done[j] = (uint8_t)force_done;
    done_row[j] = (uint8_t)force_done;
}

// rows -> (raw uid, raw pid, feats, duration) of the chosen pair; finished rows (act < 0) score item 0 and are ignored
__global__ __launch_bounds__(256) void online_pairs_kernel(const int32_t* __restrict__ env_user, const int64_t* __restrict__ act, int n,
                                                           cirs_online_reward o) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long a = act[j] < 0 ? 0 : act[j];
    o.uid_buf[j] = o.raw_uid[env_user[j]];
    o.pid_buf[j] = o.raw_pid[a];
#pragma unroll
    for (int q = 0; q < 4; ++q) o.feat_buf[(size_t)j * 4 + q] = o.item_feats[(size_t)a * 4 + q];
    o.dur_buf[j] = o.item_dur[a];
}

constexpr int kMaxGroups = 4;
// sampler scratch of one env group (cirs_policy_workspace_bytes without its slack)
static inline int64_t group_ws_bytes(const cirs_policy_cfg* cfg, int n) {
    return (int64_t)(ws_h2_floats(n) + 5 * ws_partial_elems(n, cfg->n_items)) * 4;
}

}  // namespace cirs

static int rollout_impl(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                        const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w, cirs_tracker_state* trk_st,
                        const cirs_policy_cfg* pol_cfg, const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env,
                        int32_t t_begin, int32_t t_end, uint64_t seed, uint32_t rng_base, uint32_t* visited, int32_t force_length,
                        const cirs_online_reward* online, const float* gumbel, void* workspace, int64_t workspace_bytes, void* stream,
                        const cirs_redraw* redraw = nullptr, const int32_t* init_users = nullptr);

// Collector.reset_env + collect(n_episode = n_env) from ONE call (core/collector.py:123-134,147-367): env reset, the tracker's first position (from the packed
// weight image, the first step's trunk in its launch), the max_turn vector steps.  Every trajectory entry [t][env] is written (finished envs: act -1, done 1,
// rew / ctr 0), so the caller clears nothing; the tracker lengths restart at 1.
extern "C" int cirs_rollout_collect(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                                    const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w, cirs_tracker_state* trk_st,
                                    const cirs_policy_cfg* pol_cfg, const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env,
                                    const int32_t* users, uint64_t seed, uint32_t rng_base, uint32_t* visited, int32_t force_length, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    CIRS_REQUIRE(env_cfg && env_st && users && workspace && n_env > 0, "cirs_rollout_collect: null argument");
    CIRS_REQUIRE(workspace_bytes >= (int64_t)sizeof(int64_t) * n_env, "workspace too small");
    // (env.reset rides in the setup launch of rollout_impl; its obs ids -- the users -- land in the head of the workspace: scratch nobody reads)
    CIRS_REQUIRE(env_st->user && env_st->turn && env_st->done && env_st->hist_action && env_st->cum_reward, "env state has null field");
    return rollout_impl(env_cfg, env_tab, env_st, trk_cfg, trk_w, trk_st, pol_cfg, pol_w, traj, n_env, 0, env_cfg->max_turn, seed, rng_base, visited, force_length,
                        nullptr, nullptr, workspace, workspace_bytes, stream, nullptr, users);
}

extern "C" int cirs_rollout_steps_redraw(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                                         const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w, cirs_tracker_state* trk_st,
                                         const cirs_policy_cfg* pol_cfg, const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env,
                                         int32_t t_begin, int32_t t_end, uint64_t seed, uint32_t rng_base, const cirs_redraw* redraw,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
    CIRS_REQUIRE(redraw && redraw->row_env && redraw->row_t && redraw->offsets && redraw->lens && redraw->workspace, "cirs_rollout_steps_redraw: null field");
    return rollout_impl(env_cfg, env_tab, env_st, trk_cfg, trk_w, trk_st, pol_cfg, pol_w, traj, n_env, t_begin, t_end, seed, rng_base, nullptr, 0,
                        nullptr, nullptr, workspace, workspace_bytes, stream, redraw);
}

extern "C" int cirs_rollout_steps_noise(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                                        const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w, cirs_tracker_state* trk_st,
                                        const cirs_policy_cfg* pol_cfg, const cirs_policy_weights* pol_w, const cirs_traj* traj,
                                        int32_t n_env, int32_t t_begin, int32_t t_end, const float* gumbel, uint32_t* visited,
                                        int32_t force_length, void* workspace, int64_t workspace_bytes, void* stream) {
    CIRS_REQUIRE(gumbel, "cirs_rollout_steps_noise: gumbel is null (use cirs_rollout_steps for the counter-based sampler)");
    return rollout_impl(env_cfg, env_tab, env_st, trk_cfg, trk_w, trk_st, pol_cfg, pol_w, traj, n_env, t_begin, t_end, 0, 0, visited,
                        force_length, nullptr, gumbel, workspace, workspace_bytes, stream);
}

extern "C" int cirs_rollout_steps(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                                  const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w,
                                  cirs_tracker_state* trk_st, const cirs_policy_cfg* pol_cfg,
                                  const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env,
                                  int32_t t_begin, int32_t t_end, uint64_t seed, uint32_t rng_base, uint32_t* visited,
                                  int32_t force_length, void* workspace, int64_t workspace_bytes, void* stream) {
    return cirs_rollout_steps_online(env_cfg, env_tab, env_st, trk_cfg, trk_w, trk_st, pol_cfg, pol_w, traj, n_env, t_begin, t_end,
                                     seed, rng_base, visited, force_length, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int cirs_rollout_steps_online(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                                  const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w,
                                  cirs_tracker_state* trk_st, const cirs_policy_cfg* pol_cfg,
                                  const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env,
                                  int32_t t_begin, int32_t t_end, uint64_t seed, uint32_t rng_base, uint32_t* visited,
                                  int32_t force_length, const cirs_online_reward* online, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    return rollout_impl(env_cfg, env_tab, env_st, trk_cfg, trk_w, trk_st, pol_cfg, pol_w, traj, n_env, t_begin, t_end, seed, rng_base,
                        visited, force_length, online, nullptr, workspace, workspace_bytes, stream);
}

// gumbel (nullable): harness-supplied sampler noise [max_turn][n_env][n_items] (g = -log q), row t used at vector step t
static int rollout_impl(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                        const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w, cirs_tracker_state* trk_st,
                        const cirs_policy_cfg* pol_cfg, const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env,
                        int32_t t_begin, int32_t t_end, uint64_t seed, uint32_t rng_base, uint32_t* visited, int32_t force_length,
                        const cirs_online_reward* online, const float* gumbel, void* workspace, int64_t workspace_bytes, void* stream,
                        const cirs_redraw* redraw, const int32_t* init_users) {
    using namespace cirs;
    cirs_env_tables tab_local;
    if (online) {
        CIRS_REQUIRE(env_tab && online->cfg && online->w && online->raw_uid && online->raw_pid && online->item_feats && online->item_dur &&
                     online->pred_minmax && online->uid_buf && online->pid_buf && online->feat_buf && online->dur_buf && online->pred_buf,
                     "online reward: null field");
        tab_local = *env_tab;
        tab_local.pred_online = online->pred_buf;
        tab_local.pred_minmax = online->pred_minmax;
        env_tab = &tab_local;
    }
    CIRS_REQUIRE(env_cfg && env_tab && env_st && trk_cfg && trk_w && trk_st && pol_cfg && pol_w && traj, "null argument");
    CIRS_REQUIRE(traj->obs && traj->act && traj->rew && traj->done && traj->logp && traj->value && traj->ctr, "trajectory pointer null");
    CIRS_REQUIRE(n_env > 0 && t_begin >= 0 && t_end <= env_cfg->max_turn && t_begin <= t_end, "bad step range");
    CIRS_REQUIRE(trk_cfg->n_env == n_env, "tracker n_env mismatch");
    CIRS_REQUIRE(trk_cfg->dim_state == pol_cfg->dim_state, "tracker/policy dim_state mismatch");
    CIRS_REQUIRE(trk_cfg->max_len >= env_cfg->max_turn + 1, "tracker max_len < max_turn + 1");
    const long B = n_env, S = trk_cfg->dim_state;
    hipStream_t s = (hipStream_t)stream;
    if (online) {  // per-step user-model scoring sits between the policy and the env step: unfused launch sequence
        for (int t = t_begin; t < t_end; ++t) {
            float* obs_t = traj->obs + (size_t)t * B * S;
            float* obs_n = traj->obs + (size_t)(t + 1) * B * S;
            int64_t* act_t = traj->act + (size_t)t * B;
            double* rew_t = traj->rew + (size_t)t * B;
            uint8_t* done_t = traj->done + (size_t)t * B;
            // policy(obs_t): finished envs (env_st->done) are skipped and get act = -1
            if (int rc = cirs_actor_sample(pol_cfg, pol_w, obs_t, S, n_env, gumbel ? gumbel + (size_t)t * B * pol_cfg->n_items : nullptr, seed, rng_base + (uint32_t)t, nullptr,
                                           visited, env_st->done, act_t, traj->logp + (size_t)t * B,
                                           traj->value + (size_t)t * B, workspace, workspace_bytes, stream))
                return rc;
            if (visited) {
                hipLaunchKernelGGL(mark_visited_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, s, act_t, n_env,
                                   pol_cfg->n_items, visited);
                CIRS_CHECK_LAUNCH("mark_visited_kernel");
            }
            // score the chosen (user, item) pairs with the DeepFM user model
            hipLaunchKernelGGL(online_pairs_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, s, env_st->user, act_t, n_env, *online);
            CIRS_CHECK_LAUNCH("online_pairs_kernel");
            if (int rc = cirs_deepfm_forward(online->cfg, online->w, online->uid_buf, online->pid_buf, online->feat_buf, online->dur_buf,
                                             n_env, online->pred_buf, stream))
                return rc;
            // env.step: obs_next id == action, so the int64 obs row doubles as scratch we do not keep
            if (int rc = cirs_env_step(env_cfg, env_tab, env_st, act_t, nullptr, n_env, (int64_t*)workspace, rew_t, done_t,
                                       traj->ctr + (size_t)t * B, nullptr, stream))
                return rc;
            if (force_length > 0) {
                hipLaunchKernelGGL(force_done_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, s, env_st->done, done_t, act_t,
                                   n_env, (t + 1 >= force_length) ? 1 : 0);
                CIRS_CHECK_LAUNCH("force_done_kernel");
            }
            // preprocess_fn(obs_next, rew): tracker appends one position for every env that acted this step
            if (int rc = cirs_tracker_step(trk_cfg, trk_w, trk_st, act_t, rew_t, nullptr, nullptr, n_env, obs_n, S, stream))
                return rc;
        }
        return CIRS_OK;
    }
    // ---- fused sequence: 2 launches per vector step ------------------------------------------------------------
    //   actor_head_kernel           MFMA head + Gumbel-max partials            (trunk of obs_t already in the workspace)
    //   tracker_step_kernel         one wavefront per env: merge -> act/logp, visited bit, env step, forced length (TailFuse);
    //                               tracker decode step; the policy trunk of obs_{t+1} (TrunkFuse)
    CIRS_REQUIRE(pol_cfg->hidden == kH && pol_cfg->dim_state == S && pol_cfg->n_items == env_cfg->n_items, "policy/env/tracker shape mismatch");
    CIRS_REQUIRE(pol_w->w1 && pol_w->b1 && pol_w->w2 && pol_w->b2 && pol_w->wa && pol_w->ba && pol_w->wc && pol_w->bc, "policy weight pointer null");
    CIRS_REQUIRE(workspace_bytes >= cirs_policy_workspace_bytes(pol_cfg, n_env), "workspace too small");
    CIRS_REQUIRE(env_tab->item_cats && (env_tab->normed_mat || !env_cfg->simulated) && (env_tab->mat || env_cfg->simulated), "env tables incomplete");
    if (t_begin >= t_end) return CIRS_OK;
    // ---- env groups on separate streams ---------------------------------------------------------------------------------------
    // The step kernel (one wavefront per env) is latency-bound: a chain of ~25 dependent stages that leaves the matrix / vector pipes
    // idle, while the chunk-mass kernel of the sampler is throughput-bound.  Envs are independent (noise, masks and dropout are
    // keyed by the env id), so the envs are split into G groups whose launch sequences run on G streams: one group's step kernel
    // overlaps another group's mass kernel.  Results are identical to G = 1 (tests/test_gpu_rollout.py).
    int G = 1;
    if (!gumbel) {
        static const int forced = [] { const char* ev = getenv("CIRS_ROLLOUT_GROUPS"); return ev ? atoi(ev) : 0; }();   // thread-safe magic static
        // measured at C3 (1024 envs): rollout alone 1.96 ms (G = 1), 1.81 ms (2), 1.87 ms (4); the whole step (rollout + update)
        // 8.82 / 8.93 / 10.46 ms -- the additional launches and the fork / join of the streams cost more host and queue time than the
        // overlap returns, so one group is the default and CIRS_ROLLOUT_GROUPS opts in
        G = forced > 0 ? forced : 1;
        if (G > kMaxGroups) G = kMaxGroups;
        if (redraw) G = 1;
    }
    int n_g = ((n_env + G - 1) / G + 127) / 128 * 128;       // rows per group: whole 128-row blocks of the mass kernel
    if (G > 1) {   // room for one sampler workspace per group?
        const int64_t per = (group_ws_bytes(pol_cfg, n_g) + 255) & ~(int64_t)255;
        if (per * G > workspace_bytes - kTrkImgBytes - (int64_t)ws_rplanes_bytes(pol_cfg->n_items) - 512) { G = 1; n_g = n_env; }
    } else {
        n_g = n_env;
    }
    // packed weight image of the step kernel (tracker + policy trunk), rebuilt per call (the weights change between calls) on the
    // caller's stream, before the group streams fork from it
    float* img = (float*)((char*)workspace + ((workspace_bytes - kTrkImgBytes) & ~(int64_t)255));
    // ... the fp16 planes of the actor head for the chunk-mass kernels, likewise once per call, and (cirs_rollout_collect) env.reset: one launch
    uint4* rplanes = ws_rplanes(workspace, workspace_bytes, pol_cfg->n_items);
    if (int rc = pack_tracker_image(trk_cfg, trk_w, pol_w, S, img, s, pol_w->wa, pol_cfg->n_items, gumbel ? nullptr : rplanes, init_users ? env_cfg : nullptr, env_st,
                                    init_users, n_env, (int64_t*)workspace))
        return rc;
    // group streams / events: one set per (host thread, device) -- a stream belongs to the device that was current when it was
    // created, and two host threads driving rollouts concurrently (the virtual-rank tests) must not share the event array
    constexpr int kMaxDevices = 16;
    struct GroupQueues { hipStream_t gs[kMaxGroups]; hipEvent_t gev[kMaxGroups + 1]; };
    static thread_local GroupQueues tl_queues[kMaxDevices] = {};
    int cur_dev = 0;
    if (G > 1) {
        CIRS_HIP(hipGetDevice(&cur_dev));
        CIRS_REQUIRE(cur_dev >= 0 && cur_dev < kMaxDevices, "CIRS_ROLLOUT_GROUPS > 1 supports device ids below 16");
    }
    hipStream_t* gs = tl_queues[cur_dev].gs;
    hipEvent_t* gev = tl_queues[cur_dev].gev;
    if (G > 1) {
        for (int g = 1; g < G; ++g)
            if (!gs[g]) CIRS_HIP(hipStreamCreateWithFlags(&gs[g], hipStreamNonBlocking));
        for (int g = 0; g <= G && g <= kMaxGroups; ++g)
            if (!gev[g]) CIRS_HIP(hipEventCreateWithFlags(&gev[g], hipEventDisableTiming));
        CIRS_HIP(hipEventRecord(gev[0], s));
        for (int g = 1; g < G; ++g) CIRS_HIP(hipStreamWaitEvent(gs[g], gev[0], 0));
    }
    const int64_t ws_per = (group_ws_bytes(pol_cfg, n_g) + 255) & ~(int64_t)255;
    const int n_mass_chunks = n_chunks_of(pol_cfg->n_items);
    struct Group { int base, n, n_pad; float* h2; ActorPartialView pv; HeadGrid hg; int cpw; hipStream_t st; };
    Group grp[kMaxGroups];
    int n_groups = 0;
    for (int g = 0; g < G; ++g) {
        const int base = g * n_g;
        if (base >= n_env) break;
        Group& q = grp[n_groups++];
        q.base = base; q.n = min(n_g, n_env - base); q.n_pad = n_pad_of(q.n);
        void* wsg = (char*)workspace + (size_t)g * ws_per;
        q.h2 = (float*)wsg;
        q.pv = partial_view(wsg, q.n, pol_cfg->n_items);
        q.hg = sampler_grid(pol_cfg->n_items, q.n_pad);
        q.cpw = mass_chunks_per_wg(n_mass_chunks, q.hg.n_row_blocks);
        q.st = g == 0 ? s : gs[g];
    }
    // logit store (one group, counter-based sampler, small env counts): behind the group's sampler scratch
    float* zstore = nullptr;
    if (n_groups == 1 && !gumbel && ws_zstore_floats(n_env, pol_cfg->n_items) > 0) {
        const char* ev = getenv("CIRS_ROLLOUT_ZSTORE");     // read per call (like CIRS_PPO_MERGE_KERNEL): a test may flip it between collects
        if (ev ? atoi(ev) != 0 : true) zstore = (float*)((char*)workspace + ws_per);
    }
    const char* ms_ev = getenv("CIRS_ROLLOUT_MASS_SMALL");   // per call as well
    const bool mass_small = (ms_ev ? atoi(ms_ev) != 0 : true) && n_groups == 1 && !gumbel && grp[0].n_pad <= 128;
    const uint8_t* done_all = (const uint8_t*)env_st->done;
    // Exact-redraw dropout (the reference's procedure, core/state_tracker.py:170-186,243-246): the state of vector step t is NOT the cached decode's -- it is
    // ONE batched causal pass over positions 0 .. t of every env with the masks of build_state call t (cirs_tracker_prefix_states, key = the collect's key with
    // pseudo-env ids env_base0 + t * env_stride + e), followed by the trunk of that state; the step kernel keeps writing the input slots and its own state is
    // overwritten by the next call's pass.  (cirs_hip/redraw.py ran this loop from Python: ~18 launches and torch ops per step, host-bound.)
    // with_trunk: the policy trunk of the call's states (vector step t acts on them) rides in the prefix pass's launch when that is one launch, else trunk_kernel follows
    auto redraw_state = [&](int t, bool with_trunk) -> int {
        cirs_tracker_cfg cfg_t = *trk_cfg;
        cfg_t.dropout_seed = redraw->dropout_seed;
        cfg_t.drop_env_base = (int32_t)(redraw->env_base0 + (int64_t)t * redraw->env_stride);
        const size_t start = (size_t)B * t * (t + 1) / 2;
        TrunkFuse tf{};
        if (with_trunk) { tf.on = 1; tf.cfg = *pol_cfg; tf.w = *pol_w; tf.skip = done_all; tf.h2 = grp[0].h2; tf.value = traj->value + (size_t)t * B; }
        int fused = 0;
        if (int rc = tracker_prefix_states_trunk(&cfg_t, trk_w, trk_st, redraw->row_env + start, redraw->row_t + start, redraw->offsets + (size_t)t * B,
                                                 redraw->lens + (size_t)t * B, (int32_t)(B * (t + 1)), traj->obs + (size_t)t * B * S, S, redraw->workspace,
                                                 redraw->workspace_bytes, stream, &tf, &fused))
            return rc;
        if (with_trunk && !fused) {
            hipLaunchKernelGGL(trunk_kernel, dim3(cdiv(n_env, 4)), dim3(256), 0, s, *pol_cfg, *pol_w, traj->obs + (size_t)t * B * S, (long)S, n_env, done_all, grp[0].h2,
                               traj->value + (size_t)t * B, (float*)nullptr);
            CIRS_CHECK_LAUNCH("trunk_kernel");
        }
        return CIRS_OK;
    };
    // cirs_rollout_collect: the tracker's first position (Collector.reset_env's preprocess_fn(obs = ...)) from the packed image, with the trunk of the first
    // vector step in its launch (one group) -- cirs_tracker_init's row-major weight reads were 23 us per collect, the separate trunk launch 6
    bool first_trunk_done = false;
    if (init_users) {
        TrunkFuse tf0{};
        if (n_groups == 1 && !redraw) {
            tf0.on = 1; tf0.cfg = *pol_cfg; tf0.w = *pol_w; tf0.skip = nullptr; tf0.h2 = grp[0].h2; tf0.value = traj->value + (size_t)t_begin * B;
            first_trunk_done = true;
        }
        if (int rc = tracker_step_internal(trk_cfg, trk_w, trk_st, init_users, nullptr, nullptr, nullptr, nullptr, n_env, traj->obs + (size_t)t_begin * B * S, S, &tf0, s,
                                           nullptr, img))
            return rc;
    }
    if (redraw) { if (int rc = redraw_state(t_begin, true)) return rc; }
    // trunk of the first step of this call (later ones ride on the tracker step)
    for (int gi = 0; gi < n_groups && !redraw && !first_trunk_done; ++gi) {
        const Group& q = grp[gi];
        hipLaunchKernelGGL(trunk_kernel, dim3(cdiv(q.n, 4)), dim3(256), 0, q.st, *pol_cfg, *pol_w,
                           traj->obs + ((size_t)t_begin * B + q.base) * S, (long)S, q.n, done_all + q.base, q.h2,
                           traj->value + (size_t)t_begin * B + q.base, (float*)nullptr);
    }
    CIRS_CHECK_LAUNCH("trunk_kernel");
    for (int t = t_begin; t < t_end; ++t) {
        const float* gum_t = gumbel ? gumbel + (size_t)t * B * pol_cfg->n_items : (const float*)nullptr;
        for (int gi = 0; gi < n_groups; ++gi) {
            const Group& q = grp[gi];
            float* obs_n = traj->obs + ((size_t)(t + 1) * B + q.base) * S;
            int64_t* act_t = traj->act + (size_t)t * B + q.base;
            double* rew_t = traj->rew + (size_t)t * B + q.base;
            uint8_t* done_t = traj->done + (size_t)t * B + q.base;
            if (gum_t) {   // harness-supplied noise: plain Gumbel-max over the catalogue (reference-recorded fixtures); one group
                CIRS_PROF_LAUNCH(3, q.st, hipLaunchKernelGGL(actor_head_kernel, dim3(q.hg.grid_x, q.hg.n_row_blocks), dim3(256), 0, q.st, *pol_cfg,
                                                             pol_w->wa, pol_w->ba, (const float*)q.h2, q.n, gum_t, seed,
                                                             rng_base + (uint32_t)t, (const int32_t*)nullptr, (const uint32_t*)visited,
                                                             done_all, q.pv, q.n_pad, q.hg.tiles_per_chunk));
            } else if (mass_small) {   // few envs: one workgroup per chunk, one wave per (row tile, item tile)
                CIRS_PROF_LAUNCH(3, q.st, hipLaunchKernelGGL(actor_mass_small_kernel, dim3(n_mass_chunks), dim3(q.n_pad / kTileM * 256), 0, q.st, *pol_cfg, (const uint4*)rplanes,
                                                             pol_w->ba, (const float*)q.h2, q.n, (const uint32_t*)visited, done_all + q.base, q.pv.m, q.n_pad, q.base,
                                                             zstore));
            } else {       // counter-based sampler: chunk log-masses now, chunk + item draws in the tail of the step kernel
                CIRS_PROF_LAUNCH(3, q.st, hipLaunchKernelGGL(actor_mass_kernel, dim3(cdiv(n_mass_chunks, q.cpw), q.hg.n_row_blocks), dim3(kMassThreads), 0,
                                                             q.st, *pol_cfg, (const uint4*)rplanes, pol_w->ba, (const float*)q.h2, q.n, (const int32_t*)nullptr,
                                                             (const uint32_t*)visited, done_all + q.base, q.pv.m, q.n_pad, q.cpw, 0, 0, q.base, zstore));
            }
            CIRS_CHECK_LAUNCH("sampler kernel");
            TrunkFuse tf{};
            if (t + 1 < t_end && !redraw) {
                tf.on = 1; tf.cfg = *pol_cfg; tf.w = *pol_w; tf.skip = done_all + q.base; tf.h2 = q.h2;
                tf.value = traj->value + (size_t)(t + 1) * B + q.base;
            }
            // preprocess_fn(obs_next, rew): the tracker appends one position for every env that acted this step
            // ... in the same launch as the tail of this step: action / logp, visited bit, env step, forced length
            TailFuse tl{};
            tl.on = 1; tl.cfg = *env_cfg; tl.tab = *env_tab; tl.st = *env_st; tl.n_pad = q.n_pad; tl.n_chunks = q.hg.n_chunks; tl.pv = q.pv;
            tl.env_base = q.base;
            tl.visited = visited; tl.force_length = force_length;
            if (!gum_t) {
                tl.pick_on = 1;
                tl.pick = PickArgs{q.pv.m, q.n_pad, n_mass_chunks, pol_w->wa, pol_w->ba, q.h2, visited, pol_cfg->n_items, 0, 0, seed, rng_base + (uint32_t)t, zstore};
            }
            tl.force_done = (t + 1 >= force_length) ? 1 : 0;
            // (exact redraw: obs_{t+1} is overwritten by call t + 1's prefix pass below -- under the same condition -- so the cached decode is not run)
            tl.slot_only = redraw && (t + 1 < t_end || t + 1 < trk_cfg->max_len) && !getenv("CIRS_REDRAW_FULL_DECODE") ? 1 : 0;
            tl.act_out = act_t; tl.logp_out = traj->logp + (size_t)t * B + q.base; tl.rew_out = rew_t; tl.done_out = done_t;
            tl.ctr_out = traj->ctr + (size_t)t * B + q.base;
            if (int rc = tracker_step_internal(trk_cfg, trk_w, trk_st, nullptr, act_t, rew_t, nullptr, nullptr, q.n, obs_n, S, &tf, q.st, &tl, img))
                return rc;
            if (redraw && (t + 1 < t_end || t + 1 < trk_cfg->max_len)) {      // the state of call t + 1 (the last one: obs_next of the final step)
                if (int rc = redraw_state(t + 1, t + 1 < t_end)) return rc;
            }
        }
    }
    if (n_groups > 1) {
        for (int gi = 1; gi < n_groups; ++gi) {
            CIRS_HIP(hipEventRecord(gev[gi], grp[gi].st));
            CIRS_HIP(hipStreamWaitEvent(s, gev[gi], 0));
        }
    }
    return CIRS_OK;
}

#ifdef CIRS_MASS_PROF
extern "C" int cirs_debug_mass_prof(unsigned long long* out_host32) {
    return hipMemcpyFromSymbol(out_host32, HIP_SYMBOL(cirs::g_mass_prof), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif
