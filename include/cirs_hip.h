/* cirs_hip.h -- C ABI of libcirs_hip.so: the MI355X (gfx950) implementation of the CIRS rollout + PPO hot path.
 *
 * The reference (chongminggao/CIRS-codes) is pure Python and has no FFI; the drop-in boundary is the set of
 * duck-typed Python protocols listed in SURVEY.md §8(b).  This header is the native seam underneath those
 * protocols: every entry point below replaces the chain of Python/NumPy/pandas/torch calls cited next to it
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes stub that binds it.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  All data pointers are DEVICE (HBM) pointers owned by the
 *     caller unless the name ends in _h (host).  The library allocates nothing that outlives a call, except
 *     the small launch-time scratch documented per function.
 *   - `stream` is a hipStream_t passed as void*; every call is stream-ordered and returns immediately.
 *   - return value: 0 = ok, <0 = error (CIRS_E_*); cirs_last_error() gives a thread-local message.
 *   - ids are env-encoded (LabelEncoder positions) unless stated; tables are row-major.
 */
#ifndef CIRS_HIP_H
#define CIRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CIRS_OK 0
#define CIRS_E_INVALID (-1) /* bad argument (null pointer, size out of range)          */
#define CIRS_E_LAUNCH (-2)  /* HIP launch / runtime error                              */
#define CIRS_E_UNSUPPORTED (-3)

#define CIRS_MAX_CATS_PER_ITEM 4 /* KuaiRec item_categories.json has feat0..feat3 (kuaishouEnv.py:92) */
#define CIRS_CAT_NONE 0xFFu

const char* cirs_last_error(void);
int cirs_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Environment: batched SimulatedEnv(KuaishouEnv)
 * replaces  core/env/simulatedEnv/simulated_env.py:111-193  (SimulatedEnv.step/_compute_exposure_effect/
 *           _compute_pred_reward/_add_action_to_history/_reset_history)
 *           environments/KuaishouRec/env/kuaishouEnv.py:161-231 (KuaishouEnv.step/_determine_whether_to_leave/reset)
 *           core/util.py:21-54 (compute_action_distance, compute_exposure, clip0)
 *           tianshou/env/venvs.py:175-252 (the serial DummyVectorEnv loop)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct cirs_env_cfg {
    int32_t n_users;           /* U = mat.shape[0]                                                            */
    int32_t n_items;           /* I = mat.shape[1]                                                            */
    int32_t max_turn;          /* KuaishouEnv.max_turn (kuaishouEnv.py:36)                                    */
    int32_t num_leave_compute; /* N, exit-rule window (kuaishouEnv.py:56,203)                                 */
    int32_t leave_threshold;   /* leave iff count[c] > leave_threshold (kuaishouEnv.py:212)                   */
    int32_t version;           /* 1: r/(1+e*)   2: r - e*   (simulated_env.py:102-107)                        */
    int32_t use_exposure;      /* SimulatedEnv.use_exposure_intervention (simulated_env.py:118)               */
    int32_t has_ab;            /* alpha_u/beta_i given (simulated_env.py:157)                                 */
    int32_t dist_mode;         /* 0: dense I x I float64 table (df_dist_small); 1: on-the-fly 1/Jaccard       */
    int32_t simulated;         /* 1: SimulatedEnv reward; 0: bare KuaishouEnv (reward = mat[u,a]) test envs   */
    double tau;                /* compute_exposure: tau <= 0 -> 0 (util.py:42-44)                             */
    double gamma_exposure;     /* simulated_env.py:166                                                        */
    double r_decay;            /* simulated_env.py:130-132                                                    */
} cirs_env_cfg;

typedef struct cirs_env_tables { /* read-only, shared by all envs */
    const double* mat;         /* [U,I] real watch ratio  (kuaishouEnv.py:172)                                */
    const double* normed_mat;  /* [U,I] min-max normalised DeepFM prediction (simulated_env.py:100)           */
    const double* dist;        /* [I,I] 1/Jaccard, +inf when disjoint (util.py:36); NULL when dist_mode==1   */
    const uint32_t* item_cats; /* [I] four u8 category ids per item, CIRS_CAT_NONE padded (kuaishouEnv.py:49)*/
    const double* alpha_env;   /* [U] alpha_u[raw uid of env user] widened to f64 (simulated_env.py:158-160)  */
    const double* beta_env;    /* [I] beta_i[raw pid of env item]                                             */
    /* online-reward mode (catalogues too large for a U x I table, BASELINE configs[4]): when pred_online != NULL the
     * predicted reward of row j is (pred_online[j] - pred_minmax[0]) / (pred_minmax[1] - pred_minmax[0]) in float64
     * instead of normed_mat[u,a] -- the reference's own online variant (simulated_env.py:88-98, commented out there)
     * with the normalisation of compute_normed_reward (kuaishouEnv.py:139-143) applied per pair. */
    const float* pred_online;  /* [n] raw DeepFM score of (user of row j, action of row j), or NULL              */
    const float* pred_minmax;  /* [2] global {min, max} of the raw scores                                        */
} cirs_env_tables;

typedef struct cirs_env_state { /* mutable, SoA, one entry per env (B envs) */
    int32_t* user;        /* [B]    cur_user                                                                  */
    int32_t* turn;        /* [B]    total_turn                                                                */
    uint8_t* done;        /* [B]    episode finished (stepping a finished env is a no-op)          */
    int32_t* hist_action; /* [B,T]  history_action / sequence_action                                          */
    double* cum_reward;   /* [B]    cum_reward of the reward actually returned                                */
} cirs_env_state;

/* reset envs `env_ids[0..n)` (NULL = 0..n-1) to users[0..n) (kuaishouEnv.py:182-190; the user draw itself is
 * the caller's: the reference uses an unseeded random.randint).  obs_out[n] (nullable) = user id. */
int cirs_env_reset(const cirs_env_cfg* cfg, cirs_env_state* st, const int32_t* users, const int32_t* env_ids,
                   int32_t n, int64_t* obs_out, void* stream);

/* one vector step.  actions[n] are env-encoded item ids; outputs are indexed like env_ids (position j).
 *   obs_out[n]  int64   state = last action (kuaishouEnv.py:147-153)
 *   rew_out[n]  double  reward
 *   done_out[n] uint8   leave OR t >= max_turn-1 (kuaishouEnv.py:167-169)
 *   ctr_out[n]  double  info['CTR'] = cum_reward/total_turn/10 (simulated_env.py:145); bare env: cum_reward
 *   expo_out[n] double  (nullable) exposure_gamma e* stored in history_exposure[t] (simulated_env.py:166,190) */
int cirs_env_step(const cirs_env_cfg* cfg, const cirs_env_tables* tab, cirs_env_state* st, const int64_t* actions,
                  const int32_t* env_ids, int32_t n, int64_t* obs_out, double* rew_out, uint8_t* done_out,
                  double* ctr_out, double* expo_out, void* stream);

/* dense 1/Jaccard distance table from packed categories: replaces core/util.py:225-273 (get_distance_mat /
 * get_similarity_mat restricted to the env items).  dist_out is [I,I] float64. */
int cirs_dist_jaccard(const uint32_t* item_cats, int32_t n_items, double* dist_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * State tracker: incremental (KV-cached) StateTrackerTransformer
 * replaces  core/state_tracker.py:89-115 (get_embedding), :170-186 (forward), :188-250 (build_state)
 *           torch.nn.TransformerEncoder(2 x post-norm TransformerEncoderLayer, relu, eps 1e-5), eval mode
 * The reference re-runs the causal transformer over the whole prefix every step (O(L^2) per step); because the
 * mask is causal and only output[-1] is used, caching each layer's K/V per position is exact.
 * Fixed dims of this build: dim_model == 32, d_hid == 128, dim_state <= 32, nhead in {1,2,4,8}, nlayers <= 4.
 * ---------------------------------------------------------------------------------------------------------- */
#define CIRS_MAX_TRACKER_LAYERS 4

typedef struct cirs_tracker_cfg {
    int32_t n_users, n_items;
    int32_t dim_model; /* D = 32 */
    int32_t dim_state; /* S = 20 */
    int32_t nhead;     /* 4 */
    int32_t d_hid;     /* 128 */
    int32_t nlayers;   /* 2 */
    int32_t max_len;   /* MAX_TURN + 1 (state_tracker.py:144) */
    int32_t n_env;     /* B: leading dimension of the state arrays */
    /* dropout (production mode; 0 = off, the mode every reference-recorded fixture uses).  The reference runs the tracker with
     * nn.Dropout(0.1) live in rollout, test and backward (core/state_tracker.py:155-156,176; never eval(), CIRS-RL-kuaishou.py:235-243).
     * Masks are counter-based: a pure function of (dropout_seed, drop_env_base + env, position, layer, site, element), see csrc/rng.h. */
    float dropout_p;
    int32_t drop_env_base; /* added to the local env index: global env id of the job (rank * n_env), same in forward and backward */
    uint64_t dropout_seed; /* changes per collect (seed, collect counter) */
} cirs_tracker_cfg;

typedef struct cirs_tracker_layer { /* names: transformer_encoder.layers.<l>.*  */
    const float *in_proj_w, *in_proj_b;   /* [3D,D], [3D] */
    const float *out_proj_w, *out_proj_b; /* [D,D], [D]   */
    const float *lin1_w, *lin1_b;         /* [H,D], [H]   */
    const float *lin2_w, *lin2_b;         /* [D,H], [D]   */
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b; /* [D] */
} cirs_tracker_layer;

typedef struct cirs_tracker_weights {
    const float* emb_user;                 /* embedding_dict.feat_user.weight [U,D] */
    const float* emb_item;                 /* embedding_dict.feat_item.weight [I,D] */
    const float *ffn_user_w, *ffn_user_b;  /* [D,D], [D]   */
    const float *gate_w, *gate_b;          /* fnn_gate [D,1+D] (input order [r, a], state_tracker.py:240), [D] */
    const float* pe;                       /* pos_encoder.pe [max_len, D] */
    cirs_tracker_layer layer[CIRS_MAX_TRACKER_LAYERS];
    const float *dec_w, *dec_b;            /* decoder [S,D], [S] */
} cirs_tracker_weights;

typedef struct cirs_tracker_state {
    float* x_hist;  /* [B, max_len, D]  self.data: slot 0 = user projection, slot k = gated action (state_tracker.py:198,215,242) */
    float* kcache;  /* nlayers * B * max_len * D floats; per (layer, env) laid out [D/4][max_len][4] (private to the library) */
    float* vcache;  /* [nlayers, B, max_len, D] */
    int32_t* len;   /* [B] self.len_data */
} cirs_tracker_state;

/* build_state(obs=users): len=1, slot 0, s0 -> state_out[j*state_stride + 0..S) for j in [0,n). */
int cirs_tracker_init(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, cirs_tracker_state* st,
                      const int32_t* users, const int32_t* env_ids, int32_t n, float* state_out,
                      int64_t state_stride, void* stream);
/* build_state(obs_next=items, rew): append one position, return s_t.  `rew` is the env's float64 reward (cast to
 * float32 like FloatTensor(rew), state_tracker.py:97).  state_out row stride = state_stride floats.
 * skip (nullable, [n]): rows with skip != 0 are left untouched (finished envs). */
int cirs_tracker_step(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, cirs_tracker_state* st,
                      const int64_t* items, const double* rew, const int32_t* env_ids, const uint8_t* skip, int32_t n,
                      float* state_out, int64_t state_stride, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Policy: shared trunk + actor head over the full catalogue + fused sampler + critic
 * replaces  tianshou/utils/net/common.py:87-92,184-197 (MLP/Net), utils/net/discrete.py:56-67 (Actor), :113-114 (Critic)
 *           core/policy/ppo.py:111-163 (PPOPolicy.forward: softmax, optional mask of recommended ids, Categorical.sample)
 *           core/policy/utils.py:30-58 (removed_recommended_id_from_embedding)
 * The B x I probability matrix is never written to HBM: the head GEMM (fp32 MFMA 32x32x2), the Gumbel-max
 * sampler (== torch.multinomial's exponential race) and the log-sum-exp are fused; per-chunk partials are merged
 * by a second tiny kernel.  Arithmetic order is fixed (see oracle) so action indices are bit-reproducible.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct cirs_policy_cfg {
    int32_t n_items;   /* I */
    int32_t dim_state; /* 20 */
    int32_t hidden;    /* 64 (two hidden layers of this width) */
} cirs_policy_cfg;

typedef struct cirs_policy_weights {
    const float *w1, *b1; /* actor.preprocess.model.model.0 [H,S],[H] */
    const float *w2, *b2; /* actor.preprocess.model.model.2 [H,H],[H] */
    const float *wa, *ba; /* actor.last.model.0 [I,H],[I] */
    const float *wc, *bc; /* critic.last.model.0 [1,H],[1] */
} cirs_policy_weights;

/* bytes of caller-provided scratch for cirs_actor_sample / cirs_actor_logp / cirs_rollout_steps* with batch n (h2 rows, sampler
 * partials, and -- for the fused rollout -- 256 KB for the packed weight image of its step kernel, rebuilt by every call) */
int64_t cirs_policy_workspace_bytes(const cirs_policy_cfg* cfg, int32_t n);

/* For rows j in [0,n): h2 = trunk(state[j]); value = critic; act = argmax_i (logit_i + g_i) over unmasked items,
 * logp = log(clamp(softmax(logits)[act], eps, 1-eps)) (Categorical(probs).log_prob).
 *   state        [n, state_stride] f32
 *   gumbel       nullable [n, I] f32 harness-supplied noise g = -log(q); NULL -> counter-based Philox4x32-10 keyed by
 *                (seed, rng_step) with counter (item, env_id>>2, ...) so results do not depend on batch composition
 *   env_ids      nullable [n] global env id of each row (RNG key + visited row); NULL -> j
 *   visited      nullable [B, ceil(I/32)] u32 bitmap of already recommended ids (remove_recommended_ids)
 *   skip         nullable [n]: rows with skip != 0 produce act = -1 and are not computed
 *   act_out [n] i64, logp_out [n] f32 (nullable), value_out [n] f32 (nullable), h2_out [n,H] f32 (nullable) */
int cirs_actor_sample(const cirs_policy_cfg* cfg, const cirs_policy_weights* w, const float* state,
                      int64_t state_stride, int32_t n, const float* gumbel, uint64_t seed, uint32_t rng_step,
                      const int32_t* env_ids, const uint32_t* visited, const uint8_t* skip, int64_t* act_out,
                      float* logp_out, float* value_out, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- column-sharded actor head + row-sharded tables (BASELINE configs[4]: catalogue / embedding tables split over the ranks) ----
 * The reference has no counterpart (deepctr_torch/inputs.py:31-33 only prints a notice for use_hash); the spec is SURVEY 8(e):
 * "actor head column-sharded over items with a cross-rank (max, sum-exp, sampled-candidate) reduction", "tables row-sharded by
 * id mod W, per step an all-to-all of requested ids and returned rows".  Semantics to match: cirs_actor_sample on one device.
 *
 * cirs_actor_shard_partials: THIS rank's item shard (cfg_shard->n_items items whose first global id is item_base, a multiple of
 * 128 = the sampler's chunk (chunk ids are item_base / 128: every shard but the last must hold a multiple of 128 items);
 * w_shard->wa / ba = the shard's rows, trunk / critic weights replicated) against n env rows (all envs of the job; env_ids =
 * their global ids).  Noise counters, the visited bitmap and candidate ids use GLOBAL item ids, so a shard computes exactly what
 * the full kernel computes for its items.  tuples_out [5, n] f32: {noisy score, candidate id (int32 bits), candidate logit,
 * running max, running sum-exp}.  cirs_actor_merge_shards: tuples [n_shards, 5, n] folded in shard order -> act (argmax of the
 * noisy score, ties -> lowest id: identical to the single-device action) and logp (Categorical clamp). */
/* critic(obs) for n stored states with the current parameters: the value pass of A2CPolicy._compute_returns
   (tianshou/tianshou/policy/modelfree/a2c.py:80-86) when PPOPolicy(recompute_advantage=True) recomputes the advantages at every repeat
   (core/policy/ppo.py:176-177).  workspace: cirs_policy_workspace_bytes(cfg, n). */
int cirs_critic_values(const cirs_policy_cfg* cfg, const cirs_policy_weights* w, const float* state, int64_t state_stride, int32_t n,
                       float* value_out, void* workspace, int64_t workspace_bytes, void* stream);

int cirs_actor_shard_partials(const cirs_policy_cfg* cfg_shard, const cirs_policy_weights* w_shard, const float* state,
                              int64_t state_stride, int32_t n, uint64_t seed, uint32_t rng_step, const int32_t* env_ids,
                              const uint32_t* visited, const uint8_t* skip, int32_t item_base, int32_t n_items_total,
                              float* tuples_out, float* value_out, void* workspace, int64_t workspace_bytes, void* stream);
int cirs_actor_merge_shards(const float* tuples, int32_t n_shards, int32_t n, const uint8_t* skip, int64_t* act_out,
                            float* logp_out, void* stream);
/* out[k, :] = table[idx[k], :] (zeros where idx[k] < 0) for a row-major fp32 table with row_floats (multiple of 4) floats per row: the local side of a
 * row-sharded nn.Embedding lookup (core/user_model.py:435-437, core/state_tracker.py:97-110 `embedding_dict[...](ids)`). */
int cirs_gather_rows(const float* table, int32_t row_floats, const int64_t* idx, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Device-resident rollout: the Collector hot loop with no host round trip per step
 * replaces  core/collector.py:219-317 (policy forward -> env.step -> preprocess_fn -> buffer.add, per vector step)
 *           tianshou/data/buffer/manager.py:91-142 (ReplayBufferManager.add) -- the trajectory IS the buffer
 *           core/policy/utils.py:7-27 (get_recommended_ids) via the visited bitmap
 * Trajectory tensors are TIME-MAJOR so every step reads/writes contiguous [B, .] rows (the reference's tracker
 * history `data` is (L, B, D) too, state_tracker.py:198).  Row t of act/rew/done/logp/value/ctr belongs to the
 * transition (s_t, a_t, r_t, s_{t+1}); obs has T+1 rows.  Finished envs keep act = -1 from their first idle step.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct cirs_traj {
    float* obs;     /* [T+1, B, S] tracker states; row 0 written by cirs_tracker_init                         */
    int64_t* act;   /* [T, B]     env-encoded item id, -1 = env already finished                               */
    double* rew;    /* [T, B]     */
    uint8_t* done;  /* [T, B]     */
    float* logp;    /* [T, B]     log pi(a_t | s_t) at rollout time (== logp_old of ppo.py:104-108)            */
    float* value;   /* [T, B]     V(s_t) at rollout time    (== v_s of a2c.py:83-88)                           */
    double* ctr;    /* [T, B]     info['CTR'] (simulated) / cum_reward (bare env)                              */
} cirs_traj;

/* Run vector steps t in [t_begin, t_end) for all n_env envs: actor_sample(obs[t]) -> env_step -> tracker_step -> obs[t+1].
 *   rng_base      step t draws its sampler noise with rng_step = rng_base + t
 *   visited       nullable [B, ceil(I/32)] bitmap; when given, chosen ids are masked from later draws of the same
 *                 env (remove_recommended_ids, the NX_* test collectors) and the bitmap is updated each step
 *   force_length  > 0: done is overridden to (t+1 >= force_length) for every env (core/collector.py:253-258)
 *   workspace     >= cirs_policy_workspace_bytes(policy_cfg, n_env) */
struct cirs_deepfm_cfg;
struct cirs_deepfm_weights;
/* Online scoring of the chosen (user, item) pairs with the DeepFM user model between the policy and the env step. */
typedef struct cirs_online_reward {
    const struct cirs_deepfm_cfg* cfg;
    const struct cirs_deepfm_weights* w;
    const int64_t* raw_uid;     /* [U] raw id of env user (lbe_user.classes_)        */
    const int64_t* raw_pid;     /* [I] raw id of env item (lbe_photo.classes_)       */
    const int32_t* item_feats;  /* [I,4] feat ids (0 = padding)                      */
    const float* item_dur;      /* [I]                                               */
    const float* pred_minmax;   /* [2]                                               */
    int64_t* uid_buf;           /* [B] scratch                                       */
    int64_t* pid_buf;           /* [B] scratch                                       */
    int32_t* feat_buf;          /* [B,4] scratch                                     */
    float* dur_buf;             /* [B] scratch                                       */
    float* pred_buf;            /* [B] scratch: raw scores of this step              */
} cirs_online_reward;

int cirs_rollout_steps(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                       const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w, cirs_tracker_state* trk_st,
                       const cirs_policy_cfg* pol_cfg, const cirs_policy_weights* pol_w, const cirs_traj* traj,
                       int32_t n_env, int32_t t_begin, int32_t t_end, uint64_t seed, uint32_t rng_base,
                       uint32_t* visited, int32_t force_length, void* workspace, int64_t workspace_bytes,
                       void* stream);
/* Collector.reset_env + one collect(n_episode = n_env) from ONE call (core/collector.py:123-134: reset_env -> env.reset + preprocess_fn(obs = ...);
 * :147-367: the loop): env reset with users[n_env], the tracker's first position (cirs_tracker_init's arithmetic from the step kernel's packed weight image,
 * the first vector step's trunk in the same launch), then cirs_rollout_steps(0, max_turn).  Every trajectory entry [t][env] is written (envs that finished
 * earlier: act -1, done 1, rew / ctr 0), so nothing has to be cleared before the call. */
int cirs_rollout_collect(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                         const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w, cirs_tracker_state* trk_st,
                         const cirs_policy_cfg* pol_cfg, const cirs_policy_weights* pol_w, const cirs_traj* traj,
                         int32_t n_env, const int32_t* users, uint64_t seed, uint32_t rng_base, uint32_t* visited,
                         int32_t force_length, void* workspace, int64_t workspace_bytes, void* stream);
/* same loop with HARNESS-SUPPLIED sampler noise instead of the counter-based generator: gumbel [max_turn][n_env][n_items] f32,
 * g = -log q with q ~ Exp(1) (torch.multinomial's race noise, core/policy/ppo.py:148-155 `dist.sample()`), row t is used at
 * vector step t and indexed by ORIGINAL item id also when recommended ids are masked (core/policy/utils.py:30-58): the device
 * rollout then reproduces a reference Collector.collect whose Categorical.sample was fed the same q (parity fixtures). */
int cirs_rollout_steps_noise(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                             const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w, cirs_tracker_state* trk_st,
                             const cirs_policy_cfg* pol_cfg, const cirs_policy_weights* pol_w, const cirs_traj* traj,
                             int32_t n_env, int32_t t_begin, int32_t t_end, const float* gumbel, uint32_t* visited,
                             int32_t force_length, void* workspace, int64_t workspace_bytes, void* stream);
/* same, with the predicted reward scored online (env_tab->normed_mat may be NULL) */
int cirs_rollout_steps_online(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                              const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w,
                              cirs_tracker_state* trk_st, const cirs_policy_cfg* pol_cfg,
                              const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env, int32_t t_begin,
                              int32_t t_end, uint64_t seed, uint32_t rng_base, uint32_t* visited, int32_t force_length,
                              const cirs_online_reward* online, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Learner: PPO update over the collected trajectories
 * replaces  tianshou/policy/base.py:219-244 (update), :271-313 (compute_episodic_return), :380-396 (_gae_return)
 *           tianshou/policy/modelfree/a2c.py:80-109 (_compute_returns), tianshou/utils/statistics.py:80-95 (RunningMeanStd)
 *           core/policy/ppo.py:96-109 (process_fn), :166-246 (learn), torch.optim.Adam, clip_grad_norm_
 * Rows are kept in BUFFER ORDER (VectorReplayBuffer: env-major concatenation of each env's episode), so minibatch
 * index arrays drawn from np.random.permutation(N) mean the same rows as in the reference (batch.py:734-744).
 * v_s / v_s_ / logp_old are the values recorded at rollout time: the policy has not changed since, so they equal
 * what process_fn recomputes (a2c.py:83-88, ppo.py:104-108).
 * Reference quirks kept: the shared trunk appears twice in the optimiser and in clip_grad_norm_ ->
 * its squared gradient norm counts twice, the clip coefficient is applied twice and every optimiser step runs two
 * sequential Adam sub-steps on the trunk.
 * Parameter / gradient / Adam-moment buffers are FLAT fp32 with the layout
 *   [ w1 (H*S) | b1 (H) | w2 (H*H) | b2 (H) | wa (I*H) | ba (I) | wc (H) | bc (1) ]     (trunk = first four)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct cirs_ppo_cfg {
    int32_t n_items, dim_state, hidden;
    int32_t norm_adv, value_clip, rew_norm;
    float gamma, gae_lambda, eps_clip, vf_coef, ent_coef, max_grad_norm;
    float lr, beta1, beta2, adam_eps;
    float dual_clip;   /* 0: off; > 1: dual-clip PPO as the reference computes it, -max(min(surr1, surr2), dual_clip * adv) for EVERY sign of
                          adv (core/policy/ppo.py:190-193) */
} cirs_ppo_cfg;

typedef struct cirs_ppo_batch { /* N rows, buffer order */
    float* obs;        /* [N,S]  s_t                                    */
    int32_t* act;      /* [N]                                           */
    float* adv;        /* [N]    GAE advantage (float32 of the float64) */
    float* ret;        /* [N]    normalised returns                     */
    float* v_s;        /* [N]    old value (normalised scale)           */
    float* logp_old;   /* [N]                                           */
    int32_t* row_env;  /* [N]    env of the row                         */
    int32_t* row_t;    /* [N]    turn of the row                        */
} cirs_ppo_batch;

int64_t cirs_ppo_param_count(const cirs_ppo_cfg* cfg);
int64_t cirs_ppo_workspace_bytes(const cirs_ppo_cfg* cfg, int32_t max_minibatch);

/* process_fn: GAE (float64) per env, return normalisation with the running variance, RunningMeanStd update, and
 * compaction of the time-major trajectory into buffer order.  lens[B] = episode lengths, offsets[B] = exclusive
 * prefix sum of lens (row of env b's first transition).  rms_state = {mean, var, count} (float64, device). */
int cirs_ppo_prepare(const cirs_ppo_cfg* cfg, const cirs_traj* traj, const int32_t* lens, const int32_t* offsets,
                     int32_t n_env, int32_t max_turn, int32_t n_rows, double* rms_state, const cirs_ppo_batch* out,
                     void* stream);
/* The same preparation with the row count left on the device: offsets_out [n_env] and n_rows_out [1] are formed from `lens` by the
   first kernel, so the call can be enqueued behind the rollout before the host has read the episode lengths (the read-back then
   overlaps with these kernels).  scratch: n_env * max_turn doubles; `out` sized for n_env * max_turn rows.  Same results, bit for bit. */
int cirs_ppo_prepare_async(const cirs_ppo_cfg* cfg, const cirs_traj* traj, const int32_t* lens, int32_t n_env, int32_t max_turn,
                           int32_t* offsets_out, int32_t* n_rows_out, double* rms_state, const cirs_ppo_batch* out, double* scratch,
                           void* stream);
/* cirs_ppo_prepare_async + the update's minibatch permutations (cirs_random_permutations(n_rows, perm_seed, perm_tag0, n_perm, ..), n_perm <= 8) from
 * process_fn's last launch, the row count read on the device: perm_out[c * n_rows + i], i < n_rows (buffer: n_perm * n_env * max_turn ints).
 * replaces  Batch.split(shuffle=True)'s np.random.permutation per repeat (tianshou/data/batch.py:734-744; core/policy/ppo.py:173-181). */
int cirs_ppo_prepare_async_perms(const cirs_ppo_cfg* cfg, const cirs_traj* traj, const int32_t* lens, int32_t n_env, int32_t max_turn,
                                 int32_t* offsets_out, int32_t* n_rows_out, double* rms_state, const cirs_ppo_batch* out, double* scratch,
                                 uint64_t perm_seed, uint64_t perm_tag0, int32_t n_perm, int32_t* perm_out, int32_t offsets_ready, void* stream);

/* one minibatch gradient step of learn(): forward, clipped surrogate + clipped value loss + entropy, backward,
 * clip_grad_norm_, Adam.  idx[mb] are buffer-order row ids.  opt_step = optimiser steps taken so far.
 * dobs_accum (nullable) [T+1, B, S]: d loss / d obs rows are written at (row_t, row_env) -- the gradient that flows
 * into the state tracker through the stored obs (ppo.py:215 retain_graph).  loss_out[4] = {loss, clip, vf, ent}. */
int cirs_ppo_minibatch(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v,
                       int64_t opt_step, const cirs_ppo_batch* batch, const int32_t* idx, int32_t mb,
                       float* dobs_accum, int32_t n_env, float* loss_out, void* workspace, int64_t workspace_bytes,
                       void* stream);

/* PPOPolicy.learn's loop (core/policy/ppo.py:173-233: `for step in range(repeat): for minibatch in batch.split(batch_size, merge_last=True)`) from
 * ONE call: n_repeat passes over the n_rows buffer rows, pass r in the order perms[r * n_rows .. (r + 1) * n_rows) (device; the host's
 * np.random.permutation of Batch.split, tianshou/data/batch.py:721-752), cut into minibatches of batch_size rows with the remainder merged into the
 * last one (batch.py:734-744).  Every minibatch is exactly one cirs_ppo_minibatch step -- same kernels, same bits -- with optimiser step
 * opt_step + k; what the call adds is that step k's optimiser launch also runs the head of step k + 1 (trunk forward of its rows on the updated
 * weights, its advantage statistics, the bf16 planes of the updated Wa), so the loop has one launch less per step and the host leaves the loop's
 * critical path (one ctypes call per update instead of one per step).  dobs_accum (nullable, dobs_floats floats): zeroed before the LAST pass and
 * filled by it (ppo.py:174 zero_grad at the top of every repeat: only the last pass's d loss / d obs reaches the state tracker).
 * losses [cirs_ppo_learn_steps(n_rows, batch_size, n_repeat)][4] = {loss, clip, vf, ent} per step.  workspace: cirs_ppo_workspace_bytes(cfg, m)
 * for the largest minibatch m (< 2 * batch_size). */
int32_t cirs_ppo_learn_steps(int32_t n_rows, int32_t batch_size, int32_t n_repeat);
int cirs_ppo_learn(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v, int64_t opt_step,
                   const cirs_ppo_batch* batch, const int32_t* perms, int32_t n_rows, int32_t batch_size, int32_t n_repeat,
                   float* dobs_accum, int64_t dobs_floats, int32_t n_env, float* losses, void* workspace, int64_t workspace_bytes,
                   void* stream);

/* Health of the in-launch hand-offs of the minibatch step (no reference counterpart: the reference's optimiser step, core/policy/ppo.py:215-233,
 * is a sequence of synchronous torch calls and cannot lose one).  Inside cirs_ppo_learn / cirs_ppo_minibatch* some workgroups wait for arrival
 * flags of other workgroups of the SAME launch (trunk backward rows -> gradient sums; trunk Adam -> the next step's trunk forward).  The wait is
 * bounded; a workgroup that gives up counts itself in a sticky device word.  This call enqueues (on `stream`) a copy of that count to lost_out
 * (device pointer, one int32) and, with reset != 0, clears it.  A non-zero count means the update it belongs to used incomplete sums or stale
 * weights: the host must treat the update as failed (cirs_hip/learner.py raises at the update's read-back). */
int cirs_ppo_handoff_status(int32_t* lost_out, int32_t reset, void* stream);
/* The one read-back of an update (the host needs the episode lengths to schedule the minibatches: len(buffer) in tianshou/policy/base.py:231-244) as one
 * launch: lens[n_env] and the hand-off count above written straight to PINNED host memory (device-accessible: hipHostMalloc); lost_host_pinned may be NULL.
 * offsets_out / n_rows_out (both or neither): process_fn's first job -- the buffer offsets of the envs and the row count, cirs_ppo_prepare_async's
 * outputs -- formed in the same launch; cirs_ppo_prepare_async_perms(.., offsets_ready = 1, ..) then starts at the GAE. */
int cirs_ppo_update_readback(const int32_t* lens, int32_t n_env, int32_t* lens_host_pinned, int32_t* lost_host_pinned, int32_t* offsets_out,
                             int32_t* n_rows_out, void* stream);

/* Data-parallel form of cirs_ppo_minibatch for a learner sharded over ranks.  A GLOBAL minibatch of mb_global rows
 * (idx_global) is split by rows; this rank owns idx_local[mb_local].  Advantage normalisation uses the statistics of
 * the global minibatch (every rank holds the gathered buffer) and every mean is over mb_global rows, so the SUM over
 * ranks of the gradients equals the single-device gradient of the global minibatch.
 *   phase 1: forward + backward -> grads[0 .. P) and the loss partials {clip, vf, ent, 0} in grads[P .. P+4)
 *            (P = cirs_ppo_param_count; the grads buffer must hold P+4 floats)
 *   -- caller all-reduces (sum) grads[0 .. P+4) over the ranks (RCCL) --
 *   phase 2: clip_grad_norm_ + Adam from the reduced gradients, loss_out[4] from the reduced partials
 *   phase 0: both phases back to back (single rank). */
int cirs_ppo_minibatch_dp(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v,
                          int64_t opt_step, const cirs_ppo_batch* batch, const int32_t* idx_local, int32_t mb_local,
                          const int32_t* idx_global, int32_t mb_global, float* dobs_accum, int32_t n_env,
                          float* loss_out, void* workspace, int64_t workspace_bytes, int32_t phase, void* stream);

/* cirs_ppo_minibatch_dp for a CHAIN of data-parallel steps (one update's minibatches; no reference counterpart, same semantics and bits as
 * cirs_ppo_minibatch_dp step by step): the workspace is laid out for max_mb_local rows (the update's largest local minibatch), so that consecutive
 * steps of different size share it;
 *   phase 1, head_done != 0: the head of this step (trunk forward of its rows, advantage statistics of the global minibatch, Wa planes) already ran
 *            inside the previous step's phase 2 -- three launches (head statistics, head backward, trunk backward incl. every gradient sum and this
 *            rank's loss partials) instead of seven;
 *   phase 2, next_idx_local != NULL: the optimiser launch also runs the head of the next step on the weights it forms (rows next_idx_local
 *            [next_mb_local] of this rank, statistics over next_idx_global [next_mb_global]); the caller passes head_done = 1 to that step's phase 1. */
int cirs_ppo_minibatch_dp_chain(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v, int64_t opt_step,
                                const cirs_ppo_batch* batch, const int32_t* idx_local, int32_t mb_local, const int32_t* idx_global,
                                int32_t mb_global, float* dobs_accum, int32_t n_env, float* loss_out, void* workspace, int64_t workspace_bytes,
                                int32_t phase, int32_t max_mb_local, int32_t head_done, const int32_t* next_idx_local, int32_t next_mb_local,
                                const int32_t* next_idx_global, int32_t next_mb_global, void* stream);

/* Tensor-parallel form of cirs_ppo_minibatch for an ITEM-SHARDED actor head (BASELINE configs[4]: the catalogue does not fit / is not
 * replicated; SURVEY 8(e): "actor head column-sharded over items with a cross-rank (max, sum-exp, ...) reduction").  No reference
 * counterpart (the reference is single-process); semantics = cirs_ppo_minibatch on one device with the whole catalogue.
 * Rank r holds rows [item_base, item_base + cfg->n_items) of wa / ba -- cfg, params, grads and the Adam moments describe the SHARD
 * (flat layout as above with I = the shard's item count); trunk and critic are replicated, batch->act holds GLOBAL item ids and every
 * rank processes every row of the minibatch.  Per minibatch, two collectives:
 *   phase 1  trunk forward, statistics of the local items -> stats4 [4][n_pad] = {max, sum-exp, sum exp z, logit of the row's action
 *            if this shard owns it else NaN}                                            (n_pad = rows rounded up to 32)
 *   -- all-gather; the caller hands the result back as stats_all [4][world][n_pad] (field-major) --
 *   phase 2  merge in rank order (identical on every rank) -> row losses / coefficients; fused head backward on the local items:
 *            the shard's wa|ba gradient is COMPLETE (it saw every row); red = {d h2 partial [n_pad,64], entropy clamp partial
 *            [n_pad], this rank's squared-norm partials of the wa|ba gradient in its slots of [world, 176]} =
 *            cirs_ppo_tp_exchange_floats(mb, world) floats
 *   -- all-reduce (sum) of red --
 *   phase 3  trunk / critic backward (replicated, identical inputs), clip_grad_norm_ over trunk x 2 + every shard's head + critic,
 *            Adam on the shard + the replicated trunk; loss_out[4]; dobs_accum as in cirs_ppo_minibatch. */
int64_t cirs_ppo_tp_exchange_floats(int32_t n_rows, int32_t world);
int cirs_ppo_minibatch_tp(const cirs_ppo_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v, int64_t opt_step,
                          const cirs_ppo_batch* batch, const int32_t* idx, int32_t mb, int32_t item_base, int32_t rank,
                          int32_t world, float* stats4, const float* stats_all, float* red, float* dobs_accum, int32_t n_env,
                          float* loss_out, void* workspace, int64_t workspace_bytes, int32_t phase, void* stream);

/* Sharded optimiser step of the data-parallel learner (the reduce-scatter -> sharded Adam -> all-gather form of the step above):
 * after phase 1 the caller reduce-scatters grads[0 .. P_pad) (P_pad = P + 4 rounded up to a multiple of 4 * world; the padding
 * stays zero) so that this rank holds the summed shard [shard_begin, shard_begin + shard_len) of the flat gradient.
 *   cirs_ppo_shard_norm   stats_out[cirs_ppo_shard_stat_floats()] = fixed-order partial sums of squares of the shard (trunk elements
 *                         counted twice) + the loss partials of the gradient tail where the shard holds them
 *   -- caller all-gathers the stats of all ranks: stats_all [world, cirs_ppo_shard_stat_floats()] --
 *   cirs_ppo_shard_adam   clip_grad_norm_ coefficient from stats_all (summed in rank order: identical on every rank), Adam on the
 *                         shard only (params / moments pointers are those OF THE SHARD), loss_out[4] (nullable)
 *   -- caller all-gathers the parameter shards --
 * No reference counterpart (the reference is single-process); semantics = cirs_ppo_minibatch_dp phase 2. */
int32_t cirs_ppo_shard_stat_floats(void);
int cirs_ppo_shard_norm(const cirs_ppo_cfg* cfg, const float* grads_shard, int64_t shard_begin, int64_t shard_len,
                        float* stats_out, void* stream);
int cirs_ppo_shard_adam(const cirs_ppo_cfg* cfg, float* params_shard, const float* grads_shard, float* adam_m_shard,
                        float* adam_v_shard, int64_t shard_begin, int64_t shard_len, int64_t opt_step, const float* stats_all,
                        int32_t world, float* loss_out, void* stream);

/* torch.optim.Adam single-tensor update over a flat buffer, `n_sub` sequential sub-steps with the same gradient
 * starting at step `step_before`+1; grad is multiplied by (*grad_scale)^scale_pow when grad_scale != NULL. */
int cirs_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int64_t step_before,
                   int32_t n_sub, float lr, float beta1, float beta2, float eps, const float* grad_scale,
                   int32_t scale_pow, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * State-tracker backward: the gradient PPO sends into the tracker through the stored obs
 * replaces  the retained autograd graph of core/collector.py:261-269 + core/policy/ppo.py:174,215,235
 *           (loss.backward(retain_graph=True) accumulating into state_tracker.parameters(), one optim_state.step()).
 * Because the mask is causal and dropout is off, the per-step recomputations of the reference are one causal
 * transformer pass over each episode; rows are the buffer rows (env b, position p = t), p < len_b.  The forward is
 * recomputed from the stored slots (x_hist) and back-propagated analytically down to the embedding tables.
 * Every reduction over rows (weight gradients and the embedding-table scatter) has a fixed order: no float atomics,
 * so replicated learners on different ranks produce identical bits.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct cirs_tracker_layer_grads {
    float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b, *lin1_w, *lin1_b, *lin2_w, *lin2_b;
    float *norm1_w, *norm1_b, *norm2_w, *norm2_b;
} cirs_tracker_layer_grads;

typedef struct cirs_tracker_grads { /* same tensors as cirs_tracker_weights (pe is a buffer: no gradient) */
    float *emb_user, *emb_item, *ffn_user_w, *ffn_user_b, *gate_w, *gate_b;
    float* pe_unused;
    cirs_tracker_layer_grads layer[CIRS_MAX_TRACKER_LAYERS];
    float *dec_w, *dec_b;
} cirs_tracker_grads;

int64_t cirs_tracker_backward_workspace_bytes(const cirs_tracker_cfg* cfg, int32_t n_rows);

/* users[B]; act/rew are the time-major trajectory rows [T,B]; row_env/row_t/offsets/lens describe the buffer rows
 * (see cirs_ppo_prepare); dstate [T+1,B,S] = d loss / d obs.  Every gradient tensor is OVERWRITTEN. */
int cirs_tracker_backward(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st,
                          const int32_t* users, const int64_t* act, const double* rew, const int32_t* row_env,
                          const int32_t* row_t, const int32_t* offsets, const int32_t* lens, int32_t n_rows,
                          const float* dstate, const cirs_tracker_grads* grads, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* cirs_tracker_backward for an upstream gradient that sits on the LAST row of every env only: dstate_last [n_env, S] = d loss / d state(env b, position
 * lens[b] - 1) (envs with lens[b] == 0 are ignored); same gradients as cirs_tracker_backward with a dstate that is zero everywhere else.
 * replaces  the backward of ONE build_state call under live dropout: the policy only receives s_t = the decoder of the last position of the prefix
 *           (core/state_tracker.py:243-246 `s_t = ...[:, -1, :]` behind the re-run encoder of :170-186), so in that call's graph the top encoder layer is
 *           needed at one row per env: its attention has one query, its row chain, five of its six weight-gradient problems and the decoder's run on
 *           n_env rows instead of n_rows; every layer below is the ordinary pass.  The exact-redraw learner (cirs_hip/redraw.py) runs all calls of a
 *           buffer as ONE such pass over pseudo-envs (call c, env e).
 * Workspace: cirs_tracker_backward_workspace_bytes(cfg, max(n_rows, cfg->n_env)).  Falls back to the general pass (same results, no saving) for a
 * one-layer tracker or max_len > 64. */
int cirs_tracker_backward_last(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st,
                               const int32_t* users, const int64_t* act, const double* rew, const int32_t* row_env,
                               const int32_t* row_t, const int32_t* offsets, const int32_t* lens, int32_t n_rows,
                               const float* dstate_last, const cirs_tracker_grads* grads, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* Exact-redraw dropout inside the fused rollout (core/state_tracker.py:170-186,243-246: the reference never switches the tracker to eval(), so every
 * build_state call re-runs the encoder over the WHOLE prefix with fresh masks).  cirs_rollout_steps_redraw = cirs_rollout_steps in which the state of
 * vector step t comes from cirs_tracker_prefix_states over positions 0 .. t of every env under the masks of call t (pseudo-env ids env_base0 +
 * t * env_stride + e of the key dropout_seed), instead of from the cached decode.  Row lists of all calls, concatenated: call c describes n_env * (c + 1)
 * rows (env-major) starting at n_env * c * (c + 1) / 2 of row_env / row_t, and row c of offsets / lens [max_turn + 1][n_env].
 * workspace: cirs_tracker_backward_workspace_bytes(cfg, n_env * (max_turn + 1)). */
typedef struct {
    const int32_t* row_env;
    const int32_t* row_t;
    const int32_t* offsets;
    const int32_t* lens;
    uint64_t dropout_seed;
    int64_t env_base0, env_stride;
    void* workspace;
    int64_t workspace_bytes;
} cirs_redraw;
int cirs_rollout_steps_redraw(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                              const cirs_tracker_cfg* trk_cfg, const cirs_tracker_weights* trk_w, cirs_tracker_state* trk_st,
                              const cirs_policy_cfg* pol_cfg, const cirs_policy_weights* pol_w, const cirs_traj* traj, int32_t n_env,
                              int32_t t_begin, int32_t t_end, uint64_t seed, uint32_t rng_base, const cirs_redraw* redraw,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* The forward half of cirs_tracker_backward as an entry point: ONE causal pass over the buffer rows (env b, positions 0 .. lens[b]-1) from the
 * stored input slots, with the dropout masks of the key currently set in cfg (dropout_seed / drop_env_base), and the state of every env's LAST row
 * -> state_out[b * state_stride + 0 .. dim_state) (envs with lens[b] == 0 are left untouched).  This is the reference's build_state under live
 * dropout -- the encoder re-run over the WHOLE prefix with fresh masks at every call (core/state_tracker.py:170-186, 243-246) -- as one batched
 * pass per call instead of a replay of the cached decode (cirs_hip/redraw.py).  Workspace: cirs_tracker_backward_workspace_bytes(cfg, n_rows). */
int cirs_tracker_prefix_states(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, const cirs_tracker_state* st,
                               const int32_t* row_env, const int32_t* row_t, const int32_t* offsets, const int32_t* lens,
                               int32_t n_rows, float* state_out, int64_t state_stride, void* workspace, int64_t workspace_bytes,
                               void* stream);

/* Ordered scatter of embedding-gradient rows (dim_model == 32) into a table: grad_table [n_table_rows, 32] is OVERWRITTEN with, per row k,
 * the sum of contrib[r] over the rows r with keys[r] == k, added in ascending r (stable radix sort + ordered segment sums: no float
 * atomics, cost O(n_rows) whatever the table size); keys outside [0, n_table_rows) contribute nothing.  It is the scatter of
 * cirs_tracker_backward's embedding gradients as an entry point of its own: the OWNER side of row-sharded tables (BASELINE configs[4],
 * "tables row-sharded by id mod W") after the all-to-all of (row id, gradient row) pairs.  No reference counterpart. */
int64_t cirs_embedding_scatter_workspace_bytes(int64_t n_rows);
int cirs_embedding_scatter(const int32_t* keys, const float* contrib, int64_t n_rows, int32_t n_table_rows, float* grad_table,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * DeepFM user model (UserModel_Pairwise) and the full-catalogue sweep
 * replaces  core/user_model_pairwise.py:98-154 (_deepfm/forward), core/user_model.py:419-447 (input_from_feature_columns),
 *           core/layers.py:43-72 (Linear), DeepCTR-Torch deepctr_torch/layers/interaction.py:26-34 (FM),
 *           layers/core.py:120-134 (DNN), :155-161 (PredictionLayer, regression), inputs.py:126-138 (combined_dnn_input)
 *           environments/KuaishouRec/env/kuaishouEnv.py:113-145 (compute_normed_reward)
 * y = sum_f w_f[x_f] + dur*w_d  +  FM(v_user, v_item, v_f0..3)  +  last . DNN([v_user, v_item, v_f0..3, dur]) + bias
 * Feature order in X: user_id, photo_id, feat0..feat3, photo_duration; the `feat` table is
 * shared by feat0..3 (row 0 = padding, trained to stay zero).  hidden == 64 (two DNN layers), emb_dim in {8,16,32,64}.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct cirs_deepfm_cfg {
    int32_t n_user_vocab, n_item_vocab, n_feat_vocab;
    int32_t emb_dim; /* E */
    int32_t hidden;  /* 64 */
} cirs_deepfm_cfg;

typedef struct cirs_deepfm_weights {
    const float *emb_user, *emb_item, *emb_feat; /* embedding_dict.{user_id,photo_id,feat}.weight [V,E]            */
    const float *lin_user, *lin_item, *lin_feat; /* linear.embedding_dict.*.weight [V] (Q12: `linear`, not linear_model) */
    const float* lin_dense;                      /* linear.weight [1] (photo_duration)                              */
    const float *w1, *b1;                        /* dnn.linears.0 [H, 6E+1], [H]                                     */
    const float *w2, *b2;                        /* dnn.linears.1 [H, H], [H]                                        */
    const float* last;                           /* last.weight [H]                                                  */
    const float* out_bias;                       /* out.bias [1]                                                     */
} cirs_deepfm_weights;

/* UserModel_Pairwise.forward on n (user,item) rows: ids index the vocab tables directly (raw ids), feats[n,4]. */
int cirs_deepfm_forward(const cirs_deepfm_cfg* cfg, const cirs_deepfm_weights* w, const int64_t* uid, const int64_t* pid,
                        const int32_t* feats, const float* dur, int32_t n, float* out, void* stream);

/* K1-K2 alone: embedding gather + linear logit + FM bi-interaction, no DNN -- the cache/HBM-bound stage.
 * replaces  core/user_model.py:419-447 (input_from_feature_columns), core/layers.py:59-70 (Linear.forward),
 *           DeepCTR-Torch deepctr_torch/layers/interaction.py:26-34 (FM.forward)
 * X[n,7] float32 rows = [user_id, photo_id, feat0..3, photo_duration] -- the reference's own input tensor of
 * UserModel_Pairwise.forward (ids as float32, exact below 2^24).  out[n] = linear logit + FM term.
 * cirs_deepfm_forward(x) == cirs_gather_fm(x) + (last . DNN(x) + out_bias) up to fp32 summation order. */
int cirs_gather_fm(const cirs_deepfm_cfg* cfg, const cirs_deepfm_weights* w, const float* X, int64_t n, float* out, void* stream);

int64_t cirs_deepfm_sweep_workspace_bytes(const cirs_deepfm_cfg* cfg, int32_t n_users, int32_t n_items);

/* Scores every (user, item) pair of users x items: pred_out[nu, ni] fp32 (nullable) and the global {min, max}
 * (minmax[2], device; must be initialised to {+inf, -inf} by the caller or by passing init_minmax != 0).
 * The DNN's first layer is split into a per-user and a per-item partial sum (each computed once), the 64x64 second
 * layer runs on the fp32 matrix cores for 32 pairs at a time, the FM cross term is one E-long dot per pair. */
int cirs_deepfm_sweep(const cirs_deepfm_cfg* cfg, const cirs_deepfm_weights* w, const int64_t* user_ids, int32_t nu,
                      const int64_t* item_ids, const int32_t* item_feats, const float* item_dur, int32_t ni,
                      float* pred_out, float* minmax, int32_t init_minmax, void* workspace, int64_t workspace_bytes,
                      void* stream);

/* normed = (pred - min) / (max - min) in float64 (kuaishouEnv.py:139-143) */
int cirs_normed_reward(const float* pred, int64_t n, const float* minmax, double* normed_out, void* stream);

/* ---- user-model training ---------------------------------------------------------------------------
 * One optimiser step of fit_data's inner loop (reference core/user_model.py:150-170) for UserModel_Pairwise:
 *   loss = loss_kuaishou_pairwise(y, y_pos, y_neg, exposure, alpha_u[uid], beta_i[pid])   (CIRS-UserModel-kuaishou.py:262-278,
 *                                                                                         core/user_model_pairwise.py:134-151)
 *        + get_regularization_loss()                                                      (core/user_model.py:401-417)
 *   total_loss.backward(); Adam step (lr, betas, eps of torch.optim.Adam).
 * Parameters, gradients and the two Adam moments are flat fp32 buffers of cirs_deepfm_train_param_count(cfg) floats in
 * the order  emb_user [U,E] | emb_item [I,E] | emb_feat [F,E] | lin_user [U] | lin_item [I] | lin_feat [F] | lin_dense [1]
 *          | w1 [64,6E+1] | b1 [64] | w2 [64,64] | b2 [64] | last [64] | out_bias [1] | alpha_u [U] | beta_i [I]
 *          | linear_model.{user,item,feat} [U],[I],[F] | linear_model.weight [1]      (the unused duplicate of SURVEY Q12,
 *                                                                                       which the regulariser still decays).
 * Batch columns: positive pair (uid, pid, feats[n,4], dur), negative pair (same layout), y [n], exposure [n].
 * l2_embedding / l2_linear / l2_all: the three regularisation lists of the reference (embedding_dict, linear_model, all).
 * loss_out[5] = {loss, loss_y, bpr, loss_ab, reg_loss}. */
int64_t cirs_deepfm_train_param_count(const cirs_deepfm_cfg* cfg);
int64_t cirs_deepfm_train_workspace_bytes(const cirs_deepfm_cfg* cfg, int32_t n);
int cirs_deepfm_train_step(const cirs_deepfm_cfg* cfg, float* params, float* grads, float* adam_m, float* adam_v,
                           int64_t step_before, const int64_t* uid_pos, const int64_t* pid_pos, const int32_t* feats_pos,
                           const float* dur_pos, const int64_t* uid_neg, const int64_t* pid_neg, const int32_t* feats_neg,
                           const float* dur_neg, const float* y, const float* exposure, int32_t n, int32_t use_ab,
                           float lambda_ab, float l2_embedding, float l2_linear, float l2_all, float lr, float beta1,
                           float beta2, float eps, float* loss_out, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- user-model dataset preparation ---------------------------------------------------------------
 * cirs_exposure_history replaces compute_exposure_each_user / the per-user loop of compute_exposure_effect_kuaishouRec
 * (reference core/util.py:56-76,135-169): rows are the logged interactions in file order, a user's rows contiguous;
 *   user_start[r] = index of the first row of row r's user, photo[r] item id, timestamp[r] (float64 seconds);
 *   dist [n_items, n_items] float64 (1 / similarity) or NULL -> 1 / Jaccard from item_cats [n_items] packed category words;
 *   exposure_out[r] = sum_{j in [user_start[r], r)} exp(-(max(ts_r - ts_j, ...)) * dist[photo_j, photo_r] / tau), dt == 0 -> 1.
 * cirs_find_negative replaces find_negative (core/util.py:173-196): seen_small / seen_big are bitmaps
 *   [n_users, ceil(n_items/32)] of the (user, item) pairs present in the small / big matrix; absent_id = 1225 for KuaiRec;
 *   neg_out[i] = the sampled negative item (or -1 if none exists). */
int cirs_exposure_history(const int64_t* user_start, const int32_t* photo, const double* timestamp, int64_t n_rows,
                          const double* dist, const uint32_t* item_cats, int32_t n_items, double tau, double* exposure_out,
                          void* stream);
int cirs_find_negative(const int64_t* user_ids, const int64_t* photo_ids, int64_t n, const uint32_t* seen_small,
                       const uint32_t* seen_big, int32_t n_items, int64_t absent_id, int64_t* neg_out, void* stream);

/* ---- the user model as a static recommendation policy ---------------------------------------------
 * cirs_select_items replaces the tail of UserModel.recommend_k_item (reference core/user_model.py:296-346, k = 1) for n
 * users at once, given their catalogue scores (cirs_deepfm_sweep):
 *   scores [n, row_stride >= n_items] f32;  bonus [n_items] or NULL (UCB bound, :303-314);
 *   visited [n, ceil(n_items/32)] or NULL (recommended_ids, :263-266);  skip [n] or NULL (row inactive -> act -1);
 *   softmax != 0: multinomial(softmax(u_value)) as arg-max of u_value + Gumbel noise (gumbel [n, n_items] supplied, or
 *   the counter-based generator keyed (seed, rng_step, row, item));  softmax == 0: arg-max, lowest id on ties (:331);
 *   epsilon: probability of a uniform random non-removed item instead (:333-335).
 *   act_out [n] int64 (-1: no item), value_out [n] or NULL: u_value of the chosen item (value_rec, :346).
 * cirs_rollout_static replaces the loop of interactive_evaluation (reference evaluation.py:87-120) for n_env
 * trajectories in lock-step: select -> mark visited -> env step -> forced length; traj->value holds reward_pred. */
int cirs_select_items(const float* scores, int64_t row_stride, int32_t n, int32_t n_items, int32_t softmax,
                      const float* bonus, const uint32_t* visited, const uint8_t* skip, float epsilon, const float* gumbel,
                      uint64_t seed, uint32_t rng_step, int64_t* act_out, float* value_out, void* stream);
int cirs_rollout_static(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                        const float* scores, int64_t row_stride, const float* bonus, const cirs_traj* traj, int32_t n_env,
                        int32_t t_begin, int32_t t_end, int32_t softmax, float epsilon, uint64_t seed, uint32_t rng_base,
                        uint32_t* visited, int32_t force_length, int64_t* obs_scratch, void* stream);

/* ---- feature hashing (BASELINE configs[4]: open-vocabulary ids hashed into fixed-size tables) -----------------------
 * out[i] = splitmix64(ids[i]) mod n_buckets.  No reference counterpart (DeepCTR-Torch inputs.py:31-33 only prints a notice
 * for use_hash): the mapping is this build's own and is restated bit for bit by the oracle. */
int cirs_hash_ids(const int64_t* ids, int64_t n, int64_t n_buckets, int64_t* out, void* stream);

/* ---- minibatch shuffle (Batch.split(shuffle=True): np.random.permutation(n), tianshou/data/batch.py:734-744) --------------
 * out[i] = P(i), a keyed pseudo-random permutation of [0, n): 6-round Feistel network over splitmix64 with cycle walking, one
 * thread per element, no sort and no host round trip.  Same (seed, tag) -> same permutation (ranks of a data-parallel learner
 * pass the same pair).  The oracle restates it bit for bit. */
int cirs_random_permutation(int64_t n, uint64_t seed, uint64_t tag, int32_t* out, void* stream);
/* out[c][i], c < count: the permutations of tags tag0 .. tag0 + count - 1 from one launch (the repeats of an update, core/policy/ppo.py:173-181). */
int cirs_random_permutations(int64_t n, uint64_t seed, uint64_t tag0, int32_t count, int32_t* out, void* stream);

/* ---- per-kernel timing hook (measurement only; no reference counterpart) ----------------------------------------------
 * cirs_prof_start arms HIP-event pairs around the next `max_samples` launches of one named kernel, recorded on the stream
 * the kernel is launched on; cirs_prof_stop waits for them and returns the summed duration and the sample count.
 * kernel_id: 1 = head_bwd_fused_kernel (PPO minibatch, fused actor-head backward), 2 = head_stats_kernel (PPO
 * minibatch, forward statistics), 3 = actor_head_kernel (rollout sampler).  bench.py's `roofline` object uses it so that
 * `seconds_per_launch` is the same quantity as the kernel's average in the rocprofv3 --kernel-trace --stats summary. */
int cirs_prof_start(int32_t kernel_id, int32_t max_samples);
int cirs_prof_stop(double* total_seconds, int32_t* n_samples);

/* ---- evaluation metrics on device trajectories ---------------------------------------------------
 * Replaces the buffer walks of Callback_Coverage_Count.on_epoch_end (reference evaluation.py:303-352) and the
 * row test of get_feat_dominate_dict (evaluation.py:36-44):
 *   act       [n] int64: the time-major act tensor of a rollout ([T,B], -1 where an env had finished)
 *   item_flag [n_items] u8 or NULL: 1 if the item has one of the dominating feature values
 *   bitmap    [ceil(n_items/32)] scratch
 *   out3      {hit_item, n_acts, n_flagged}: CV = hit_item / n_items, CV_turn = hit_item / n_acts,
 *             ifeat_feat = n_flagged / n_acts */
int cirs_eval_coverage(const int64_t* act, int64_t n, int32_t n_items, const uint8_t* item_flag, uint32_t* bitmap,
                       int64_t* out3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CIRS_HIP_H */
