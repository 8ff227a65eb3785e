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

#ifdef __cplusplus
}
#endif
#endif /* CIRS_HIP_H */
