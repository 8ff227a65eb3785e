// static_policy.hip -- the user model as a (static) recommendation policy inside the env loop: the static baselines'
// `UserModel.recommend_k_item` (reference core/user_model.py:254-348) and `interactive_evaluation`
// (reference evaluation.py:79-151), i.e. the north-star's "score the full item catalogue per step through DeepFM".
//
//   scores[n, I]  (one catalogue sweep per user, cirs_deepfm_sweep)  ->  one item per row:
//     u_value = score (+ UCB bound, user_model.py:303-314)
//     is_softmax: multinomial(softmax(u_value), 1) == argmax_i (u_value_i + Gumbel_i)       (:317-319; same exponential-race
//                 identity as the actor sampler, noise from the counter-based generator of rng.h or supplied by the caller)
//     else:       topk(u_value, 1) == argmax, lowest id on ties                               (:331)
//     epsilon:    with probability epsilon a uniform random non-removed item instead          (:333-335)
//     removed ids (recommended_ids, :263-266) come as the per-row visited bitmap of the rollout
//   One workgroup per row; lanes stride the catalogue in groups of four consecutive items (one Philox block each),
//   (key, id) pairs are reduced in a fixed order, so the CPU oracle reproduces the choice bit for bit.
#include "common.h"
#include "rng.h"

#define CIRS_RNG_STREAM_SELECT 0x53454C43u /* 'SELC' */
#define CIRS_RNG_STREAM_EPS 0x45505347u    /* 'EPSG' */

namespace cirs {

__global__ __launch_bounds__(256) void select_items_kernel(const float* __restrict__ scores, long row_stride, int n, int I, int softmax,
                                                           const float* __restrict__ bonus, const uint32_t* __restrict__ visited,
                                                           const uint8_t* __restrict__ skip, float epsilon,
                                                           const float* __restrict__ gumbel, uint64_t seed, uint32_t rng_step,
                                                           int64_t* __restrict__ act_out, float* __restrict__ value_out) {
    __shared__ float s_key[256];
    __shared__ int s_idx[256];
    __shared__ int s_cnt[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    if (skip && skip[row]) {
        if (tid == 0) { act_out[row] = -1; if (value_out) value_out[row] = 0.f; }
        return;
    }
    const int words = (I + 31) / 32;
    const uint32_t* vis = visited ? visited + (size_t)row * words : nullptr;
    const float* sc = scores + (size_t)row * row_stride;
    float best = -INFINITY;
    int best_i = 0x7FFFFFFF;
    int n_free = 0;
    for (int g = tid; g * 4 < I; g += 256) {
        const int i0 = g * 4;
        float g4[4] = {0.f, 0.f, 0.f, 0.f};
        if (softmax && !gumbel) {
            const u32x4 r = philox4x32_10((uint32_t)g, (uint32_t)row, rng_step, CIRS_RNG_STREAM_SELECT, (uint32_t)seed, (uint32_t)(seed >> 32));
            g4[0] = gumbel_from_bits(r.x); g4[1] = gumbel_from_bits(r.y); g4[2] = gumbel_from_bits(r.z); g4[3] = gumbel_from_bits(r.w);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q;
            if (i >= I) break;
            if (vis && ((vis[i >> 5] >> (i & 31)) & 1u)) continue;
            ++n_free;
            float key = sc[i] + (bonus ? bonus[i] : 0.f);
            if (softmax) key += gumbel ? gumbel[(size_t)row * I + i] : g4[q];
            if (key > best) { best = key; best_i = i; }  // ascending ids within a thread: strict > keeps the lowest id
        }
    }
    s_key[tid] = best; s_idx[tid] = best_i; s_cnt[tid] = n_free;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float ok = s_key[tid + s];
            const int oi = s_idx[tid + s];
            if (ok > s_key[tid] || (ok == s_key[tid] && oi < s_idx[tid])) { s_key[tid] = ok; s_idx[tid] = oi; }
            s_cnt[tid] += s_cnt[tid + s];
        }
        __syncthreads();
    }
    if (tid != 0) return;
    int pick = s_idx[0] == 0x7FFFFFFF ? -1 : s_idx[0];
    const int free_items = s_cnt[0];
    if (epsilon > 0.f && free_items > 0) {
        const u32x4 r = philox4x32_10((uint32_t)row, rng_step, 0u, CIRS_RNG_STREAM_EPS, (uint32_t)seed, (uint32_t)(seed >> 32));
        if (u01_from_bits(r.x) < epsilon) {  // np.random.random() < epsilon -> torch.randint over the preserved items
            int target = (int)(u01_from_bits(r.y) * (float)free_items);
            if (target >= free_items) target = free_items - 1;
            int seen = 0;
            pick = -1;
            for (int w = 0; w < words && pick < 0; ++w) {
                uint32_t freebits = ~(vis ? vis[w] : 0u);
                if (w == words - 1 && (I & 31)) freebits &= (1u << (I & 31)) - 1u;
                const int c = __popc(freebits);
                if (seen + c > target) {
                    for (int b = 0; b < 32; ++b)
                        if ((freebits >> b) & 1u) {
                            if (seen == target) { pick = w * 32 + b; break; }
                            ++seen;
                        }
                } else {
                    seen += c;
                }
            }
        }
    }
    act_out[row] = pick;
    if (value_out) value_out[row] = pick >= 0 ? sc[pick] + (bonus ? bonus[pick] : 0.f) : 0.f;
}

// defined in rollout.hip
__global__ void mark_visited_kernel(const int64_t* __restrict__ act, int n, int n_items, uint32_t* __restrict__ visited);
__global__ void force_done_kernel(uint8_t* __restrict__ st_done, uint8_t* __restrict__ done_row, const int64_t* __restrict__ act, int n,
                                  int force_done);

}  // namespace cirs

extern "C" int cirs_select_items(const float* scores, int64_t row_stride, int32_t n, int32_t n_items, int32_t softmax,
                                 const float* bonus, const uint32_t* visited, const uint8_t* skip, float epsilon,
                                 const float* gumbel, uint64_t seed, uint32_t rng_step, int64_t* act_out, float* value_out,
                                 void* stream) {
    using namespace cirs;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(scores && act_out && n_items > 0 && row_stride >= n_items, "bad arguments");
    CIRS_REQUIRE(epsilon >= 0.f && epsilon <= 1.f, "epsilon outside [0, 1]");
    hipLaunchKernelGGL(select_items_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, scores, (long)row_stride, n, n_items, softmax, bonus,
                       visited, skip, epsilon, gumbel, seed, rng_step, act_out, value_out);
    CIRS_CHECK_LAUNCH("select_items_kernel");
    return CIRS_OK;
}

extern "C" int cirs_rollout_static(const cirs_env_cfg* env_cfg, const cirs_env_tables* env_tab, cirs_env_state* env_st,
                                   const float* scores, int64_t row_stride, const float* bonus, const cirs_traj* traj, int32_t n_env,
                                   int32_t t_begin, int32_t t_end, int32_t softmax, float epsilon, uint64_t seed, uint32_t rng_base,
                                   uint32_t* visited, int32_t force_length, int64_t* obs_scratch, void* stream) {
    using namespace cirs;
    CIRS_REQUIRE(env_cfg && env_tab && env_st && scores && traj && obs_scratch, "null argument");
    CIRS_REQUIRE(traj->act && traj->rew && traj->done && traj->value && traj->ctr, "trajectory pointer null");
    CIRS_REQUIRE(n_env > 0 && t_begin >= 0 && t_end <= env_cfg->max_turn && t_begin <= t_end, "bad step range");
    hipStream_t s = (hipStream_t)stream;
    const long B = n_env;
    for (int t = t_begin; t < t_end; ++t) {
        int64_t* act_t = traj->act + (size_t)t * B;
        uint8_t* done_t = traj->done + (size_t)t * B;
        if (int rc = cirs_select_items(scores, row_stride, n_env, env_cfg->n_items, softmax, bonus, visited, env_st->done, epsilon, nullptr,
                                       seed, rng_base + (uint32_t)t, act_t, traj->value + (size_t)t * B, stream))
            return rc;
        if (visited) {
            hipLaunchKernelGGL(mark_visited_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, s, act_t, n_env, env_cfg->n_items, visited);
            CIRS_CHECK_LAUNCH("mark_visited_kernel");
        }
        if (int rc = cirs_env_step(env_cfg, env_tab, env_st, act_t, nullptr, n_env, obs_scratch, traj->rew + (size_t)t * B, done_t,
                                   traj->ctr + (size_t)t * B, nullptr, stream))
            return rc;
        if (force_length > 0) {
            hipLaunchKernelGGL(force_done_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, s, env_st->done, done_t, act_t, n_env,
                               (t + 1 >= force_length) ? 1 : 0);
            CIRS_CHECK_LAUNCH("force_done_kernel");
        }
    }
    return CIRS_OK;
}
