// internal.h -- entry points shared between translation units of libcirs_hip.so (not part of the C ABI).
#pragma once
#include "common.h"

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

// per-kernel timing hook (api.hip): ids 1 = head_bwd_fused_kernel, 2 = actor_head_kernel<stats>, 3 = actor_head_kernel<sample>
bool prof_before(int kernel_id, hipStream_t s);
void prof_after(hipStream_t s);
#define CIRS_PROF_LAUNCH(ID, STREAM, ...)                       \
    do {                                                        \
        const bool prof_ = ::cirs::prof_before(ID, STREAM);     \
        __VA_ARGS__;                                            \
        if (prof_) ::cirs::prof_after(STREAM);                  \
    } while (0)

int tracker_step_internal(const cirs_tracker_cfg* cfg, const cirs_tracker_weights* w, cirs_tracker_state* st, const int32_t* users,
                          const int64_t* items, const double* rew, const int32_t* env_ids, const uint8_t* skip, int n, float* state_out,
                          long state_stride, const TrunkFuse* tf, hipStream_t s);

}  // namespace cirs
