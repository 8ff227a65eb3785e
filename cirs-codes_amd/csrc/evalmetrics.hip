// evalmetrics.hip -- coverage / feature-domination counts of the CIRS evaluation callbacks on device trajectories
// (reference evaluation.py:286-371 Callback_Coverage_Count, :10-77 get_feat_dominate_dict).  Integer work only:
//   hit_item   = number of distinct recommended items   (CV = hit_item / n_items, CV_turn = hit_item / n_acts)
//   n_acts     = number of recommendations in the collected episodes
//   n_flagged  = recommendations whose item carries one of the dominating feature values (ifeat_feat = n_flagged / n_acts)
// The trajectory is the time-major act tensor of the rollout ([T,B] int64, -1 once an env has finished).  Distinct items
// are marked in a bitmap with integer atomics and counted with popcount: bit-exact, independent of scheduling.
#include "common.h"

namespace cirs {

__global__ __launch_bounds__(256) void coverage_mark_kernel(const int64_t* __restrict__ act, long n, int n_items,
                                                            const uint8_t* __restrict__ item_flag, uint32_t* __restrict__ bitmap,
                                                            unsigned long long* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long a = i < n ? act[i] : -1;
    const bool valid = a >= 0 && a < n_items;
    if (valid) atomicOr(&bitmap[a >> 5], 1u << (a & 31));
    const bool flagged = valid && item_flag && item_flag[a] != 0;
    const unsigned long long mv = __ballot(valid), mf = __ballot(flagged);
    if ((threadIdx.x & 63) == 0) {
        if (mv) atomicAdd(&out[1], (unsigned long long)__popcll(mv));
        if (mf) atomicAdd(&out[2], (unsigned long long)__popcll(mf));
    }
}

__global__ __launch_bounds__(256) void coverage_count_kernel(const uint32_t* __restrict__ bitmap, int n_words,
                                                             unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = i < n_words ? __popc(bitmap[i]) : 0;
    c = wave_sum_i32(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&out[0], (unsigned long long)c);
}

}  // namespace cirs

extern "C" int cirs_eval_coverage(const int64_t* act, int64_t n, int32_t n_items, const uint8_t* item_flag, uint32_t* bitmap,
                                  int64_t* out3, void* stream) {
    using namespace cirs;
    CIRS_REQUIRE(bitmap && out3 && n_items > 0 && n >= 0, "bad arguments");
    CIRS_REQUIRE(n == 0 || act, "act is null");
    hipStream_t s = (hipStream_t)stream;
    const int n_words = (n_items + 31) / 32;
    CIRS_HIP(hipMemsetAsync(bitmap, 0, sizeof(uint32_t) * (size_t)n_words, s));
    CIRS_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(int64_t), s));
    if (n > 0) {
        hipLaunchKernelGGL(coverage_mark_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, act, (long)n, n_items, item_flag, bitmap,
                           (unsigned long long*)out3);
        hipLaunchKernelGGL(coverage_count_kernel, dim3(cdiv(n_words, 256)), dim3(256), 0, s, bitmap, n_words, (unsigned long long*)out3);
    }
    CIRS_CHECK_LAUNCH("cirs_eval_coverage");
    return CIRS_OK;
}
