// dataprep.hip -- the two O(big) loops of the user-model dataset preparation:
//   cirs_exposure_history   compute_exposure_each_user (reference core/util.py:56-76, numba): for the i-th logged interaction
//                           of a user, exposure = sum_{j<i} exp(-(ts_i - ts_j) * dist[photo_j, photo_i] / tau) in float64, a
//                           zero time difference counting as 1; the user's first interaction keeps 0.  O(L^2) per user
//                           (KuaiRec big matrix: ~1.1e10 terms) -- one wavefront per interaction, lanes stride the history.
//   cirs_find_negative      find_negative (core/util.py:173-196, numba): the nearest item id above (else below) the positive
//                           one that the user has interacted with in neither matrix, skipping the absent id 1225.  Integer.
// Distances come from a table (row-major [n_items, n_items], the reference's 1 / similarity matrix) or, when it is null,
// from the packed category words (1 / Jaccard, inf when disjoint), exactly like the env kernel.
#include "common.h"

namespace cirs {

__global__ __launch_bounds__(256) void exposure_history_kernel(const int64_t* __restrict__ user_start, const int32_t* __restrict__ photo,
                                                               const double* __restrict__ timestamp, long n_rows, const double* __restrict__ dist,
                                                               const uint32_t* __restrict__ item_cats, int n_items, double tau,
                                                               double* __restrict__ exposure_out) {
    const int lane = threadIdx.x & 63;
    const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const long s0 = user_start[r];   // first row of this row's user (rows of a user are contiguous, util.py:158-160)
    const int pi = photo[r];
    const double ti = timestamp[r];
    const uint32_t ci = item_cats ? item_cats[pi] : 0u;
    double acc = 0.0;
    for (long j = s0 + lane; j < r; j += CIRS_WAVE) {
        double td = ti - timestamp[j];
        if (td == 0.0) td = 1.0;     // util.py:67 "important!"
        const int pj = photo[j];
        const double d = dist ? dist[(size_t)pj * n_items + pi] : jaccard_dist(item_cats[pj], ci);
        acc += exp(-td * d / tau);
    }
    acc = wave_sum_f64(acc);
    if (lane == 0) exposure_out[r] = acc;
}

// seen_small / seen_big: bitmaps [n_users, words], bit (u, i) = the user interacted with item i
__global__ __launch_bounds__(256) void find_negative_kernel(const int64_t* __restrict__ user_ids, const int64_t* __restrict__ photo_ids, long n,
                                                            const uint32_t* __restrict__ seen_small, const uint32_t* __restrict__ seen_big,
                                                            int words, long max_item, long absent_id, int64_t* __restrict__ neg_out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long user = user_ids[i], item = photo_ids[i];
    const uint32_t* a = seen_small + (size_t)user * words;
    const uint32_t* b = seen_big + (size_t)user * words;
    auto seen = [&](long k) { return (((a[k >> 5] | b[k >> 5]) >> (k & 31)) & 1u) != 0; };
    long found = -1;
    long neg = item + 1;
    while (neg <= max_item) {
        if (neg == absent_id) neg = absent_id + 1;
        if (neg > max_item) break;
        if (seen(neg)) ++neg;
        else { found = neg; break; }
    }
    if (found < 0) {
        neg = item - 1;
        while (neg >= 0) {
            if (neg == absent_id) neg = absent_id - 1;
            if (neg < 0) break;
            if (seen(neg)) --neg;
            else { found = neg; break; }
        }
    }
    neg_out[i] = found;   // -1: every other item was seen (the reference leaves the row at its zero initialisation)
}

}  // namespace cirs

extern "C" int cirs_exposure_history(const int64_t* user_start, const int32_t* photo, const double* timestamp, int64_t n_rows,
                                     const double* dist, const uint32_t* item_cats, int32_t n_items, double tau, double* exposure_out,
                                     void* stream) {
    using namespace cirs;
    if (n_rows <= 0) return CIRS_OK;
    CIRS_REQUIRE(user_start && photo && timestamp && exposure_out && n_items > 0, "null argument");
    CIRS_REQUIRE(dist || item_cats, "need a distance table or the packed category words");
    CIRS_REQUIRE(tau > 0.0, "tau must be positive (tau == 0 means no exposure effect: the caller skips the computation, CIRS-UserModel-kuaishou.py:139-141)");
    hipLaunchKernelGGL(exposure_history_kernel, dim3(cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, user_start, photo, timestamp,
                       (long)n_rows, dist, item_cats, n_items, tau, exposure_out);
    CIRS_CHECK_LAUNCH("exposure_history_kernel");
    return CIRS_OK;
}

extern "C" int cirs_find_negative(const int64_t* user_ids, const int64_t* photo_ids, int64_t n, const uint32_t* seen_small,
                                  const uint32_t* seen_big, int32_t n_items, int64_t absent_id, int64_t* neg_out, void* stream) {
    using namespace cirs;
    if (n <= 0) return CIRS_OK;
    CIRS_REQUIRE(user_ids && photo_ids && seen_small && seen_big && neg_out && n_items > 0, "null argument");
    hipLaunchKernelGGL(find_negative_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, user_ids, photo_ids, (long)n, seen_small,
                       seen_big, (n_items + 31) / 32, (long)n_items - 1, (long)absent_id, neg_out);
    CIRS_CHECK_LAUNCH("find_negative_kernel");
    return CIRS_OK;
}
