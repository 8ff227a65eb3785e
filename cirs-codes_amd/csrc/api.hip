// api.hip -- error plumbing + version of libcirs_hip.
#include <vector>
#include "common.h"

namespace cirs {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace cirs

// ---- per-kernel timing hook (bench.py's roofline leg) -----------------------------------------------------------------
// cirs_prof_start(kernel_id, max_samples) arms HIP-event pairs around every launch of ONE named kernel on the stream it is
// launched on; cirs_prof_stop synchronises those events and returns the summed duration.  Off by default: zero cost.
namespace cirs {
static int g_prof_id = 0, g_prof_max = 0, g_prof_n = 0;
static std::vector<hipEvent_t> g_prof_ev;
bool prof_before(int kernel_id, hipStream_t s) {
    if (kernel_id != g_prof_id || g_prof_n >= g_prof_max) return false;
    return hipEventRecord(g_prof_ev[2 * g_prof_n], s) == hipSuccess;
}
void prof_after(hipStream_t s) {
    (void)hipEventRecord(g_prof_ev[2 * g_prof_n + 1], s);
    ++g_prof_n;
}
}  // namespace cirs

extern "C" int cirs_prof_start(int32_t kernel_id, int32_t max_samples) {
    using namespace cirs;
    CIRS_REQUIRE(kernel_id >= 1 && kernel_id <= 3 && max_samples >= 1 && max_samples <= 65536, "bad arguments");
    while ((int)g_prof_ev.size() < 2 * max_samples) {
        hipEvent_t e;
        CIRS_HIP(hipEventCreate(&e));
        g_prof_ev.push_back(e);
    }
    g_prof_id = kernel_id; g_prof_max = max_samples; g_prof_n = 0;
    return CIRS_OK;
}

extern "C" int cirs_prof_stop(double* total_seconds, int32_t* n_samples) {
    using namespace cirs;
    CIRS_REQUIRE(total_seconds && n_samples, "null argument");
    double tot = 0.0;
    for (int i = 0; i < g_prof_n; ++i) {
        CIRS_HIP(hipEventSynchronize(g_prof_ev[2 * i + 1]));
        float ms = 0.f;
        CIRS_HIP(hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]));
        tot += (double)ms * 1e-3;
    }
    *total_seconds = tot; *n_samples = g_prof_n;
    g_prof_id = 0; g_prof_max = 0; g_prof_n = 0;
    return CIRS_OK;
}

extern "C" const char* cirs_last_error(void) { return cirs::g_last_error.c_str(); }
extern "C" int cirs_version(void) { return 100; }
