// api.hip -- error plumbing + version of libcirs_hip.
#include "common.h"

namespace cirs {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace cirs

extern "C" const char* cirs_last_error(void) { return cirs::g_last_error.c_str(); }
extern "C" int cirs_version(void) { return 100; }
