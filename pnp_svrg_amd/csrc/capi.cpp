// capi.cpp -- error plumbing of the C ABI (include/pnp_hip.h).
#include "common.h"

namespace {
thread_local std::string g_last_error;
}

namespace pnp {
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace pnp

extern "C" int pnp_version(void) { return 100; }
extern "C" const char* pnp_last_error(void) { return g_last_error.c_str(); }
