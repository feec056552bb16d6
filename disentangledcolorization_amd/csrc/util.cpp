// util.cpp — error reporting shared by the translation units.
#include <cstdarg>
#include <cstdio>
#include "common.h"

namespace disco {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return DISCO_EHIP;
}
const char* last_error() { return g_err; }
}  // namespace disco

extern "C" const char* disco_last_error(void) { return disco::last_error(); }
extern "C" int disco_abi_version(void) { return DISCO_ABI_VERSION; }
