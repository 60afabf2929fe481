// Error plumbing + version for libmirage_hip.so.
#include "common.h"
#include "mirage_hip.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

int mrg_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int mrg_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" const char* mrg_last_error_string() { return g_err; }
extern "C" int mrg_version() { return MRG_ABI_VERSION; }
