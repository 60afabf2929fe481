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

// ---- phase timestamps inside a captured step (profiling aid): one lane stores the constant-rate wall clock (100 MHz) when the
// stream reaches this point.  A graph-safe alternative to rocprofv3 for the phase structure of a replay, whose host-side
// interception makes short replays host-bound (tools/step_phases.py).
namespace {
__global__ void timestamp_kernel(unsigned long long* slot) { if (threadIdx.x == 0) *slot = wall_clock64(); }
}
// one wave that samples {wall clock (100 MHz), shader clock} every `period` wall ticks, `samples` times: the clock its CU runs at while
// other streams do the work (tools/step_clock.py: shader clock by phase of a replayed step)
__global__ void clock_monitor_kernel(unsigned long long* buf, int samples, int period) {
    if (threadIdx.x != 0) return;
    unsigned long long next = wall_clock64();
    for (int i = 0; i < samples; ++i) {
        unsigned long long w;
        while ((w = wall_clock64()) < next) __builtin_amdgcn_s_sleep(8);
        buf[2 * i] = w;
        buf[2 * i + 1] = clock64();
        next += period;
    }
}
extern "C" int mrg_debug_clock_monitor(unsigned long long* buf, int samples, int period, hipStream_t stream) {
    MRG_REQUIRE(buf != nullptr && samples > 0 && period > 0, MRG_ERR_ARG, "mrg_debug_clock_monitor: bad arguments");
    hipLaunchKernelGGL(clock_monitor_kernel, dim3(1), dim3(64), 0, stream, buf, samples, period);
    return mrg_check_launch("mrg_debug_clock_monitor");
}
extern "C" int mrg_debug_timestamp(unsigned long long* slot, hipStream_t stream) {
    MRG_REQUIRE(slot != nullptr, MRG_ERR_ARG, "mrg_debug_timestamp: null pointer");
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(64), 0, stream, slot);
    return mrg_check_launch("mrg_debug_timestamp");
}
