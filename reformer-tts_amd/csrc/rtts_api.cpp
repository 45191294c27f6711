// Version and thread-local error reporting of librtts_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <atomic>
#include "../../include/rtts.h"

static thread_local char g_err[512] = "";

extern "C" void rtts_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* rtts_last_error(void) { return g_err; }
extern "C" int rtts_version(void) { return RTTS_VERSION; }

// Test-only: force the run length of the walking LSH attention kernels (chunks a workgroup works in a row).  -1 = the library's
// own pick (the default, and what every product call gets), 0 = the one-chunk kernel, n >= 1 = runs of n where n divides the
// ring.  Process-wide and atomic; the launch path reads two words and no environment.
static std::atomic<int> g_walk[2] = {{-1}, {-1}};
int rtts_walk_override(int backward) { return g_walk[backward ? 1 : 0].load(std::memory_order_relaxed); }
extern "C" int rtts_debug_set_walk(int fwd_run, int bwd_run) {
    if (fwd_run < -1 || fwd_run > 16 || bwd_run < -1 || bwd_run > 64) {
        rtts_set_error("rtts_debug_set_walk: run lengths are -1 (library's pick), 0 (one-chunk kernel) or 1..16 / 1..64 (got %d, %d)", fwd_run, bwd_run);
        return -1;
    }
    g_walk[0].store(fwd_run, std::memory_order_relaxed);
    g_walk[1].store(bwd_run, std::memory_order_relaxed);
    return 0;
}
