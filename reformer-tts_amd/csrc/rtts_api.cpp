// Version and thread-local error reporting of librtts_hip.so.
#include <stdarg.h>
#include <stdio.h>
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
