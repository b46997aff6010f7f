// Error reporting + version of libframeino_hip.so (C ABI: include/frameino_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/frameino_hip.h"

static thread_local char g_err[512] = "";

void fino_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fino_last_error(void) { return g_err; }
extern "C" int fino_version(void) { return FINO_VERSION; }
