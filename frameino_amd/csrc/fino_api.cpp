// Error reporting + version of libframeino_hip.so (C ABI: include/frameino_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/frameino_hip.h"

static thread_local char g_err[512] = "";

void fino_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fino_last_error(void) { return g_err; }
#ifdef FINO_EXPERIMENT
extern "C" int fino_version(void) { return -FINO_VERSION; }   // timing-experiment build: never the product (see the header)
#else
extern "C" int fino_version(void) { return FINO_VERSION; }
#endif

// ---- tuning knobs (diagnostics / A-B timing in one process; never needed for correct results) ----
static std::atomic<int> g_tune[FINO_TUNE_COUNT];

extern "C" int fino_tune_set(int key, int value) {
    if (key < 0 || key >= FINO_TUNE_COUNT) {
        fino_set_error("fino_tune_set: unknown key %d", key);
        return FINO_ERR_ARG;
    }
    g_tune[key].store(value, std::memory_order_relaxed);
    return FINO_OK;
}
extern "C" int fino_tune_get(int key) {
    return (key < 0 || key >= FINO_TUNE_COUNT) ? 0 : g_tune[key].load(std::memory_order_relaxed);
}
