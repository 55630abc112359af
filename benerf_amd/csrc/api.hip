// Library-level entry points: version + thread-local last-error text.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void benerf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int benerf_version(void) { return 100; }
extern "C" const char* benerf_last_error(void) { return g_err; }

static int g_mlp_precision = BENERF_MLP_SPLIT;
extern "C" int benerf_set_mlp_precision(int mode) {
    if (mode != BENERF_MLP_F32 && mode != BENERF_MLP_SPLIT) {
        benerf_set_error("set_mlp_precision: unknown mode %d", mode);
        return BENERF_EBADARG;
    }
    g_mlp_precision = mode;
    return BENERF_OK;
}
extern "C" int benerf_get_mlp_precision(void) { return g_mlp_precision; }
