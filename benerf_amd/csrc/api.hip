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

// Range guard of the split-f16 MLP mode (include/benerf_hip.h, benerf_mlp_status_check): the only entry point that
// synchronises - it has to, a device-side condition cannot reach a return code otherwise.
extern "C" int benerf_mlp_status_check(const uint32_t* status, benerf_stream_t stream) {
    BENERF_REQUIRE(status, "mlp_status_check: null pointer");
    uint32_t h[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, as_stream(stream)) != hipSuccess ||
        hipStreamSynchronize(as_stream(stream)) != hipSuccess) {
        benerf_set_error("mlp_status_check: copy failed: %s", hipGetErrorString(hipGetLastError()));
        return BENERF_EHIP;
    }
    float act, grad;
    memcpy(&act, &h[0], 4);
    memcpy(&grad, &h[1], 4);
    if (h[2]) {
        benerf_set_error("mlp: a backward launch was handed activation buffers written in another precision mode");
        return BENERF_EBADARG;
    }
    if (!(act < 65504.f) || !(grad < 65504.f)) {
        benerf_set_error("mlp(split): %s magnitude %g left the f16 range (65504); re-run with BENERF_MLP_F32 or BENERF_MLP_AUTO",
                         !(act < 65504.f) ? "activation" : "scaled gradient", (double)(!(act < 65504.f) ? act : grad));
        return BENERF_ERANGE;
    }
    return BENERF_OK;
}
