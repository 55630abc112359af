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

extern "C" int benerf_version(void) { return BENERF_ABI_VERSION; }      // include/benerf_hip.h
extern "C" const char* benerf_last_error(void) { return g_err; }

// Range guard of the split-f16 MLP mode (include/benerf_hip.h, benerf_mlp_status_check): the only entry point that
// synchronises - it has to, a device-side condition cannot reach a return code otherwise.
extern "C" int benerf_mlp_status_check(const uint32_t* status, benerf_stream_t stream) {
    BENERF_REQUIRE(status, "mlp_status_check: null pointer");
    uint32_t h[BENERF_ST_WORDS] = {0};
    if (hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, as_stream(stream)) != hipSuccess ||
        hipStreamSynchronize(as_stream(stream)) != hipSuccess) {
        benerf_set_error("mlp_status_check: copy failed: %s", hipGetErrorString(hipGetLastError()));
        return BENERF_EHIP;
    }
    float act, grad;
    memcpy(&act, &h[BENERF_ST_ACT], 4);
    memcpy(&grad, &h[BENERF_ST_GRAD], 4);
    if (h[BENERF_ST_SKIPPED] && act < 65504.f && grad < 65504.f) {   // steps gated since the host last cleared the words
        memcpy(&act, &h[BENERF_ST_LAST_ACT], 4);
        memcpy(&grad, &h[BENERF_ST_LAST_GRAD], 4);
        benerf_set_error("mlp: %u of %u training steps were skipped (%u in a row at the end) - an activation (max %g) or a scaled "
                         "gradient (max %g; inf: the loss gradient itself was not finite) left the f16 range (65504) of the split mode on "
                         "this or another rank; train with BENERF_MLP_F32 unless the loss is NaN",
                         h[BENERF_ST_SKIPPED], h[BENERF_ST_STEPS], h[BENERF_ST_CONSECUTIVE], (double)act, (double)grad);
        return BENERF_ERANGE;
    }
    if (h[BENERF_ST_MODE]) {
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

// One size query for every caller-owned scratch buffer (include/benerf_hip.h).
extern "C" size_t benerf_workspace_bytes(int which, int64_t n_points, int n_poses, int n_pix) {
    switch (which) {
        case BENERF_WS_MLP_ACTS: return sizeof(float) * benerf_mlp_act_floats(n_points);
        case BENERF_WS_MLP_DACTS: return sizeof(float) * benerf_mlp_dact_floats(n_points);
        case BENERF_WS_MLP_DW: return sizeof(float) * benerf_mlp_dw_workspace_floats(n_points);
        case BENERF_WS_MLP_PACKED: return sizeof(float) * benerf_mlp_packed_floats();
        case BENERF_WS_RAYS_BWD: return sizeof(float) * benerf_rays_bwd_workspace_floats(n_poses, n_pix);
        default: return 0;
    }
}
