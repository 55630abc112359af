// Shared host/device helpers for libbenerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/benerf_hip.h"

void benerf_set_error(const char* fmt, ...);

#define BENERF_REQUIRE(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            benerf_set_error(__VA_ARGS__);                          \
            return BENERF_EBADARG;                                  \
        }                                                           \
    } while (0)

#define BENERF_LAUNCH_CHECK(name)                                                        \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            benerf_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return BENERF_EHIP;                                                          \
        }                                                                                \
    } while (0)

static inline hipStream_t as_stream(benerf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the function ON THE CURRENT DEVICE: set once per (call site,
// device), not once per process - a process that launches on cuda:1 after cuda:0 needs it there too.
struct BenerfLdsAttr { unsigned long long done; };
static inline bool benerf_lds_attr(BenerfLdsAttr& st, const void* fn, int bytes) {
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (known && ((st.done >> dev) & 1ull)) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (known) st.done |= 1ull << dev;      // benign race: the call is idempotent
    return true;
}

// ---------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (speed mode; parity mode passes explicit draw tensors).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) {  // [0,1), 24 bits
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
// element `idx` of stream (seed, offset): one uniform in [0,1)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint64_t c = idx >> 2;
    uint4 r = philox4x32_10(make_uint4((uint32_t)c, (uint32_t)(c >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)),
                            make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    uint32_t v = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
    return u32_to_unit(v);
}
// element `idx` of stream (seed, offset): one standard normal (Box-Muller on two uniforms)
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint64_t c = idx >> 1;
    uint4 r = philox4x32_10(make_uint4((uint32_t)c, (uint32_t)(c >> 32), (uint32_t)offset, (uint32_t)(offset >> 32) ^ 0x80000000u),
                            make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    uint32_t a = (idx & 1) ? r.z : r.x, b = (idx & 1) ? r.w : r.y;
    float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
    float u2 = u32_to_unit(b);
    float rad = sqrtf(-2.0f * logf(u1));
    return rad * cosf(6.28318530717958647692f * u2);
}
