// Shared host-side helpers of the kanvit C-ABI library (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/kanvit.h"

extern thread_local char g_kanvit_err[512];

static inline int kv_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kanvit_err, sizeof(g_kanvit_err), fmt, ap);
    va_end(ap);
    return code;
}

#define KV_HIP_CHECK(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return kv_fail(KANVIT_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                                 \
    } while (0)

// Launch-time check: peek (not get) so that a sticky error of an unrelated earlier call is not
// swallowed, and nothing synchronises (graph capture safe).
#define KV_LAUNCH_CHECK(what)                                                                   \
    do {                                                                                        \
        hipError_t _e = hipGetLastError();                                                      \
        if (_e != hipSuccess)                                                                   \
            return kv_fail(KANVIT_EDEVICE, "launch of %s failed: %s", what, hipGetErrorString(_e)); \
    } while (0)

template <typename K>
static inline hipError_t kv_allow_lds(K kernel, size_t bytes) {
    // dynamic LDS above the 64 KiB default needs an explicit opt-in (gfx950 has 160 KiB per CU)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Row of a 32x32 MFMA accumulator tile held in register `reg` of a lane in half `hf` (= lane >> 5);
// the column is lane & 31 (cdna guide section 3, C/D layout of the 32x32 shapes).
__device__ __forceinline__ int kv_acc_row(int reg, int hf) { return (reg & 3) + 8 * (reg >> 2) + 4 * hf; }
