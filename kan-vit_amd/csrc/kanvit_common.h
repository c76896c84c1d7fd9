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

// Run-time switches of the library (fallback-kernel selection for the parity tests, two tuning knobs).  They are read from
// the KANVIT_* environment variables ONCE, when the library is first used -- never on the launch path -- and the active set
// is reported by kanvit_config() so that a measurement can state what it ran.  kanvit_config_reload() re-reads them (tests).
struct KvConfig {
    int no_reg;          // KANVIT_NO_REG          register-form KAN kernels off (LDS-tile kernels everywhere)
    int no_reg_bw;       // KANVIT_NO_REG_BW       register-form weight gradient off
    int bw_no_t16;       // KANVIT_BW_NO_T16       B-spline weight gradient on 32-row tiles (5 + 4 windows) instead of the 16-row-tile kernel
    int no_fast;         // KANVIT_NO_FAST         predicate-free variants of the LDS-tile kernels off
    int no_pipe;         // KANVIT_NO_PIPE         fp32 register kernels without the one-step-ahead LDS fragment prefetch (round-1 form)
    int no_ws;           // KANVIT_NO_WS           W-stationary bf16 forward off
    int no_bf16;         // KANVIT_NO_BF16         ignore KANVIT_FLAG_BF16_MFMA (exact fp32 kernels)
    int no_tiny;         // KANVIT_NO_TINY         the vector-pipe kernels for tiny per-head layers (I, O <= 16) off
    int no_fused_ln;     // KANVIT_NO_FUSED_LN     kanvit_layer_ln_fusable() answers 0: FastKAN's LayerNorm stays a separate op
    int attn_v1;         // KANVIT_ATTN_V1         first-form attention kernels
    int attn_v2;         // KANVIT_ATTN_V2         second-form fp32 attention kernels (round 1) instead of the pipelined third form
    int attn_v3;         // KANVIT_ATTN_V3         third-form fp32 attention kernels (round 2) instead of the LDS-DMA ring of the fourth form
    int attn_v4;         // KANVIT_ATTN_V4         fourth-form fp32 attention kernels (round 3: 32-row tiles) instead of the 16-row-tile kernels of round 4
    int attn_no_ds;      // KANVIT_ATTN_NO_DS      fp32 attention backward without the dS hand-off
    int ff_grid;         // KANVIT_FF_GRID         work-groups of the fused small feed-forward backward (tuning; 0 = default)
    int attn_grid;       // KANVIT_ATTN_GRID       work-groups of the persistent attention kernels (tuning; 0 = one round of resident ones)
    int bf16_nsh;        // KANVIT_BF16_NSH        LDS-tile bf16 forward: groups per basis tile (tuning)
    int bf16_ic;         // KANVIT_BF16_IC         LDS-tile bf16 forward: feature chunk cap (tuning)
    int bs_bw_bf16;      // KANVIT_BSPLINE_BW_BF16 B-spline weight gradient in bf16 mode: 0 = default (32-row bf16 register kernel), 1 = LDS-tile bf16 kernel, 2 = exact 16-row kernel (A/B)
    int bw_no_dma;       // KANVIT_BW_NO_DMA       ChebyKAN weight gradient in the register-ring form of rounds 1-3 instead of the LDS-DMA form (A/B)
    int bw_dma_force;    // KANVIT_BW_DMA_FORCE    the LDS-DMA weight gradient also where its launch would not fill the chip (small shapes: the parity tests)
    int ws_no_strip;     // KANVIT_WS_NO_STRIP     W-stationary bf16 forward stores straight from the accumulators (rounds 1-3) instead of through the LDS strips (A/B)
    int bi_no_res;       // KANVIT_BI_NO_RES       bf16 input gradient of the per-head layers in the streaming form of rounds 2-3 instead of the dY-resident form (A/B)
    int tail;            // KANVIT_TAIL            row tiles of a q|k|v launch that run as sub-divided work-groups at the end of the grid: -1 = automatic (default), 0 = off, k = k tiles (tuning)
    char text[544];
};
const KvConfig& kv_config();

template <typename K>
static inline hipError_t kv_allow_lds(K kernel, size_t bytes) {
    // dynamic LDS above the 64 KiB default needs an explicit opt-in (gfx950 has 160 KiB per CU)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

// hipFuncSetAttribute is per DEVICE: remember which devices of this process have had the opt-in for this kernel
// (one 64-bit mask per call site; the call itself is idempotent, so a lost race only repeats it).
#define KV_ALLOW_LDS(bytes, ...)                                                                  \
    do {                                                                                          \
        static unsigned long long kv_mask_ = 0ull;                                                \
        int kv_dev_ = 0;                                                                          \
        KV_HIP_CHECK(hipGetDevice(&kv_dev_));                                                     \
        const unsigned long long kv_bit_ = 1ull << (kv_dev_ & 63);                                \
        if (!(__atomic_load_n(&kv_mask_, __ATOMIC_RELAXED) & kv_bit_)) {                          \
            KV_HIP_CHECK(kv_allow_lds(__VA_ARGS__, bytes));                                       \
            __atomic_fetch_or(&kv_mask_, kv_bit_, __ATOMIC_RELAXED);                              \
        }                                                                                         \
    } while (0)

// tiny per-head layers on the vector pipe (csrc/kan_tiny.hip), called by kan_layer.hip's entry points
struct KvTinyArgs {
    const float* x;
    const float* w;
    const float* bp;
    const float* bias;
    const float* dy;
    float* y;
    float* dx;
    float* slab;
    long long M, ldx, ldy, bp_stride, rows_per_slab;
    int family, I, O, groups, xmod, G, GP, K, order, nk, has_base, flags, slabs;
};
bool kv_tiny_ok(const kanvit_layer_desc* d);
int kv_tiny_slabs(const kanvit_layer_desc* d);
int kv_tiny_fwd(const KvTinyArgs& a, hipStream_t st);
int kv_tiny_bwd_input(const KvTinyArgs& a, hipStream_t st);
int kv_tiny_bwd_weight(const KvTinyArgs& a, hipStream_t st);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Row of a 32x32 MFMA accumulator tile held in register `reg` of a lane in half `hf` (= lane >> 5);
// the column is lane & 31 (cdna guide section 3, C/D layout of the 32x32 shapes).
__device__ __forceinline__ int kv_acc_row(int reg, int hf) { return (reg & 3) + 8 * (reg >> 2) + 4 * hf; }
