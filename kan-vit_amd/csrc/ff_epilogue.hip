// Backward epilogue of the feed-forward's first Linear + ReLU (reference model.py:25-29): the ReLU mask
// of the incoming gradient and the bias gradient in ONE pass over [M, N].
//     dh[m][n]  = y[m][n] <= 0 ? 0 : dy[m][n]         (what autograd's threshold_backward does for ReLU(inplace) on the saved y)
//     dbias[n]  = sum_m dh[m][n]
// Stock torch runs two passes (threshold_backward: read dy, y, write dh; then sum(0): read dh again -- 12 x 310 MB per ViT-B
// step for the second one).  Here a work-group owns a band of rows x 1024 columns: each thread streams one float4 column group
// down the band (EP_ROWS rows in flight), writes the masked gradient and keeps its four column sums in registers; the bands'
// partial sums go to the workspace and are added in band order by a second, tiny kernel -- deterministic, no float atomics.
// HBM-bound: the sums ride for free on the traffic of the mask pass.  Measured on [25216, 3072] (tools/time_ff_epilogue.py): stock
// 227 us (157 + 60), this 177 us = 930 MB at 5.2 TB/s.  Rejected: whole rows per work-group (up to 4 column groups per thread:
// 224-281 us), 16 rows in flight with 2 work-groups per CU (same time, more registers), 8 work-groups per CU (186 us).
#include "../../include/kanvit.h"
#include "kanvit_common.h"

namespace {

#ifndef EP_ROWS
#define EP_ROWS 8                             // rows in flight per thread (tools/build_variant.sh -DEP_ROWS=4 for A/B timing)
#endif
#ifndef EP_WGS_PER_CU
#define EP_WGS_PER_CU 4
#endif
constexpr int EP_THREADS = 256;
constexpr int EP_COLS = EP_THREADS * 4;       // columns per work-group

// Every byte is touched once: non-temporal loads and stores (measured 185 -> 177 us on [25216, 3072]; stores alone, 187).
// (In the ViT-B step, 30 steps, A/B in one process each: stock epilogue 89.32 ms, this kernel with plain accesses 89.03, loads
// non-temporal 88.89, both 88.80.)
#define EP_LD(p) __builtin_nontemporal_load(p)
#define EP_ST(v, p) __builtin_nontemporal_store(v, p)
__global__ __launch_bounds__(EP_THREADS) void relu_bwd_bias_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                   float* __restrict__ dh, float* __restrict__ part, long long M, int N,
                                                                   long long rows_per_band) {
    const int c = (blockIdx.x * EP_THREADS + threadIdx.x) * 4;
    if (c >= N) return;
    const long long r0 = (long long)blockIdx.y * rows_per_band;
    long long r1 = r0 + rows_per_band;
    if (r1 > M) r1 = M;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    long long r = r0;
    for (; r + EP_ROWS <= r1; r += EP_ROWS) {             // EP_ROWS rows in flight
        f32x4 g[EP_ROWS], a[EP_ROWS];
#pragma unroll
        for (int q = 0; q < EP_ROWS; ++q) {
            g[q] = EP_LD(reinterpret_cast<const f32x4*>(dy + (r + q) * N + c));
            a[q] = EP_LD(reinterpret_cast<const f32x4*>(y + (r + q) * N + c));
        }
#pragma unroll
        for (int q = 0; q < EP_ROWS; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[q][e] = a[q][e] <= 0.0f ? 0.0f : g[q][e];      // threshold_backward's own comparison (a NaN activation passes the gradient)
            EP_ST(g[q], reinterpret_cast<f32x4*>(dh + (r + q) * N + c));
            s += g[q];                         // rows added in order
        }
    }
    for (; r < r1; ++r) {
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * N + c);
        const f32x4 a = *reinterpret_cast<const f32x4*>(y + r * N + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = a[e] <= 0.0f ? 0.0f : g[e];
        *reinterpret_cast<f32x4*>(dh + r * N + c) = g;
        s += g;
    }
    *reinterpret_cast<f32x4*>(part + (long long)blockIdx.y * N + c) = s;
}

// The same pass on bf16 tensors (torch.autocast: dy, y and dh are bf16; the column sums stay fp32, as torch's
// `dy.sum(0, dtype=float32)`): 8-byte loads of four bf16 per thread, the same 1024 columns per work-group and band partials.
typedef unsigned short ep_u16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(EP_THREADS) void relu_bwd_bias_bf16_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ y,
                                                                        unsigned short* __restrict__ dh, float* __restrict__ part, long long M,
                                                                        int N, long long rows_per_band) {
    const int c = (blockIdx.x * EP_THREADS + threadIdx.x) * 4;
    if (c >= N) return;
    const long long r0 = (long long)blockIdx.y * rows_per_band;
    long long r1 = r0 + rows_per_band;
    if (r1 > M) r1 = M;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int RB = 2 * EP_ROWS;                         // rows in flight (half the bytes per row)
    long long r = r0;
    for (; r + RB <= r1; r += RB) {
        ep_u16x4 g[RB], a[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            g[q] = EP_LD(reinterpret_cast<const ep_u16x4*>(dy + (r + q) * N + c));
            a[q] = EP_LD(reinterpret_cast<const ep_u16x4*>(y + (r + q) * N + c));
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // !(y <= 0) on the bf16 bit pattern (threshold_backward's comparison): positive and not zero, or a NaN of either sign
                const unsigned short yb = a[q][e];
                const bool pos = ((yb & 0x8000u) == 0 && yb != 0) || ((yb & 0x7f80u) == 0x7f80u && (yb & 0x007fu));
                g[q][e] = pos ? g[q][e] : (unsigned short)0;
                s[e] += __builtin_bit_cast(float, (unsigned)g[q][e] << 16);      // rows added in order
            }
            EP_ST(g[q], reinterpret_cast<ep_u16x4*>(dh + (r + q) * N + c));
        }
    }
    for (; r < r1; ++r) {
        ep_u16x4 g = *reinterpret_cast<const ep_u16x4*>(dy + r * N + c);
        const ep_u16x4 a = *reinterpret_cast<const ep_u16x4*>(y + r * N + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned short yb = a[e];
            const bool pos = ((yb & 0x8000u) == 0 && yb != 0) || ((yb & 0x7f80u) == 0x7f80u && (yb & 0x007fu));
            g[e] = pos ? g[e] : (unsigned short)0;
            s[e] += __builtin_bit_cast(float, (unsigned)g[e] << 16);
        }
        *reinterpret_cast<ep_u16x4*>(dh + r * N + c) = g;
    }
    *reinterpret_cast<f32x4*>(part + (long long)blockIdx.y * N + c) = s;
}

// out[c] = sum over the bands of part[band][c], in a fixed order: 32 columns x 8 band groups per work-group (group k adds bands
// k, k + 8, ... with eight loads in flight, then the eight group sums are added in order).  One thread per column walking all
// the bands is a serial chain of ~40 dependent round trips on 12 work-groups -- it took longer than the pass it finishes.
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int bands) {
    __shared__ float sub[8][33];
    const int cl = threadIdx.x & 31, k = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.0f;
    if (c < N) {
        int w = k;
        for (; w + 56 < bands; w += 64) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = part[(long long)(w + 8 * j) * N + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += t[j];
        }
        for (; w < bands; w += 8) s += part[(long long)w * N + c];
    }
    sub[k][cl] = s;
    __syncthreads();
    if (k == 0 && c < N) {
        float t = sub[0][cl];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += sub[j][cl];
        out[c] = t;
    }
}

int ep_bands(long long M, int N) {
    const long long gx = (N + EP_COLS - 1) / EP_COLS;
    long long bands = ((long long)EP_WGS_PER_CU * 256 + gx - 1) / gx;          // ~EP_WGS_PER_CU work-groups per CU over the chip
    const long long maxb = (M + 15) / 16;                  // at least 16 rows per band
    if (bands > maxb) bands = maxb;
    if (bands < 1) bands = 1;
    if (bands > 65535) bands = 65535;
    return (int)bands;
}

}  // namespace

extern "C" {

size_t kanvit_relu_bwd_bias_workspace(int64_t M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return sizeof(float) * (size_t)ep_bands(M, N) * (size_t)N;
}

static int relu_bwd_bias_impl(bool bf16, int64_t M, int N, const void* dy, const void* y, void* dh, float* dbias, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (M < 0 || N < 1 || (N & 3)) return kv_fail(KANVIT_EINVAL, "kanvit_relu_bwd_bias: N=%d must be a positive multiple of 4 (M=%lld)", N, (long long)M);
    if (!dbias) return kv_fail(KANVIT_EINVAL, "kanvit_relu_bwd_bias: null dbias");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        KV_HIP_CHECK(hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)N, st));
        return 0;
    }
    if (!dy || !y || !dh) return kv_fail(KANVIT_EINVAL, "kanvit_relu_bwd_bias: null dy/y/dh");
    if (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dh) & 15) return kv_fail(KANVIT_EINVAL, "kanvit_relu_bwd_bias: dy / y / dh must be 16-byte aligned");
    const size_t need = kanvit_relu_bwd_bias_workspace(M, N);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15))
        return kv_fail(KANVIT_ENOMEM, "kanvit_relu_bwd_bias: workspace %zu bytes < required %zu (or not 16-byte aligned)", workspace_bytes, need);
    const int bands = ep_bands(M, N);
    const long long rpb = (M + bands - 1) / bands;
    dim3 grid((unsigned)((N + EP_COLS - 1) / EP_COLS), (unsigned)((M + rpb - 1) / rpb), 1);
    if (bf16)
        hipLaunchKernelGGL(relu_bwd_bias_bf16_kernel, grid, dim3(EP_THREADS), 0, st, (const unsigned short*)dy, (const unsigned short*)y,
                           (unsigned short*)dh, (float*)workspace, (long long)M, N, rpb);
    else
        hipLaunchKernelGGL(relu_bwd_bias_kernel, grid, dim3(EP_THREADS), 0, st, (const float*)dy, (const float*)y, (float*)dh, (float*)workspace,
                           (long long)M, N, rpb);
    KV_LAUNCH_CHECK("relu_bwd_bias_kernel");
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, st, (const float*)workspace, dbias, N, (int)grid.y);
    KV_LAUNCH_CHECK("colsum_reduce_kernel");
    return 0;
}

int kanvit_relu_bwd_bias(int64_t M, int N, const float* dy, const float* y, float* dh, float* dbias, void* workspace,
                         size_t workspace_bytes, void* stream) {
    return relu_bwd_bias_impl(false, M, N, dy, y, dh, dbias, workspace, workspace_bytes, stream);
}

int kanvit_relu_bwd_bias_bf16(int64_t M, int N, const void* dy, const void* y, void* dh, float* dbias, void* workspace,
                              size_t workspace_bytes, void* stream) {
    return relu_bwd_bias_impl(true, M, N, dy, y, dh, dbias, workspace, workspace_bytes, stream);
}

}  // extern "C"
