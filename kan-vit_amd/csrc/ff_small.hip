// Fused feed-forward for the small ("T" / "C", d = 64) geometries -- SURVEY.md section 8(f)1, reference model.py:25-29,36:
//     y = relu(x W1^T + b1) W2^T + b2          x[M][D], W1[F][D], W2[D][F]   (two nn.Linear around an in-place ReLU)
// At M = 6400 rows the two stock GEMMs and the dozen backward ops around them (ReLU mask, two dx GEMMs, two token-split
// weight-gradient bmm + sums, two bias sums, memsets) are ~5 us launches that do ~1 us of work each: a third of the
// MNIST-tiny step.  Here the forward is ONE launch and the backward ONE launch + an ordered reduce.
//
// fp32 on v_mfma_f32_32x32x2_f32, everything TRANSPOSED so that no accumulator ever has to be re-laid out:
//   a work-group owns a tile of 32 rows; its NW waves split the F hidden units (32*NTH each).
//   h^T[n][m] = sum_k W1[n][k] x[m][k]        A = W1 rows (lane = n), B = x rows (lane = m); the K order is permuted
//                                               (half h takes k = h*D/2 + s) so both operands are contiguous float4 loads
//   the accumulator holds h^T with lane = m, register = n.  For the second product
//   y^T[o][m] = sum_n W2[o][n] h^T[n][m]      the contraction index is the REGISTER index: register r of lane half h IS
//                                               the B operand of k-step (n = 8*(r>>2) + 4*h + (r&3)) -- bias + ReLU are applied
//                                               in registers and the value goes straight back into the matrix pipe.
//   Each wave holds a partial y^T over its hidden slice; the NW partials meet in LDS and are added in wave order.
// Backward recomputes h (cheaper than saving [M][F]), forms dh^T = W2^T dy^T the same way, masks it in registers, feeds it
// to dx^T = W1^T dh^T (again register = contraction index), and stages h^T / dh^T / x / dy tiles in LDS once for the two
// weight gradients (contraction over the 32 rows).  dW1, dW2^T, db1, db2 accumulate in registers over the tiles a
// work-group walks, go to one partial per work-group and are summed in fixed order by ff_small_reduce_kernel
// (deterministic; no atomics).
#include "../../include/kanvit.h"
#include "kanvit_common.h"

namespace {

struct FfArgs {
    const float* x;
    const float* w1;
    const float* b1;
    const float* w2;
    const float* b2;
    const float* dy;
    float* y;
    float* dx;
    float* part;      // [work-groups][2*F*D + F + D]: dW1[F][D] | dW2^T[F][D] | db1[F] | db2[D]
    long long M;
    int tiles;
};

template <int N>
__device__ __forceinline__ void ff_load_row(float (&r)[N], const float* p) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
        r[4 * i] = v[0];
        r[4 * i + 1] = v[1];
        r[4 * i + 2] = v[2];
        r[4 * i + 3] = v[3];
    }
}

__device__ __forceinline__ f32x16 ff_zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

// h^T tile of hidden units [n0, n0 + 32) for the 32 rows whose half-rows sit in xr: pre-activation sums
template <int D>
__device__ __forceinline__ f32x16 ff_hidden_tile(const float* __restrict__ w1, int n0, int l31, int hf, const float (&xr)[D / 2]) {
    float wv[D / 2];
    ff_load_row<D / 2>(wv, w1 + (size_t)(n0 + l31) * D + hf * (D / 2));
    f32x16 acc = ff_zero16();
#pragma unroll
    for (int s = 0; s < D / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[s], xr[s], acc, 0, 0, 0);
    return acc;
}

template <int D, int F, int NW>
__global__ __launch_bounds__(64 * NW) void ff_small_fwd_kernel(const FfArgs a) {
    constexpr int NTH = F / (32 * NW), NDT = D / 32, DH = D / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int nb = wave * 32 * NTH;
    extern __shared__ __attribute__((aligned(16))) float red[];      // [NW][D][33] partial y^T per wave
    for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        const long long m0 = (long long)tile * 32;
        long long mr = m0 + l31;
        if (mr >= a.M) mr = a.M - 1;
        float xr[DH];
        ff_load_row<DH>(xr, a.x + mr * D + hf * DH);
        f32x16 h[NTH];
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            h[t] = ff_hidden_tile<D>(a.w1, nb + t * 32, l31, hf, xr);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b1 + nb + t * 32 + 8 * q + 4 * hf);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[t][4 * q + e] = fmaxf(h[t][4 * q + e] + bv[e], 0.0f);
            }
        }
        f32x16 yacc[NDT];
#pragma unroll
        for (int ot = 0; ot < NDT; ++ot) {
            yacc[ot] = ff_zero16();
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.w2 + (size_t)(ot * 32 + l31) * F + nb + t * 32 + 8 * q + 4 * hf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) yacc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[e], h[t][4 * q + e], yacc[ot], 0, 0, 0);
                }
        }
#pragma unroll
        for (int ot = 0; ot < NDT; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * D + ot * 32 + kv_acc_row(r, hf)) * 33 + l31] = yacc[ot][r];
        __syncthreads();
        for (int idx = tid; idx < 32 * (D / 4); idx += 64 * NW) {
            const int m = idx / (D / 4), c = idx % (D / 4);
            if (m0 + m < a.M) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b2 + 4 * c);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = red[(4 * c + e) * 33 + m];
#pragma unroll
                    for (int w = 1; w < NW; ++w) s += red[(w * D + 4 * c + e) * 33 + m];
                    v[e] = s + bv[e];
                }
                *reinterpret_cast<f32x4*>(a.y + (m0 + m) * D + 4 * c) = v;
            }
        }
        __syncthreads();
    }
}

template <int D, int F, int NW>
__global__ __launch_bounds__(64 * NW) void ff_small_bwd_kernel(const FfArgs a) {
    constexpr int NTH = F / (32 * NW), NDT = D / 32, DH = D / 2, DS = D + 32;
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31_ = lane & 31, hf_ = lane >> 5;
    const int nb = wave * 32 * NTH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hs = smem;                      // [F][33]   relu(h)^T of the tile, hidden-major
    float* dhs = hs + F * 33;              // [F][33]   masked dh^T
    float* xs = dhs + F * 33;              // [32][DS]  x tile (rows past M: zero)
    float* dys = xs + 32 * DS;             // [32][DS]  dy tile
    float* red = dys + 32 * DS;            // [NW][D][33] partial dx^T per wave

    f32x16 gw1[NTH][NDT], gw2[NTH][NDT];   // dW1[n][k], dW2^T[n][o] of this wave's hidden slice
    float gb1[NTH][16];
    float gb2 = 0.0f;
#pragma unroll
    for (int t = 0; t < NTH; ++t) {
#pragma unroll
        for (int j = 0; j < NDT; ++j) {
            gw1[t][j] = ff_zero16();
            gw2[t][j] = ff_zero16();
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) gb1[t][r] = 0.0f;
    }

    // Tiles blockIdx.x, blockIdx.x + gridDim.x, ...: the weight-gradient accumulators stay in registers across them (one
    // partial per work-group instead of one per tile: 132 KB each).  Everything else of an iteration is tile-invariant to the
    // compiler -- weight fragments, ~100 addresses -- and loop-invariant code motion would hoist it all into registers and
    // spill the kernel (measured: 256 VGPRs + 400 bytes of scratch); the pointers and the lane index therefore pass through
    // an empty asm each iteration, which makes them opaque.
#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        const float *pw1 = a.w1, *pw2 = a.w2, *pb1 = a.b1;
        int l31 = l31_, hf = hf_;
        asm volatile("" : "+s"(pw1), "+s"(pw2), "+s"(pb1), "+v"(l31), "+v"(hf));
        const long long m0 = (long long)tile * 32;
        const bool live = m0 + l31 < a.M;
        const long long mr = live ? m0 + l31 : a.M - 1;
        // stage the x and dy tiles (the weight gradients contract over rows: lane = column there)
        for (int idx = tid; idx < 32 * (D / 4); idx += NT) {
            const int m = idx / (D / 4), c = idx % (D / 4);
            f32x4 xv = {0.0f, 0.0f, 0.0f, 0.0f}, dv = xv;
            if (m0 + m < a.M) {
                xv = *reinterpret_cast<const f32x4*>(a.x + (m0 + m) * D + 4 * c);
                dv = *reinterpret_cast<const f32x4*>(a.dy + (m0 + m) * D + 4 * c);
            }
            *reinterpret_cast<f32x4*>(xs + m * DS + 4 * c) = xv;
            *reinterpret_cast<f32x4*>(dys + m * DS + 4 * c) = dv;
        }
        float xr[DH], dyr[DH];
        ff_load_row<DH>(xr, a.x + mr * D + hf * DH);
        ff_load_row<DH>(dyr, a.dy + mr * D + hf * DH);
        if (!live) {
#pragma unroll
            for (int s = 0; s < DH; ++s) dyr[s] = 0.0f;
        }
        f32x16 dxacc[NDT];
#pragma unroll
        for (int kt = 0; kt < NDT; ++kt) dxacc[kt] = ff_zero16();
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            const int n0 = nb + t * 32;
            f32x16 h = ff_hidden_tile<D>(pw1, n0, l31, hf, xr);
            // dh^T[n][m] = sum_o W2[o][n] dy[m][o]      (half h takes o = h*D/2 + s)
            f32x16 dh = ff_zero16();
            {       // addresses as a wave-uniform base (scalar registers) + ONE per-lane 32-bit offset: a 64-bit address per load
                    // (the offsets do not fit the 12-bit immediate) costs two registers each and spills the kernel
                const int voff = hf * DH * F + l31;
#pragma unroll
                for (int s = 0; s < DH; ++s) {
                    const float* wrow = pw2 + (size_t)s * F + n0;      // uniform
                    dh = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[voff], dyr[s], dh, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(pb1 + n0 + 8 * q + 4 * hf);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    const float pre = h[r] + bv[e];
                    h[r] = fmaxf(pre, 0.0f);
                    dh[r] = pre > 0.0f ? dh[r] : 0.0f;       // ReLU mask (threshold_backward: gradient passes where the output is > 0)
                    gb1[t][r] += dh[r];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                hs[(n0 + kv_acc_row(r, hf)) * 33 + l31] = h[r];
                dhs[(n0 + kv_acc_row(r, hf)) * 33 + l31] = dh[r];
            }
            // dx^T[k][m] += sum_n W1[n][k] dh^T[n][m]: register r of dh is the operand of k-step n = n0 + row(r)
            {
                const int voff = 4 * hf * D + l31;
#pragma unroll
                for (int kt = 0; kt < NDT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float* wrow = pw1 + (size_t)(n0 + (r & 3) + 8 * (r >> 2)) * D + kt * 32;      // uniform
                        dxacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[voff], dh[r], dxacc[kt], 0, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int kt = 0; kt < NDT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * D + kt * 32 + kv_acc_row(r, hf)) * 33 + l31] = dxacc[kt][r];
        __syncthreads();
        for (int idx = tid; idx < 32 * (D / 4); idx += NT) {
            const int m = idx / (D / 4), c = idx % (D / 4);
            if (m0 + m < a.M) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = red[(4 * c + e) * 33 + m];
#pragma unroll
                    for (int w = 1; w < NW; ++w) s += red[(w * D + 4 * c + e) * 33 + m];
                    v[e] = s;
                }
                *reinterpret_cast<f32x4*>(a.dx + (m0 + m) * D + 4 * c) = v;
            }
        }
        if (tid < D) {
            float s = 0.0f;
#pragma unroll 8
            for (int m = 0; m < 32; ++m) s += dys[m * DS + tid];
            gb2 += s;
        }
        // weight gradients: contraction over the 32 rows of the tile (16 k-steps, half h takes row 2s + h)
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            const float* hrow = hs + (nb + t * 32 + l31) * 33 + hf;
            const float* drow = dhs + (nb + t * 32 + l31) * 33 + hf;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float ha = hrow[2 * s], da = drow[2 * s];
#pragma unroll
                for (int j = 0; j < NDT; ++j) {
                    gw2[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ha, dys[(2 * s + hf) * DS + j * 32 + l31], gw2[t][j], 0, 0, 0);
                    gw1[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(da, xs[(2 * s + hf) * DS + j * 32 + l31], gw1[t][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    const int l31 = l31_, hf = hf_;
    float* part = a.part + (size_t)blockIdx.x * (2 * F * D + F + D);
#pragma unroll
    for (int t = 0; t < NTH; ++t) {
#pragma unroll
        for (int j = 0; j < NDT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb + t * 32 + kv_acc_row(r, hf);
                part[(size_t)n * D + j * 32 + l31] = gw1[t][j][r];
                part[(size_t)F * D + (size_t)n * D + j * 32 + l31] = gw2[t][j][r];
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = gb1[t][r];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
            if (l31 == 0) part[2 * F * D + nb + t * 32 + kv_acc_row(r, hf)] = v;
        }
    }
    if (tid < D) part[2 * F * D + F + tid] = gb2;
}

// out[idx] = sum_p part[p][idx] in fixed order (8 interleaved sub-sums joined in order), scattered to dW1 / dW2 / db1 / db2
template <int D, int F>
__global__ __launch_bounds__(256) void ff_small_reduce_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw1,
                                                              float* __restrict__ dw2, float* __restrict__ db1, float* __restrict__ db2) {
    constexpr int PART = 2 * F * D + F + D;
    __shared__ float sub[8][33];
    const int cl = threadIdx.x & 31, k = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + cl;
    float s = 0.0f;
    if (idx < PART) {
        int p = k;
        for (; p + 24 < nparts; p += 32) {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = part[(size_t)(p + 8 * j) * PART + idx];
#pragma unroll
            for (int j = 0; j < 4; ++j) s += t[j];
        }
        for (; p < nparts; p += 8) s += part[(size_t)p * PART + idx];
    }
    sub[k][cl] = s;
    __syncthreads();
    if (k == 0 && idx < PART) {
        float t = sub[0][cl];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += sub[j][cl];
        if (idx < F * D) dw1[idx] = t;
        else if (idx < 2 * F * D) {
            const int j = idx - F * D;
            dw2[(size_t)(j % D) * F + j / D] = t;
        } else if (idx < 2 * F * D + F) db1[idx - 2 * F * D] = t;
        else db2[idx - 2 * F * D - F] = t;
    }
}

constexpr int FF_D = 64, FF_F = 256, FF_NW = 8, FF_MAX_TILES = 2048;

int ff_bwd_grid(long long M) {       // one work-group per CU and tile up to 256 tiles, then ~2..8 tiles per work-group
    const long long tiles = (M + 31) / 32;
    const int cap = kv_config().ff_grid > 0 ? kv_config().ff_grid : 256;
    return (int)(tiles < cap ? tiles : cap);
}
constexpr size_t FF_BWD_LDS = sizeof(float) * (2 * FF_F * 33 + 2 * 32 * (FF_D + 32) + FF_NW * FF_D * 33);

int ff_check(const char* who, long long M, int D, int F) {
    if (D != FF_D || F != FF_F) return kv_fail(KANVIT_EINVAL, "%s: only D=%d, F=%d is instantiated (got D=%d, F=%d)", who, FF_D, FF_F, D, F);
    if (M < 0 || M > 32LL * FF_MAX_TILES) return kv_fail(KANVIT_EINVAL, "%s: M=%lld outside [0, %d] (small-geometry path; larger M: stock GEMMs)", who, (long long)M, 32 * FF_MAX_TILES);
    return 0;
}

}  // namespace

extern "C" {

int kanvit_ff_small_supported(int D, int F) { return D == FF_D && F == FF_F; }
int64_t kanvit_ff_small_max_rows(void) { return 32LL * FF_MAX_TILES; }

int kanvit_ff_small_fwd(int64_t M, int D, int F, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                        float* y, void* stream) {
    if (int rc = ff_check("kanvit_ff_small_fwd", M, D, F)) return rc;
    if (M == 0) return 0;
    if (!x || !w1 || !b1 || !w2 || !b2 || !y) return kv_fail(KANVIT_EINVAL, "kanvit_ff_small_fwd: null argument");
    if (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)w2 | (uintptr_t)b2 | (uintptr_t)y) & 15)
        return kv_fail(KANVIT_EINVAL, "kanvit_ff_small_fwd: pointers must be 16-byte aligned");
    FfArgs a{};
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.y = y; a.M = M; a.tiles = (int)((M + 31) / 32);
    KV_ALLOW_LDS(160 * 1024, (ff_small_fwd_kernel<FF_D, FF_F, FF_NW>));
    hipLaunchKernelGGL((ff_small_fwd_kernel<FF_D, FF_F, FF_NW>), dim3(a.tiles < 1024 ? a.tiles : 1024), dim3(64 * FF_NW),
                       sizeof(float) * FF_NW * FF_D * 33, (hipStream_t)stream, a);
    KV_LAUNCH_CHECK("ff_small_fwd_kernel");
    return 0;
}

size_t kanvit_ff_small_bwd_workspace(int64_t M, int D, int F) {
    if (M <= 0 || D != FF_D || F != FF_F) return 0;
    return sizeof(float) * (size_t)ff_bwd_grid(M) * (2 * FF_F * FF_D + FF_F + FF_D);
}

int kanvit_ff_small_bwd(int64_t M, int D, int F, const float* x, const float* w1, const float* b1, const float* w2, const float* dy,
                        float* dx, float* dw1, float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = ff_check("kanvit_ff_small_bwd", M, D, F)) return rc;
    if (!dw1 || !db1 || !dw2 || !db2) return kv_fail(KANVIT_EINVAL, "kanvit_ff_small_bwd: null gradient output");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        KV_HIP_CHECK(hipMemsetAsync(dw1, 0, sizeof(float) * F * D, st));
        KV_HIP_CHECK(hipMemsetAsync(dw2, 0, sizeof(float) * F * D, st));
        KV_HIP_CHECK(hipMemsetAsync(db1, 0, sizeof(float) * F, st));
        KV_HIP_CHECK(hipMemsetAsync(db2, 0, sizeof(float) * D, st));
        return 0;
    }
    if (!x || !w1 || !b1 || !w2 || !dy || !dx) return kv_fail(KANVIT_EINVAL, "kanvit_ff_small_bwd: null argument");
    if (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)w2 | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)workspace) & 15)
        return kv_fail(KANVIT_EINVAL, "kanvit_ff_small_bwd: pointers must be 16-byte aligned");
    const size_t need = kanvit_ff_small_bwd_workspace(M, D, F);
    if (!workspace || workspace_bytes < need)
        return kv_fail(KANVIT_ENOMEM, "kanvit_ff_small_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
    FfArgs a{};
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.dy = dy; a.dx = dx; a.part = (float*)workspace; a.M = M; a.tiles = (int)((M + 31) / 32);
    const int grid = ff_bwd_grid(M);
    KV_ALLOW_LDS(160 * 1024, (ff_small_bwd_kernel<FF_D, FF_F, FF_NW>));
    hipLaunchKernelGGL((ff_small_bwd_kernel<FF_D, FF_F, FF_NW>), dim3(grid), dim3(64 * FF_NW), FF_BWD_LDS, st, a);
    KV_LAUNCH_CHECK("ff_small_bwd_kernel");
    constexpr int PART = 2 * FF_F * FF_D + FF_F + FF_D;
    hipLaunchKernelGGL((ff_small_reduce_kernel<FF_D, FF_F>), dim3((PART + 31) / 32), dim3(256), 0, st, (const float*)workspace, grid, dw1, dw2, db1, db2);
    KV_LAUNCH_CHECK("ff_small_reduce_kernel");
    return 0;
}

}  // extern "C"
