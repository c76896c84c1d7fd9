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

#include <initializer_list>

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
    float* part;      // [work-groups][2*F*D + F + 3*D]: dW1[F][D] | dW2^T[F][D] | db1[F] | db2[D] | dgamma[D] | dbeta[D]
    long long M;
    int tiles;
    // LN variants: the block's second LayerNorm fused in front (model.py:36: x = x + FF(LN2(x)) with x = x_in + delta)
    const float* delta;   // forward: added to x first (may be null)
    const float* gamma;
    const float* beta;
    float* s;             // forward out / backward in: s = x + delta [M][D]
    float* mean;          // forward out / backward in: row statistics of s
    float* rstd;
    const float* ds_in;   // backward: gradient arriving on s from later uses (may be null)
    float eps;
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

template <int D, int F, int NW, bool LN>
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
        if constexpr (LN) {       // s = x + delta; h2 = LayerNorm(s) -- every wave forms it for its own operand, wave 0 stores s and the statistics
            if (a.delta) {
                float dr[DH];
                ff_load_row<DH>(dr, a.delta + mr * D + hf * DH);
#pragma unroll
                for (int i = 0; i < DH; ++i) xr[i] += dr[i];
            }
            const bool owner = wave == 0 && m0 + l31 < a.M;
            if (owner) {
#pragma unroll
                for (int i = 0; i < DH / 4; ++i) {
                    const f32x4 v = {xr[4 * i], xr[4 * i + 1], xr[4 * i + 2], xr[4 * i + 3]};
                    *reinterpret_cast<f32x4*>(a.s + mr * D + hf * DH + 4 * i) = v;
                }
            }
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < DH; ++i) sum += xr[i];
            sum += __shfl_xor(sum, 32);
            const float mu = sum * (1.0f / D);
            float sq = 0.0f;
#pragma unroll
            for (int i = 0; i < DH; ++i) sq += (xr[i] - mu) * (xr[i] - mu);       // two-pass: no E[x^2] - E[x]^2 cancellation
            sq += __shfl_xor(sq, 32);
            const float rs = rsqrtf(sq * (1.0f / D) + a.eps);
            if (owner && hf == 0) {
                a.mean[mr] = mu;
                a.rstd[mr] = rs;
            }
            float gr[DH], br[DH];
            ff_load_row<DH>(gr, a.gamma + hf * DH);
            ff_load_row<DH>(br, a.beta + hf * DH);
#pragma unroll
            for (int i = 0; i < DH; ++i) xr[i] = (xr[i] - mu) * rs * gr[i] + br[i];
        }
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

template <int D, int F, int NW, bool LN>
__global__ __launch_bounds__(64 * NW) void ff_small_bwd_kernel(const FfArgs a) {
    static_assert(!LN || 32 * (D / 4) == 64 * NW, "the LayerNorm backward maps one float4 of the dx tile to each thread");
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
    f32x4 gdg = {0.0f, 0.0f, 0.0f, 0.0f}, gdb = gdg;      // LN: column sums for dgamma / dbeta (thread = 4 columns, fixed over tiles)
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
        // the weight fragments of the first two products do not depend on the staged tiles: request them before the staging
        // barrier so their L2 latency runs under it (they would otherwise queue up behind the barrier, one phase each)
        float w1f[DH], w2f[DH];
        ff_load_row<DH>(w1f, pw1 + (size_t)(nb + l31) * D + hf * DH);
        {
            const int voff = hf * DH * F + l31;
#pragma unroll
            for (int s = 0; s < DH; ++s) w2f[s] = (pw2 + (size_t)s * F + nb)[voff];
        }
        // stage the x and dy tiles (rows past M: zero); LN: x is rebuilt from s and its row statistics
        for (int idx = tid; idx < 32 * (D / 4); idx += NT) {
            const int m = idx / (D / 4), c = idx % (D / 4);
            f32x4 xv = {0.0f, 0.0f, 0.0f, 0.0f}, dv = xv;
            if (m0 + m < a.M) {
                dv = *reinterpret_cast<const f32x4*>(a.dy + (m0 + m) * D + 4 * c);
                if constexpr (LN) {
                    const f32x4 sv = *reinterpret_cast<const f32x4*>(a.s + (m0 + m) * D + 4 * c);
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gamma + 4 * c), bv = *reinterpret_cast<const f32x4*>(a.beta + 4 * c);
                    const float mu = a.mean[m0 + m], rs = a.rstd[m0 + m];
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = (sv[e] - mu) * rs * gv[e] + bv[e];
                } else {
                    xv = *reinterpret_cast<const f32x4*>(a.x + (m0 + m) * D + 4 * c);
                }
            }
            *reinterpret_cast<f32x4*>(xs + m * DS + 4 * c) = xv;
            *reinterpret_cast<f32x4*>(dys + m * DS + 4 * c) = dv;
        }
        __syncthreads();
        float xr[DH], dyr[DH];      // this lane's half rows (operands of the row-indexed products) come back out of the staged tiles
        ff_load_row<DH>(xr, xs + l31 * DS + hf * DH);
        ff_load_row<DH>(dyr, dys + l31 * DS + hf * DH);
        f32x16 dxacc[NDT];
#pragma unroll
        for (int kt = 0; kt < NDT; ++kt) dxacc[kt] = ff_zero16();
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            const int n0 = nb + t * 32;
            f32x16 h, dh = ff_zero16();
            // dh^T[n][m] = sum_o W2[o][n] dy[m][o]      (half h takes o = h*D/2 + s); addresses are a wave-uniform base (scalar
            // registers) + ONE per-lane 32-bit offset: a 64-bit address per load (the offsets do not fit the 12-bit immediate)
            // costs two registers each
            if (t == 0) {
                h = ff_zero16();
#pragma unroll
                for (int s = 0; s < DH; ++s) h = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[s], xr[s], h, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < DH; ++s) dh = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[s], dyr[s], dh, 0, 0, 0);
            } else {
                h = ff_hidden_tile<D>(pw1, n0, l31, hf, xr);
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
            const bool rowok = m0 + m < a.M;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (rowok || LN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = red[(4 * c + e) * 33 + m];
#pragma unroll
                    for (int w = 1; w < NW; ++w) s += red[(w * D + 4 * c + e) * 33 + m];
                    v[e] = s;
                }
            }
            if constexpr (LN) {
                // v = d loss / d LN2(s) for 4 columns of row m; the 16 lanes of a row meet by xor-shuffles (every thread is here: one
                // float4 each).  ds = rstd * (v*g - mean(v*g) - xhat * mean(v*g*xhat)) + the gradient arriving on s itself.
                f32x4 sv = {0.0f, 0.0f, 0.0f, 0.0f}, xh, gy;
                float mu = 0.0f, rs = 0.0f;
                if (rowok) {
                    sv = *reinterpret_cast<const f32x4*>(a.s + (m0 + m) * D + 4 * c);
                    mu = a.mean[m0 + m];
                    rs = a.rstd[m0 + m];
                }
                const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gamma + 4 * c);
                float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[e] = (sv[e] - mu) * rs;
                    gy[e] = v[e] * gv[e];
                    t1 += gy[e];
                    t2 += gy[e] * xh[e];
                    gdg[e] += v[e] * xh[e];
                    gdb[e] += v[e];
                }
#pragma unroll
                for (int o = 1; o < D / 4; o <<= 1) {
                    t1 += __shfl_xor(t1, o);
                    t2 += __shfl_xor(t2, o);
                }
                if (rowok) {
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = rs * (gy[e] - t1 * (1.0f / D) - xh[e] * t2 * (1.0f / D));
                    if (a.ds_in) o4 += *reinterpret_cast<const f32x4*>(a.ds_in + (m0 + m) * D + 4 * c);
                    *reinterpret_cast<f32x4*>(a.dx + (m0 + m) * D + 4 * c) = o4;
                }
            } else {
                if (rowok) *reinterpret_cast<f32x4*>(a.dx + (m0 + m) * D + 4 * c) = v;
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
    float* part = a.part + (size_t)blockIdx.x * (2 * F * D + F + 3 * D);
    if constexpr (LN) {       // dgamma / dbeta: thread (m, c) -> the 4 rows of a wave by shuffles, the NW waves through LDS (red is free now)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gdg[e] += __shfl_xor(gdg[e], 16);
            gdg[e] += __shfl_xor(gdg[e], 32);
            gdb[e] += __shfl_xor(gdb[e], 16);
            gdb[e] += __shfl_xor(gdb[e], 32);
        }
        if (lane < 16) {
            *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * D + 4 * lane) = gdg;
            *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * D + 4 * lane) = gdb;
        }
        __syncthreads();
        if (tid < 2 * D) {
            float t = red[tid];
#pragma unroll
            for (int w = 1; w < NW; ++w) t += red[w * 2 * D + tid];
            part[2 * F * D + F + D + tid] = t;
        }
    }
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
                                                              float* __restrict__ dw2, float* __restrict__ db1, float* __restrict__ db2,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int STRIDE = 2 * F * D + F + 3 * D;
    const int PART = dgamma ? STRIDE : STRIDE - 2 * D;
    __shared__ float sub[8][33];
    const int cl = threadIdx.x & 31, k = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + cl;
    float s = 0.0f;
    if (idx < PART) {
        int p = k;
        for (; p + 24 < nparts; p += 32) {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = part[(size_t)(p + 8 * j) * STRIDE + idx];
#pragma unroll
            for (int j = 0; j < 4; ++j) s += t[j];
        }
        for (; p < nparts; p += 8) s += part[(size_t)p * STRIDE + idx];
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
        else if (idx < 2 * F * D + F + D) db2[idx - 2 * F * D - F] = t;
        else if (idx < 2 * F * D + F + 2 * D) dgamma[idx - 2 * F * D - F - D] = t;
        else dbeta[idx - 2 * F * D - F - 2 * D] = t;
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

constexpr int FF_PART = 2 * FF_F * FF_D + FF_F + 3 * FF_D;

template <bool LN>
int ff_launch_fwd(FfArgs& a, hipStream_t st) {
    a.tiles = (int)((a.M + 31) / 32);
    KV_ALLOW_LDS(160 * 1024, (ff_small_fwd_kernel<FF_D, FF_F, FF_NW, LN>));
    hipLaunchKernelGGL((ff_small_fwd_kernel<FF_D, FF_F, FF_NW, LN>), dim3(a.tiles < 1024 ? a.tiles : 1024), dim3(64 * FF_NW),
                       sizeof(float) * FF_NW * FF_D * 33, st, a);
    KV_LAUNCH_CHECK("ff_small_fwd_kernel");
    return 0;
}

template <bool LN>
int ff_launch_bwd(FfArgs& a, float* dw1, float* db1, float* dw2, float* db2, float* dgamma, float* dbeta, hipStream_t st) {
    a.tiles = (int)((a.M + 31) / 32);
    const int grid = ff_bwd_grid(a.M);
    KV_ALLOW_LDS(160 * 1024, (ff_small_bwd_kernel<FF_D, FF_F, FF_NW, LN>));
    hipLaunchKernelGGL((ff_small_bwd_kernel<FF_D, FF_F, FF_NW, LN>), dim3(grid), dim3(64 * FF_NW), FF_BWD_LDS, st, a);
    KV_LAUNCH_CHECK("ff_small_bwd_kernel");
    const int outs = LN ? FF_PART : FF_PART - 2 * FF_D;
    hipLaunchKernelGGL((ff_small_reduce_kernel<FF_D, FF_F>), dim3((outs + 31) / 32), dim3(256), 0, st, (const float*)a.part, grid, dw1, dw2, db1, db2,
                       LN ? dgamma : (float*)nullptr, LN ? dbeta : (float*)nullptr);
    KV_LAUNCH_CHECK("ff_small_reduce_kernel");
    return 0;
}

bool ff_misaligned(std::initializer_list<const void*> ps, unsigned mask = 15) {
    for (const void* q : ps)
        if (q && ((uintptr_t)q & mask)) return true;
    return false;
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
    if (ff_misaligned({x, w1, b1, w2, b2, y})) return kv_fail(KANVIT_EINVAL, "kanvit_ff_small_fwd: pointers must be 16-byte aligned");
    FfArgs a{};
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.y = y; a.M = M;
    return ff_launch_fwd<false>(a, (hipStream_t)stream);
}

size_t kanvit_ff_small_bwd_workspace(int64_t M, int D, int F) {
    if (M <= 0 || D != FF_D || F != FF_F) return 0;
    return sizeof(float) * (size_t)ff_bwd_grid(M) * FF_PART;
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
    if (ff_misaligned({x, w1, b1, w2, dy, dx, workspace})) return kv_fail(KANVIT_EINVAL, "kanvit_ff_small_bwd: pointers must be 16-byte aligned");
    const size_t need = kanvit_ff_small_bwd_workspace(M, D, F);
    if (!workspace || workspace_bytes < need)
        return kv_fail(KANVIT_ENOMEM, "kanvit_ff_small_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
    FfArgs a{};
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.dy = dy; a.dx = dx; a.part = (float*)workspace; a.M = M;
    return ff_launch_bwd<false>(a, dw1, db1, dw2, db2, nullptr, nullptr, st);
}

/* s = x + delta;  y = FF(LayerNorm(s)): the second half of a TransformerBlock up to (not including) the residual add of y */
int kanvit_lnff_small_fwd(int64_t M, int D, int F, float eps, const float* x, const float* delta, const float* gamma, const float* beta,
                          const float* w1, const float* b1, const float* w2, const float* b2, float* s, float* mean, float* rstd, float* y,
                          void* stream) {
    if (int rc = ff_check("kanvit_lnff_small_fwd", M, D, F)) return rc;
    if (M == 0) return 0;
    if (!x || !gamma || !beta || !w1 || !b1 || !w2 || !b2 || !s || !mean || !rstd || !y) return kv_fail(KANVIT_EINVAL, "kanvit_lnff_small_fwd: null argument");
    if (ff_misaligned({x, delta, gamma, beta, w1, b1, w2, b2, s, y})) return kv_fail(KANVIT_EINVAL, "kanvit_lnff_small_fwd: pointers must be 16-byte aligned");
    FfArgs a{};
    a.x = x; a.delta = delta; a.gamma = gamma; a.beta = beta; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.s = s; a.mean = mean; a.rstd = rstd; a.y = y; a.M = M; a.eps = eps;
    return ff_launch_fwd<true>(a, (hipStream_t)stream);
}

/* ds = ds_in + LayerNorm-backward(FF-backward(dy)) (the gradient of x AND of delta), plus every parameter gradient */
int kanvit_lnff_small_bwd(int64_t M, int D, int F, const float* s, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          const float* w1, const float* b1, const float* w2, const float* dy, const float* ds_in, float* ds, float* dgamma,
                          float* dbeta, float* dw1, float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = ff_check("kanvit_lnff_small_bwd", M, D, F)) return rc;
    if (!dw1 || !db1 || !dw2 || !db2 || !dgamma || !dbeta) return kv_fail(KANVIT_EINVAL, "kanvit_lnff_small_bwd: null gradient output");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        KV_HIP_CHECK(hipMemsetAsync(dw1, 0, sizeof(float) * F * D, st));
        KV_HIP_CHECK(hipMemsetAsync(dw2, 0, sizeof(float) * F * D, st));
        KV_HIP_CHECK(hipMemsetAsync(db1, 0, sizeof(float) * F, st));
        KV_HIP_CHECK(hipMemsetAsync(db2, 0, sizeof(float) * D, st));
        KV_HIP_CHECK(hipMemsetAsync(dgamma, 0, sizeof(float) * D, st));
        KV_HIP_CHECK(hipMemsetAsync(dbeta, 0, sizeof(float) * D, st));
        return 0;
    }
    if (!s || !mean || !rstd || !gamma || !beta || !w1 || !b1 || !w2 || !dy || !ds) return kv_fail(KANVIT_EINVAL, "kanvit_lnff_small_bwd: null argument");
    if (ff_misaligned({s, gamma, beta, w1, b1, w2, dy, ds_in, ds, workspace})) return kv_fail(KANVIT_EINVAL, "kanvit_lnff_small_bwd: pointers must be 16-byte aligned");
    const size_t need = kanvit_ff_small_bwd_workspace(M, D, F);
    if (!workspace || workspace_bytes < need)
        return kv_fail(KANVIT_ENOMEM, "kanvit_lnff_small_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
    FfArgs a{};
    a.s = const_cast<float*>(s); a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.gamma = gamma; a.beta = beta;
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.dy = dy; a.ds_in = ds_in; a.dx = ds; a.part = (float*)workspace; a.M = M;
    return ff_launch_bwd<true>(a, dw1, db1, dw2, db2, dgamma, dbeta, st);
}

}  // extern "C"
