// Fused residual-add + LayerNorm for the TransformerBlock assembly (reference model.py:31-37:
//   x = x + MSA(LN1(x));  x = x + FF(LN2(x))  -- an add and an nn.LayerNorm, each a separate pass over [B*N, d]).
//
//   forward :  s = x + delta (delta optional);  y = (s - mean(s)) * rstd(s) * gamma + beta;   writes s, y, mean, rstd
//   backward:  dx = LayerNorm backward of dy w.r.t. s  (+ dres, the gradient arriving on the residual stream);
//              dgamma = sum_rows dy * xhat,  dbeta = sum_rows dy
//
// HBM-bound row kernels: one wave per row, D <= 1024 columns held in registers (float4 per lane, lane + 64 v), two-pass
// variance in registers (no E[x^2] - E[x]^2 cancellation), wave reductions by DPP/shuffle.  The backward keeps per-lane
// partial sums of dgamma / dbeta over all rows a wave processes; the waves of a work-group meet in LDS and write ONE
// partial per work-group; a second kernel sums the partials in a fixed order (deterministic, no atomics).  Stock torch runs LN backward as three kernels at ~3x the
// algorithmic traffic (0.124 ms per ViT-B LayerNorm at B = 128) plus a separate add kernel per residual.
#include "../../include/kanvit.h"
#include "kanvit_common.h"

#include <type_traits>

namespace {

constexpr int LN_MAXV = 4;           // float4 per lane: D <= 1024
constexpr int LN_WAVES = 8;          // waves per work-group

__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct LnArgs {
    const float* x;
    const float* delta;
    const float* gamma;
    const float* beta;
    float* xsum;
    float* y;
    float* mean;
    float* rstd;
    const float* dy;
    const float* dres;
    float* dx;
    float* part;      // [work-groups][2][D]
    unsigned short* dx_bf16;      // optional second copy of dx, rounded to bf16 (the gradient of a bf16 `delta`)
    long long M;
    int D;
    float eps;
};

// bf16 tensors at the boundary (torch.autocast: the feed-forward's output / input / input gradient are bf16): 4 values = 8 bytes
typedef unsigned short ln_u16x4 __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ f32x4 ln_ld4(const float* base, long long idx) {      // idx in elements; `base` is a bf16 pointer when BF
    if constexpr (BF) {
        const ln_u16x4 u = *reinterpret_cast<const ln_u16x4*>(reinterpret_cast<const unsigned short*>(base) + idx);
        return f32x4{__builtin_bit_cast(float, (unsigned)u[0] << 16), __builtin_bit_cast(float, (unsigned)u[1] << 16),
                     __builtin_bit_cast(float, (unsigned)u[2] << 16), __builtin_bit_cast(float, (unsigned)u[3] << 16)};
    } else {
        return *reinterpret_cast<const f32x4*>(base + idx);
    }
}
__device__ __forceinline__ ln_u16x4 ln_to_bf16(const f32x4& v) {               // round to nearest even, as torch's .to(bfloat16)
    typedef __bf16 ln_bf16x4 __attribute__((ext_vector_type(4)));
    const ln_bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(ln_u16x4, b);
}

template <int NV, bool DB = false, bool YB = false>      // DB: delta is bf16; YB: y is written as bf16
__global__ __launch_bounds__(64 * LN_WAVES) void addln_fwd_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wid = (long long)blockIdx.x * LN_WAVES + wave, nw = (long long)gridDim.x * LN_WAVES;
    const int D = a.D;
    const float invd = 1.0f / (float)D;
    f32x4 g[NV], bt[NV];
    bool ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = 4 * (lane + 64 * v);
        ok[v] = c < D;
        g[v] = ok[v] ? *reinterpret_cast<const f32x4*>(a.gamma + c) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        bt[v] = ok[v] ? *reinterpret_cast<const f32x4*>(a.beta + c) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    for (long long r = wid; r < a.M; r += nw) {
        f32x4 s[NV];
        float sum = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * (lane + 64 * v);
            f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
            if (ok[v]) {
                t = *reinterpret_cast<const f32x4*>(a.x + r * D + c);
                if (a.delta) t += ln_ld4<DB>(a.delta, r * D + c);
                if (a.xsum) *reinterpret_cast<f32x4*>(a.xsum + r * D + c) = t;
            }
            s[v] = t;
            sum += (t[0] + t[1]) + (t[2] + t[3]);
        }
        const float mu = ln_wave_sum(sum) * invd;
        float sq = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (ok[v]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = s[v][e] - mu;
                    sq += d * d;
                }
            }
        }
        const float rs = rsqrtf(ln_wave_sum(sq) * invd + a.eps);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (ok[v]) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (s[v][e] - mu) * rs * g[v][e] + bt[v][e];
                if constexpr (YB) *reinterpret_cast<ln_u16x4*>(reinterpret_cast<unsigned short*>(a.y) + r * D + 4 * (lane + 64 * v)) = ln_to_bf16(o);
                else *reinterpret_cast<f32x4*>(a.y + r * D + 4 * (lane + 64 * v)) = o;
            }
        }
        if (lane == 0) {
            a.mean[r] = mu;
            a.rstd[r] = rs;
        }
    }
}

template <int NV, bool GB = false>      // GB: dy is bf16
__global__ __launch_bounds__(64 * LN_WAVES) void addln_bwd_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wid = (long long)blockIdx.x * LN_WAVES + wave, nw = (long long)gridDim.x * LN_WAVES;
    const int D = a.D;
    const float invd = 1.0f / (float)D;
    f32x4 g[NV], dg[NV], db[NV];
    bool ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = 4 * (lane + 64 * v);
        ok[v] = c < D;
        g[v] = ok[v] ? *reinterpret_cast<const f32x4*>(a.gamma + c) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        dg[v] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        db[v] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    for (long long r = wid; r < a.M; r += nw) {
        const float mu = a.mean[r], rs = a.rstd[r];
        f32x4 xh[NV], gy[NV];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * (lane + 64 * v);
            f32x4 xs = {0.0f, 0.0f, 0.0f, 0.0f}, dyv = xs;
            if (ok[v]) {
                xs = *reinterpret_cast<const f32x4*>(a.x + r * D + c);
                dyv = ln_ld4<GB>(a.dy, r * D + c);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = ok[v] ? (xs[e] - mu) * rs : 0.0f;
                xh[v][e] = h;
                gy[v][e] = dyv[e] * g[v][e];
                s1 += gy[v][e];
                s2 += gy[v][e] * h;
                dg[v][e] += dyv[e] * h;
                db[v][e] += dyv[e];
            }
        }
        const float m1 = ln_wave_sum(s1) * invd, m2 = ln_wave_sum(s2) * invd;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (ok[v]) {
                const int c = 4 * (lane + 64 * v);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (gy[v][e] - m1 - xh[v][e] * m2);
                if (a.dres) o += *reinterpret_cast<const f32x4*>(a.dres + r * D + c);
                *reinterpret_cast<f32x4*>(a.dx + r * D + c) = o;
                if (a.dx_bf16) *reinterpret_cast<ln_u16x4*>(a.dx_bf16 + r * D + c) = ln_to_bf16(o);
            }
        }
    }
    // one partial per work-group: the waves' sums meet in LDS and are added in wave order
    extern __shared__ __attribute__((aligned(16))) float lsm[];      // [LN_WAVES][2][D]
    float* p = lsm + wave * 2 * D;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (ok[v]) {
            const int c = 4 * (lane + 64 * v);
            *reinterpret_cast<f32x4*>(p + c) = dg[v];
            *reinterpret_cast<f32x4*>(p + D + c) = db[v];
        }
    }
    __syncthreads();
    float* out = a.part + (long long)blockIdx.x * 2 * D;
    for (int c = threadIdx.x; c < 2 * D; c += 64 * LN_WAVES) {
        float sacc = lsm[c];
#pragma unroll
        for (int w = 1; w < LN_WAVES; ++w) sacc += lsm[w * 2 * D + c];
        out[c] = sacc;
    }
}

// dgamma[c] = sum_w part[w][0][c], dbeta[c] = sum_w part[w][1][c], fixed order: thread (column, k) adds partials k, k+8, ...;
// the 8 sub-sums of a column are then added in order through LDS
__global__ __launch_bounds__(256) void addln_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int D, int nparts) {
    __shared__ float sub[8][33];
    const int cl = threadIdx.x & 31, k = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.0f;
    if (c < 2 * D) {
        int w = k;
        for (; w + 56 < nparts; w += 64) {            // 8 independent loads in flight per thread, added in order
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = part[(long long)(w + 8 * j) * 2 * D + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += t[j];
        }
        for (; w < nparts; w += 8) s += part[(long long)w * 2 * D + c];
    }
    sub[k][cl] = s;
    __syncthreads();
    if (k == 0 && c < 2 * D) {
        float t = sub[0][cl];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += sub[j][cl];
        if (c < D) dgamma[c] = t;
        else dbeta[c - D] = t;
    }
}

// ---- LayerNorm backward of the FUSED FastKAN LayerNorm (KANVIT_FLAG_FUSED_LN, models/fastkan.py:68) --------------------------
// The KAN kernels formed u[m][g][i] = (x[m][gx][i] - mean) * rstd * gamma[g][i] + beta[g][i] on the fly (g = s * xmod + gx, the
// ns = groups / xmod projections share the statistics of their x slice) and the input-gradient kernel returned du.  This kernel
// finishes the chain rule in ONE pass over du (instead of torch's einsum / sum / mean sequence: 2.8 ms per ViT-S block):
//   dxhat = sum_s du[s] * gamma[s];   dx += rstd * (dxhat - mean_i(dxhat) - xhat * mean_i(dxhat * xhat))     (in place)
//   dgamma[g] = sum_m du * xhat,  dbeta[g] = sum_m du
// A (row, x slice) is owned by LPR = 2^k >= I/4 lanes (float4 each, NCH chunks for I > 256), so a wave covers 64 / LPR rows at
// once and the two means are xor-shuffles inside the lane group; per-lane column sums over all rows the wave walks, then lane
// groups -> waves (LDS) -> one partial per work-group, summed in fixed order by addln_reduce_kernel.  HBM-bound: du, x read
// once, dx read + written once.
struct LnKanArgs {
    const float* x;
    const float* stats;
    const float* bp;
    const float* du;
    float* dx;
    float* part;          // [gridDim.x][2][groups][I]
    long long M, ldx, ldu, bp_stride;
    int I, G, xmod, groups, lpr_log2;
};

template <int NCH, int NS>
__global__ __launch_bounds__(256) void kan_ln_bwd_kernel(const LnKanArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int LPR = 1 << a.lpr_log2, RPW = 64 >> a.lpr_log2;
    const int sub = lane & (LPR - 1), rg = lane >> a.lpr_log2;
    const int gx = blockIdx.y, I = a.I;
    const float invI = 1.0f / (float)I;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 gam[NS][NCH], dg[NS][NCH], db[NS][NCH];
    bool ok[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 4 * (sub + LPR * c);
        ok[c] = col < I;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            gam[s][c] = ok[c] ? *reinterpret_cast<const f32x4*>(a.bp + (long long)(s * a.xmod + gx) * a.bp_stride + a.G + col) : zero;
            dg[s][c] = zero;
            db[s][c] = zero;
        }
    }
    const long long step = (long long)gridDim.x * 4 * RPW;
    const long long iters = (a.M + step - 1) / step;               // same trip count for every lane: the shuffles need them all
    long long r = ((long long)blockIdx.x * 4 + wave) * RPW + rg;
    for (long long it = 0; it < iters; ++it, r += step) {
        const bool live = r < a.M;
        const long long rc = live ? r : a.M - 1;
        const float2 st = *reinterpret_cast<const float2*>(a.stats + (rc * a.xmod + gx) * 2);
        const float mu = st.x, rs = st.y;
        f32x4 xh[NCH], dxh[NCH];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = 4 * (sub + LPR * c);
            const bool act = ok[c] && live;
            f32x4 xs = zero;
            if (act) xs = *reinterpret_cast<const f32x4*>(a.x + rc * a.ldx + (long long)gx * I + col);
            dxh[c] = zero;
#pragma unroll
            for (int e = 0; e < 4; ++e) xh[c][e] = act ? (xs[e] - mu) * rs : 0.0f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                f32x4 duv = zero;
                if (act) duv = *reinterpret_cast<const f32x4*>(a.du + rc * a.ldu + (long long)(s * a.xmod + gx) * I + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dxh[c][e] += duv[e] * gam[s][c][e];
                    dg[s][c][e] += duv[e] * xh[c][e];
                    db[s][c][e] += duv[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1 += dxh[c][e];
                s2 += dxh[c][e] * xh[c][e];
            }
        }
        for (int o = LPR >> 1; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        if (a.dx && live) {
            const float m1 = s1 * invI, m2 = s2 * invI;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (ok[c]) {
                    float* dp = a.dx + r * a.ldx + (long long)gx * I + 4 * (sub + LPR * c);
                    f32x4 o = *reinterpret_cast<const f32x4*>(dp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += rs * (dxh[c][e] - m1 - xh[c][e] * m2);
                    *reinterpret_cast<f32x4*>(dp) = o;
                }
            }
        }
    }
    // column sums: lane groups of the wave (xor over the row-group bits), then the 4 waves through LDS, in fixed order
    for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dg[s][c][e] += __shfl_xor(dg[s][c][e], o);
                    db[s][c][e] += __shfl_xor(db[s][c][e], o);
                }
    }
    extern __shared__ __attribute__((aligned(16))) float lsm[];      // [4 waves][2][NS][I]
    if (rg == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ok[c]) {
                    const int col = 4 * (sub + LPR * c);
                    *reinterpret_cast<f32x4*>(lsm + ((wave * 2 + 0) * NS + s) * I + col) = dg[s][c];
                    *reinterpret_cast<f32x4*>(lsm + ((wave * 2 + 1) * NS + s) * I + col) = db[s][c];
                }
    }
    __syncthreads();
    const int per = 2 * NS * I;
    for (int idx = threadIdx.x; idx < per; idx += 256) {
        const float t = ((lsm[idx] + lsm[per + idx]) + lsm[2 * per + idx]) + lsm[3 * per + idx];
        const int k = idx / (NS * I), s = (idx / I) % NS, i = idx % I;
        a.part[(((long long)blockIdx.x * 2 + k) * a.groups + s * a.xmod + gx) * I + i] = t;
    }
}

int ln_kan_check(const kanvit_layer_desc* d, const char* who) {
    if (!d) return kv_fail(KANVIT_EINVAL, "%s: null descriptor", who);
    const int ns = d->x_group_mod > 0 ? d->groups / d->x_group_mod : 0;
    if (d->family != KANVIT_RBF || !(d->flags & KANVIT_FLAG_FUSED_LN)) return kv_fail(KANVIT_EINVAL, "%s: descriptor is not a KANVIT_FLAG_FUSED_LN FastKAN launch", who);
    if (d->groups < 1 || d->x_group_mod < 1 || d->groups % d->x_group_mod || (ns != 1 && ns != 3))
        return kv_fail(KANVIT_EINVAL, "%s: groups / x_group_mod must be 1 or 3", who);
    if (d->I < 32 || d->I % 32 || d->I > (ns == 3 ? 512 : 1024)) return kv_fail(KANVIT_EINVAL, "%s: I=%d unsupported (multiple of 32, <= 1024; <= 512 with 3 projections)", who, d->I);
    if ((d->G & 3) || (d->bparam_stride & 3) || d->bparam_stride < d->G + 2 * (int64_t)d->I || (d->ldx & 3) || (d->ldu & 3))
        return kv_fail(KANVIT_EINVAL, "%s: G, bparam_stride, ldx, ldu must be multiples of 4 and bparam_stride >= G + 2*I", who);
    if (d->ldx < (int64_t)d->x_group_mod * d->I || d->ldu < (int64_t)d->groups * d->I) return kv_fail(KANVIT_EINVAL, "%s: ldx / ldu too small", who);
    return 0;
}

int ln_kan_lpr_log2(int I) {
    int k = 3;
    while ((1 << k) < I / 4 && k < 6) ++k;
    return k;
}

int ln_kan_grid(const kanvit_layer_desc* d) {
    const int rpw = 64 >> ln_kan_lpr_log2(d->I);
    long long nb = (d->M + 4 * rpw - 1) / (4 * rpw);
    long long cap = 2048 / d->x_group_mod;
    if (cap < 1) cap = 1;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

int ln_grid(long long M) {
    long long wgs = (M + LN_WAVES - 1) / LN_WAVES;
    if (wgs > 2 * 256) wgs = 2 * 256;      // 2 work-groups of 8 waves per CU; each wave then walks M / 4096 rows
    if (wgs < 1) wgs = 1;
    return (int)wgs;
}

int ln_check(const char* who, long long M, int D) {
    if (M < 0 || D < 4 || (D & 3) || D > 256 * LN_MAXV)
        return kv_fail(KANVIT_EINVAL, "%s: D=%d must be a multiple of 4 in [4, %d] and M >= 0", who, D, 256 * LN_MAXV);
    return 0;
}

template <typename F>
int ln_dispatch(int D, F&& f) {
    const int nv = (D + 255) / 256;
    switch (nv) {
        case 1: return f(std::integral_constant<int, 1>{});
        case 2: return f(std::integral_constant<int, 2>{});
        case 3: return f(std::integral_constant<int, 3>{});
        default: return f(std::integral_constant<int, 4>{});
    }
}

}  // namespace

extern "C" {

int kanvit_addln_fwd_ex(int64_t M, int D, float eps, const float* x, const void* delta, int delta_bf16, const float* gamma, const float* beta,
                        float* xsum, void* y, int y_bf16, float* mean, float* rstd, void* stream) {
    if (int rc = ln_check("kanvit_addln_fwd", M, D)) return rc;
    if (M == 0) return 0;
    if (!x || !gamma || !beta || !y || !mean || !rstd) return kv_fail(KANVIT_EINVAL, "kanvit_addln_fwd: null argument");
    if (((uintptr_t)x | (uintptr_t)(delta ? delta : x) | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)(xsum ? xsum : (const float*)y) | (uintptr_t)y) & 15)
        return kv_fail(KANVIT_EINVAL, "kanvit_addln_fwd: pointers must be 16-byte aligned");
    LnArgs a{};
    a.x = x; a.delta = (const float*)delta; a.gamma = gamma; a.beta = beta; a.xsum = xsum; a.y = (float*)y; a.mean = mean; a.rstd = rstd;
    a.M = M; a.D = D; a.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    const bool db = delta && delta_bf16, yb = y_bf16 != 0;
    return ln_dispatch(D, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        const dim3 grid(ln_grid(M)), block(64 * LN_WAVES);
        if (db && yb) hipLaunchKernelGGL((addln_fwd_kernel<NV, true, true>), grid, block, 0, st, a);
        else if (db) hipLaunchKernelGGL((addln_fwd_kernel<NV, true, false>), grid, block, 0, st, a);
        else if (yb) hipLaunchKernelGGL((addln_fwd_kernel<NV, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((addln_fwd_kernel<NV, false, false>), grid, block, 0, st, a);
        KV_LAUNCH_CHECK("addln_fwd_kernel");
        return 0;
    });
}

int kanvit_addln_fwd(int64_t M, int D, float eps, const float* x, const float* delta, const float* gamma, const float* beta,
                     float* xsum, float* y, float* mean, float* rstd, void* stream) {
    return kanvit_addln_fwd_ex(M, D, eps, x, delta, 0, gamma, beta, xsum, y, 0, mean, rstd, stream);
}

size_t kanvit_addln_bwd_workspace(int64_t M, int D) {
    if (M <= 0 || D < 4) return 0;
    return sizeof(float) * (size_t)ln_grid(M) * 2 * D;      // one [2][D] partial per work-group
}

int kanvit_addln_bwd(int64_t M, int D, const float* xsum, const float* gamma, const float* mean, const float* rstd,
                     const float* dy, const float* dres, float* dx, float* dgamma, float* dbeta, void* workspace,
                     size_t workspace_bytes, void* stream) {
    return kanvit_addln_bwd_ex(M, D, xsum, gamma, mean, rstd, dy, 0, dres, dx, nullptr, dgamma, dbeta, workspace, workspace_bytes, stream);
}

int kanvit_addln_bwd_ex(int64_t M, int D, const float* xsum, const float* gamma, const float* mean, const float* rstd,
                        const void* dy, int dy_bf16, const float* dres, float* dx, void* dx_bf16, float* dgamma, float* dbeta, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (int rc = ln_check("kanvit_addln_bwd", M, D)) return rc;
    if (!dgamma || !dbeta) return kv_fail(KANVIT_EINVAL, "kanvit_addln_bwd: null dgamma/dbeta");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        KV_HIP_CHECK(hipMemsetAsync(dgamma, 0, sizeof(float) * D, st));
        KV_HIP_CHECK(hipMemsetAsync(dbeta, 0, sizeof(float) * D, st));
        return 0;
    }
    if (!xsum || !gamma || !mean || !rstd || !dy || !dx) return kv_fail(KANVIT_EINVAL, "kanvit_addln_bwd: null argument");
    if (((uintptr_t)xsum | (uintptr_t)gamma | (uintptr_t)dy | (uintptr_t)(dres ? dres : xsum) | (uintptr_t)dx | (uintptr_t)workspace | (uintptr_t)dx_bf16) & 15)
        return kv_fail(KANVIT_EINVAL, "kanvit_addln_bwd: pointers must be 16-byte aligned");
    const size_t need = kanvit_addln_bwd_workspace(M, D);
    if (!workspace || workspace_bytes < need)
        return kv_fail(KANVIT_ENOMEM, "kanvit_addln_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
    LnArgs a{};
    a.x = xsum; a.gamma = gamma; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
    a.dy = (const float*)dy; a.dres = dres; a.dx = dx; a.dx_bf16 = (unsigned short*)dx_bf16; a.part = (float*)workspace; a.M = M; a.D = D;
    const int grid = ln_grid(M);
    int rc = ln_dispatch(D, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        if (dy_bf16) hipLaunchKernelGGL((addln_bwd_kernel<NV, true>), dim3(grid), dim3(64 * LN_WAVES), sizeof(float) * LN_WAVES * 2 * D, st, a);
        else hipLaunchKernelGGL((addln_bwd_kernel<NV, false>), dim3(grid), dim3(64 * LN_WAVES), sizeof(float) * LN_WAVES * 2 * D, st, a);
        KV_LAUNCH_CHECK("addln_bwd_kernel");
        return 0;
    });
    if (rc) return rc;
    hipLaunchKernelGGL(addln_reduce_kernel, dim3((2 * D + 31) / 32), dim3(256), 0, st, (const float*)workspace, dgamma, dbeta, D, grid);
    KV_LAUNCH_CHECK("addln_reduce_kernel");
    return 0;
}

size_t kanvit_layer_ln_bwd_workspace(const kanvit_layer_desc* d) {
    if (!d || d->M <= 0 || d->groups < 1 || d->x_group_mod < 1 || d->I < 32) return 0;
    return sizeof(float) * (size_t)ln_kan_grid(d) * 2 * d->groups * d->I;
}

int kanvit_layer_ln_bwd(const kanvit_layer_desc* d, const float* x, const float* stats, const float* bparams, const float* du,
                        float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = ln_kan_check(d, "kanvit_layer_ln_bwd")) return rc;
    if (!dgamma || !dbeta) return kv_fail(KANVIT_EINVAL, "kanvit_layer_ln_bwd: null dgamma/dbeta");
    hipStream_t st = (hipStream_t)stream;
    const int D = d->groups * d->I;
    if (d->M == 0) {
        KV_HIP_CHECK(hipMemsetAsync(dgamma, 0, sizeof(float) * D, st));
        KV_HIP_CHECK(hipMemsetAsync(dbeta, 0, sizeof(float) * D, st));
        return 0;
    }
    if (!x || !stats || !bparams || !du) return kv_fail(KANVIT_EINVAL, "kanvit_layer_ln_bwd: null argument");
    if (((uintptr_t)x | (uintptr_t)bparams | (uintptr_t)du | (uintptr_t)(dx ? dx : (float*)x) | (uintptr_t)workspace) & 15 || ((uintptr_t)stats & 7))
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_ln_bwd: pointers must be 16-byte aligned (stats: 8)");
    const size_t need = kanvit_layer_ln_bwd_workspace(d);
    if (!workspace || workspace_bytes < need)
        return kv_fail(KANVIT_ENOMEM, "kanvit_layer_ln_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
    LnKanArgs a{};
    a.x = x; a.stats = stats; a.bp = bparams; a.du = du; a.dx = dx; a.part = (float*)workspace;
    a.M = d->M; a.ldx = d->ldx; a.ldu = d->ldu; a.bp_stride = d->bparam_stride;
    a.I = d->I; a.G = d->G; a.xmod = d->x_group_mod; a.groups = d->groups; a.lpr_log2 = ln_kan_lpr_log2(d->I);
    const int ns = d->groups / d->x_group_mod, nch = (d->I / 4 + 63) / 64;
    const dim3 grid(ln_kan_grid(d), d->x_group_mod);
    const size_t lds = sizeof(float) * 4 * 2 * ns * d->I;
#define KV_LNK(NCH_, NS_)                                                                                         \
    do {                                                                                                          \
        hipLaunchKernelGGL((kan_ln_bwd_kernel<NCH_, NS_>), grid, dim3(256), lds, st, a);                          \
    } while (0)
    if (ns == 1) {
        switch (nch) { case 1: KV_LNK(1, 1); break; case 2: KV_LNK(2, 1); break; case 3: KV_LNK(3, 1); break; default: KV_LNK(4, 1); }
    } else {
        switch (nch) { case 1: KV_LNK(1, 3); break; case 2: KV_LNK(2, 3); break; default: return kv_fail(KANVIT_EINVAL, "kanvit_layer_ln_bwd: internal (nch)"); }
    }
#undef KV_LNK
    KV_LAUNCH_CHECK("kan_ln_bwd_kernel");
    hipLaunchKernelGGL(addln_reduce_kernel, dim3((2 * D + 31) / 32), dim3(256), 0, st, (const float*)workspace, dgamma, dbeta, D, (int)grid.x);
    KV_LAUNCH_CHECK("addln_reduce_kernel");
    return 0;
}

}  // extern "C"
