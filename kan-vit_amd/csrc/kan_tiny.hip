// Tiny per-head KAN layers (I <= 16, O <= 16: train.py's own default geometry has d = 64, 8 heads -> per-head layers of
// 8 -> 8 features, 24 of them per q|k|v launch, models/cheby.py:36-48 under attention.py:188-197) -- SURVEY.md section 8(f)1.
// A 32x32 MFMA tile is 94 % padding for an 8 x 8 head, and the general LDS-tile kernels spend 10 / 19 / 23 us per launch on a
// producer/consumer pipeline built for K = 320 x O = 64 tiles.  Here the contraction runs on the VECTOR pipe:
//   forward / input gradient: one THREAD per (row, group); a work-group's rows share the group, so the weights are wave-uniform
//     (scalar loads, SGPR operands of the FMAs) and the basis values never leave registers (BasisGen / BasisDGen, kan_basis.h);
//   weight gradient: per (slab, group) a work-group stages phi[64 rows][K] and dY[64 rows][O] in LDS and every thread owns
//     outputs (k, o); partial slabs in the layout of the register kernel, summed by kan_slab_reduce_kernel (deterministic).
// Exact fp32 (fmaf chains).  Families without extra gradients only: LINEAR, CHEBY, uniform cubic BSPLINE, FOURIER.
#include "../../include/kanvit.h"
#include "kan_basis.h"
#include "kanvit_common.h"

#include <type_traits>

namespace {

__device__ __forceinline__ BasisArgs tiny_basis(const KvTinyArgs& a, int g) {
    BasisArgs b;
    b.G = a.G;
    b.GP = a.GP;
    b.order = a.order;
    b.nk = a.nk;
    b.has_base = a.has_base;
    b.inv_h = 0.0f;
    b.bp = a.bp ? a.bp + (long long)g * a.bp_stride : nullptr;
    b.uniform = (a.flags & KANVIT_FLAG_UNIFORM_KNOTS) && a.order == 3;
    return b;
}

// grid (ceil(M / 256), groups): thread = row
template <int FAM, int OT>
__global__ __launch_bounds__(256) void kan_tiny_fwd_kernel(const KvTinyArgs a) {
    const int g = blockIdx.y, gx = g % a.xmod;
    // the group's weights ([K][O] <= 4 KB) go to LDS once: every thread then reads them as broadcasts (a chain of scalar
    // loads -- one s_load + wait per (i, j) -- measured 20 us for this launch)
    __shared__ __attribute__((aligned(16))) float w_s[200 * OT];      // [K][OT], columns past O zero: the inner loop needs no guard
    for (int e = threadIdx.x; e < a.K * OT; e += 256) {               // (a guard per output column compiled to a branch + LDS wait each)
        const int k = e / OT, o = e - k * OT;
        w_s[e] = o < a.O ? a.w[((long long)g * a.K + k) * a.O + o] : 0.0f;
    }
    __syncthreads();
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= a.M) return;
    const BasisArgs b = tiny_basis(a, g);
    const float* xr = a.x + m * a.ldx + (long long)gx * a.I;
    const float* wg = w_s;
    float y[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) y[o] = (a.bias && o < a.O) ? a.bias[(long long)g * a.O + o] : 0.0f;
    for (int i = 0; i < a.I; ++i) {
        BasisGen<FAM> gen;
        gen.init(b, xr[i], 0.0f, i);
#pragma unroll 5
        for (int j = 0; j < a.GP; ++j) {           // partial unroll: several LDS reads in flight instead of read -> wait -> 8 FMAs
            const float phi = gen.next(j);
            const float* wr = wg + (i * a.GP + j) * OT;                 // LDS broadcast reads
#pragma unroll
            for (int o = 0; o < OT; ++o) y[o] = __builtin_fmaf(phi, wr[o], y[o]);
        }
    }
    float* yr = a.y + m * a.ldy + (long long)g * a.O;
#pragma unroll
    for (int o = 0; o < OT; ++o)
        if (o < a.O) yr[o] = y[o];
}

// grid (ceil(M / 256), xmod): thread = (row, x slice); the groups sharing the slice are summed in registers
template <int FAM, int OT, int IT>
__global__ __launch_bounds__(256) void kan_tiny_bwd_input_kernel(const KvTinyArgs a) {
    const int gx = blockIdx.y, nshare = a.groups / a.xmod;
    extern __shared__ __attribute__((aligned(16))) float w_s[];   // [nshare][K][OT]: weights of the groups sharing this x slice, zero padded
    const int kw = a.K * OT;
    for (int e = threadIdx.x; e < nshare * kw; e += 256) {
        const int p = e / kw, r = e - p * kw, k = r / OT, o = r - k * OT;
        w_s[e] = o < a.O ? a.w[((long long)(p * a.xmod + gx) * a.K + k) * a.O + o] : 0.0f;
    }
    __syncthreads();
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= a.M) return;
    const float* xr = a.x + m * a.ldx + (long long)gx * a.I;
    float xv[IT], dxv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        xv[i] = i < a.I ? xr[i] : 0.0f;
        dxv[i] = 0.0f;
    }
    for (int p = 0; p < nshare; ++p) {
        const int g = p * a.xmod + gx;
        const BasisArgs b = tiny_basis(a, g);
        const float* dyr = a.dy + m * a.ldy + (long long)g * a.O;
        const float* wg = w_s + p * kw;
        float dyv[OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) dyv[o] = o < a.O ? dyr[o] : 0.0f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            if (i < a.I) {
                BasisDGen<FAM> gen;
                gen.init(b, xv[i], 0.0f, i);
                float acc = 0.0f;
#pragma unroll 5
                for (int j = 0; j < a.GP; ++j) {
                    const float d = gen.next(j);
                    const float* wr = wg + (i * a.GP + j) * OT;
                    float dphi = 0.0f;
#pragma unroll
                    for (int o = 0; o < OT; ++o) dphi = __builtin_fmaf(dyv[o], wr[o], dphi);
                    acc = __builtin_fmaf(dphi, d, acc);
                }
                dxv[i] += acc;
            }
        }
    }
    float* dxr = a.dx + m * a.ldx + (long long)gx * a.I;
#pragma unroll
    for (int i = 0; i < IT; ++i)
        if (i < a.I) dxr[i] = dxv[i];
}

// grid (slabs, groups), 256 threads; LDS: phi_s[64][K + 1] | dy_s[64][OT]
template <int FAM, int OT>
__global__ __launch_bounds__(256) void kan_tiny_bwd_weight_kernel(const KvTinyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, g = blockIdx.y, gx = g % a.xmod, K = a.K, KS = K + 1;
    float* phi_s = smem;
    float* dy_s = smem + 64 * KS;
    const BasisArgs b = tiny_basis(a, g);
    const long long ms = (long long)blockIdx.x * a.rows_per_slab;
    long long me = ms + a.rows_per_slab;
    if (me > a.M) me = a.M;
    const int nout = K * a.O;                       // <= 4 outputs per thread (K * O <= 1024, host-checked)
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (long long r0 = ms; r0 < me; r0 += 64) {
        // basis values of 64 rows x I features (rows past the slab end: zero), dY rows
        for (int e = tid; e < 64 * a.I; e += 256) {
            const int r = e / a.I, i = e - r * a.I;
            const long long m = r0 + r;
            float* dst = phi_s + r * KS + i * a.GP;
            if (m < me) {
                BasisGen<FAM> gen;
                gen.init(b, a.x[m * a.ldx + (long long)gx * a.I + i], 0.0f, i);
                for (int j = 0; j < a.GP; ++j) dst[j] = gen.next(j);
            } else {
                for (int j = 0; j < a.GP; ++j) dst[j] = 0.0f;
            }
        }
        for (int e = tid; e < 64 * OT; e += 256) {
            const int r = e / OT, o = e - r * OT;
            const long long m = r0 + r;
            dy_s[e] = (m < me && o < a.O) ? a.dy[m * a.ldy + (long long)g * a.O + o] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q;
            if (idx < nout) {
                const int k = idx / a.O, o = idx - k * a.O;
                float s = acc[q];
#pragma unroll 8
                for (int r = 0; r < 64; ++r) s = __builtin_fmaf(phi_s[r * KS + k], dy_s[r * OT + o], s);
                acc[q] = s;
            }
        }
        __syncthreads();
    }
    float* out = a.slab + ((long long)blockIdx.x * a.groups + g) * nout;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q;
        if (idx < nout) out[idx] = acc[q];
    }
}

template <typename F>
int tiny_family(int family, F&& f) {
    switch (family) {
        case KANVIT_LINEAR: return f(std::integral_constant<int, KV_LINEAR>{});
        case KANVIT_CHEBY: return f(std::integral_constant<int, KV_CHEBY>{});
        case KANVIT_BSPLINE: return f(std::integral_constant<int, KV_BSPLINE>{});
        case KANVIT_FOURIER: return f(std::integral_constant<int, KV_FOURIER>{});
        default: return kv_fail(KANVIT_EINVAL, "internal: tiny-head dispatch (family %d)", family);
    }
}

}  // namespace

bool kv_tiny_ok(const kanvit_layer_desc* d) {
    if (kv_config().no_tiny || d->I > 16 || d->O > 16 || d->I < 1 || d->O < 1) return false;
    const int fam = d->family;
    int gp;
    if (fam == KANVIT_LINEAR) gp = 1;
    else if (fam == KANVIT_CHEBY) gp = d->G;
    else if (fam == KANVIT_FOURIER) gp = 2 * d->G;
    else if (fam == KANVIT_BSPLINE && (d->flags & KANVIT_FLAG_UNIFORM_KNOTS) && d->spline_order == 3) gp = d->G + d->has_base;
    else return false;
    if (gp < 1 || d->I * gp > 200 || d->I * gp * d->O > 1024) return false;      // K <= 200: the weight gradient's phi tile [64][K + 1] stays under 64 KB of LDS
    if (d->x_group_mod < 1 || d->groups % d->x_group_mod || (d->groups / d->x_group_mod) * d->I * gp * 16 > 15 * 1024) return false;
    return d->M >= 64;                               // fewer rows: the general kernels' single tile is as good
}

int kv_tiny_slabs(const kanvit_layer_desc* d) {
    long long s = (d->M + 127) / 128;             // 128 rows (two staged chunks) per slab: the chunks of a work-group run back to back
    if (s > 64) s = 64;                            // behind a global-load latency each, so few chunks per group beats few partials
    if (s < 1) s = 1;
    return (int)s;
}

int kv_tiny_fwd(const KvTinyArgs& a, hipStream_t st) {
    const dim3 grid((unsigned)((a.M + 255) / 256), (unsigned)a.groups);
    return tiny_family(a.family, [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        if (a.O <= 8) hipLaunchKernelGGL((kan_tiny_fwd_kernel<F, 8>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((kan_tiny_fwd_kernel<F, 16>), grid, dim3(256), 0, st, a);
        KV_LAUNCH_CHECK("kan_tiny_fwd_kernel");
        return 0;
    });
}

int kv_tiny_bwd_input(const KvTinyArgs& a, hipStream_t st) {
    const dim3 grid((unsigned)((a.M + 255) / 256), (unsigned)a.xmod);
    return tiny_family(a.family, [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        const int nshare = a.groups / a.xmod;
        if (a.O <= 8 && a.I <= 8)
            hipLaunchKernelGGL((kan_tiny_bwd_input_kernel<F, 8, 8>), grid, dim3(256), sizeof(float) * nshare * a.K * 8, st, a);
        else
            hipLaunchKernelGGL((kan_tiny_bwd_input_kernel<F, 16, 16>), grid, dim3(256), sizeof(float) * nshare * a.K * 16, st, a);
        KV_LAUNCH_CHECK("kan_tiny_bwd_input_kernel");
        return 0;
    });
}

int kv_tiny_bwd_weight(const KvTinyArgs& a, hipStream_t st) {
    const dim3 grid((unsigned)a.slabs, (unsigned)a.groups);
    return tiny_family(a.family, [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        if (a.O <= 8) {
            const size_t lds = sizeof(float) * (64 * (size_t)(a.K + 1) + 64 * 8);
            hipLaunchKernelGGL((kan_tiny_bwd_weight_kernel<F, 8>), grid, dim3(256), lds, st, a);
        } else {
            const size_t lds = sizeof(float) * (64 * (size_t)(a.K + 1) + 64 * 16);
            hipLaunchKernelGGL((kan_tiny_bwd_weight_kernel<F, 16>), grid, dim3(256), lds, st, a);
        }
        KV_LAUNCH_CHECK("kan_tiny_bwd_weight_kernel");
        return 0;
    });
}
