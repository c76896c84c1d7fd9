// Fused KAN layer kernels for gfx950 (MI355X): basis evaluation + coefficient contraction in
// one pass, forward and backward, for `groups` independent layers per launch.
//
// Shape of the work: every family is
//     Y[M x O] = Phi(X)[M x K] . W[K x O],   K = I*GP,  Phi generated on the fly from X[M x I]
// so the three kernels are GEMMs whose generated operand never exists in HBM:
//     fwd         Y   = Phi(X)   . W            (A operand generated into LDS, K-major)
//     bwd_input   dPhi = dY . W^T, then dX = sum_j dPhi_j * phi_j'(X) on the LDS-resident tile
//     bwd_weight  dW  = Phi(X)^T . dY           (split over row ranges -> slabs -> ordered reduce)
// The contraction runs on the fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products,
// fp32 accumulate -> bitwise a k-ordered fmaf chain), which is what lets the result sit within
// 1e-4 of the reference's fp32 CPU output.  One wave owns a 32-row strip of the block tile.
//
// LDS images (all fp32, ds_read_b32 / ds_write_b32 only, every access pattern below is bank
// conflict free because the row strides are odd):
//     x_s [BM][IC|1]      input tile, written coalesced (feature fastest), read row fastest
//     A_s [KC][BM+1]      generated basis values, K-major: lane = row for both the writer
//                         (basis evaluation) and the MFMA A-operand reader
//     W_s [KC][BN]        weight chunk, lane = output column
#include "kan_basis.h"
#include "kanvit_common.h"

namespace {

constexpr int BM = 128;          // rows per block in fwd / bwd_input (4 waves x 32 rows)
constexpr int NTHR = 256;
constexpr int AS = BM + 1;       // row stride of the K-major LDS tiles
constexpr int BIN_NC = 32;       // dY columns staged per step in bwd_input
constexpr int BW_ROWS = 32;      // rows staged per step in bwd_weight
constexpr int BW_AS = BW_ROWS + 1;
constexpr int BW_NT = 2;         // 64 output columns per bwd_weight block
constexpr int BW_TPW = 5;        // max 32x32 MFMA tiles per wave in bwd_weight
constexpr int BW_KC_MAX = 288;   // (BW_TPW*4 tiles / BW_NT) * 32 = 320 >= 288
constexpr int N_CU = 256;

struct LayerArgs {
    const float* x;
    const float* u;
    const float* w;
    const float* bp;
    const float* bias;
    float* y;
    const float* dy;
    float* dx;
    float* du;
    float* dparam;
    float* slab;
    long long M, ldx, ldu, ldy, bp_stride, rows_per_split;
    int I, O, groups, xmod, G, GP, order, nk, has_base, K, IC, msplit, nchunks_n;
    float rbf_inv_h;
};

__device__ __forceinline__ BasisArgs make_basis(const LayerArgs& a, int g) {
    BasisArgs b;
    b.G = a.G;
    b.GP = a.GP;
    b.order = a.order;
    b.nk = a.nk;
    b.has_base = a.has_base;
    b.inv_h = a.rbf_inv_h;
    b.bp = a.bp ? a.bp + (long long)g * a.bp_stride : nullptr;
    return b;
}

// =============================================================================================
// forward
// grid (ceil(M/BM), ceil(O/BN), groups), 256 threads
// =============================================================================================
template <int FAM, int NT>
__global__ __launch_bounds__(NTHR) void kan_fwd_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int g = blockIdx.z;
    const int IC = a.IC, GP = a.GP, ICP = IC | 1;
    const int KC = IC * GP, KCP = (KC + 1) & ~1;

    float* x_s = smem;
    float* u_s = x_s + BM * ICP;
    float* A_s = u_s + (FAM == KV_RBF ? BM * ICP : 0);
    float* W_s = A_s + KCP * AS;

    const BasisArgs b = make_basis(a, g);
    const float* xg = a.x + (long long)(g % a.xmod) * a.I;
    const float* ug = (FAM == KV_RBF && a.u) ? a.u + (long long)g * a.I : nullptr;
    const float* wg = a.w + (long long)g * a.K * a.O;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;

    if (KCP != KC)
        for (int r = tid; r < AS; r += NTHR) A_s[KC * AS + r] = 0.0f;

    for (int i0 = 0; i0 < a.I; i0 += IC) {
        // (1) x tile, coalesced along the feature axis; (2) weight chunk, coalesced along O
        for (int idx = tid; idx < BM * IC; idx += NTHR) {
            const int r = idx / IC, il = idx - r * IC;
            const long long m = m0 + r;
            const int i = i0 + il;
            const bool ok = (m < a.M) && (i < a.I);
            const float xv = ok ? xg[m * a.ldx + i] : 0.0f;
            x_s[r * ICP + il] = xv;
            if (FAM == KV_RBF) u_s[r * ICP + il] = ok ? (ug ? ug[m * a.ldu + i] : xv) : 0.0f;
        }
        for (int idx = tid; idx < KCP * BN; idx += NTHR) {
            const int kk = idx / BN, n = idx - kk * BN;
            const int kg = i0 * GP + kk;
            const bool ok = (kk < KC) && (kg < a.K) && (n0 + n < a.O);
            W_s[idx] = ok ? wg[(long long)kg * a.O + n0 + n] : 0.0f;
        }
        __syncthreads();
        // (3) basis values -> A_s (lane = row)
        for (int idx = tid; idx < BM * IC; idx += NTHR) {
            const int il = idx / BM, r = idx - il * BM;
            const int i = i0 + il;
            float* dst = A_s + (il * GP) * AS + r;
            if (i < a.I) {
                basis_fwd<FAM>(b, x_s[r * ICP + il], FAM == KV_RBF ? u_s[r * ICP + il] : 0.0f, i, dst, AS);
            } else {
                for (int j = 0; j < GP; ++j) dst[j * AS] = 0.0f;
            }
        }
        __syncthreads();
        // (4) contraction: A[row = l31][k = hf] from A_s, B[k = hf][col = l31] from W_s
        const float* ap = A_s + hf * AS + wave * 32 + l31;
        const float* wp = W_s + hf * BN + l31;
        for (int s = 0; s < KCP / 2; ++s) {
            const float av = ap[(2 * s) * AS];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float bv = wp[(2 * s) * BN + nt * 32];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    float* yg = a.y + (long long)g * a.O;
    const float* bg = a.bias ? a.bias + (long long)g * a.O : nullptr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + nt * 32 + l31;
        if (col < a.O) {
            const float bv = bg ? bg[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wave * 32 + kv_acc_row(r, hf);
                if (m < a.M) yg[m * a.ldy + col] = acc[nt][r] + bv;
            }
        }
    }
}

// =============================================================================================
// backward w.r.t. the input.  grid (ceil(M/BM), xmod); each block walks the `nshare` groups
// that read the same x columns (q, k and v of one head) and sums their contributions in LDS.
// =============================================================================================
template <int FAM, int KT>
__global__ __launch_bounds__(NTHR) void kan_bwd_input_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KCT = 32 * KT;
    constexpr int WS = KCT + 1;
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr bool SINE = (FAM == KV_SINE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const long long m0 = (long long)blockIdx.x * BM;
    const int gx = blockIdx.y;
    const int nshare = a.groups / a.xmod;
    const int IC = a.IC, GP = a.GP, ICP = IC | 1;
    const int KC = IC * GP;

    float* x_s = smem;
    float* dx_s = x_s + BM * ICP;
    float* u_s = dx_s + BM * ICP;
    float* du_s = u_s + (RBF ? BM * ICP : 0);
    float* dfq_s = du_s + (RBF ? BM * ICP : 0);                 // SINE: [nshare][4 waves][G]
    float* un = dfq_s + (SINE ? nshare * 4 * a.G : 0);
    float* dA_s = un;                                           // [KCT][AS]   (after the MFMA loop)
    float* dY_s = un;                                           // [BIN_NC][AS] (during the MFMA loop)
    float* Wt_s = un + BIN_NC * AS;                             // [BIN_NC][WS]

    const float* xg = a.x + (long long)gx * a.I;
    float* dxg = a.dx + (long long)gx * a.I;

    if (SINE)
        for (int j = tid; j < nshare * 4 * a.G; j += NTHR) dfq_s[j] = 0.0f;

    for (int i0 = 0; i0 < a.I; i0 += IC) {
        for (int idx = tid; idx < BM * IC; idx += NTHR) {
            const int r = idx / IC, il = idx - r * IC;
            const long long m = m0 + r;
            const int i = i0 + il;
            x_s[r * ICP + il] = ((m < a.M) && (i < a.I)) ? xg[m * a.ldx + i] : 0.0f;
            dx_s[r * ICP + il] = 0.0f;
        }
        for (int p = 0; p < nshare; ++p) {
            const int g = p * a.xmod + gx;
            const BasisArgs b = make_basis(a, g);
            const float* wg = a.w + (long long)g * a.K * a.O;
            const float* dyg = a.dy + (long long)g * a.O;
            f32x16 acc[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kt][r] = 0.0f;

            for (int n0 = 0; n0 < a.O; n0 += BIN_NC) {
                // dY tile transposed to [n][row]; W chunk transposed to [n][k]
                for (int idx = tid; idx < BM * BIN_NC; idx += NTHR) {
                    const int r = idx / BIN_NC, n = idx - r * BIN_NC;
                    const long long m = m0 + r;
                    const bool ok = (m < a.M) && (n0 + n < a.O);
                    dY_s[n * AS + r] = ok ? dyg[m * a.ldy + n0 + n] : 0.0f;
                }
                for (int idx = tid; idx < KCT * BIN_NC; idx += NTHR) {
                    const int kk = idx / BIN_NC, n = idx - kk * BIN_NC;
                    const int kg = i0 * GP + kk;
                    const bool ok = (kk < KC) && (kg < a.K) && (n0 + n < a.O);
                    Wt_s[n * WS + kk] = ok ? wg[(long long)kg * a.O + n0 + n] : 0.0f;
                }
                __syncthreads();
                const float* ap = dY_s + hf * AS + wave * 32 + l31;
                const float* wp = Wt_s + hf * WS + l31;
#pragma unroll 4
                for (int s = 0; s < BIN_NC / 2; ++s) {
                    const float av = ap[(2 * s) * AS];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        const float bv = wp[(2 * s) * WS + kt * 32];
                        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[kt], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
            // accumulators -> dA_s[k][row] (lane = k: stride AS, conflict free)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dA_s[(kt * 32 + l31) * AS + wave * 32 + kv_acc_row(r, hf)] = acc[kt][r];
            if (RBF) {
                const float* ug = a.u ? a.u + (long long)g * a.I : nullptr;
                for (int idx = tid; idx < BM * IC; idx += NTHR) {
                    const int r = idx / IC, il = idx - r * IC;
                    const long long m = m0 + r;
                    const int i = i0 + il;
                    const bool ok = (m < a.M) && (i < a.I);
                    u_s[r * ICP + il] = ok ? (ug ? ug[m * a.ldu + i] : xg[m * a.ldx + i]) : 0.0f;
                }
            }
            __syncthreads();
            // chain rule through the basis; uniform trip count so SINE can wave-reduce dfreq
            for (int base = 0; base < BM * IC; base += NTHR) {
                const int idx = base + tid;
                const int il = idx / BM, r = idx - il * BM;
                const bool valid = (idx < BM * IC) && (i0 + il < a.I);
                const int ilc = valid ? il : 0;
                const int rc = valid ? r : 0;
                float dxv, duv;
                basis_bwd<FAM>(b, x_s[rc * ICP + ilc], RBF ? u_s[rc * ICP + ilc] : 0.0f, i0 + ilc, valid,
                               dA_s + (ilc * GP) * AS + rc, AS, dxv, duv, SINE ? dfq_s + (p * 4 + wave) * a.G : nullptr);
                if (valid) {
                    dx_s[r * ICP + il] += dxv;
                    if (RBF) du_s[r * ICP + il] = duv;
                }
            }
            __syncthreads();
            if (RBF && a.du) {
                float* dug = a.du + (long long)g * a.I;
                for (int idx = tid; idx < BM * IC; idx += NTHR) {
                    const int r = idx / IC, il = idx - r * IC;
                    const long long m = m0 + r;
                    const int i = i0 + il;
                    if ((m < a.M) && (i < a.I)) dug[m * a.ldu + i] = du_s[r * ICP + il];
                }
            }
        }
        // same idx -> element map as the zeroing loop above: no barrier needed before the next chunk
        for (int idx = tid; idx < BM * IC; idx += NTHR) {
            const int r = idx / IC, il = idx - r * IC;
            const long long m = m0 + r;
            const int i = i0 + il;
            if ((m < a.M) && (i < a.I)) dxg[m * a.ldx + i] = dx_s[r * ICP + il];
        }
    }
    if (SINE) {
        __syncthreads();
        for (int j = tid; j < nshare * a.G; j += NTHR) {
            const int p = j / a.G, gg = j - p * a.G;
            const float* src = dfq_s + (p * 4) * a.G + gg;
            const float v = ((src[0] + src[a.G]) + src[2 * a.G]) + src[3 * a.G];
            a.dparam[((long long)blockIdx.x * a.groups + (p * a.xmod + gx)) * a.G + gg] = v;
        }
    }
}

// =============================================================================================
// backward w.r.t. the packed weights.  grid (feature chunks, msplit, groups * nchunks_n)
// =============================================================================================
template <int FAM>
__global__ __launch_bounds__(NTHR) void kan_bwd_weight_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * BW_NT;
    constexpr bool RBF = (FAM == KV_RBF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int IC = a.IC, GP = a.GP, ICP = IC | 1;
    const int i0 = blockIdx.x * IC;
    const int ms = blockIdx.y;
    const int g = blockIdx.z / a.nchunks_n;
    const int n0 = (blockIdx.z - g * a.nchunks_n) * BN;
    const int KC = IC * GP;
    const int KT = (KC + 31) / 32;
    const int ntiles = KT * BW_NT;

    float* x_s = smem;
    float* u_s = x_s + BW_ROWS * ICP;
    float* dY_s = u_s + (RBF ? BW_ROWS * ICP : 0);
    float* A_s = dY_s + BW_ROWS * BN;              // [KT*32][BW_AS]

    const BasisArgs b = make_basis(a, g);
    const float* xg = a.x + (long long)(g % a.xmod) * a.I;
    const float* ug = (RBF && a.u) ? a.u + (long long)g * a.I : nullptr;
    const float* dyg = a.dy + (long long)g * a.O;

    f32x16 acc[BW_TPW];
#pragma unroll
    for (int j = 0; j < BW_TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    for (int idx = KC * BW_AS + tid; idx < KT * 32 * BW_AS; idx += NTHR) A_s[idx] = 0.0f;   // pad rows

    const long long mbeg = (long long)ms * a.rows_per_split;
    const long long mend = (mbeg + a.rows_per_split < a.M) ? mbeg + a.rows_per_split : a.M;
    for (long long mr = mbeg; mr < mend; mr += BW_ROWS) {
        for (int idx = tid; idx < BW_ROWS * IC; idx += NTHR) {
            const int r = idx / IC, il = idx - r * IC;
            const long long m = mr + r;
            const int i = i0 + il;
            const bool ok = (m < mend) && (i < a.I);
            const float xv = ok ? xg[m * a.ldx + i] : 0.0f;
            x_s[r * ICP + il] = xv;
            if (RBF) u_s[r * ICP + il] = ok ? (ug ? ug[m * a.ldu + i] : xv) : 0.0f;
        }
        for (int idx = tid; idx < BW_ROWS * BN; idx += NTHR) {
            const int r = idx / BN, n = idx - r * BN;
            const long long m = mr + r;
            const bool ok = (m < mend) && (n0 + n < a.O);
            dY_s[idx] = ok ? dyg[m * a.ldy + n0 + n] : 0.0f;      // zero rows kill the padded rows of A_s
        }
        __syncthreads();
        for (int idx = tid; idx < BW_ROWS * IC; idx += NTHR) {
            const int il = idx / BW_ROWS, r = idx - il * BW_ROWS;
            const int i = i0 + il;
            float* dst = A_s + (il * GP) * BW_AS + r;
            if (i < a.I) {
                basis_fwd<FAM>(b, x_s[r * ICP + il], RBF ? u_s[r * ICP + il] : 0.0f, i, dst, BW_AS);
            } else {
                for (int j = 0; j < GP; ++j) dst[j * BW_AS] = 0.0f;
            }
        }
        __syncthreads();
        // dW tile[k][o] += sum_rows A[row][k] * dY[row][o]: MFMA "row" index = k, contraction = row
#pragma unroll
        for (int j = 0; j < BW_TPW; ++j) {
            const int t = wave + 4 * j;
            if (t < ntiles) {
                const int kt = t / BW_NT, nt = t - kt * BW_NT;
                const float* ap = A_s + (kt * 32 + l31) * BW_AS + hf;
                const float* bp2 = dY_s + hf * BN + nt * 32 + l31;
#pragma unroll 4
                for (int s = 0; s < BW_ROWS / 2; ++s)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp2[(2 * s) * BN], acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    float* slab = a.slab + ((long long)ms * a.groups + g) * a.K * a.O;
#pragma unroll
    for (int j = 0; j < BW_TPW; ++j) {
        const int t = wave + 4 * j;
        if (t < ntiles) {
            const int kt = t / BW_NT, nt = t - kt * BW_NT;
            const int col = n0 + nt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kt * 32 + kv_acc_row(r, hf);
                const int k = i0 * GP + kk;
                if (kk < KC && k < a.K && col < a.O) slab[(long long)k * a.O + col] = acc[j][r];
            }
        }
    }
}

// ordered sum of the msplit partial slabs (deterministic; no float atomics)
__global__ __launch_bounds__(256) void kan_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                              long long total, int msplit) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        float s = slab[e];
        for (int ms = 1; ms < msplit; ++ms) s += slab[(long long)ms * total + e];
        dw[e] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int gp_of(const kanvit_layer_desc* d) {
    switch (d->family) {
        case KANVIT_LINEAR: return 1;
        case KANVIT_CHEBY: return d->G;
        case KANVIT_BSPLINE: return d->G + (d->has_base ? 1 : 0);
        case KANVIT_RBF: return d->G + (d->has_base ? 1 : 0);
        case KANVIT_SINE: return d->G;
        case KANVIT_FOURIER: return 2 * d->G;
        default: return -1;
    }
}

int validate(const kanvit_layer_desc* d, const char* who) {
    if (!d) return kv_fail(KANVIT_EINVAL, "%s: null descriptor", who);
    const int gp = gp_of(d);
    if (gp < 1) return kv_fail(KANVIT_EINVAL, "%s: unknown family %d or G=%d", who, d->family, d->G);
    if (gp > 80) return kv_fail(KANVIT_EINVAL, "%s: %d generated columns per feature exceeds the supported 80", who, gp);
    if (d->groups < 1 || d->x_group_mod < 1 || d->groups % d->x_group_mod != 0)
        return kv_fail(KANVIT_EINVAL, "%s: groups=%d must be a positive multiple of x_group_mod=%d", who, d->groups,
                       d->x_group_mod);
    if (d->groups > 65535) return kv_fail(KANVIT_EINVAL, "%s: groups=%d exceeds 65535", who, d->groups);
    if (d->I < 1 || d->O < 1 || d->M < 0) return kv_fail(KANVIT_EINVAL, "%s: bad sizes M=%lld I=%d O=%d", who,
                                                          (long long)d->M, d->I, d->O);
    if ((long long)d->I * gp > 0x7fffffffLL / 4) return kv_fail(KANVIT_EINVAL, "%s: K too large", who);
    if (d->ldx < (int64_t)d->x_group_mod * d->I) return kv_fail(KANVIT_EINVAL, "%s: ldx=%lld < x_group_mod*I", who,
                                                                 (long long)d->ldx);
    if (d->ldy < (int64_t)d->groups * d->O) return kv_fail(KANVIT_EINVAL, "%s: ldy=%lld < groups*O", who,
                                                            (long long)d->ldy);
    if (d->family == KANVIT_BSPLINE) {
        const int nk = d->G + d->spline_order + 1;
        if (d->spline_order < 0 || d->G < 1 || nk > KV_MAX_KNOTS)
            return kv_fail(KANVIT_EINVAL, "%s: bspline G=%d order=%d unsupported (knots %d > %d)", who, d->G,
                           d->spline_order, nk, KV_MAX_KNOTS);
        if (d->bparam_stride < (int64_t)d->I * nk) return kv_fail(KANVIT_EINVAL, "%s: bparam_stride too small", who);
    }
    if (d->family == KANVIT_RBF && d->bparam_stride < d->G) return kv_fail(KANVIT_EINVAL, "%s: bparam_stride too small", who);
    if (d->family == KANVIT_SINE && d->bparam_stride < (int64_t)d->G * (1 + d->I))
        return kv_fail(KANVIT_EINVAL, "%s: bparam_stride too small", who);
    return 0;
}

LayerArgs base_args(const kanvit_layer_desc* d) {
    LayerArgs a{};
    a.M = d->M;
    a.ldx = d->ldx;
    a.ldu = d->ldu;
    a.ldy = d->ldy;
    a.bp_stride = d->bparam_stride;
    a.I = d->I;
    a.O = d->O;
    a.groups = d->groups;
    a.xmod = d->x_group_mod;
    a.G = d->G;
    a.GP = gp_of(d);
    a.order = d->spline_order;
    a.nk = d->G + d->spline_order + 1;
    a.has_base = d->has_base;
    a.K = d->I * a.GP;
    a.rbf_inv_h = d->rbf_inv_h;
    return a;
}

int needs_bparams(int family) { return family == KANVIT_BSPLINE || family == KANVIT_RBF || family == KANVIT_SINE; }

// ---- forward -----------------------------------------------------------------------------------
template <int FAM, int NT>
int launch_fwd(const LayerArgs& a, hipStream_t st) {
    constexpr int BN = 32 * NT;
    const int ICP = a.IC | 1;
    const int KC = a.IC * a.GP, KCP = (KC + 1) & ~1;
    const size_t lds = sizeof(float) * ((size_t)BM * ICP * (FAM == KV_RBF ? 2 : 1) + (size_t)KCP * AS + (size_t)KCP * BN);
    static bool attr_done = false;   // benign race: idempotent
    if (!attr_done) {
        KV_HIP_CHECK(kv_allow_lds(kan_fwd_kernel<FAM, NT>, 160 * 1024));
        attr_done = true;
    }
    dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.O + BN - 1) / BN), (unsigned)a.groups);
    hipLaunchKernelGGL((kan_fwd_kernel<FAM, NT>), grid, dim3(NTHR), lds, st, a);
    KV_LAUNCH_CHECK("kan_fwd_kernel");
    return 0;
}

template <int FAM>
int dispatch_fwd(LayerArgs& a, hipStream_t st) {
    const int nt = a.O <= 32 ? 1 : (a.O <= 64 ? 2 : 4);
    const int kcmax = nt == 4 ? 64 : 80;
    int ic = kcmax / a.GP;
    if (ic < 1) ic = 1;
    if (ic > a.I) ic = a.I;
    a.IC = ic;
    if (nt == 1) return launch_fwd<FAM, 1>(a, st);
    if (nt == 2) return launch_fwd<FAM, 2>(a, st);
    return launch_fwd<FAM, 4>(a, st);
}

// ---- backward input ------------------------------------------------------------------------------
template <int FAM, int KT>
int launch_bwd_input(const LayerArgs& a, hipStream_t st) {
    constexpr int KCT = 32 * KT;
    const int ICP = a.IC | 1;
    const int nshare = a.groups / a.xmod;
    const size_t un = (size_t)KCT * AS > (size_t)BIN_NC * AS + (size_t)BIN_NC * (KCT + 1)
                          ? (size_t)KCT * AS
                          : (size_t)BIN_NC * AS + (size_t)BIN_NC * (KCT + 1);
    const size_t lds = sizeof(float) * ((size_t)BM * ICP * (FAM == KV_RBF ? 4 : 2) +
                                        (FAM == KV_SINE ? (size_t)nshare * 4 * a.G : 0) + un);
    static bool attr_done = false;
    if (!attr_done) {
        KV_HIP_CHECK(kv_allow_lds(kan_bwd_input_kernel<FAM, KT>, 160 * 1024));
        attr_done = true;
    }
    dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)a.xmod, 1);
    hipLaunchKernelGGL((kan_bwd_input_kernel<FAM, KT>), grid, dim3(NTHR), lds, st, a);
    KV_LAUNCH_CHECK("kan_bwd_input_kernel");
    return 0;
}

template <int FAM>
int dispatch_bwd_input(LayerArgs& a, hipStream_t st) {
    int ic = 96 / a.GP;
    if (ic < 1) ic = 1;
    if (ic > a.I) ic = a.I;
    a.IC = ic;
    const int kt = (ic * a.GP + 31) / 32;
    if (kt == 1) return launch_bwd_input<FAM, 1>(a, st);
    if (kt == 2) return launch_bwd_input<FAM, 2>(a, st);
    return launch_bwd_input<FAM, 3>(a, st);
}

// ---- backward weight -----------------------------------------------------------------------------
struct BwPlan {
    int ic, nfchunks, nchunks_n, msplit;
    long long rows_per_split;
};

BwPlan plan_bwd_weight(const kanvit_layer_desc* d) {
    BwPlan p;
    const int gp = gp_of(d);
    int ic = 64;
    while (ic > 1 && ic * gp > BW_KC_MAX) ic >>= 1;
    if (ic > d->I) ic = d->I;
    p.ic = ic;
    p.nfchunks = (d->I + ic - 1) / ic;
    p.nchunks_n = (d->O + 32 * BW_NT - 1) / (32 * BW_NT);
    const long long base = (long long)p.nfchunks * p.nchunks_n * d->groups;
    long long want = (4LL * N_CU + base - 1) / base;                      // ~4 blocks per CU over the chip
    const long long maxsplit = (d->M + 4 * BW_ROWS - 1) / (4 * BW_ROWS);  // at least 128 rows per split
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    if (want > 65535) want = 65535;
    long long rps = (d->M + want - 1) / want;
    rps = (rps + BW_ROWS - 1) / BW_ROWS * BW_ROWS;
    if (rps < BW_ROWS) rps = BW_ROWS;
    p.rows_per_split = rps;
    p.msplit = (int)((d->M + rps - 1) / rps);
    if (p.msplit < 1) p.msplit = 1;
    return p;
}

template <int FAM>
int launch_bwd_weight(const LayerArgs& a, const BwPlan& p, hipStream_t st) {
    const int ICP = a.IC | 1;
    const int KT = (a.IC * a.GP + 31) / 32;
    const size_t lds = sizeof(float) * ((size_t)BW_ROWS * ICP * (FAM == KV_RBF ? 2 : 1) + (size_t)BW_ROWS * 32 * BW_NT +
                                        (size_t)KT * 32 * BW_AS);
    static bool attr_done = false;
    if (!attr_done) {
        KV_HIP_CHECK(kv_allow_lds(kan_bwd_weight_kernel<FAM>, 160 * 1024));
        attr_done = true;
    }
    dim3 grid((unsigned)p.nfchunks, (unsigned)p.msplit, (unsigned)(a.groups * p.nchunks_n));
    hipLaunchKernelGGL((kan_bwd_weight_kernel<FAM>), grid, dim3(NTHR), lds, st, a);
    KV_LAUNCH_CHECK("kan_bwd_weight_kernel");
    return 0;
}

#define KV_FAMILY_SWITCH(fam, CALL)                                   \
    switch (fam) {                                                    \
        case KANVIT_LINEAR: return CALL(KV_LINEAR);                   \
        case KANVIT_CHEBY: return CALL(KV_CHEBY);                     \
        case KANVIT_BSPLINE: return CALL(KV_BSPLINE);                 \
        case KANVIT_RBF: return CALL(KV_RBF);                         \
        case KANVIT_SINE: return CALL(KV_SINE);                       \
        case KANVIT_FOURIER: return CALL(KV_FOURIER);                 \
        default: return kv_fail(KANVIT_EINVAL, "unknown family %d", fam); \
    }

}  // namespace

thread_local char g_kanvit_err[512] = "";

extern "C" {

const char* kanvit_last_error(void) { return g_kanvit_err; }
int kanvit_abi_version(void) { return KANVIT_ABI_VERSION; }
int kanvit_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return kv_fail(KANVIT_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int kanvit_layer_fwd(const kanvit_layer_desc* d, const float* x, const float* u, const float* w, const float* bparams,
                     const float* bias, float* y, void* stream) {
    if (int rc = validate(d, "kanvit_layer_fwd")) return rc;
    if (d->M == 0) return 0;
    if (!x || !w || !y) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: null x/w/y");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: family %d needs bparams", d->family);
    if (d->family == KANVIT_RBF && u && d->ldu < (int64_t)d->groups * d->I)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: ldu < groups*I");
    if (d->M == 0) return 0;
    LayerArgs a = base_args(d);
    a.x = x;
    a.u = u;
    a.w = w;
    a.bp = bparams;
    a.bias = bias;
    a.y = y;
    hipStream_t st = (hipStream_t)stream;
#define KV_CALL(F) dispatch_fwd<F>(a, st)
    KV_FAMILY_SWITCH(d->family, KV_CALL)
#undef KV_CALL
}

int64_t kanvit_layer_dparam_tiles(const kanvit_layer_desc* d) {
    if (!d || d->family != KANVIT_SINE) return 0;
    return (d->M + BM - 1) / BM;
}

int kanvit_layer_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,
                           const float* bparams, const float* dy, float* dx, float* du, float* dparam, void* stream) {
    if (int rc = validate(d, "kanvit_layer_bwd_input")) return rc;
    if (d->M == 0) return 0;
    if (!x || !w || !dy || !dx) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: null x/w/dy/dx");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: family %d needs bparams", d->family);
    if (d->family == KANVIT_SINE && !dparam) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: SINE needs dparam");
    if (d->family == KANVIT_RBF && (u || du) && d->ldu < (int64_t)d->groups * d->I)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: ldu < groups*I");
    if (d->family == KANVIT_SINE && (d->groups / d->x_group_mod) * 4 * d->G > 4096)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: SINE G too large");
    if (d->M == 0) return 0;
    LayerArgs a = base_args(d);
    a.x = x;
    a.u = u;
    a.w = w;
    a.bp = bparams;
    a.dy = dy;
    a.dx = dx;
    a.du = du;
    a.dparam = dparam;
    hipStream_t st = (hipStream_t)stream;
#define KV_CALL(F) dispatch_bwd_input<F>(a, st)
    KV_FAMILY_SWITCH(d->family, KV_CALL)
#undef KV_CALL
}

size_t kanvit_layer_bwd_weight_workspace(const kanvit_layer_desc* d) {
    if (!d || gp_of(d) < 1 || d->groups < 1 || d->I < 1 || d->O < 1) return 0;
    const BwPlan p = plan_bwd_weight(d);
    if (p.msplit <= 1) return 0;
    return sizeof(float) * (size_t)p.msplit * d->groups * ((size_t)d->I * gp_of(d)) * d->O;
}

int kanvit_layer_bwd_weight(const kanvit_layer_desc* d, const float* x, const float* u, const float* bparams,
                            const float* dy, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = validate(d, "kanvit_layer_bwd_weight")) return rc;
    if (d->M == 0) {   // no rows: the gradient is exactly zero
        if (!dw) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: null dw");
        KV_HIP_CHECK(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->groups * d->I * gp_of(d) * d->O, (hipStream_t)stream));
        return 0;
    }
    if (!x || !dy || !dw) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: null x/dy/dw");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: family %d needs bparams", d->family);
    if (d->family == KANVIT_RBF && u && d->ldu < (int64_t)d->groups * d->I)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: ldu < groups*I");
    const BwPlan p = plan_bwd_weight(d);
    const size_t need = kanvit_layer_bwd_weight_workspace(d);
    if (need > 0 && (!workspace || workspace_bytes < need))
        return kv_fail(KANVIT_ENOMEM, "kanvit_layer_bwd_weight: workspace %zu bytes < required %zu", workspace_bytes, need);
    LayerArgs a = base_args(d);
    a.x = x;
    a.u = u;
    a.bp = bparams;
    a.dy = dy;
    a.IC = p.ic;
    a.msplit = p.msplit;
    a.nchunks_n = p.nchunks_n;
    a.rows_per_split = p.rows_per_split;
    a.slab = (p.msplit > 1) ? (float*)workspace : dw;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (d->family) {
        case KANVIT_LINEAR: rc = launch_bwd_weight<KV_LINEAR>(a, p, st); break;
        case KANVIT_CHEBY: rc = launch_bwd_weight<KV_CHEBY>(a, p, st); break;
        case KANVIT_BSPLINE: rc = launch_bwd_weight<KV_BSPLINE>(a, p, st); break;
        case KANVIT_RBF: rc = launch_bwd_weight<KV_RBF>(a, p, st); break;
        case KANVIT_SINE: rc = launch_bwd_weight<KV_SINE>(a, p, st); break;
        case KANVIT_FOURIER: rc = launch_bwd_weight<KV_FOURIER>(a, p, st); break;
        default: return kv_fail(KANVIT_EINVAL, "unknown family %d", d->family);
    }
    if (rc) return rc;
    if (p.msplit > 1) {
        const long long total = (long long)d->groups * a.K * d->O;
        long long nb = (total + 255) / 256;
        if (nb > 8 * N_CU) nb = 8 * N_CU;
        hipLaunchKernelGGL(kan_slab_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float*)workspace, dw, total,
                           p.msplit);
        KV_LAUNCH_CHECK("kan_slab_reduce_kernel");
    }
    return 0;
}

// ---- per-family named entry points ---------------------------------------------------------------
#define KV_DEFINE_FAMILY(name, FAMID)                                                                               \
    int kanvit_##name##_fwd(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,            \
                            const float* bp, const float* bias, float* y, void* s) {                               \
        if (!d || d->family != FAMID) return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_fwd: descriptor family mismatch"); \
        return kanvit_layer_fwd(d, x, u, w, bp, bias, y, s);                                                        \
    }                                                                                                               \
    int kanvit_##name##_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,      \
                                  const float* bp, const float* dy, float* dx, float* du, float* dp, void* s) {    \
        if (!d || d->family != FAMID) return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_bwd_input: descriptor family mismatch"); \
        return kanvit_layer_bwd_input(d, x, u, w, bp, dy, dx, du, dp, s);                                           \
    }                                                                                                               \
    int kanvit_##name##_bwd_weight(const kanvit_layer_desc* d, const float* x, const float* u, const float* bp,    \
                                   const float* dy, float* dw, void* ws, size_t wsb, void* s) {                    \
        if (!d || d->family != FAMID) return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_bwd_weight: descriptor family mismatch"); \
        return kanvit_layer_bwd_weight(d, x, u, bp, dy, dw, ws, wsb, s);                                            \
    }                                                                                                               \
    int kanvit_##name##_qkv_fwd(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,        \
                                const float* bp, const float* bias, float* y, void* s) {                           \
        if (!d || d->family != FAMID || d->groups != 3 * d->x_group_mod)                                            \
            return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_qkv_fwd: need family match and groups == 3*x_group_mod"); \
        return kanvit_layer_fwd(d, x, u, w, bp, bias, y, s);                                                        \
    }                                                                                                               \
    int kanvit_##name##_qkv_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,  \
                                      const float* bp, const float* dy, float* dx, float* du, float* dp, void* s) { \
        if (!d || d->family != FAMID || d->groups != 3 * d->x_group_mod)                                            \
            return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_qkv_bwd_input: need family match and groups == 3*x_group_mod"); \
        return kanvit_layer_bwd_input(d, x, u, w, bp, dy, dx, du, dp, s);                                           \
    }                                                                                                               \
    int kanvit_##name##_qkv_bwd_weight(const kanvit_layer_desc* d, const float* x, const float* u, const float* bp, \
                                       const float* dy, float* dw, void* ws, size_t wsb, void* s) {                \
        if (!d || d->family != FAMID || d->groups != 3 * d->x_group_mod)                                            \
            return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_qkv_bwd_weight: need family match and groups == 3*x_group_mod"); \
        return kanvit_layer_bwd_weight(d, x, u, bp, dy, dw, ws, wsb, s);                                            \
    }
KV_DEFINE_FAMILY(linear, KANVIT_LINEAR)
KV_DEFINE_FAMILY(cheby, KANVIT_CHEBY)
KV_DEFINE_FAMILY(bspline, KANVIT_BSPLINE)
KV_DEFINE_FAMILY(rbf, KANVIT_RBF)
KV_DEFINE_FAMILY(sine, KANVIT_SINE)
KV_DEFINE_FAMILY(fourier, KANVIT_FOURIER)

}  // extern "C"
