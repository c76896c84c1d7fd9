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

#include <type_traits>
#include "kanvit_common.h"

#include <stdlib.h>

namespace {

constexpr int BM = 128;          // rows per block in fwd / bwd_input (4 consumer waves x 32 rows)
constexpr int NTHR = 512;        // 4 consumer (MFMA) waves + 4 producer (load / basis) waves
constexpr int NPROD = 256;       // producer threads
constexpr int AS = BM + 1;       // row stride of the K-major LDS tiles
constexpr int BIN_NC = 32;       // dY columns staged per step in bwd_input
constexpr int BW_ROWS = 32;      // rows staged per step in bwd_weight
constexpr int BW_AS = BW_ROWS + 1;
constexpr int BW_NT = 2;         // 64 output columns per bwd_weight block
constexpr int BW_TPW = 5;        // max 32x32 MFMA tiles per consumer wave in bwd_weight
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
    const unsigned short* wb;   // bf16 fragment-major repack of w (KANVIT_FLAG_BF16_MFMA)
    const unsigned short* wb2;  // bf16 repack for the input-gradient kernel: [g][chunk][O/8][KCT][8 n]
    long long M, ldx, ldu, ldy, bp_stride, rows_per_split;
    int I, O, groups, xmod, G, GP, order, nk, has_base, K, IC, msplit, nchunks_n;
    float rbf_inv_h;
    int flags;
    // patch gather (kanvit_patch_embed_*): x is an NCHW image batch, row m = (sample m / P, patch m % P), feature i = (c, iy, ix)
    // of the patch (model.py:111-126); y rows are shifted behind `pg_pre` prepended rows per sample (the class token), `pos`
    // ([P + pg_pre][O]) is added and the class-token row cls + pos[0] is written by the lanes that own a sample's first patch
    int pg, pg_C, pg_H, pg_W, pg_n, pg_pre;
    const float* cls;
    const float* pos;
    // KANVIT_FLAG_FUSED_LN (RBF): the spline-path input u = LayerNorm(x slice) * gamma + beta (models/fastkan.py:68) is formed
    // in the kernels; bparams of a group = [centres(G) | gamma(I) | beta(I)]; stats[M][xmod][2] = (mean, rstd) per row and x
    // slice, written by the forward kernel and read by the two backward kernels
    int vcols;            // bf16 input gradient of ONE wide layer (O = 64*v): the v column chunks run as "groups" sharing x, basis and chain rule
    int ln;
    float ln_eps;
    float* stats;
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
    b.uniform = (a.flags & KANVIT_FLAG_UNIFORM_KNOTS) && a.order == 3;
    return b;
}

// LayerNorm statistics of one row of I features held by a lane pair: this lane sees ICH consecutive features of every
// chunk of 2*ICH (xh points at its first one), its partner lane (l ^ 32) the others.  Two passes (mean, then centred
// sum of squares: the accuracy of torch's Welford kernel), biased variance, rstd = rsqrt(var + eps) as nn.LayerNorm.
template <int ICH>
__device__ __forceinline__ void kv_ln_row_stats(const float* __restrict__ xh, int nch, int I, float eps, float& mean, float& rstd) {
    float s = 0.0f;
    for (int c = 0; c < nch; ++c)
#pragma unroll
        for (int e = 0; e < ICH; ++e) s += xh[c * 2 * ICH + e];
    s += __shfl_xor(s, 32);
    mean = s / (float)I;
    float q = 0.0f;
    for (int c = 0; c < nch; ++c)
#pragma unroll
        for (int e = 0; e < ICH; ++e) {
            const float d = xh[c * 2 * ICH + e] - mean;
            q = fmaf(d, d, q);
        }
    q += __shfl_xor(q, 32);
    rstd = rsqrtf(q / (float)I + eps);
}

__device__ __forceinline__ int kv_pow2_ge(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ---------------------------------------------------------------------------------------------
// Producer helpers.  All global loads of a pass are issued before the first LDS store so that
// their latencies overlap (the compiler keeps the order of the two unrolled loops).
// ---------------------------------------------------------------------------------------------
// rows x IC tile of a row-major matrix -> dst[r*ICP + il]; thread map (il = pt % ICR, r = pt / ICR)
// with ICR = pow2 >= IC keeps the global reads coalesced along the feature axis without a division.
template <int ROWS>
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src, long long ld,
                                           long long row0, long long row_end, int i0, int I, int IC, int ICP, int pt) {
    const int ICR = kv_pow2_ge(IC);
    const int lg = __builtin_ctz(ICR);
    const int il = pt & (ICR - 1);
    const int rstep = NPROD >> lg;            // ICR <= 128 guaranteed by the host (IC <= 96)
    const int r0 = pt >> lg;
    const bool col_ok = (il < IC) && (i0 + il < I);
    const float* base = src + row0 * ld + i0;                      // uniform
    const int ldi = (int)ld;
    const int nrows = (row_end - row0 < ROWS) ? (int)(row_end - row0) : ROWS;   // valid rows (may be <= 0)
    const int off = r0 * ldi + il;
    float* d = dst + r0 * ICP + il;
    constexpr int NB = ROWS >= 128 ? 8 : 4;      // loads in flight per round: one memory latency per round
    for (int rb = 0; rb < ROWS; rb += NB * rstep) {
        float v[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int r = rb + q * rstep;
            v[q] = (col_ok && r0 + r < nrows) ? base[off + r * ldi] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int r = rb + q * rstep;
            if (il < IC && r0 + r < ROWS) d[r * ICP] = v[q];
        }
    }
}

// =============================================================================================
// forward.  grid (nsets * ceil(O/BN), ceil(M/BM)), 512 threads; nsets = groups / NSH.
// NSH = number of groups that share ONE generated basis tile: for families without per-layer
// basis parameters (LINEAR, CHEBY, FOURIER) the q, k and v mappings of a head read the same x
// columns and therefore the same Phi(x), so one block evaluates Phi once and contracts it against
// the three weight sets (NSH = 3); otherwise NSH = 1.
// Measured on MI355X (tools/coissue_probe*.hip): the fp32-input MFMA does NOT overlap with VALU
// work of either wave on its SIMD (time adds), unlike the bf16 MFMA.  The producer code below is
// therefore written for minimum instruction count: per-thread pointers are set up once, bounds
// checks collapse to wave-uniform flags on interior tiles, and all loads of a pass are issued
// before the first LDS store.
// Pipeline (one __syncthreads per feature chunk c):
//   consumers: MFMA on A_s/W_s[c&1]
//   producers: W chunk c+1 -> W_s[(c+1)&1]; basis(x_s[(c+1)&1]) -> A_s[(c+1)&1]; x chunk c+2 -> x_s[c&1]
// =============================================================================================
template <int FAM, int NT, int NSH, bool FAST>
__global__ __launch_bounds__(NTHR) void kan_fwd_kernel(const LayerArgs a) {
    // FAST (host-checked): IC is a power of two dividing I, O % BN == 0 -- every chunk and column tile is
    // interior, so the only remaining bounds question is the last row tile (wave-uniform flag full_m).
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * NT;
    constexpr int WROW = NSH * BN;              // floats per W_s row
    constexpr int V4 = BN / 4;                  // float4 per W row of one group (power of two)
    constexpr int WRS = NPROD / V4;             // W rows per staging pass
    constexpr bool RBF = (FAM == KV_RBF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const bool consumer = wave < 4;
    const int pt = tid & (NPROD - 1);
    const int ntn = (a.O + BN - 1) / BN;
    const int gs = blockIdx.x / ntn;
    const int n0 = (blockIdx.x - gs * ntn) * BN;
    const int nsets = a.groups / NSH;
    const long long m0 = (long long)blockIdx.y * BM;
    const int IC = a.IC, GP = a.GP, ICP = IC | 1;
    const int KC = IC * GP, KCP = (KC + 1) & ~1;
    const int XS = BM * ICP, ASZ = KCP * AS, WSZ = KCP * WROW;
    const int nch = (a.I + IC - 1) / IC;
    const bool full_m = (m0 + BM <= a.M);
    const bool full_n = FAST || ((n0 + BN <= a.O) && ((a.O & 3) == 0));
    const int mrem = full_m ? BM : (int)(a.M - m0);      // valid rows of this tile

    float* x_s = smem;                          // [2][XS]
    float* u_s = x_s + 2 * XS;                  // [2][XS]  (RBF)
    float* A_s = u_s + (RBF ? 2 * XS : 0);      // [2][ASZ]
    float* W_s = A_s + 2 * ASZ;                 // [2][WSZ]

    const BasisArgs b = make_basis(a, gs);      // NSH > 1 only for families without basis parameters
    const int xcol = (NSH == 1 ? gs % a.xmod : gs) * a.I;
    const int ldx = (int)a.ldx, ldu = (RBF && a.u) ? (int)a.ldu : (int)a.ldx;

    // ---- producer state: uniform bases + 32-bit per-thread offsets, set up once ----
    const int ICR = FAST ? IC : kv_pow2_ge(IC);
    const int xlg = __builtin_ctz(ICR);
    const int xl = pt & (ICR - 1), xr0 = pt >> xlg, xrs = NPROD >> xlg;
    const float* xbase = a.x + m0 * a.ldx + xcol;                                        // uniform
    const float* ubase = (RBF && a.u) ? a.u + m0 * a.ldu + (long long)gs * a.I : xbase;  // uniform
    const int xoff = xr0 * ldx + xl, uoff = xr0 * ldu + xl;
    const int xsoff = xr0 * ICP + xl;
    const int wk = pt / V4, wc = (pt & (V4 - 1)) * 4;
    const int woff = wk * a.O + wc;
    const int gr = pt & (BM - 1), gl0 = pt >> 7;              // basis: fixed row gr, features gl0, gl0+2, ...

    auto stage_x = [&](int c, int buf) {
        float* dxs = x_s + buf * XS + xsoff;
        float* dus = u_s + buf * XS + xsoff;
        const float* sx = xbase + c * IC;
        const float* su = ubase + c * IC;
        const int npass = (BM + xrs - 1) / xrs;
        if (FAST && full_m) {
            for (int q0 = 0; q0 < npass; q0 += 4) {
                float v[4], w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = sx[xoff + (q0 + q) * xrs * ldx];
                    if (RBF) w[q] = su[uoff + (q0 + q) * xrs * ldu];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dxs[(q0 + q) * xrs * ICP] = v[q];
                    if (RBF) dus[(q0 + q) * xrs * ICP] = w[q];
                }
            }
        } else {
            const bool col_ok = (xl < IC) && (c * IC + xl < a.I);
            for (int q0 = 0; q0 < npass; q0 += 4) {
                float v[4], w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = xr0 + (q0 + q) * xrs;
                    const bool ok = col_ok && (r < mrem);
                    v[q] = ok ? sx[xoff + (q0 + q) * xrs * ldx] : 0.0f;
                    if (RBF) w[q] = ok ? su[uoff + (q0 + q) * xrs * ldu] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = xr0 + (q0 + q) * xrs;
                    if (xl < IC && r < BM) {
                        dxs[(q0 + q) * xrs * ICP] = v[q];
                        if (RBF) dus[(q0 + q) * xrs * ICP] = w[q];
                    }
                }
            }
        }
    };
    auto stage_w = [&](int c, int buf) {
        float* dst = W_s + buf * WSZ + wk * WROW + wc;
        const int k0 = c * KC;
        const int npass = (KCP + WRS - 1) / WRS;
#pragma unroll
        for (int p = 0; p < NSH; ++p) {
            const int g = (NSH == 1) ? gs : p * nsets + gs;
            const float* src = a.w + ((long long)g * a.K + k0) * a.O + n0;               // uniform
            for (int q0 = 0; q0 < npass; q0 += 4) {
                f32x4 val[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = wk + (q0 + q) * WRS;
                    f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (kk < KC && (FAST || k0 + kk < a.K)) {
                        const float* sp = src + woff + (q0 + q) * WRS * a.O;
                        if (full_n) {
                            t = *reinterpret_cast<const f32x4*>(sp);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n0 + wc + e < a.O) t[e] = sp[e];
                        }
                    }
                    val[q] = t;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = wk + (q0 + q) * WRS;
                    if (kk < KCP) *reinterpret_cast<f32x4*>(dst + (q0 + q) * WRS * WROW + p * BN) = val[q];
                }
            }
        }
    };
    auto gen_a = [&](int c, int buf) {
        const float* xs = x_s + buf * XS + gr * ICP;
        const float* us = u_s + buf * XS + gr * ICP;
        float* As = A_s + buf * ASZ + gr;
        for (int il = gl0; il < IC; il += 2) {
            const int i = c * IC + il;
            float* dst = As + (il * GP) * AS;
            if (FAST || i < a.I) {
                basis_fwd<FAM>(b, xs[il], RBF ? us[il] : 0.0f, i, dst, AS);
            } else {
                for (int j = 0; j < GP; ++j) dst[j * AS] = 0.0f;
            }
        }
    };

    f32x16 acc[NSH * NT];
#pragma unroll
    for (int t = 0; t < NSH * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // prologue: x chunk 0; then operands of chunk 0 and x chunk 1
    if (!consumer) {
        stage_x(0, 0);
        if (KCP != KC)
            for (int r = pt; r < 2 * AS; r += NPROD) A_s[(r / AS) * ASZ + KC * AS + (r % AS)] = 0.0f;
    }
    __syncthreads();
    if (!consumer) {
        stage_w(0, 0);
        gen_a(0, 0);
        if (nch > 1) stage_x(1, 1);
    }
    __syncthreads();

    for (int c = 0; c < nch; ++c) {
        if (consumer) {
            // A[row = l31][k = hf] from A_s (lane = row), B[k = hf][col = l31] from W_s (lane = column)
            const float* ap = A_s + (c & 1) * ASZ + hf * AS + wave * 32 + l31;
            const float* wp = W_s + (c & 1) * WSZ + hf * WROW + l31;
#pragma unroll 2
            for (int s = 0; s < KCP / 2; ++s) {
                const float av = ap[(2 * s) * AS];
#pragma unroll
                for (int t = 0; t < NSH * NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wp[(2 * s) * WROW + t * 32], acc[t], 0, 0, 0);
            }
        } else if (!consumer && c + 1 < nch) {
            stage_w(c + 1, (c + 1) & 1);
            gen_a(c + 1, (c + 1) & 1);
            if (c + 2 < nch) stage_x(c + 2, c & 1);
        }
        __syncthreads();
    }

    if constexpr (FAST) {
        // Epilogue: the accumulator layout gives each lane one dword per store (2 x 128 B per wave-instruction, 96
    // instructions per lane, consumer waves only) -- store-issue bound (0.12 of 0.33 ms in the bf16 kernel).
    // Instead park the tile in the idle operand buffers and let all 8 waves write float4 rows.
    {
        constexpr int OS = WROW + 4;                 // row stride: 16-byte aligned, lane = column -> conflict free
        float* O_s = A_s;
        if (consumer) {
#pragma unroll
            for (int t = 0; t < NSH * NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) O_s[(wave * 32 + kv_acc_row(r, hf)) * OS + t * 32 + l31] = acc[t][r];
        }
        __syncthreads();
        constexpr int V4R = WROW / 4;                // float4 per tile row
        for (int v = tid; v < BM * V4R; v += NTHR) {
            const int row = v / V4R, c4 = (v - row * V4R) * 4;
            const int p = c4 / BN, cl = c4 - p * BN;
            if (row < mrem) {
                const int g = (NSH == 1) ? gs : p * nsets + gs;
                f32x4 val = *reinterpret_cast<const f32x4*>(O_s + row * OS + c4);
                if (a.bias) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + (long long)g * a.O + n0 + cl);
                    val += bv;
                }
                *reinterpret_cast<f32x4*>(a.y + (m0 + row) * a.ldy + (long long)g * a.O + n0 + cl) = val;
            }
        }
    }
    } else {
    if (consumer) {
        const int ldy = (int)a.ldy;
        const int rbase = wave * 32 + 4 * hf;                 // kv_acc_row(r, hf) = (r&3) + 8*(r>>2) + 4*hf
#pragma unroll
        for (int p = 0; p < NSH; ++p) {
            const int g = (NSH == 1) ? gs : p * nsets + gs;
            float* yt = a.y + m0 * a.ldy + (long long)g * a.O + n0;                      // uniform
            const float* bg = a.bias ? a.bias + (long long)g * a.O + n0 : nullptr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int cl = nt * 32 + l31;
                if (FAST || n0 + cl < a.O) {
                    const float bv = bg ? bg[cl] : 0.0f;
                    const int yo = rbase * ldy + cl;
                    if (full_m) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            yt[yo + ((r & 3) + 8 * (r >> 2)) * ldy] = acc[p * NT + nt][r] + bv;
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (rbase + (r & 3) + 8 * (r >> 2) < mrem)
                                yt[yo + ((r & 3) + 8 * (r >> 2)) * ldy] = acc[p * NT + nt][r] + bv;
                    }
                }
            }
        }
    }
    }
}

// =============================================================================================
// forward, register-operand form (fp32 exact).  Because the fp32 MFMA and VALU work serialise on a SIMD no matter which
// wave issues them (section header above), nothing is gained by evaluating the basis in other waves -- and the LDS basis
// tile, its barriers and the producer waves are pure overhead.  Here every lane generates its own MFMA A operand:
//   the K index is permuted so that lane half hf owns whole features: k-step s of a chunk <-> (feature j = s / GP of the
//   half's ICH features, basis index g = s % GP); A[row = lane&31][k = hf] = phi_g(x[row][i0 + hf*ICH + j]) comes straight
//   from a BasisGen in registers, and the W chunk is staged with the same permutation (row (s, hf) <- k = feature*GP + g).
// 256 threads = 4 waves x 32 rows; LDS holds only two W chunk buffers (float4 global loads prefetched into registers one
// chunk ahead); x is read by each lane directly (ICH consecutive floats of its row per chunk); one barrier per chunk.
// Epilogue: each wave transposes its 32x32 tiles through a private LDS patch and writes float4 row segments.
// Requirements (host-checked): O % (32*NT) == 0, I % IC == 0, IC in {8, 4, 2}, 16-byte aligned rows when IC == 8.
// =============================================================================================
// GPC > 0: the number of basis functions per feature is a compile-time constant (the shapes the reference instantiates).  The
// chunk body is then fully unrolled and the W fragments are read from LDS ONE K-STEP AHEAD into a second register set, with
// scheduling fences pinning "reads of step s+1, then MFMAs of step s": an LDS read takes ~100 cycles from issue to use and
// the round-1 form (read -> s_waitcnt lgkmcnt(0) -> two MFMAs, the same destination registers every time) left the matrix
// pipe idle for most of that on every second MFMA (59 % busy in the PMC pass).  GPC == 0 keeps the runtime-GP loop.
template <int FAM, int NT, int NSH, int ICH, int GPC = 0>
__global__ __launch_bounds__(256, GPC ? 2 : 1) void kan_fwd_reg_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * NT;
    constexpr int WROW = NSH * BN;
    constexpr int V4 = BN / 4;
    constexpr int IC = 2 * ICH;
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr int TS = 36;                        // staging patch row stride (floats): 16-byte aligned, conflict free
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int ntn = a.O / BN;
    const int gs = blockIdx.x / ntn;
    const int n0 = (blockIdx.x - gs * ntn) * BN;
    const int nsets = a.groups / NSH;
    const long long m0 = (long long)blockIdx.y * BM;
    const int GP = a.GP, KC = IC * GP;            // k rows per chunk (even)
    const int nch = a.I / IC;
    const int WSZ = KC * WROW;
    const int mrem = (m0 + BM <= a.M) ? BM : (int)(a.M - m0);
    float* W_s = smem;                            // [2][KC][WROW], row (2s + hf)

    const BasisArgs b = make_basis(a, gs);
    const int xcol = (NSH == 1 ? gs % a.xmod : gs) * a.I;
    const int row = wave * 32 + l31;
    const bool row_ok = row < mrem;
    const float* xrow = a.x + (m0 + (row_ok ? row : 0)) * a.ldx + xcol + hf * ICH;
    const float* urow = (RBF && a.u) ? a.u + (m0 + (row_ok ? row : 0)) * a.ldu + (long long)gs * a.I + hf * ICH : xrow;
    // patch gather: the lane's row is a patch of an NCHW image; its ICH features of a chunk are ICH consecutive pixels of one
    // image line (host-checked: the chunk width divides the patch width), and the chunks are visited in order, so the
    // position inside the patch advances incrementally: no division in the chunk loop
    long long yrow = m0 + row;                    // output row of this lane's input row
    const float* posrow = nullptr;                // position embedding row added to it
    bool cls_owner = false;                       // this lane also writes its sample's class-token row
    float ln_mean = 0.0f, ln_rstd = 1.0f;        // KANVIT_FLAG_FUSED_LN: statistics of this lane's row over the group's x slice
    const float* ln_gb = nullptr;                 // gamma of this lane's first feature (beta I floats further on)
    if constexpr (RBF) {
        if (a.ln) {
            kv_ln_row_stats<ICH>(xrow, nch, a.I, a.ln_eps, ln_mean, ln_rstd);
            ln_gb = b.bp + a.G + hf * ICH;
            if (hf == 0 && row_ok && gs < a.xmod) {
                float2 st = {ln_mean, ln_rstd};
                *reinterpret_cast<float2*>(a.stats + ((m0 + row) * a.xmod + gs) * 2) = st;
            }
        }
    }
    int pg_ix = 0, pg_iy = 0, pg_off = 0, pg_pw = 0, pg_ph = 0;
    if (a.pg) {
        const int P = a.pg_n * a.pg_n;
        pg_ph = a.pg_H / a.pg_n;
        pg_pw = a.pg_W / a.pg_n;
        const long long m = m0 + (row_ok ? row : 0);
        const long long smp = m / P;
        const int pidx = (int)(m - smp * P);
        const int py = pidx / a.pg_n, px = pidx - py * a.pg_n;
        xrow = a.x + ((smp * a.pg_C) * a.pg_H + (long long)py * pg_ph) * a.pg_W + px * pg_pw;      // patch origin in channel 0
        pg_ix = pg_off = hf * ICH;
        yrow = m + (smp + 1) * a.pg_pre;
        posrow = a.pos ? a.pos + (long long)(pidx + a.pg_pre) * a.O : nullptr;
        cls_owner = a.pg_pre && a.cls && pidx == 0;
    }

    // W staging: thread -> (LDS row lr = tid / V4 (+ 256/V4 per pass), 4 columns wc); LDS row (s, h) <- global k
    const int wc = (tid & (V4 - 1)) * 4, wr0 = tid / V4;
    constexpr int WRS = 256 / V4;
    constexpr int WQ = (NSH == 3) ? 4 : 8;        // passes held in registers (host guarantees ceil(KC / WRS) <= WQ)
    f32x4 wreg[NSH][WQ];
    int koff[WQ];                                 // natural k offset (times O) of the LDS rows this thread stages, or -1
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int lr = wr0 + q * WRS;             // LDS row = 2*s + h
        const int s_ = lr >> 1, h_ = lr & 1;
        const int j_ = s_ / GP, g_ = s_ - j_ * GP;
        koff[q] = (lr < KC) ? ((h_ * ICH + j_) * GP + g_) * a.O : -1;
    }
    auto load_w = [&](int c) {
#pragma unroll
        for (int p = 0; p < NSH; ++p) {
            const int g = (NSH == 1) ? gs : p * nsets + gs;
            const float* src = a.w + ((long long)g * a.K + (long long)c * KC) * a.O + n0 + wc;      // chunk base (natural k order)
#pragma unroll
            for (int q = 0; q < WQ; ++q)
                if (koff[q] >= 0) wreg[p][q] = *reinterpret_cast<const f32x4*>(src + koff[q]);
        }
    };
    auto store_w = [&](int buf) {
        float* dst = W_s + buf * WSZ + wc;
#pragma unroll
        for (int p = 0; p < NSH; ++p)
#pragma unroll
            for (int q = 0; q < WQ; ++q)
                if (koff[q] >= 0) *reinterpret_cast<f32x4*>(dst + (wr0 + q * WRS) * WROW + p * BN) = wreg[p][q];
    };

    f32x16 acc[NSH * NT];
#pragma unroll
    for (int t = 0; t < NSH * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    float xv[ICH], uv[ICH];
    auto load_x = [&](int c) {                    // called for c = 0, 1, 2, ... in order
        const float* xs = a.pg ? xrow + pg_off : xrow + c * IC;
        if constexpr (ICH == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xs);
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = v[e];
            if (RBF && !a.ln) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(urow + c * IC);
#pragma unroll
                for (int e = 0; e < 4; ++e) uv[e] = w4[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < ICH; ++e) {
                xv[e] = xs[e];
                if (RBF && !a.ln) uv[e] = urow[c * IC + e];
            }
        }
        if constexpr (RBF) {
            if (a.ln) {                           // u = (x - mean) * rstd * gamma + beta, the operation order of nn.LayerNorm
#pragma unroll
                for (int e = 0; e < ICH; ++e) uv[e] = (xv[e] - ln_mean) * ln_rstd * ln_gb[c * IC + e] + ln_gb[a.I + c * IC + e];
            }
        }
        if (a.pg) {                               // next chunk: IC pixels further along the line, then next line, then next channel
            pg_ix += IC;
            pg_off += IC;
            if (pg_ix >= pg_pw) {
                pg_ix -= pg_pw;
                pg_off += a.pg_W - pg_pw;
                if (++pg_iy == pg_ph) {
                    pg_iy = 0;
                    pg_off += (a.pg_H - pg_ph) * a.pg_W;
                }
            }
        }
    };

    load_w(0);
    load_x(0);
    store_w(0);
    __syncthreads();

    for (int c = 0; c < nch; ++c) {
        float xc[ICH], uc[ICH];
#pragma unroll
        for (int e = 0; e < ICH; ++e) {
            xc[e] = xv[e];
            uc[e] = RBF ? uv[e] : 0.0f;
        }
        if (c + 1 < nch) {                        // prefetch the next chunk; lands while this chunk's MFMAs run
            load_w(c + 1);
            load_x(c + 1);
        }
        const float* wp = W_s + (c & 1) * WSZ + hf * WROW + l31;
        if constexpr (GPC > 0) {
            constexpr int VH = ICH * GPC;               // k-steps of this chunk (one generated value per lane and step)
            constexpr int NTT = NSH * NT;
            float phi[VH];
#pragma unroll
            for (int j = 0; j < ICH; ++j) {
                BasisGen<FAM> gen;
                gen.init(b, xc[j], uc[j], c * IC + hf * ICH + j);
#pragma unroll
                for (int g = 0; g < GPC; ++g) phi[j * GPC + g] = gen.next(g);
            }
            float wa[2][NTT];
#pragma unroll
            for (int t = 0; t < NTT; ++t) wa[0][t] = wp[t * 32];
#pragma unroll
            for (int s2 = 0; s2 < VH; ++s2) {
                if (s2 + 1 < VH) {
#pragma unroll
                    for (int t = 0; t < NTT; ++t) wa[(s2 + 1) & 1][t] = wp[(2 * (s2 + 1)) * WROW + t * 32];
                }
                __builtin_amdgcn_sched_barrier(0);      // the reads of step s2+1 are issued before the MFMAs of step s2 ...
#pragma unroll
                for (int t = 0; t < NTT; ++t)           // flipped product Y^T = W^T . Phi^T: accumulator rows = y columns
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s2 & 1][t], phi[s2], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // ... and nothing else is hoisted across (register budget: 2 waves per SIMD)
            }
        } else {
#pragma unroll
        for (int j = 0; j < ICH; ++j) {
            BasisGen<FAM> gen;
            gen.init(b, xc[j], uc[j], c * IC + hf * ICH + j);
            const float* wj = wp + (2 * j * GP) * WROW;
            for (int g = 0; g < GP; ++g) {
                const float av = gen.next(g);
#pragma unroll
                for (int t = 0; t < NSH * NT; ++t)      // flipped product Y^T = W^T . Phi^T: accumulator rows = y columns
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wj[(2 * g) * WROW + t * 32], av, acc[t], 0, 0, 0);
            }
        }
        }
        if (c + 1 < nch) store_w((c + 1) & 1);
        __syncthreads();
    }

    // epilogue: with the flipped product, accumulator registers 4q..4q+3 of a tile are 4 consecutive y columns of the lane's
    // OWN row (column 8q + 4hf + 0..3 of the tile): float4 stores straight from registers, no staging tile, no barrier
    if (row < mrem) {
#pragma unroll
        for (int t = 0; t < NSH * NT; ++t) {
            const int p = t / NT, nt = t - p * NT;
            const int g = (NSH == 1) ? gs : p * nsets + gs;
            const int col = n0 + nt * 32 + 4 * hf;
            float* yp = a.y + yrow * a.ldy + (long long)g * a.O + col;
            const float* bp = a.bias ? a.bias + (long long)g * a.O + col : nullptr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                if (bp) v += *reinterpret_cast<const f32x4*>(bp + 8 * q);
                if (posrow) v += *reinterpret_cast<const f32x4*>(posrow + col + 8 * q);
                *reinterpret_cast<f32x4*>(yp + 8 * q) = v;
                if (cls_owner) {                  // the class-token row of this sample: cls + pos[0] (model.py:150-152), this column tile
                    f32x4 cv = *reinterpret_cast<const f32x4*>(a.cls + col + 8 * q);
                    if (a.pos) cv += *reinterpret_cast<const f32x4*>(a.pos + col + 8 * q);
                    *reinterpret_cast<f32x4*>(yp - a.ldy + 8 * q) = cv;
                }
            }
        }
    }
}

// =============================================================================================
// backward w.r.t. the input, register form (fp32 exact).  dPhi^T = W . dY^T: the MFMA row index is k, the column index
// the token row, so a lane holds dPhi for ITS token row.  The k rows are permuted so that lane half hf owns whole features:
//   slot q = 16*kt + r of half hf (accumulator kt, register r; tile row kv_acc_row(r, hf)) <-> (feature jq = q / GP of the
//   half's FPH features, basis index g = q % GP);  W_s row kt*32 + rho holds w[(i0 + hf*FPH + jq)*GP + g][:] (or zeros).
// After the contraction every lane applies the chain rule to its own FPH features in registers and writes dx straight to
// global -- no dPhi tile in LDS, no parking barriers, no producer waves.  dY and x are read per lane (its own row).
// The contraction index n is split per 32-column chunk as n = n0 + hf*16 + s so a lane consumes 16 CONSECUTIVE dY values.
// 256 threads = 4 waves x 32 rows; LDS = two W^T chunk buffers [32 n][KCT+1]; one barrier per 32 dY columns.
// SHARED: q, k, v of a head are summed in the accumulators (one chain rule per head); otherwise one chain rule per group.
// Requirements (host-checked): GP compile time, I % (2*FPH) == 0, O % 32 == 0, 16-byte aligned rows.
// SINE (GP = 5): d loss / d freq is summed per lane over the features of a step, wave-reduced into per-wave LDS slots and
// written as this row tile's partials to dparam (same protocol as the LDS-tile kernel).
// =============================================================================================
template <int FAM, int GP, int KT, bool SHARED>
__global__ __launch_bounds__(256, (FAM == KV_SINE || FAM == KV_FOURIER) ? 1 : 2) void kan_bwd_input_reg_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KCT = 32 * KT;
    // W^T image: element (n = h*16 + s, k) at s*WS2 + h*HOFF + k.  HOFF = 32 (mod 64) puts the two lane halves of a fragment
    // read on disjoint bank halves (with the old [2s+h][KCT+1] rows the halves were 33 banks apart: one bank shared, every
    // ds_read2_b32 took 3 cycles instead of 2 -- the 16.7 % LDS conflict rate of profiles/r02_sq_pmc_fp32.md); WS2 = 2 (mod 16)
    // spreads the staging writes (8 lanes: s = e + 4j', h = 0/1, 8 consecutive k) over all 64 banks.
    constexpr int HOFF = (KCT % 64 == 32) ? KCT : KCT + 32;
    constexpr int WS2 = ((HOFF + KCT + 13) / 16) * 16 + 2;
    constexpr int FPH = (16 * KT) / GP;           // features per lane half and chunk
    constexpr int IC = 2 * FPH;
    constexpr bool RBF = (FAM == KV_RBF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int gx = blockIdx.x;
    const long long m0 = (long long)blockIdx.y * BM;
    const int nshare = a.groups / a.xmod;
    const int nci = a.I / IC, ncn = a.O / 32;
    const int mrem = (m0 + BM <= a.M) ? BM : (int)(a.M - m0);
    const int row = wave * 32 + l31;
    const bool row_ok = row < mrem;
    const long long grow = m0 + (row_ok ? row : 0);
    float* W_s = smem;                            // [2][16][WS2]
    constexpr int WSZ = 16 * WS2;
    constexpr bool SINE = (FAM == KV_SINE);
    float* dfq_s = W_s + 2 * WSZ;                 // SINE: [nshare][4 waves][GP] partial d loss / d freq of this row tile
    if constexpr (SINE) {
        for (int j = tid; j < nshare * 4 * GP; j += 256) dfq_s[j] = 0.0f;
    }

    const float* xrow = a.x + grow * a.ldx + (long long)gx * a.I + hf * FPH;
    float* dxrow = a.dx + grow * a.ldx + (long long)gx * a.I + hf * FPH;
    const float* dyrow = a.dy + grow * a.ldy + hf * 16;

    // W^T staging: a thread loads float4 along n for one k row and scatters it to 4 LDS n-rows.  LDS k row kr <-> slot.
    constexpr int NV = KCT * 8;                   // float4 per chunk (KCT rows x 32 n / 4)
    constexpr int WQ = (NV + 255) / 256;
    int kofs[WQ];                                 // (natural k offset inside the feature chunk) * O, or -1 for zero rows
    int nofs[WQ];
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int v = tid + q * 256;
        const int kr = v >> 3, n4 = (v & 7) * 4;  // LDS k row, first of 4 n
        const int rho = kr & 31, kt = kr >> 5;
        const int h_ = (rho >> 2) & 1, r_ = (rho & 3) + 4 * (rho >> 3);
        const int slot = kt * 16 + r_;
        const int jq = slot / GP, g_ = slot - jq * GP;
        kofs[q] = (v < NV && jq < FPH) ? ((h_ * FPH + jq) * GP + g_) * a.O : -1;
        nofs[q] = n4;
    }
    f32x4 wreg[WQ];
    auto load_w = [&](int ci, int g, int cn) {
        const float* src = a.w + ((long long)g * a.K + (long long)ci * IC * GP) * a.O + cn * 32;
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
            if (kofs[q] >= 0) t = *reinterpret_cast<const f32x4*>(src + kofs[q] + nofs[q]);
            wreg[q] = t;
        }
    };
    auto store_w = [&](int buf) {
        float* dst = W_s + buf * WSZ;
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int v = tid + q * 256;
            if (v < NV) {
                const int kr = v >> 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = nofs[q] + e;                    // n within the 32-column chunk: n = h*16 + s
                    dst[(n & 15) * WS2 + (n >> 4) * HOFF + kr] = wreg[q][e];
                }
            }
        }
    };
    f32x4 dyreg[4];
    auto load_dy = [&](int g, int cn) {
        const float* src = dyrow + (long long)g * a.O + cn * 32;
#pragma unroll
        for (int e = 0; e < 4; ++e) dyreg[e] = *reinterpret_cast<const f32x4*>(src + 4 * e);
    };

    const int spc = nshare * ncn;                 // steps per feature chunk
    const int T = nci * spc;
    f32x16 acc[KT];
    float dxacc[FPH];
    float xv[FPH];

    int ci = 0, p = 0, cn = 0;
    load_w(0, gx, 0);
    load_dy(gx, 0);
    store_w(0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (p == 0 && cn == 0) {                  // new feature chunk
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kt][r] = 0.0f;
#pragma unroll
            for (int j = 0; j < FPH; ++j) dxacc[j] = 0.0f;
            if constexpr (FPH % 4 == 0) {
#pragma unroll
                for (int j4 = 0; j4 < FPH / 4; ++j4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + ci * IC + 4 * j4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[4 * j4 + e] = v[e];
                }
            } else {
#pragma unroll
                for (int j = 0; j < FPH; ++j) xv[j] = xrow[ci * IC + j];
            }
        }
        // this step's dY values (register copy), then prefetch the next step's operands
        float dyv[16];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) dyv[4 * e + c4] = dyreg[e][c4];
        int cin = ci, pn = p, cnn = cn + 1;
        if (cnn == ncn) { cnn = 0; ++pn; }
        if (pn == nshare) { pn = 0; ++cin; }
        if (t + 1 < T) {
            load_w(cin, pn * a.xmod + gx, cnn);
            load_dy(pn * a.xmod + gx, cnn);
        }
        // contraction over this chunk's 32 dY columns: A = W^T rows (LDS, lane = k row), B = dY of this lane's row
        // W^T fragments are read ONE k-step ahead into a second register set (same reasoning as the forward kernel: an LDS
        // read -> wait -> MFMA chain on one register pair leaves the matrix pipe idle for the read latency on every pair)
        const float* wp = W_s + (t & 1) * WSZ + hf * HOFF + l31;
        float wa[2][KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) wa[0][kt] = wp[kt * 32];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            if (s2 + 1 < 16) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) wa[(s2 + 1) & 1][kt] = wp[(s2 + 1) * WS2 + kt * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s2 & 1][kt], dyv[s2], acc[kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool ends = SHARED ? (p == nshare - 1 && cn == ncn - 1) : (cn == ncn - 1);
        if (ends) {
            // chain rule in registers: this lane's FPH features of its own row
            const int g = p * a.xmod + gx;
            const BasisArgs b = make_basis(a, g);
            float duv[RBF ? FPH : 1];
            float uvv[RBF ? FPH : 1];
            if constexpr (RBF) {
                if (a.ln) {                       // KANVIT_FLAG_FUSED_LN: u from x, the saved row statistics and this group's gamma / beta
                    const float2 st = *reinterpret_cast<const float2*>(a.stats + (grow * a.xmod + gx) * 2);
                    const float* gb = b.bp + a.G + ci * IC + hf * FPH;
#pragma unroll
                    for (int j = 0; j < FPH; ++j) uvv[j] = (xv[j] - st.x) * st.y * gb[j] + gb[a.I + j];
                } else {
                    const float* urow = a.u ? a.u + grow * a.ldu + (long long)g * a.I + hf * FPH + ci * IC : xrow + ci * IC;
#pragma unroll
                    for (int j = 0; j < FPH; ++j) uvv[j] = urow[j];
                }
            }
            float dfq[SINE ? GP : 1];
            if constexpr (SINE) {
#pragma unroll
                for (int g_ = 0; g_ < GP; ++g_) dfq[g_] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < FPH; ++j) {
                BasisDGen<FAM> gen;
                gen.init(b, xv[j], RBF ? uvv[j] : 0.0f, ci * IC + hf * FPH + j);
                float dsum = 0.0f, usum = 0.0f;
#pragma unroll
                for (int g_ = 0; g_ < GP; ++g_) {
                    const int slot = j * GP + g_;
                    const float d = gen.next(g_);
                    const float v = acc[slot / 16][slot % 16];
                    if (RBF && g_ < GP - 1) usum += v * d;       // RBF: the last column is the silu base path (has_base)
                    else dsum += v * d;
                    if constexpr (SINE) dfq[g_] += v * gen.lastc * xv[j];
                }
                if (RBF && !a.has_base) { usum += dsum; dsum = 0.0f; }
                dxacc[j] += dsum;
                if constexpr (RBF) duv[j] = usum;
            }
            if constexpr (SINE) {                 // one wave reduction per grid point and step; rows past M contribute nothing
#pragma unroll
                for (int g_ = 0; g_ < GP; ++g_) {
                    const float part = kv_wave_sum(row_ok ? dfq[g_] : 0.0f);
                    if (lane == 0) dfq_s[(p * 4 + wave) * GP + g_] += part;
                }
            }
            if constexpr (RBF) {
                if (a.du && row_ok) {
                    float* durow = a.du + grow * a.ldu + (long long)g * a.I + hf * FPH + ci * IC;
#pragma unroll
                    for (int j = 0; j < FPH; ++j) durow[j] = duv[j];
                }
            }
            if (p == nshare - 1) {                // last group sharing these x columns: dx is complete
                if (row_ok) {
                    if constexpr (FPH % 4 == 0) {
#pragma unroll
                        for (int j4 = 0; j4 < FPH / 4; ++j4) {
                            const f32x4 v = {dxacc[4 * j4], dxacc[4 * j4 + 1], dxacc[4 * j4 + 2], dxacc[4 * j4 + 3]};
                            *reinterpret_cast<f32x4*>(dxrow + ci * IC + 4 * j4) = v;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < FPH; ++j) dxrow[ci * IC + j] = dxacc[j];
                    }
                }
            }
            if (!SHARED || p == nshare - 1) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[kt][r] = 0.0f;
            }
        }
        if (t + 1 < T) store_w((t + 1) & 1);
        __syncthreads();
        ci = cin; p = pn; cn = cnn;
    }
    if constexpr (SINE) {                         // combine the 4 waves in a fixed order (last loop barrier orders the adds)
        for (int j = tid; j < nshare * GP; j += 256) {
            const int pp = j / GP, gg = j - pp * GP;
            const float* src = dfq_s + (pp * 4) * GP + gg;
            const float v = ((src[0] + src[GP]) + src[2 * GP]) + src[3 * GP];
            a.dparam[((long long)blockIdx.y * a.groups + (pp * a.xmod + gx)) * a.G + gg] = v;
        }
    }
}

// =============================================================================================
// bf16 matrix-core variants (KANVIT_FLAG_BF16_MFMA; the bf16 configurations of BASELINE.json).
// Same producer/consumer pipeline and the same fp32 LDS basis tile as above; what changes is the
// contraction: the consumer gathers 8 consecutive k of its row from the K-major fp32 tile
// (8 ds_read_b32), rounds them to bf16 (v_cvt_pk_bf16_f32) and issues ONE v_mfma_f32_32x32x16_bf16
// where the exact path issues eight v_mfma_f32_32x32x2_f32 (32 vs 512 cycles per 16 k).  The
// weights are repacked once per call (kan_pack_w_fwd_kernel) into bf16 in exactly the LDS image
// the B operand wants -- [chunk][k/8][n][8 k] -- so staging is a 16-byte copy and the B fragment
// a single ds_read_b128 with lane = column (conflict free).  Unlike the fp32 MFMA, the bf16 MFMA
// does overlap with the producers' VALU work (tools/coissue_probe.hip), so the pipeline finally
// hides the basis evaluation; the kernel becomes HBM / latency bound instead of matrix bound.
// =============================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned kv_pack_bf16(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};        // hipcc -O3: one v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}

// w[groups][K][O] fp32 -> wb[groups][nch][KCP/8][O][8] bf16 (k inside a chunk padded to KCP with zeros)
__global__ __launch_bounds__(256) void kan_pack_w_fwd_kernel(const float* __restrict__ w, unsigned short* __restrict__ wb,
                                                             int K, int O, int KC, int KCP, int nch, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one (g, c, kb, n) per thread
    if (e >= total) return;
    const int n = (int)(e % O);
    long long r = e / O;
    const int kb = (int)(r % (KCP / 8));
    r /= (KCP / 8);
    const int c = (int)(r % nch);
    const long long g = r / nch;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kk = kb * 8 + j, k = c * KC + kk;
        v[j] = (kk < KC && k < K) ? w[(g * K + k) * O + n] : 0.0f;
    }
    u32x4 out = {kv_pack_bf16(v[0], v[1]), kv_pack_bf16(v[2], v[3]), kv_pack_bf16(v[4], v[5]), kv_pack_bf16(v[6], v[7])};
    *reinterpret_cast<u32x4*>(wb + e * 8) = out;
}

// w[groups][K][O] fp32 -> wb2[groups][nci][O/8][KCT][8] bf16: element (nb, kk, e) = w[g][ci*KC + kk][nb*8 + e]
__global__ __launch_bounds__(256) void kan_pack_w_bwd_kernel(const float* __restrict__ w, unsigned short* __restrict__ wb2,
                                                             int K, int O, int KC, int KCT, int nci, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one (g, ci, nb, kk) per thread
    if (e >= total) return;
    const int kk = (int)(e % KCT);
    long long r = e / KCT;
    const int nb = (int)(r % (O / 8));
    r /= (O / 8);
    const int ci = (int)(r % nci);
    const long long g = r / nci;
    const int k = ci * KC + kk;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (kk < KC && k < K) ? w[(g * K + k) * O + nb * 8 + j] : 0.0f;
    u32x4 out = {kv_pack_bf16(v[0], v[1]), kv_pack_bf16(v[2], v[3]), kv_pack_bf16(v[4], v[5]), kv_pack_bf16(v[6], v[7])};
    *reinterpret_cast<u32x4*>(wb2 + e * 8) = out;
}

// Requirements (host-checked): IC is a power of two >= 8 dividing I; O % (32*NT) == 0; tile-local offsets fit 32 bits.
template <int FAM, int NT, int NSH>
__global__ __launch_bounds__(NTHR) void kan_fwd_bf16_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * NT;
    constexpr int WROW = NSH * BN;
    constexpr bool RBF = (FAM == KV_RBF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const bool consumer = wave < 4;
    const int pt = tid & (NPROD - 1);
    const int ntn = a.O / BN;
    const int gs = blockIdx.x / ntn;
    const int n0 = (blockIdx.x - gs * ntn) * BN;
    const int nsets = a.groups / NSH;
    const long long m0 = (long long)blockIdx.y * BM;
    const int IC = a.IC, GP = a.GP, ICP = IC | 1;
    const int KC = IC * GP, KCP = (KC + 15) & ~15;
    const int XS = BM * ICP, ASZ = KCP * AS;
    const int WSZ = KCP * WROW / 2;               // W_s buffer size in floats (bf16 elements / 2)
    const int nch = a.I / IC;
    const bool full_m = (m0 + BM <= a.M);
    const int mrem = full_m ? BM : (int)(a.M - m0);

    float* x_s = smem;                          // [2][XS]
    float* u_s = x_s + 2 * XS;                  // [2][XS]  (RBF)
    float* A_s = u_s + (RBF ? 2 * XS : 0);      // [2][ASZ]   fp32, K-major
    unsigned short* W_s = reinterpret_cast<unsigned short*>(A_s + 2 * ASZ);   // [2][KCP/8][WROW][8] bf16

    const BasisArgs b = make_basis(a, gs);
    const int xcol = (NSH == 1 ? gs % a.xmod : gs) * a.I;
    const int ldx = (int)a.ldx, ldu = (RBF && a.u) ? (int)a.ldu : (int)a.ldx;

    const int xlg = __builtin_ctz(IC);
    const int xl = pt & (IC - 1), xr0 = pt >> xlg, xrs = NPROD >> xlg;
    const float* xbase = a.x + m0 * a.ldx + xcol;
    const float* ubase = (RBF && a.u) ? a.u + m0 * a.ldu + (long long)gs * a.I : xbase;
    const int xoff = xr0 * ldx + xl, uoff = xr0 * ldu + xl;
    const int xsoff = xr0 * ICP + xl;
    const int gr = pt & (BM - 1), gl0 = pt >> 7;

    // Staging is split into "issue the global loads" and "store to LDS" so that a producer step can put
    // all its loads in flight, evaluate the basis while they travel, and only then wait for them.
    constexpr int XQ = 8;                          // x loads per thread and chunk (BM*IC/NPROD = IC/2 <= 8 for IC <= 16)
    constexpr int WQ = 8;                          // 16-byte W loads per thread, group and chunk (KCP/8*BN/NPROD)
    float xr[XQ], ur[XQ];
    u32x4 wr[NSH][WQ];
    const int xnp = BM / xrs;                      // passes: IC/2 (host guarantees <= XQ)
    const int nvec = (KCP / 8) * BN;               // 16-byte vectors per group and chunk (host guarantees <= WQ*NPROD)
    auto load_x = [&](int c) {
        const float* sx = xbase + c * IC;
        const float* su = ubase + c * IC;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const bool ok = (q < xnp) && (full_m || (xr0 + q * xrs < mrem));
            xr[q] = ok ? sx[xoff + q * xrs * ldx] : 0.0f;
            if (RBF) ur[q] = ok ? su[uoff + q * xrs * ldu] : 0.0f;
        }
    };
    auto store_x = [&](int buf) {
        float* dxs = x_s + buf * XS + xsoff;
        float* dus = u_s + buf * XS + xsoff;
#pragma unroll
        for (int q = 0; q < XQ; ++q)
            if (q < xnp) {
                dxs[q * xrs * ICP] = xr[q];
                if (RBF) dus[q * xrs * ICP] = ur[q];
            }
    };
    auto load_w = [&](int c) {
#pragma unroll
        for (int p = 0; p < NSH; ++p) {
            const int g = (NSH == 1) ? gs : p * nsets + gs;
            const unsigned short* src = a.wb + ((((long long)g * nch + c) * (KCP / 8)) * a.O + n0) * 8;   // uniform
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                const int v = pt + q * NPROD;
                const int kb = v / BN, n = v & (BN - 1);
                if (v < nvec) wr[p][q] = *reinterpret_cast<const u32x4*>(src + ((long long)kb * a.O + n) * 8);
            }
        }
    };
    auto store_w = [&](int buf) {
        unsigned short* dst = W_s + (size_t)buf * WSZ * 2;
#pragma unroll
        for (int p = 0; p < NSH; ++p)
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                const int v = pt + q * NPROD;
                const int kb = v / BN, n = v & (BN - 1);
                if (v < nvec) *reinterpret_cast<u32x4*>(dst + ((size_t)kb * WROW + p * BN + n) * 8) = wr[p][q];
            }
    };
    auto gen_a = [&](int c, int buf) {
        const float* xs = x_s + buf * XS + gr * ICP;
        const float* us = u_s + buf * XS + gr * ICP;
        float* As = A_s + buf * ASZ + gr;
        for (int il = gl0; il < IC; il += 2)
            basis_fwd<FAM>(b, xs[il], RBF ? us[il] : 0.0f, c * IC + il, As + (il * GP) * AS, AS);
    };

    f32x16 acc[NSH * NT];
#pragma unroll
    for (int t = 0; t < NSH * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    if (!consumer) {
        load_x(0);
        load_w(0);
        for (int r = KC * AS + pt; r < ASZ; r += NPROD) {          // zero the k-padding rows of both A buffers
            A_s[r] = 0.0f;
            A_s[ASZ + r] = 0.0f;
        }
        store_x(0);
        if (nch > 1) load_x(1);
        store_w(0);
    }
    __syncthreads();
    if (!consumer) {
        gen_a(0, 0);
        if (nch > 1) store_x(1);
    }
    __syncthreads();

    for (int c = 0; c < nch; ++c) {
        if (consumer) {
            const float* ap = A_s + (c & 1) * ASZ + (8 * hf) * AS + wave * 32 + l31;
            const unsigned short* wp = W_s + (size_t)(c & 1) * WSZ * 2 + ((size_t)hf * WROW + l31) * 8;
            for (int ks = 0; ks < KCP / 16; ++ks) {
                float af[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) af[j] = ap[(16 * ks + j) * AS];
                const u32x4 au = {kv_pack_bf16(af[0], af[1]), kv_pack_bf16(af[2], af[3]), kv_pack_bf16(af[4], af[5]),
                                  kv_pack_bf16(af[6], af[7])};
                const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, au);
#pragma unroll
                for (int t = 0; t < NSH * NT; ++t) {
                    const bf16x8_t b8 = *reinterpret_cast<const bf16x8_t*>(wp + ((size_t)(2 * ks) * WROW + t * 32) * 8);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[t], 0, 0, 0);
                }
            }
        } else if (!consumer && c + 1 < nch) {
            load_w(c + 1);                 // loads in flight ...
            if (c + 2 < nch) load_x(c + 2);
            gen_a(c + 1, (c + 1) & 1);     // ... while the basis is evaluated ...
            store_w((c + 1) & 1);          // ... and only then waited for
            if (c + 2 < nch) store_x(c & 1);
        }
        __syncthreads();
    }

    // Epilogue: the accumulator layout gives each lane one dword per store (2 x 128 B per wave-instruction, 96
    // instructions per lane, consumer waves only) -- store-issue bound (0.12 of 0.33 ms in the bf16 kernel).
    // Instead park the tile in the idle operand buffers and let all 8 waves write float4 rows.
    {
        constexpr int OS = WROW + 4;                 // row stride: 16-byte aligned, lane = column -> conflict free
        float* O_s = A_s;
        if (consumer) {
#pragma unroll
            for (int t = 0; t < NSH * NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) O_s[(wave * 32 + kv_acc_row(r, hf)) * OS + t * 32 + l31] = acc[t][r];
        }
        __syncthreads();
        constexpr int V4R = WROW / 4;                // float4 per tile row
        for (int v = tid; v < BM * V4R; v += NTHR) {
            const int row = v / V4R, c4 = (v - row * V4R) * 4;
            const int p = c4 / BN, cl = c4 - p * BN;
            if (row < mrem) {
                const int g = (NSH == 1) ? gs : p * nsets + gs;
                f32x4 val = *reinterpret_cast<const f32x4*>(O_s + row * OS + c4);
                if (a.bias) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + (long long)g * a.O + n0 + cl);
                    val += bv;
                }
                *reinterpret_cast<f32x4*>(a.y + (m0 + row) * a.ldy + (long long)g * a.O + n0 + cl) = val;
            }
        }
    }
}

// =============================================================================================
// forward, register-operand form on the bf16 matrix cores (KANVIT_FLAG_BF16_MFMA).  Same idea as kan_fwd_reg_kernel: lane
// half hf owns ICH whole features per chunk and generates their ICH*GP basis values in order; eight consecutive values are
// rounded to bf16 and form the A fragment of one v_mfma_f32_32x32x16_bf16.  The weights are repacked once per call
// (kan_pack_w_fwd_reg_kernel) into the matching image [chunk][k-step][half][n][8 values], so the B fragment is a single
// ds_read_b128 with lane = column.  GP is a template parameter so the (feature, basis) bookkeeping unrolls statically.
// =============================================================================================
// w[groups][K][O] fp32 -> wb[groups][nch][VSTEPS][2][O][8] bf16; value v = 8*ks + e of half h is (j = v / GP, g = v % GP),
// i.e. natural k = (c*IC + h*ICH + j)*GP + g; v >= ICH*GP pads with zeros.
__global__ __launch_bounds__(256) void kan_pack_w_fwd_reg_kernel(const float* __restrict__ w, unsigned short* __restrict__ wb,
                                                                 int K, int O, int GP, int ICH, int vsteps, int nch, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one (g, c, ks, h, n) per thread
    if (e >= total) return;
    const int n = (int)(e % O);
    long long r = e / O;
    const int h = (int)(r & 1);
    r >>= 1;
    const int ks = (int)(r % vsteps);
    r /= vsteps;
    const int c = (int)(r % nch);
    const long long g = r / nch;
    float v[8];
#pragma unroll
    for (int j8 = 0; j8 < 8; ++j8) {
        const int vi = ks * 8 + j8, j = vi / GP, gg = vi - j * GP;
        const int k = (c * 2 * ICH + h * ICH + j) * GP + gg;
        v[j8] = (j < ICH && k < K) ? w[(g * K + k) * O + n] : 0.0f;
    }
    u32x4 out = {kv_pack_bf16(v[0], v[1]), kv_pack_bf16(v[2], v[3]), kv_pack_bf16(v[4], v[5]), kv_pack_bf16(v[6], v[7])};
    *reinterpret_cast<u32x4*>(wb + e * 8) = out;
}

template <int FAM, int GP, int NT, int NSH, int ICH>
__global__ __launch_bounds__(256) void kan_fwd_reg_bf16_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * NT;
    constexpr int WROW = NSH * BN;
    constexpr int IC = 2 * ICH;
    constexpr int VH = ICH * GP;                  // values generated per lane and chunk
    constexpr int VS = (VH + 7) / 8;              // MFMA k-steps per chunk
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr int TS = 36;
    constexpr int NV = VS * 2 * BN;               // 16-byte vectors per group and chunk
    constexpr int WQ = (NV + 255) / 256;
    constexpr int WSZ = VS * 2 * WROW * 8;        // bf16 elements per W buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int ntn = a.O / BN;
    const int gs = blockIdx.x / ntn;
    const int n0 = (blockIdx.x - gs * ntn) * BN;
    const int nsets = a.groups / NSH;
    const long long m0 = (long long)blockIdx.y * BM;
    const int nch = a.I / IC;
    const int mrem = (m0 + BM <= a.M) ? BM : (int)(a.M - m0);
    unsigned short* W_s = reinterpret_cast<unsigned short*>(smem);     // [2][VS][2][WROW][8]

    const BasisArgs b = make_basis(a, gs);
    const int xcol = (NSH == 1 ? gs % a.xmod : gs) * a.I;
    const int row = wave * 32 + l31;
    const bool row_ok = row < mrem;
    const float* xrow = a.x + (m0 + (row_ok ? row : 0)) * a.ldx + xcol + hf * ICH;
    const float* urow = (RBF && a.u) ? a.u + (m0 + (row_ok ? row : 0)) * a.ldu + (long long)gs * a.I + hf * ICH : xrow;
    float ln_mean = 0.0f, ln_rstd = 1.0f;        // KANVIT_FLAG_FUSED_LN (see kan_fwd_reg_kernel)
    const float* ln_gb = nullptr;
    if constexpr (RBF) {
        if (a.ln) {
            kv_ln_row_stats<ICH>(xrow, nch, a.I, a.ln_eps, ln_mean, ln_rstd);
            ln_gb = b.bp + a.G + hf * ICH;
            if (hf == 0 && row_ok && gs < a.xmod) {
                float2 st = {ln_mean, ln_rstd};
                *reinterpret_cast<float2*>(a.stats + ((m0 + row) * a.xmod + gs) * 2) = st;
            }
        }
    }

    u32x4 wreg[NSH][WQ];
    auto load_w = [&](int c) {
#pragma unroll
        for (int p = 0; p < NSH; ++p) {
            const int g = (NSH == 1) ? gs : p * nsets + gs;
            const unsigned short* src = a.wb + (((long long)g * nch + c) * (VS * 2)) * a.O * 8;     // [VS*2][O][8]
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                const int v = tid + q * 256;
                const int kr = v / BN, n = v & (BN - 1);          // kr = ks*2 + h
                if (v < NV) wreg[p][q] = *reinterpret_cast<const u32x4*>(src + ((long long)kr * a.O + n0 + n) * 8);
            }
        }
    };
    auto store_w = [&](int buf) {
        unsigned short* dst = W_s + (size_t)buf * WSZ;
#pragma unroll
        for (int p = 0; p < NSH; ++p)
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                const int v = tid + q * 256;
                const int kr = v / BN, n = v & (BN - 1);
                if (v < NV) *reinterpret_cast<u32x4*>(dst + ((size_t)kr * WROW + p * BN + n) * 8) = wreg[p][q];
            }
    };

    f32x16 acc[NSH * NT];
#pragma unroll
    for (int t = 0; t < NSH * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    float xv[ICH], uv[ICH];
    auto load_x = [&](int c) {
        if constexpr (ICH % 4 == 0) {
#pragma unroll
            for (int j4 = 0; j4 < ICH / 4; ++j4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + c * IC + 4 * j4);
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[4 * j4 + e] = v[e];
                if (RBF && !a.ln) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(urow + c * IC + 4 * j4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) uv[4 * j4 + e] = w4[e];
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < ICH; ++e) {
                xv[e] = xrow[c * IC + e];
                if (RBF && !a.ln) uv[e] = urow[c * IC + e];
            }
        }
        if constexpr (RBF) {
            if (a.ln) {
#pragma unroll
                for (int e = 0; e < ICH; ++e) uv[e] = (xv[e] - ln_mean) * ln_rstd * ln_gb[c * IC + e] + ln_gb[a.I + c * IC + e];
            }
        }
    };

    load_w(0);
    load_x(0);
    store_w(0);
    __syncthreads();

    for (int c = 0; c < nch; ++c) {
        float xc[ICH], uc[ICH];
#pragma unroll
        for (int e = 0; e < ICH; ++e) {
            xc[e] = xv[e];
            uc[e] = RBF ? uv[e] : 0.0f;
        }
        if (c + 1 < nch) {
            load_w(c + 1);
            load_x(c + 1);
        }
        const unsigned short* wp = W_s + (size_t)(c & 1) * WSZ + ((size_t)hf * WROW + l31) * 8;
        BasisGen<FAM> gen;
#pragma unroll
        for (int ks = 0; ks < VS; ++ks) {
            float av[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                constexpr int dummy = 0;
                (void)dummy;
                const int vi = ks * 8 + e;                        // compile-time after unrolling
                const int j = vi / GP, g = vi - j * GP;
                if (vi < VH) {
                    if (g == 0) gen.init(b, xc[j], uc[j], c * IC + hf * ICH + j);
                    av[e] = gen.next(g);
                } else {
                    av[e] = 0.0f;
                }
            }
            const u32x4 au = {kv_pack_bf16(av[0], av[1]), kv_pack_bf16(av[2], av[3]), kv_pack_bf16(av[4], av[5]),
                              kv_pack_bf16(av[6], av[7])};
            const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, au);
#pragma unroll
            for (int t = 0; t < NSH * NT; ++t) {
                const bf16x8_t b8 = *reinterpret_cast<const bf16x8_t*>(wp + ((size_t)(2 * ks) * WROW + t * 32) * 8);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[t], 0, 0, 0);
            }
        }
        if (c + 1 < nch) store_w((c + 1) & 1);
        __syncthreads();
    }

    float* T_w = smem + wave * 32 * TS;
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int t = 0; t < NSH * NT; ++t) {
        const int p = t / NT, nt = t - p * NT;
        const int g = (NSH == 1) ? gs : p * nsets + gs;
#pragma unroll
        for (int r = 0; r < 16; ++r) T_w[kv_acc_row(r, hf) * TS + l31] = acc[t][r];
        f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + (long long)g * a.O + n0 + nt * 32 + ec);
        float* yt = a.y + (m0 + wave * 32) * a.ldy + (long long)g * a.O + n0 + nt * 32 + ec;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = er + q * 8;
            if (wave * 32 + rr < mrem) {
                f32x4 v = *reinterpret_cast<const f32x4*>(T_w + rr * TS + ec);
                v += bv;
                *reinterpret_cast<f32x4*>(yt + (long long)rr * a.ldy) = v;
            }
        }
    }
}

// =============================================================================================
// forward, bf16, W-STATIONARY persistent form (when the whole bf16 image of the NSH groups' weights fits the LDS: the
// per-head q|k|v launches).  The HBM roofline form of kan_fwd_reg_bf16_kernel:
//   * a work-group loads its [all chunks][VS][2][NSH*BN][8] weight image into LDS ONCE and then walks row tiles
//     (grid.y work-groups per column set, stride grid.y) -- no per-tile weight staging, no barrier after the first;
//   * the product is flipped, Y^T = W^T . Phi^T: the weight fragment is the A operand (lane = output column: the same
//     ds_read_b128 as before), the generated basis values are the B operand (lane = token).  The accumulator then holds
//     4 consecutive output columns of the lane's OWN row per register quad -> float4 stores straight from registers,
//     no staging tile;
//   * x of the next (tile, chunk) is prefetched while the current chunk is contracted; 8 waves (2 per SIMD) of 32 rows.
// =============================================================================================
constexpr int KV_WS_THREADS = 512;   // 8 waves (12 measured slower: 66 row tiles over 21 work-groups per head quantise to 79 %)
template <int FAM, int GP, int NT, int NSH, int ICH, int NCH>
__global__ __launch_bounds__(KV_WS_THREADS, KV_WS_THREADS / 256) void kan_fwd_ws_bf16_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * NT;
    constexpr int WROW = NSH * BN;
    constexpr int IC = 2 * ICH;
    constexpr int VH = ICH * GP;
    constexpr int VS = (VH + 7) / 8;
    constexpr int NK = NCH * VS;                  // MFMA k-steps over the whole K
    constexpr int NTT = NSH * NT;                 // column tiles per row
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr int ROWS = KV_WS_THREADS / 2;       // rows per work-group iteration (32 per wave)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int ntn = a.O / BN;
    const int gs = blockIdx.x / ntn;
    const int n0 = (blockIdx.x - gs * ntn) * BN;
    const int nsets = a.groups / NSH;
    unsigned short* W_s = reinterpret_cast<unsigned short*>(smem);     // [NK][2][WROW][8]
    float* bias_s = reinterpret_cast<float*>(W_s + (size_t)NK * 2 * WROW * 8);   // [WROW] (zeros without a bias)

    // ---- stage the whole weight image (and the bias of this column set) once
    {
        constexpr int nvec_g = NK * 2 * BN;       // 16-byte vectors per group
        for (int p = 0; p < NSH; ++p) {
            const int g = (NSH == 1) ? gs : p * nsets + gs;
            const unsigned short* src = a.wb + ((long long)g * (NK * 2)) * a.O * 8;     // [NK*2][O][8]
            for (int v = tid; v < nvec_g; v += KV_WS_THREADS) {
                const int kr = v / BN, n = v - kr * BN;           // kr = (c*VS + ks)*2 + h
                const u32x4 t = *reinterpret_cast<const u32x4*>(src + ((long long)kr * a.O + n0 + n) * 8);
                *reinterpret_cast<u32x4*>(W_s + ((size_t)kr * WROW + p * BN + n) * 8) = t;
            }
            if (tid < BN) bias_s[p * BN + tid] = a.bias ? a.bias[(long long)g * a.O + n0 + tid] : 0.0f;
        }
    }
    __syncthreads();

    const BasisArgs b = make_basis(a, gs);
    const int xcol = (NSH == 1 ? gs % a.xmod : gs) * a.I;
    const long long ntiles = (a.M + ROWS - 1) / ROWS;
    const bool has_bias = a.bias != nullptr;

    // Per row tile: (1) the basis fragments of the whole K are generated once into registers (NK bf16x8 values);
    // (2) x of the NEXT tile is requested into the now dead x registers -- a full tile of MFMA work to land; (3) the column
    // tiles are contracted TWO at a time over the whole K (weight fragments double-buffered in registers, one k-step ahead)
    // and stored as soon as their pair is done, so stores drain under the MFMAs of the next pair.
    float xcur[NCH][ICH], ucur[RBF ? NCH : 1][RBF ? ICH : 1];
    auto load_tile_x = [&](long long tile) {
        long long r = tile * ROWS + wave * 32 + l31;
        if (r > a.M - 1) r = a.M - 1;
        const float* xrow = a.x + r * a.ldx + xcol + hf * ICH;
        const float* urow = (RBF && a.u) ? a.u + r * a.ldu + (long long)gs * a.I + hf * ICH : xrow;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if constexpr (ICH % 4 == 0) {
#pragma unroll
                for (int j4 = 0; j4 < ICH / 4; ++j4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + c * IC + 4 * j4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xcur[c][4 * j4 + e] = v[e];
                    if constexpr (RBF) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(urow + c * IC + 4 * j4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) ucur[c][4 * j4 + e] = w4[e];
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < ICH; ++e) {
                    xcur[c][e] = xrow[c * IC + e];
                    if constexpr (RBF) ucur[c][e] = urow[c * IC + e];
                }
            }
        }
    };

    // The row-tile body exists twice: FULL tiles store unconditionally (a store under a branch makes the compiler's
    // waitcnt bookkeeping pessimistic: later waits become vmcnt(0) and also wait for every store to retire).
    auto do_tile = [&](long long tile, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        bf16x8_t phi[NK];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            BasisGen<FAM> gen;
#pragma unroll
            for (int ks = 0; ks < VS; ++ks) {
                float av[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int vi = ks * 8 + e;                        // compile-time after unrolling
                    const int j = vi / GP, g = vi - j * GP;
                    if (vi < VH) {
                        if (g == 0) gen.init(b, xcur[c][j], RBF ? ucur[RBF ? c : 0][RBF ? j : 0] : 0.0f, c * IC + hf * ICH + j);
                        av[e] = gen.next(g);
                    } else {
                        av[e] = 0.0f;
                    }
                }
                const u32x4 au = {kv_pack_bf16(av[0], av[1]), kv_pack_bf16(av[2], av[3]), kv_pack_bf16(av[4], av[5]),
                                  kv_pack_bf16(av[6], av[7])};
                phi[c * VS + ks] = __builtin_bit_cast(bf16x8_t, au);
            }
        }
        if (tile + gridDim.y < ntiles) load_tile_x(tile + gridDim.y);

        const long long r = tile * ROWS + wave * 32 + l31;
        const bool st_ok = FULL || (r < a.M);
        const unsigned short* wp = W_s + ((size_t)hf * WROW + l31) * 8;
#pragma unroll 1
        for (int t0 = 0; t0 < NTT; t0 += 2) {     // a real loop: unrolled, the scheduler hoists every ds_read of every pair and spills
            constexpr bool PAIR = true;
            const bool two = t0 + 1 < NTT;
            f32x16 acc0, acc1;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0[q] = 0.0f;
                acc1[q] = 0.0f;
            }
            bf16x8_t w0[2], w1[2];
            w0[0] = *reinterpret_cast<const bf16x8_t*>(wp + ((size_t)0 * WROW + t0 * 32) * 8);
            w1[0] = *reinterpret_cast<const bf16x8_t*>(wp + ((size_t)0 * WROW + (two ? t0 + 1 : t0) * 32) * 8);
#pragma unroll
            for (int s2 = 0; s2 < NK; ++s2) {
                if (s2 + 1 < NK) {
                    w0[(s2 + 1) & 1] = *reinterpret_cast<const bf16x8_t*>(wp + ((size_t)(2 * (s2 + 1)) * WROW + t0 * 32) * 8);
                    w1[(s2 + 1) & 1] = *reinterpret_cast<const bf16x8_t*>(wp + ((size_t)(2 * (s2 + 1)) * WROW + (two ? t0 + 1 : t0) * 32) * 8);
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[s2 & 1], phi[s2], acc0, 0, 0, 0);     // Y^T tile: rows = columns of y
                if (PAIR && two) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[s2 & 1], phi[s2], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);            // keep the one-step-ahead fragment prefetch, no further hoisting
            }
            // accumulator registers 4q..4q+3 of a tile are y[row][.. + 8q + 4hf + 0..3] of this lane's row
            if (st_ok) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = t0 + tt;
                    if (t < NTT) {
                        const int p = t / NT, nt = t - p * NT;
                        const int g = (NSH == 1) ? gs : p * nsets + gs;
                        float* yp = a.y + r * a.ldy + (long long)g * a.O + n0 + nt * 32 + 4 * hf;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v = tt == 0 ? f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]}
                                              : f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
                            if (has_bias) v += *reinterpret_cast<const f32x4*>(bias_s + t * 32 + 8 * q + 4 * hf);
                            *reinterpret_cast<f32x4*>(yp + 8 * q) = v;
                        }
                    }
                }
            }
        }
    };
    const long long nfull = a.M / ROWS;           // tiles with all 256 rows present
    long long tile = blockIdx.y;
    if (tile < ntiles) load_tile_x(tile);
    for (; tile < nfull; tile += gridDim.y) do_tile(tile, std::true_type{});
    if (tile < ntiles) do_tile(tile, std::false_type{});
}

// =============================================================================================
// backward w.r.t. the input, register form on the bf16 matrix cores.  Same slot <-> (feature, basis) permutation and
// in-register chain rule as kan_bwd_input_reg_kernel; the contraction over the dY columns uses v_mfma_f32_32x32x16_bf16:
// k-step ks covers 16 columns, lane half h the 8 columns 16*ks + 8*h .. +7 -- the B fragment is 8 consecutive dY values
// of the lane's own row (registers, rounded to bf16), the A fragment one ds_read_b128 from the repacked image
// [chunk][ks][h][k row][8 n] (kan_pack_w_bwd_reg_kernel).  One step = all O columns of one group: T = nci * nshare steps.
// Requirements: as the fp32 register kernel, plus O in {32, 64}.
// =============================================================================================
// wb2[g][nci][O/16][2][KCT][8]: element (ks, h, kr, e) = w[g][k(kr)][16*ks + 8*h + e], k(kr) by the slot permutation
// (ldw, gstride): row stride of w and the offset between two "groups" -- (O, K*O) for real groups, (O_real, 64) when the groups
// are the 64-column chunks of one wide layer
__global__ __launch_bounds__(256) void kan_pack_w_bwd_reg_kernel(const float* __restrict__ w, unsigned short* __restrict__ wb2,
                                                                 int K, int O, int GP, int FPH, int KCT, int nci, long long total,
                                                                 long long ldw, long long gstride) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one (g, ci, ks, h, kr) per thread
    if (e >= total) return;
    const int kr = (int)(e % KCT);
    long long r = e / KCT;
    const int h = (int)(r & 1);
    r >>= 1;
    const int ks = (int)(r % (O / 16));
    r /= (O / 16);
    const int ci = (int)(r % nci);
    const long long g = r / nci;
    const int rho = kr & 31, kt = kr >> 5;
    const int h_ = (rho >> 2) & 1, r_ = (rho & 3) + 4 * (rho >> 3);
    const int slot = kt * 16 + r_;
    const int jq = slot / GP, g_ = slot - jq * GP;
    const int k = (ci * 2 * FPH + h_ * FPH + jq) * GP + g_;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (jq < FPH && k < K) ? w[g * gstride + (long long)k * ldw + 16 * ks + 8 * h + j] : 0.0f;
    u32x4 out = {kv_pack_bf16(v[0], v[1]), kv_pack_bf16(v[2], v[3]), kv_pack_bf16(v[4], v[5]), kv_pack_bf16(v[6], v[7])};
    *reinterpret_cast<u32x4*>(wb2 + e * 8) = out;
}

template <int FAM, int GP, int KT, bool SHARED>
__global__ __launch_bounds__(256, FAM == KV_SINE ? 1 : 2) void kan_bwd_input_reg_bf16_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KCT = 32 * KT;
    constexpr int FPH = (16 * KT) / GP;
    constexpr int IC = 2 * FPH;
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr bool SINE = (FAM == KV_SINE);
    constexpr int MAXKS = 4;                      // O <= 64
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int gx = blockIdx.x;
    const long long m0 = (long long)blockIdx.y * BM;
    const int nshare = a.groups / a.xmod;
    const int nci = a.I / IC, nks = a.O / 16;
    const int mrem = (m0 + BM <= a.M) ? BM : (int)(a.M - m0);
    const int row = wave * 32 + l31;
    const bool row_ok = row < mrem;
    const long long grow = m0 + (row_ok ? row : 0);
    unsigned short* W_s = reinterpret_cast<unsigned short*>(smem);     // [2][nks][2][KCT][8]
    const int WSZ = nks * 2 * KCT * 8;            // bf16 elements per buffer
    const int NV = nks * 2 * KCT;                 // 16-byte vectors per buffer
    // SINE: [real groups sharing x][4 waves][GP] partial d loss / d freq of this row tile (as the fp32 register kernel)
    const int ndf = a.vcols ? 1 : nshare;
    float* dfq_s = reinterpret_cast<float*>(W_s + 2 * (size_t)WSZ);
    if constexpr (SINE) {
        for (int j = tid; j < ndf * 4 * GP; j += 256) dfq_s[j] = 0.0f;
    }

    const float* xrow = a.x + grow * a.ldx + (long long)gx * a.I + hf * FPH;
    float* dxrow = a.dx + grow * a.ldx + (long long)gx * a.I + hf * FPH;
    const float* dyrow = a.dy + grow * a.ldy + hf * 8;

    constexpr int WQ = (MAXKS * 2 * KCT + 255) / 256;
    u32x4 wreg[WQ];
    auto load_w = [&](int ci, int g) {
        const unsigned short* src = a.wb2 + (((long long)g * nci + ci) * NV) * 8;
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int v = tid + q * 256;
            if (v < NV) wreg[q] = *reinterpret_cast<const u32x4*>(src + (long long)v * 8);
        }
    };
    auto store_w = [&](int buf) {
        unsigned short* dst = W_s + (size_t)buf * WSZ;
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int v = tid + q * 256;
            if (v < NV) *reinterpret_cast<u32x4*>(dst + (size_t)v * 8) = wreg[q];
        }
    };
    f32x4 dyreg[MAXKS][2];
    auto load_dy = [&](int g) {
        const float* src = dyrow + (long long)g * a.O;
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks)
            if (ks < nks) {
                dyreg[ks][0] = *reinterpret_cast<const f32x4*>(src + 16 * ks);
                dyreg[ks][1] = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4);
            }
    };

    const int T = nci * nshare;
    f32x16 acc[KT];
    float dxacc[FPH];
    float xv[FPH];

    int ci = 0, p = 0;
    load_w(0, gx);
    load_dy(gx);
    store_w(0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (p == 0) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kt][r] = 0.0f;
#pragma unroll
            for (int j = 0; j < FPH; ++j) dxacc[j] = 0.0f;
            if constexpr (FPH % 4 == 0) {
#pragma unroll
                for (int j4 = 0; j4 < FPH / 4; ++j4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + ci * IC + 4 * j4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[4 * j4 + e] = v[e];
                }
            } else {
#pragma unroll
                for (int j = 0; j < FPH; ++j) xv[j] = xrow[ci * IC + j];
            }
        }
        bf16x8_t dyb[MAXKS];
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            const u32x4 u = {kv_pack_bf16(dyreg[ks][0][0], dyreg[ks][0][1]), kv_pack_bf16(dyreg[ks][0][2], dyreg[ks][0][3]),
                             kv_pack_bf16(dyreg[ks][1][0], dyreg[ks][1][1]), kv_pack_bf16(dyreg[ks][1][2], dyreg[ks][1][3])};
            dyb[ks] = __builtin_bit_cast(bf16x8_t, u);
        }
        int cin = ci, pn = p + 1;
        if (pn == nshare) { pn = 0; ++cin; }
        if (t + 1 < T) {
            load_w(cin, pn * a.xmod + gx);
            load_dy(pn * a.xmod + gx);
        }
        const unsigned short* wp = W_s + (size_t)(t & 1) * WSZ + ((size_t)hf * KCT + l31) * 8;
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks)
            if (ks < nks) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const bf16x8_t a8 = *reinterpret_cast<const bf16x8_t*>(wp + ((size_t)(2 * ks) * KCT + kt * 32) * 8);
                    acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, dyb[ks], acc[kt], 0, 0, 0);
                }
            }
        const bool ends = SHARED ? (p == nshare - 1) : true;
        if (ends) {
            const int g = a.vcols ? gx : p * a.xmod + gx;      // column chunks of one wide layer: ONE real group
            const BasisArgs b = make_basis(a, g);
            float duv[RBF ? FPH : 1];
            float uvv[RBF ? FPH : 1];
            if constexpr (RBF) {
                if (a.ln) {                       // KANVIT_FLAG_FUSED_LN: u from x, the saved row statistics and this group's gamma / beta
                    const float2 st = *reinterpret_cast<const float2*>(a.stats + (grow * a.xmod + gx) * 2);
                    const float* gb = b.bp + a.G + ci * IC + hf * FPH;
#pragma unroll
                    for (int j = 0; j < FPH; ++j) uvv[j] = (xv[j] - st.x) * st.y * gb[j] + gb[a.I + j];
                } else {
                    const float* urow = a.u ? a.u + grow * a.ldu + (long long)g * a.I + hf * FPH + ci * IC : xrow + ci * IC;
#pragma unroll
                    for (int j = 0; j < FPH; ++j) uvv[j] = urow[j];
                }
            }
            float dfq[SINE ? GP : 1];
            if constexpr (SINE) {
#pragma unroll
                for (int g_ = 0; g_ < GP; ++g_) dfq[g_] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < FPH; ++j) {
                BasisDGen<FAM> gen;
                gen.init(b, xv[j], RBF ? uvv[j] : 0.0f, ci * IC + hf * FPH + j);
                float dsum = 0.0f, usum = 0.0f;
#pragma unroll
                for (int g_ = 0; g_ < GP; ++g_) {
                    const int slot = j * GP + g_;
                    const float d = gen.next(g_);
                    const float v = acc[slot / 16][slot % 16];
                    if (RBF && g_ < GP - 1) usum += v * d;
                    else dsum += v * d;
                    if constexpr (SINE) dfq[g_] += v * gen.lastc * xv[j];
                }
                dxacc[j] += dsum;
                if constexpr (RBF) duv[j] = usum;
            }
            if constexpr (SINE) {                 // one wave reduction per grid point and step; rows past M contribute nothing
                const int pg = a.vcols ? 0 : p;
#pragma unroll
                for (int g_ = 0; g_ < GP; ++g_) {
                    const float part = kv_wave_sum(row_ok ? dfq[g_] : 0.0f);
                    if (lane == 0) dfq_s[(pg * 4 + wave) * GP + g_] += part;
                }
            }
            if constexpr (RBF) {
                if (a.du && row_ok) {
                    float* durow = a.du + grow * a.ldu + (long long)g * a.I + hf * FPH + ci * IC;
#pragma unroll
                    for (int j = 0; j < FPH; ++j) durow[j] = duv[j];
                }
            }
            if (p == nshare - 1 && row_ok) {
                if constexpr (FPH % 4 == 0) {
#pragma unroll
                    for (int j4 = 0; j4 < FPH / 4; ++j4) {
                        const f32x4 v = {dxacc[4 * j4], dxacc[4 * j4 + 1], dxacc[4 * j4 + 2], dxacc[4 * j4 + 3]};
                        *reinterpret_cast<f32x4*>(dxrow + ci * IC + 4 * j4) = v;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < FPH; ++j) dxrow[ci * IC + j] = dxacc[j];
                }
            }
            if (!SHARED || p == nshare - 1) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[kt][r] = 0.0f;
            }
        }
        if (t + 1 < T) store_w((t + 1) & 1);
        __syncthreads();
        ci = cin; p = pn;
    }
    if constexpr (SINE) {                         // combine the 4 waves in a fixed order (the last loop barrier orders the adds)
        const int rgroups = a.vcols ? 1 : a.groups;
        for (int j = tid; j < ndf * GP; j += 256) {
            const int pp = j / GP, gg = j - pp * GP;
            const float* src = dfq_s + (pp * 4) * GP + gg;
            const float v = ((src[0] + src[GP]) + src[2 * GP]) + src[3 * GP];
            a.dparam[((long long)blockIdx.y * rgroups + (a.vcols ? 0 : pp * a.xmod + gx)) * a.G + gg] = v;
        }
    }
}

// =============================================================================================
// backward w.r.t. the input.  grid (xmod, ceil(M/BM)), 512 threads.
// Steps t = (feature chunk ci, sharing group p, dY column chunk cn), cn fastest.  Per step the
// consumers contract dY[:, cn] with W^T into dPhi accumulators; when a contraction ends they park
// the tile in dA_s (between two barriers).  SHARED (families without basis parameters): the
// nshare groups that read the same x columns also share phi', so their dPhi tiles are summed in
// the accumulators (one contraction over (p, n) per feature chunk) and the chain rule runs once.
// Producers run one step ahead on the operands and one step behind on the chain rule:
//   iteration t: operands of step t+1 -> ops[(t+1)&1]; x chunk ci at its first step;
//                basis_bwd of the tile parked at the end of iteration t-1;
//                du / dx write-out one iteration after the basis_bwd that produced them.
// =============================================================================================
// BF (KANVIT_FLAG_BF16_MFMA, O in {16, 32, 64}): a step covers ALL dY columns of one group; the operands live in LDS as
// bf16 in MFMA-ready images -- dY rows [row][O+8] converted while staging (A fragment = one ds_read_b128), W^T from the
// pre-packed [O/8][KCT][8] image (B fragment = one ds_read_b128, lane = k) -- and the contraction runs on
// v_mfma_f32_32x32x16_bf16.  Half the barriers of the fp32 schedule, no conversions in the consumer.
template <int FAM, int KT, bool SHARED, bool BF>
__global__ __launch_bounds__(NTHR) void kan_bwd_input_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KCT = 32 * KT;
    constexpr int WS = KCT + 1;
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr bool SINE = (FAM == KV_SINE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const bool consumer = wave < 4;
    const int pt = tid & (NPROD - 1);
    const int pw = wave & 3;
    const int gx = blockIdx.x;
    const long long m0 = (long long)blockIdx.y * BM;
    const int nshare = a.groups / a.xmod;
    const int IC = a.IC, GP = a.GP, ICP = IC | 1;
    const int KC = IC * GP;
    const int XS = BM * ICP;
    const int nci = (a.I + IC - 1) / IC;
    const int ncn = BF ? 1 : (a.O + BIN_NC - 1) / BIN_NC;
    const int spc = nshare * ncn;                   // steps per feature chunk
    const int T = nci * spc;
    const bool full_m = (m0 + BM <= a.M);
    const int mrem = full_m ? BM : (int)(a.M - m0);
    const bool vec_n = ((a.O & 3) == 0) && (a.O % BIN_NC == 0) && ((a.ldy & 3) == 0);   // float4 operand loads
    const int OP = a.O + 8;                         // BF: bf16 elements per dY row image (16-byte aligned, odd 16-B slot count)
    // one operand buffer, in floats: fp32 path dY_s then Wt_s; BF path dYb[BM][OP] then Wtb[O/8][KCT][8] (bf16)
    const int OPS = BF ? (BM * OP / 2 + (a.O / 8) * KCT * 4) : (BIN_NC * AS + BIN_NC * WS);

    float* x_s = smem;                              // [2][XS]  by ci parity
    float* dx_s = x_s + 2 * XS;                     // [2][XS]
    float* u_s = dx_s + 2 * XS;                     // [2][XS]  by (ci*nshare+p) parity  (RBF)
    float* du_s = u_s + (RBF ? 2 * XS : 0);         // [2][XS]                            (RBF)
    float* dfq_s = du_s + (RBF ? 2 * XS : 0);       // [nshare][4][G]                     (SINE)
    float* dA_s = dfq_s + (SINE ? nshare * 4 * a.G : 0);   // [KCT][AS]
    float* ops = dA_s + KCT * AS;                   // [2][OPS]

    const float* xg = a.x + (long long)gx * a.I;
    float* dxg = a.dx + (long long)gx * a.I;
    const int ldy = (int)a.ldy;
    const float* dyb = a.dy + m0 * a.ldy;           // uniform: this tile's dY rows

    // ---- producer tasks ----
    auto stage_ops = [&](int t, int ci, int p, int cn) {
        const int g = p * a.xmod + gx, n0 = cn * BIN_NC, k0 = ci * KC;
        if constexpr (BF) {
            unsigned short* dYb = reinterpret_cast<unsigned short*>(ops + (t & 1) * OPS);
            unsigned short* Wtb = dYb + BM * OP;
            const float* dyt = dyb + (long long)g * a.O;                                         // uniform
            const int o4 = a.O >> 2, lg4 = __builtin_ctz(o4);                                    // float4 per row (4, 8 or 16)
            const int c4 = (pt & (o4 - 1)) * 4, r0 = pt >> lg4, rpp = NPROD >> lg4;              // rows per pass
            const int nps = a.O >> 3;                                                            // passes: BM / rpp
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = r0 + q * rpp;
                f32x4 tv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (q < nps && (full_m || r < mrem)) tv = *reinterpret_cast<const f32x4*>(dyt + r * ldy + c4);
                v[q] = tv;
            }
            const int nvw = (a.O >> 3) * KCT;                                                    // 16-byte vectors of the W image
            const unsigned short* wsrc = a.wb2 + (((long long)g * nci + ci) * nvw) * 8;          // uniform
            u32x4 wv[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int vi = pt + q * NPROD;
                if (vi < nvw) wv[q] = *reinterpret_cast<const u32x4*>(wsrc + (long long)vi * 8);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < nps) {
                    const unsigned lo = kv_pack_bf16(v[q][0], v[q][1]), hi = kv_pack_bf16(v[q][2], v[q][3]);
                    unsigned* d2 = reinterpret_cast<unsigned*>(dYb + (r0 + q * rpp) * OP + c4);
                    d2[0] = lo;
                    d2[1] = hi;
                }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int vi = pt + q * NPROD;
                if (vi < nvw) *reinterpret_cast<u32x4*>(Wtb + (size_t)vi * 8) = wv[q];
            }
            return;
        }
        float* dY_s = ops + (t & 1) * OPS;
        float* Wt_s = dY_s + BIN_NC * AS;
        const float* dyt = dyb + (long long)g * a.O + n0;                       // uniform
        const float* wt = a.w + ((long long)g * a.K + k0) * a.O + n0;           // uniform
        if (vec_n) {
            // dY tile [BM x 32]: thread (row r = pt >> 3 (+32 per pass), 4 columns c4 = (pt & 7) * 4)
            const int c4 = (pt & 7) * 4, r0 = pt >> 3;
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q * 32;
                f32x4 tv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (full_m || r < mrem) tv = *reinterpret_cast<const f32x4*>(dyt + r * ldy + c4);
                v[q] = tv;
            }
            // W chunk [KC x 32] -> Wt_s[n][kk]: rows kk = r0 + q*32 < KCT
            f32x4 wv[KT];
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                const int kk = r0 + q * 32;
                f32x4 tv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (kk < KC && k0 + kk < a.K) tv = *reinterpret_cast<const f32x4*>(wt + kk * a.O + c4);
                wv[q] = tv;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) dY_s[(c4 + e) * AS + r0 + q * 32] = v[q][e];
#pragma unroll
            for (int q = 0; q < KT; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) Wt_s[(c4 + e) * WS + r0 + q * 32] = wv[q][e];
        } else {
            const int n = pt & 31, rr = pt >> 5;
            const bool nok = n0 + n < a.O;
#pragma unroll
            for (int rb = 0; rb < BM; rb += 32) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = rb + q * 8 + rr;
                    v[q] = (nok && r < mrem) ? dyt[r * ldy + n] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) dY_s[n * AS + rb + q * 8 + rr] = v[q];
            }
#pragma unroll
            for (int kb = 0; kb < KCT; kb += 32) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = kb + q * 8 + rr;
                    v[q] = (nok && kk < KC && k0 + kk < a.K) ? wt[kk * a.O + n] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) Wt_s[n * WS + kb + q * 8 + rr] = v[q];
            }
        }
    };
    auto chain_rule = [&](int ci, int p) {          // dA_s holds the (ci, p) tile (SHARED: summed over p)
        const int g = p * a.xmod + gx;
        const int q = ci * nshare + p;
        const BasisArgs b = make_basis(a, g);
        const float* xs = x_s + (ci & 1) * XS;
        float* dxs = dx_s + (ci & 1) * XS;
        const float* us = u_s + (q & 1) * XS;
        float* dus = du_s + (q & 1) * XS;
        const int r = pt & (BM - 1);
        if constexpr (SINE) {
            if (a.G <= KV_SINE_REG_G) {
                // d loss / d freq[g] = sum over (row, feature) of dA * cos(.) * x: per-lane partial sums in registers over
                // the whole tile, ONE wave reduction per grid point at the end (not one per element)
                float dfq[KV_SINE_REG_G];
#pragma unroll
                for (int g2 = 0; g2 < KV_SINE_REG_G; ++g2) dfq[g2] = 0.0f;
                for (int il = pt >> 7; il < IC; il += 2) {
                    if (ci * IC + il >= a.I) break;
                    float dxv;
                    basis_bwd_sine_reg(b, xs[r * ICP + il], ci * IC + il, dA_s + (il * GP) * AS + r, AS, dxv, dfq);
                    dxs[r * ICP + il] += dxv;
                }
                float* dst = dfq_s + (p * 4 + pw) * a.G;
#pragma unroll
                for (int g2 = 0; g2 < KV_SINE_REG_G; ++g2) {
                    if (g2 < a.G) {
                        const float part = kv_wave_sum(dfq[g2]);
                        if ((threadIdx.x & 63) == 0) dst[g2] += part;
                    }
                }
                return;
            }
        }
        for (int il = pt >> 7; il < ((IC + 1) & ~1); il += 2) {   // uniform trip count (SINE wave-reduces)
            const bool valid = (il < IC) && (ci * IC + il < a.I);
            const int ilc = valid ? il : 0;
            float dxv, duv;
            basis_bwd<FAM>(b, xs[r * ICP + ilc], RBF ? us[r * ICP + ilc] : 0.0f, ci * IC + ilc, valid,
                           dA_s + (ilc * GP) * AS + r, AS, dxv, duv, SINE ? dfq_s + (p * 4 + pw) * a.G : nullptr);
            if (valid) {
                dxs[r * ICP + il] += dxv;
                if (RBF) dus[r * ICP + il] = duv;
            }
        }
    };
    // write a [BM x IC] LDS tile back to global (coalesced along the feature axis), optionally zeroing it
    const int ICR = kv_pow2_ge(IC);
    const int wlg = __builtin_ctz(ICR);
    const int wl = pt & (ICR - 1), wr0 = pt >> wlg, wrs = NPROD >> wlg;
    auto write_rows = [&](float* __restrict__ src, float* __restrict__ dstg, int ld, int i0, bool zero) {
        if (wl < IC && i0 + wl < a.I) {
            float* dt = dstg + m0 * ld + i0 + wl;
            for (int r = wr0; r < BM; r += wrs) {
                if (r < mrem) dt[r * ld] = src[r * ICP + wl];
                if (zero) src[r * ICP + wl] = 0.0f;
            }
        } else if (zero && wl < IC) {
            for (int r = wr0; r < BM; r += wrs) src[r * ICP + wl] = 0.0f;
        }
    };

    f32x16 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.0f;

    if (!consumer) {
        for (int j = pt; j < 2 * XS; j += NPROD) dx_s[j] = 0.0f;
        if (SINE)
            for (int j = pt; j < nshare * 4 * a.G; j += NPROD) dfq_s[j] = 0.0f;
        stage_ops(0, 0, 0, 0);
    }
    __syncthreads();

    // (ci, p, cn) of steps t, t+1, t-1, t-2 are carried incrementally: runtime integer division costs ~25 scalar
    // instructions on this ISA and the loop needed eight of them per iteration.
    int ci = 0, p = 0, cn = 0;                 // step t
    int ci1 = 0, p1 = 0, cn1 = 0;              // step t-1 (valid for t >= 1)
    int ci2 = 0, p2 = 0, cn2 = 0;              // step t-2 (valid for t >= 2)
    for (int t = 0; t < T + 2; ++t) {
        const int rem = p * ncn + cn;
        int cin = ci, pn = p, cnn = cn + 1;    // step t+1
        if (cnn == ncn) { cnn = 0; ++pn; }
        if (pn == nshare) { pn = 0; ++cin; }
        // step t completes a contraction: per (ci, p), or per ci when the groups share the basis
        const bool ends = (t < T) && (SHARED ? (rem == spc - 1) : (cn == ncn - 1));
        if (consumer) {
            if (t < T) {
                if constexpr (BF) {
                    const unsigned short* dYb = reinterpret_cast<const unsigned short*>(ops + (t & 1) * OPS);
                    const unsigned short* ap = dYb + (wave * 32 + l31) * OP + 8 * hf;
                    const unsigned short* bp = dYb + BM * OP + ((size_t)hf * KCT + l31) * 8;
                    for (int ks = 0; ks < (a.O >> 4); ++ks) {
                        const bf16x8_t a8 = *reinterpret_cast<const bf16x8_t*>(ap + 16 * ks);
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt) {
                            const bf16x8_t b8 = *reinterpret_cast<const bf16x8_t*>(bp + ((size_t)(2 * ks) * KCT + kt * 32) * 8);
                            acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[kt], 0, 0, 0);
                        }
                    }
                } else {
                    const float* ap = ops + (t & 1) * OPS + hf * AS + wave * 32 + l31;
                    const float* wp = ops + (t & 1) * OPS + BIN_NC * AS + hf * WS + l31;
#pragma unroll 4
                    for (int s2 = 0; s2 < BIN_NC / 2; ++s2) {
                        const float av = ap[(2 * s2) * AS];
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt)
                            acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wp[(2 * s2) * WS + kt * 32], acc[kt], 0, 0, 0);
                    }
                }
            }
        } else {
            if (t + 1 < T) stage_ops(t + 1, cin, pn, cnn);
            if (t < T && rem == 0) {             // first step of chunk ci: its x tile
                stage_rows<BM>(x_s + (ci & 1) * XS, xg, a.ldx, m0, a.M, ci * IC, a.I, IC, ICP, pt);
            }
            if (RBF && t < T && cn == 0) {                        // first step of (ci, p): its u tile
                const int g = p * a.xmod + gx;
                const float* ug = a.u ? a.u + (long long)g * a.I : xg;
                stage_rows<BM>(u_s + ((ci * nshare + p) & 1) * XS, ug, a.u ? a.ldu : a.ldx, m0, a.M, ci * IC, a.I, IC, ICP, pt);
            }
            // (t-2): write-outs of what the chain rule of iteration t-1 produced
            if (t >= 2) {
                const int rem2 = p2 * ncn + cn2;
                const bool ended2 = SHARED ? (rem2 == spc - 1) : (cn2 == ncn - 1);
                if (ended2) {
                    if (RBF && a.du)
                        write_rows(du_s + ((ci2 * nshare + p2) & 1) * XS, a.du + (long long)(p2 * a.xmod + gx) * a.I, (int)a.ldu,
                                   ci2 * IC, false);
                    if (p2 == nshare - 1) write_rows(dx_s + (ci2 & 1) * XS, dxg, (int)a.ldx, ci2 * IC, true);
                }
            }
            // (t-1): chain rule on the tile parked at the end of iteration t-1
            if (t >= 1 && t - 1 < T) {
                const int rem1 = p1 * ncn + cn1;
                const bool ended1 = SHARED ? (rem1 == spc - 1) : (cn1 == ncn - 1);
                if (ended1) chain_rule(ci1, p1);
            }
        }
        if (ends) {
            __syncthreads();                                      // producers are done reading dA_s
            if (consumer) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        dA_s[(kt * 32 + l31) * AS + wave * 32 + kv_acc_row(r, hf)] = acc[kt][r];
                        acc[kt][r] = 0.0f;
                    }
            }
        }
        __syncthreads();
        ci2 = ci1; p2 = p1; cn2 = cn1;
        ci1 = ci; p1 = p; cn1 = cn;
        ci = cin; p = pn; cn = cnn;
    }

    if (SINE && !consumer) {
        // every producer wave added into its own slots; combine the 4 waves in a fixed order
        for (int j = pt; j < nshare * a.G; j += NPROD) {
            const int p = j / a.G, gg = j - p * a.G;
            const float* src = dfq_s + (p * 4) * a.G + gg;
            const float v = ((src[0] + src[a.G]) + src[2 * a.G]) + src[3 * a.G];
            a.dparam[((long long)blockIdx.y * a.groups + (p * a.xmod + gx)) * a.G + gg] = v;
        }
    }
}

// =============================================================================================
// backward w.r.t. the packed weights.  grid (feature chunks, msplit, nsets * nchunks_n), 512 thr.
// Steps = 32-row slices of this block's row range; producers prepare slice s+1 (dY tile, basis
// tile) and slice s+2's x tile while the consumers contract slice s.  NSH > 1: the groups that
// share the basis tile (q, k, v of a head for LINEAR / CHEBY / FOURIER) are contracted against ONE
// generated tile (dY tile is [32 x NSH*64]).
// =============================================================================================
template <int FAM, int NSH, bool BF>
__global__ __launch_bounds__(NTHR) void kan_bwd_weight_kernel(const LayerArgs a) {
    // BF (KANVIT_FLAG_BF16_MFMA): both operands are gathered from the fp32 LDS tiles (8 ds_read_b32 each), rounded to
    // bf16 and contracted by v_mfma_f32_32x32x16_bf16 -- 16 rows per MFMA instead of 2; LDS-read bound, ~4x the fp32 rate.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * BW_NT;
    constexpr int YROW = NSH * BN;                  // floats per dY_s row
    constexpr int TPW = (NSH == 1) ? BW_TPW : 8;    // max tiles per consumer wave
    constexpr bool RBF = (FAM == KV_RBF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const bool consumer = wave < 4;
    const int pt = tid & (NPROD - 1);
    const int IC = a.IC, GP = a.GP, ICP = IC | 1;
    const int i0 = blockIdx.x * IC;
    const int ms = blockIdx.y;
    const int gs = blockIdx.z / a.nchunks_n;
    const int n0 = (blockIdx.z - gs * a.nchunks_n) * BN;
    const int nsets = a.groups / NSH;
    const int KC = IC * GP;
    const int KT = (KC + 31) / 32;
    const int ntiles = KT * BW_NT * NSH;
    const int XS = BW_ROWS * ICP, YS = BW_ROWS * YROW, ASZ = KT * 32 * BW_AS;

    float* x_s = smem;                              // [2][XS]
    float* u_s = x_s + 2 * XS;                      // [2][XS] (RBF)
    float* dY_s = u_s + (RBF ? 2 * XS : 0);         // [2][YS]
    float* A_s = dY_s + 2 * YS;                     // [2][ASZ]

    const BasisArgs b = make_basis(a, gs);
    const int xcol = (NSH == 1 ? gs % a.xmod : gs) * a.I;
    const float* xg = a.x + xcol;
    const float* ug = (RBF && a.u) ? a.u + (long long)gs * a.I : xg;
    const long long ldu = (RBF && a.u) ? a.ldu : a.ldx;
    const long long mbeg = (long long)ms * a.rows_per_split;
    const long long mend = (mbeg + a.rows_per_split < a.M) ? mbeg + a.rows_per_split : a.M;
    const int nst = (mend > mbeg) ? (int)((mend - mbeg + BW_ROWS - 1) / BW_ROWS) : 0;
    const int ldy = (int)a.ldy;
    const bool vec_n = ((a.O & 3) == 0) && (n0 + BN <= a.O) && ((a.ldy & 3) == 0);

    auto stage_x = [&](int s, int buf) {
        stage_rows<BW_ROWS>(x_s + buf * XS, xg, a.ldx, mbeg + (long long)s * BW_ROWS, mend, i0, a.I, IC, ICP, pt);
        if (RBF) stage_rows<BW_ROWS>(u_s + buf * XS, ug, ldu, mbeg + (long long)s * BW_ROWS, mend, i0, a.I, IC, ICP, pt);
    };
    auto stage_dy = [&](int s, int buf) {           // [32 x NSH*64] tile, rows past the range are zero
        float* dst = dY_s + buf * YS;
        const long long mr = mbeg + (long long)s * BW_ROWS;
        const int rows = (mend - mr < BW_ROWS) ? (int)(mend - mr) : BW_ROWS;
        const float* dyt = a.dy + mr * a.ldy + n0;                              // uniform
        if (vec_n) {
            const int c4 = (pt & 15) * 4, r0 = pt >> 4;                          // 16 float4 per 64-column row, 16 rows per pass
#pragma unroll
            for (int p = 0; p < NSH; ++p) {
                const int g = (NSH == 1) ? gs : p * nsets + gs;
                f32x4 v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = r0 + q * 16;
                    f32x4 tv = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (r < rows) tv = *reinterpret_cast<const f32x4*>(dyt + (long long)g * a.O + r * ldy + c4);
                    v[q] = tv;
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) *reinterpret_cast<f32x4*>(dst + (r0 + q * 16) * YROW + p * BN + c4) = v[q];
            }
        } else {
            const int n = pt & (BN - 1), rr = pt / BN;  // 4 rows per pass
            const bool nok = n0 + n < a.O;
#pragma unroll
            for (int p = 0; p < NSH; ++p) {
                const int g = (NSH == 1) ? gs : p * nsets + gs;
#pragma unroll
                for (int rb = 0; rb < BW_ROWS; rb += 16) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = rb + q * 4 + rr;
                        v[q] = (nok && r < rows) ? dyt[(long long)g * a.O + r * ldy + n] : 0.0f;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[(rb + q * 4 + rr) * YROW + p * BN + n] = v[q];
                }
            }
        }
    };
    auto gen_a = [&](int buf) {
        const int r = pt & (BW_ROWS - 1);
        const float* xs = x_s + buf * XS + r * ICP;
        const float* us = u_s + buf * XS + r * ICP;
        float* As = A_s + buf * ASZ + r;
        for (int il = pt >> 5; il < IC; il += NPROD / BW_ROWS) {
            const int i = i0 + il;
            float* dst = As + (il * GP) * BW_AS;
            if (i < a.I) {
                basis_fwd<FAM>(b, xs[il], RBF ? us[il] : 0.0f, i, dst, BW_AS);
            } else {
                for (int j = 0; j < GP; ++j) dst[j * BW_AS] = 0.0f;
            }
        }
    };

    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    if (!consumer) {
        for (int bufi = 0; bufi < 2; ++bufi)
            for (int idx = KC * BW_AS + pt; idx < ASZ; idx += NPROD) A_s[bufi * ASZ + idx] = 0.0f;   // pad rows
        if (nst > 0) stage_x(0, 0);
    }
    __syncthreads();
    if (!consumer && nst > 0) {
        stage_dy(0, 0);
        gen_a(0);
        if (nst > 1) stage_x(1, 1);
    }
    __syncthreads();

    constexpr int NTC = BW_NT * NSH;                // column tiles per k tile
    for (int s = 0; s < nst; ++s) {
        if (consumer) {
            // dW tile[k][o] += sum_rows A[row][k] * dY[row][o]: MFMA row index = k, contraction = row
            const float* Ab = A_s + (s & 1) * ASZ;
            const float* Yb = dY_s + (s & 1) * YS;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int t = wave + 4 * j;
                if (t < ntiles) {
                    const int kt = t / NTC, nt = t - kt * NTC;
                    if constexpr (BF) {
                        const float* ap = Ab + (kt * 32 + l31) * BW_AS + 8 * hf;
                        const float* bp2 = Yb + (8 * hf) * YROW + nt * 32 + l31;
#pragma unroll
                        for (int ks = 0; ks < BW_ROWS / 16; ++ks) {
                            float af[8], bf[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                af[e] = ap[16 * ks + e];
                                bf[e] = bp2[(16 * ks + e) * YROW];
                            }
                            const u32x4 au = {kv_pack_bf16(af[0], af[1]), kv_pack_bf16(af[2], af[3]), kv_pack_bf16(af[4], af[5]),
                                              kv_pack_bf16(af[6], af[7])};
                            const u32x4 bu = {kv_pack_bf16(bf[0], bf[1]), kv_pack_bf16(bf[2], bf[3]), kv_pack_bf16(bf[4], bf[5]),
                                              kv_pack_bf16(bf[6], bf[7])};
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, au),
                                                                             __builtin_bit_cast(bf16x8_t, bu), acc[j], 0, 0, 0);
                        }
                    } else {
                        const float* ap = Ab + (kt * 32 + l31) * BW_AS + hf;
                        const float* bp2 = Yb + hf * YROW + nt * 32 + l31;
#pragma unroll 4
                        for (int k2 = 0; k2 < BW_ROWS / 2; ++k2)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * k2], bp2[(2 * k2) * YROW], acc[j], 0, 0, 0);
                    }
                }
            }
        } else if (s + 1 < nst) {
            stage_dy(s + 1, (s + 1) & 1);
            gen_a((s + 1) & 1);
            if (s + 2 < nst) stage_x(s + 2, s & 1);
        }
        __syncthreads();
    }

    if (consumer) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int t = wave + 4 * j;
            if (t < ntiles) {
                const int kt = t / NTC, nt = t - kt * NTC;
                const int p = nt / BW_NT, ntl = nt - p * BW_NT;
                const int g = (NSH == 1) ? gs : p * nsets + gs;
                float* slab = a.slab + ((long long)ms * a.groups + g) * a.K * a.O;
                const int col = n0 + ntl * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kk = kt * 32 + kv_acc_row(r, hf);
                    const int k = i0 * GP + kk;
                    if (kk < KC && k < a.K && col < a.O) slab[(long long)k * a.O + col] = acc[j][r];
                }
            }
        }
    }
}

// =============================================================================================
// backward w.r.t. the weights, register form: a barrier-free, LDS-free streaming kernel.
//   dW[g][i*GP + j][o] = sum_m Phi_j(x[m][i]) * dY[m][g*O + o]
// The contraction runs over tokens, so the MFMA lane index of the Phi operand is the K row.  K is tiled so that k-tile j of
// a 32-feature block holds basis function j of 32 DIFFERENT features: lane (l & 31) owns one feature, evaluates its GP
// basis functions at its token(s) once (BasisGen, a recurrence for Chebyshev) and those GP values ARE its A fragments of
// the GP k-tiles.  The dY operand is a plain dword load (lane = output column).  Each wave accumulates a
// [32 features x GP] x [NOT column tiles] block of dW in registers (GP*NOT*16 accumulators) over its slab of tokens;
// operands are software-prefetched PD blocks ahead (one wave per SIMD: latency is hidden by the prefetch, not occupancy).
// Work-group = 4 consecutive wave units (feature block fastest), so neighbouring waves share dY (and x across tile sets).
// fp32: v_mfma_f32_32x32x2f32, 2 tokens per step (lane half = token parity).  bf16 flag: v_mfma_f32_32x32x16_bf16, 16
// tokens per step, lane half h owns tokens 8h..8h+7 of the step.  Rows beyond the slab end are clamped for x and zeroed
// for dY.  Partials go to slab[s][g][k][o]; kan_slab_reduce_kernel sums them in order.
// grid ceil(units * slabs / 4), 256 threads = 4 (slab, wave unit) pairs, unit fastest.
// =============================================================================================
template <int FAM, int GP, int NOT, bool BF, int JC = GP>
__global__ __launch_bounds__(256) void kan_bwd_weight_reg_kernel(const LayerArgs a, int nfb, int nos, int tiles_per_bg,
                                                                 int shared, int nbg) {
    constexpr int NJC = GP / JC;                  // wide bases (G = 28) are contracted in NJC windows of JC basis functions,
                                                  // each its own wave unit (every window regenerates only its own values)
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr int TS = BF ? 8 : 1;            // tokens per lane per step
    constexpr int UB = BF ? 1 : 4;            // steps per prefetch block
    constexpr int PD = BF ? 3 : 2;            // blocks in flight
    constexpr int NTOK = TS * UB;             // tokens per lane per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
    // wave unit u = (basis group, column-tile set, feature block), feature block fastest: the 4 waves of a work-group are
    // always 4 live units (a partly populated work-group would leave SIMDs idle: one wave fills a SIMD's register file,
    // so the next work-group cannot start until ALL four SIMDs are free)
    const int units = nfb * nos * nbg * NJC;
    const long long gw = (long long)blockIdx.x * 4 + wave;       // global wave index over (slab, unit), unit fastest
    if (gw >= (long long)units * a.msplit) return;
    const int u = (int)(gw % units), slab = (int)(gw / units);
    const int fb = u % nfb, os = (u / nfb) % nos, jc = (u / (nfb * nos)) % NJC, bg = u / (nfb * nos * NJC);
    const int j0 = jc * JC;
    const long long ms = (long long)slab * a.rows_per_split;
    long long me = ms + a.rows_per_split;
    if (me > a.M) me = a.M;
    const int len = (int)(me - ms);
    if (len <= 0) return;
    const int otpg = a.O / 32;
    const int f = fb * 32 + l31;
    const int gx = bg % a.xmod;

    // this wave's column tiles
    int tg[NOT];          // group of tile i (or -1)
    long long tcol[NOT];  // column offset of tile i in a dY row
#pragma unroll
    for (int i = 0; i < NOT; ++i) {
        const int tt = os * NOT + i;
        if (tt < tiles_per_bg) {
            const int p = tt / otpg;
            tg[i] = shared ? p * a.xmod + bg : bg;
            tcol[i] = (long long)tg[i] * a.O + (tt - p * otpg) * 32;
        } else {            // past the last tile: recompute the first tile (no branch around the MFMAs), never stored
            tg[i] = -1;
            tcol[i] = (long long)(shared ? ((os * NOT) / otpg) * a.xmod + bg : bg) * a.O + ((os * NOT) % otpg) * 32;
        }
    }
    const int g0 = bg;                 // basis parameters: identical for every group of a shared launch
    const BasisArgs b = make_basis(a, g0);
    BasisGenP<FAM, JC> proto;          // knots / centres / frequencies / phases of this lane's feature, loaded once
    proto.prepare(b, f, j0);

    // Addressing: wave-uniform 64-bit bases (start of this slab) + 32-bit per-lane offsets, so that a load costs one or two
    // VALU instructions for its address instead of a 64-bit multiply-add chain: the PMC pass of round 1 counted 3.5 VALU
    // instructions per MFMA in this kernel, mostly address arithmetic, and the fp32 matrix pipe waits for every one of them
    // (DESIGN.md section 4.1).  The host guarantees rows_per_split * max(ldx, ldu, ldy) < 2^29 elements.
    const float* xbase = a.x + ms * a.ldx + (long long)gx * a.I;                                  // uniform
    const float* ubase = RBF ? (a.u ? a.u + ms * a.ldu + (long long)g0 * a.I : xbase) : xbase;      // uniform
    const int ldx32 = (int)a.ldx, ldu32 = RBF ? (a.u ? (int)a.ldu : (int)a.ldx) : (int)a.ldx, ldy32 = (int)a.ldy;
    const float* dybase = a.dy + ms * a.ldy;                                                        // uniform
    int dyo[NOT];
#pragma unroll
    for (int i = 0; i < NOT; ++i) dyo[i] = (int)tcol[i] + l31;

    f32x16 acc[JC][NOT];
#pragma unroll
    for (int j = 0; j < JC; ++j)
#pragma unroll
        for (int i = 0; i < NOT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

    // token of (block, step u, e): fp32: 2*(blk*UB + u) + hf ; bf16: 16*blk + 8*hf + e
    float rx[PD][NTOK], ru[RBF ? PD : 1][RBF ? NTOK : 1], rdy[PD][NTOK][NOT];
    // KANVIT_FLAG_FUSED_LN: u = (x - mean) * rstd * gamma + beta is formed when a block leaves the ring.  The slab's (mean, rstd)
    // pairs sit in a wave-private LDS strip (the register file is full: a second ring for them spills), filled once up front
    // and read back as two-address broadcasts; no barrier -- the strip belongs to this wave alone.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool ln = RBF && a.ln;
    float ln_g = 1.0f, ln_b = 0.0f;               // gamma / beta of this lane's feature
    const float2* st_w = nullptr;
    if constexpr (RBF) {
        if (ln) {
            ln_g = b.bp[a.G + f];
            ln_b = b.bp[a.G + a.I + f];
            float2* strip = reinterpret_cast<float2*>(smem) + (size_t)wave * a.rows_per_split;
            const float* stbase = a.stats + (ms * a.xmod + gx) * 2;      // uniform
            for (int i = lane; i < len; i += 64) strip[i] = *reinterpret_cast<const float2*>(stbase + (size_t)i * (2 * a.xmod));
            st_w = strip;
        }
    }
    auto tok_of = [&](int blk, int t) -> int { return BF ? (16 * blk + 8 * hf + t) : (2 * (blk * UB + t) + hf); };
    auto load_block = [&](int q, int blk) {
#pragma unroll
        for (int t = 0; t < NTOK; ++t) {
            int tk = tok_of(blk, t);
            if (tk > len - 1) tk = len - 1;
            rx[q][t] = xbase[tk * ldx32 + f];
            if constexpr (RBF) {
                if (!ln) ru[q][t] = ubase[tk * ldu32 + f];
            }
            const int dyr = tk * ldy32;
#pragma unroll
            for (int i = 0; i < NOT; ++i) rdy[q][t][i] = dybase[dyr + dyo[i]];
        }
    };
    const int tok_per_blk = BF ? 16 : 2 * UB;
    const int nblk = (len + tok_per_blk - 1) / tok_per_blk;
#pragma unroll
    for (int q = 0; q < PD; ++q)
        if (q < nblk) load_block(q, q);

    for (int blk0 = 0; blk0 < nblk; blk0 += PD) {
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            const int blk = blk0 + q;
            if (blk < nblk) {
                // take the block out of the ring (this is where the loads are waited for), zero dY of rows past the slab
                float cx[NTOK], cu[RBF ? NTOK : 1], cdy[NTOK][NOT];
#pragma unroll
                for (int t = 0; t < NTOK; ++t) {
                    const bool ok = tok_of(blk, t) < len;
                    cx[t] = rx[q][t];
                    if constexpr (RBF) {
                        if (ln) {
                            const int tk = tok_of(blk, t);
                            const float2 st = st_w[tk < len ? tk : len - 1];
                            cu[t] = (rx[q][t] - st.x) * st.y * ln_g + ln_b;
                        } else {
                            cu[t] = ru[q][t];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NOT; ++i) cdy[t][i] = ok ? rdy[q][t][i] : 0.0f;
                }
                if (blk + PD < nblk) load_block(q, blk + PD);
                if constexpr (!BF) {
#pragma unroll
                    for (int t = 0; t < NTOK; ++t) {
                        BasisGenP<FAM, JC> gen = proto;
                        gen.init(cx[t], RBF ? cu[t] : 0.0f);
#pragma unroll
                        for (int j = 0; j < JC; ++j) {
                            const float av = gen.next(j);
#pragma unroll
                            for (int i = 0; i < NOT; ++i)
                                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, cdy[t][i], acc[j][i], 0, 0, 0);
                        }
                    }
                } else {
                    unsigned af[JC][4];
#pragma unroll
                    for (int ep = 0; ep < 4; ++ep) {
                        BasisGenP<FAM, JC> g0_ = proto, g1_ = proto;
                        g0_.init(cx[2 * ep], RBF ? cu[2 * ep] : 0.0f);
                        g1_.init(cx[2 * ep + 1], RBF ? cu[2 * ep + 1] : 0.0f);
#pragma unroll
                        for (int j = 0; j < JC; ++j) af[j][ep] = kv_pack_bf16(g0_.next(j), g1_.next(j));
                    }
                    bf16x8_t bfr[NOT];
#pragma unroll
                    for (int i = 0; i < NOT; ++i) {
                        const u32x4 u4 = {kv_pack_bf16(cdy[0][i], cdy[1][i]), kv_pack_bf16(cdy[2][i], cdy[3][i]),
                                          kv_pack_bf16(cdy[4][i], cdy[5][i]), kv_pack_bf16(cdy[6][i], cdy[7][i])};
                        bfr[i] = __builtin_bit_cast(bf16x8_t, u4);
                    }
#pragma unroll
                    for (int j = 0; j < JC; ++j) {
                        const u32x4 a4 = {af[j][0], af[j][1], af[j][2], af[j][3]};
                        const bf16x8_t afr = __builtin_bit_cast(bf16x8_t, a4);
#pragma unroll
                        for (int i = 0; i < NOT; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[i], acc[j][i], 0, 0, 0);
                    }
                }
            }
        }
    }

    // dW partial of this slab: row k = (fb*32 + acc row)*GP + j, 32 contiguous columns per row
    float* base = a.slab + (long long)slab * ((long long)a.groups * a.K * a.O);
#pragma unroll
    for (int i = 0; i < NOT; ++i) {
        if (tg[i] < 0) continue;
        const int tt = os * NOT + i;
        const int col0 = (tt % otpg) * 32;
        float* gb = base + (long long)tg[i] * a.K * a.O + col0 + l31;
#pragma unroll
        for (int j = 0; j < JC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int fr = fb * 32 + kv_acc_row(r, hf);
                gb[((long long)fr * GP + j0 + j) * a.O] = acc[j][i][r];
            }
    }
}

// ordered sum of the msplit partial slabs (deterministic; no float atomics)
__global__ __launch_bounds__(256) void kan_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                              long long total, int msplit) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        float s = slab[e];
        int ms = 1;
        for (; ms + 7 < msplit; ms += 8) {            // eight loads in flight, added in slab order
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = slab[(long long)(ms + j) * total + e];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += t[j];
        }
        for (; ms < msplit; ++ms) s += slab[(long long)ms * total + e];
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

}  // namespace
extern "C" int kanvit_layer_ln_fusable(const kanvit_layer_desc* d);
namespace {

int validate(const kanvit_layer_desc* d, const char* who) {
    if (!d) return kv_fail(KANVIT_EINVAL, "%s: null descriptor", who);
    const int gp = gp_of(d);
    if (gp < 1) return kv_fail(KANVIT_EINVAL, "%s: unknown family %d or G=%d", who, d->family, d->G);
    if (gp > 80) return kv_fail(KANVIT_EINVAL, "%s: %d generated columns per feature exceeds the supported 80", who, gp);
    if (d->groups < 1 || d->x_group_mod < 1 || d->groups % d->x_group_mod != 0)
        return kv_fail(KANVIT_EINVAL, "%s: groups=%d must be a positive multiple of x_group_mod=%d", who, d->groups,
                       d->x_group_mod);
    if (d->groups > 65535) return kv_fail(KANVIT_EINVAL, "%s: groups=%d exceeds 65535", who, d->groups);
    if ((d->M + BM - 1) / BM > 65535) return kv_fail(KANVIT_EINVAL, "%s: M=%lld exceeds %d rows per launch", who,
                                                     (long long)d->M, 65535 * BM);
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
    if (d->flags & KANVIT_FLAG_FUSED_LN) {
        if (d->family != KANVIT_RBF) return kv_fail(KANVIT_EINVAL, "%s: KANVIT_FLAG_FUSED_LN is an RBF (FastKAN) flag", who);
        if (d->bparam_stride < (int64_t)d->G + 2 * (int64_t)d->I)
            return kv_fail(KANVIT_EINVAL, "%s: KANVIT_FLAG_FUSED_LN needs bparams = [centres(G) | gamma(I) | beta(I)] per group", who);
        if (!kanvit_layer_ln_fusable(d)) return kv_fail(KANVIT_EINVAL, "%s: shape not covered by the register kernels that fuse the LayerNorm", who);
    }
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
    a.flags = d->flags;
    a.ln = (d->family == KANVIT_RBF && (d->flags & KANVIT_FLAG_FUSED_LN)) ? 1 : 0;
    a.ln_eps = d->ln_eps;
    return a;
}

int needs_bparams(int family) { return family == KANVIT_BSPLINE || family == KANVIT_RBF || family == KANVIT_SINE; }

// ---- forward -----------------------------------------------------------------------------------
// RBF in the register kernels: only FastKAN's own uniform 8-centre grid (kv_rbf8: two exp anchors + recurrence); the caller
// vouches with KANVIT_FLAG_UNIFORM_KNOTS, anything else takes the LDS-tile kernels (direct exp per centre)
inline bool kv_rbf_reg_ok(int flags, int G) { return (flags & KANVIT_FLAG_UNIFORM_KNOTS) && G == 8; }

// families that get shared-basis (NSH = 3 / SHARED) kernel instantiations ...
template <int FAM>
constexpr bool kv_shared_basis() { return FAM == KV_LINEAR || FAM == KV_CHEBY || FAM == KV_FOURIER || FAM == KV_BSPLINE; }
// ... and whether a given launch may use them: parameter-free families always, BSPLINE when the caller vouches that the
// groups sharing x also share the knot table (KANVIT_FLAG_SHARED_BPARAMS)
inline bool kv_share_ok(int family, int flags) {
    return family == KANVIT_LINEAR || family == KANVIT_CHEBY || family == KANVIT_FOURIER ||
           (family == KANVIT_BSPLINE && (flags & KANVIT_FLAG_SHARED_BPARAMS));
}

template <int FAM>
size_t fwd_lds(int ic, int gp, int nt, int nsh) {
    const int kcp = (ic * gp + 1) & ~1;
    const size_t xarea = 2 * (size_t)BM * (ic | 1) * (FAM == KV_RBF ? 2 : 1);
    const size_t opnd = 2 * ((size_t)kcp * AS + (size_t)kcp * 32 * nt * nsh);
    const size_t otile = (size_t)BM * (32 * nt * nsh + 4);          // staged output tile (FAST epilogue) aliases the operands
    return sizeof(float) * (xarea + (opnd > otile ? opnd : otile));
}

template <int FAM, int NT, int NSH, bool FAST>
int launch_fwd(const LayerArgs& a, hipStream_t st) {
    constexpr int BN = 32 * NT;
    const size_t lds = fwd_lds<FAM>(a.IC, a.GP, NT, NSH);
    KV_ALLOW_LDS(160 * 1024, kan_fwd_kernel<FAM, NT, NSH, FAST>);
    dim3 grid((unsigned)((a.groups / NSH) * ((a.O + BN - 1) / BN)), (unsigned)((a.M + BM - 1) / BM), 1);
    hipLaunchKernelGGL((kan_fwd_kernel<FAM, NT, NSH, FAST>), grid, dim3(NTHR), lds, st, a);
    KV_LAUNCH_CHECK("kan_fwd_kernel");
    return 0;
}

template <int FAM, int NT, int NSH>
int launch_fwd_sel(const LayerArgs& a, bool fast, hipStream_t st) {
    return fast ? launch_fwd<FAM, NT, NSH, true>(a, st) : launch_fwd<FAM, NT, NSH, false>(a, st);
}

// ---- bf16 register-operand forward ------------------------------------------------------------------
struct FwdRegBf16Plan {
    bool ok;
    int gp, nt, nsh, ich, vs, nch;
    size_t lds, ws_bytes;
};

FwdRegBf16Plan plan_fwd_reg_bf16(const kanvit_layer_desc* d) {
    FwdRegBf16Plan p{};
    if (kv_config().no_reg) return p;
    p.gp = gp_of(d);
    const int fam = d->family;
    const bool gp_ok = (fam == KANVIT_LINEAR && p.gp == 1) || (fam == KANVIT_CHEBY && p.gp == 5) ||
                       (fam == KANVIT_BSPLINE && p.gp == 9 && (d->flags & KANVIT_FLAG_UNIFORM_KNOTS) && d->spline_order == 3) ||
                       (fam == KANVIT_RBF && p.gp == 9 && kv_rbf_reg_ok(d->flags, d->G)) || (fam == KANVIT_SINE && (p.gp == 4 || p.gp == 28)) ||
                       (fam == KANVIT_FOURIER && p.gp == 56);
    if (!gp_ok) return p;
    p.nt = d->O <= 32 ? 1 : (d->O <= 64 ? 2 : 4);
    if (d->O % (32 * p.nt) || (d->O & 3) || (d->ldy & 3)) return p;
    const int nshare = d->groups / d->x_group_mod;
    p.nsh = (kv_share_ok(fam, d->flags) && nshare == 3 && p.nt <= 2) ? 3 : 1;
    p.ich = p.gp >= 28 ? 1 : 8;                   // features per lane half and chunk (instantiated: 8, or 1 for the wide bases)
    if (d->I % (2 * p.ich)) return p;
    if (p.ich == 8 && ((d->ldx & 3) || (d->I & 3) || (fam == KANVIT_RBF && (d->ldu & 3)))) return p;
    p.vs = (p.ich * p.gp + 7) / 8;
    p.nch = d->I / (2 * p.ich);
    p.lds = (size_t)2 * p.vs * 2 * 32 * p.nt * p.nsh * 16;
    if (p.lds < sizeof(float) * 4 * 32 * 36) p.lds = sizeof(float) * 4 * 32 * 36;
    if (p.lds > 160 * 1024) return p;
    if ((p.vs * 2 * 32 * p.nt + 255) / 256 > 12) return p;
    p.ws_bytes = (size_t)d->groups * p.nch * p.vs * 2 * d->O * 16;
    p.ok = true;
    return p;
}

template <int FAM, int GP, int NT, int NSH, int ICH>
int launch_fwd_reg_bf16(const LayerArgs& a, const FwdRegBf16Plan& p, hipStream_t st) {
    // W-stationary persistent form when the whole weight image of a column set fits the LDS and there are enough row tiles
    // (instantiated for I = 64 per group: 4 chunks of 16 features -- the per-head q|k|v launches of ViT-B/S)
    if constexpr (ICH == 8 && (FAM == KV_LINEAR || FAM == KV_CHEBY)) {     // the families whose basis fragments fit the register file
        const size_t wlds = (size_t)p.nch * p.vs * 2 * 32 * NT * NSH * 16 + sizeof(float) * 32 * NT * NSH;
        const int gx = (a.groups / NSH) * (a.O / (32 * NT));
        if (wlds <= 150 * 1024 && p.nch == 4 && a.M >= 4096 && gx <= N_CU && !((uintptr_t)a.y & 15) && !kv_config().no_ws) {
            KV_ALLOW_LDS(160 * 1024, (kan_fwd_ws_bf16_kernel<FAM, GP, NT, NSH, ICH, 4>));
            const long long ntiles = (a.M + KV_WS_THREADS / 2 - 1) / (KV_WS_THREADS / 2);
            long long py = N_CU / gx;             // one work-group per CU (the image fills the LDS)
            if (py > ntiles) py = ntiles;
            if (py < 1) py = 1;
            hipLaunchKernelGGL((kan_fwd_ws_bf16_kernel<FAM, GP, NT, NSH, ICH, 4>), dim3((unsigned)gx, (unsigned)py, 1), dim3(KV_WS_THREADS), wlds, st, a);
            KV_LAUNCH_CHECK("kan_fwd_ws_bf16_kernel");
            return 0;
        }
    }
    KV_ALLOW_LDS(160 * 1024, (kan_fwd_reg_bf16_kernel<FAM, GP, NT, NSH, ICH>));
    dim3 grid((unsigned)((a.groups / NSH) * (a.O / (32 * NT))), (unsigned)((a.M + BM - 1) / BM), 1);
    hipLaunchKernelGGL((kan_fwd_reg_bf16_kernel<FAM, GP, NT, NSH, ICH>), grid, dim3(256), p.lds, st, a);
    KV_LAUNCH_CHECK("kan_fwd_reg_bf16_kernel");
    return 0;
}

template <int FAM, int GP, int ICH>
int launch_fwd_reg_bf16_shape(const LayerArgs& a, const FwdRegBf16Plan& p, hipStream_t st) {
    if (p.nsh == 3) {
        if constexpr (kv_shared_basis<FAM>()) {
            if (p.nt == 1) return launch_fwd_reg_bf16<FAM, GP, 1, 3, ICH>(a, p, st);
            return launch_fwd_reg_bf16<FAM, GP, 2, 3, ICH>(a, p, st);
        }
    }
    if (p.nt == 1) return launch_fwd_reg_bf16<FAM, GP, 1, 1, ICH>(a, p, st);
    if (p.nt == 2) return launch_fwd_reg_bf16<FAM, GP, 2, 1, ICH>(a, p, st);
    return launch_fwd_reg_bf16<FAM, GP, 4, 1, ICH>(a, p, st);
}

template <int FAM>
int dispatch_fwd_reg_bf16(LayerArgs& a, const FwdRegBf16Plan& p, void* ws, hipStream_t st) {
    unsigned short* wb = (unsigned short*)ws;
    const long long total = (long long)a.groups * p.nch * p.vs * 2 * a.O;
    hipLaunchKernelGGL(kan_pack_w_fwd_reg_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.w, wb, a.K, a.O, p.gp,
                       p.ich, p.vs, p.nch, total);
    KV_LAUNCH_CHECK("kan_pack_w_fwd_reg_kernel");
    a.wb = wb;
    if constexpr (FAM == KV_LINEAR) return launch_fwd_reg_bf16_shape<FAM, 1, 8>(a, p, st);
    if constexpr (FAM == KV_CHEBY) return launch_fwd_reg_bf16_shape<FAM, 5, 8>(a, p, st);
    if constexpr (FAM == KV_BSPLINE) return launch_fwd_reg_bf16_shape<FAM, 9, 8>(a, p, st);
    if constexpr (FAM == KV_RBF) return launch_fwd_reg_bf16_shape<FAM, 9, 8>(a, p, st);
    if constexpr (FAM == KV_SINE) {
        if (p.gp == 4) return launch_fwd_reg_bf16_shape<FAM, 4, 8>(a, p, st);
        return launch_fwd_reg_bf16_shape<FAM, 28, 1>(a, p, st);
    }
    if constexpr (FAM == KV_FOURIER) return launch_fwd_reg_bf16_shape<FAM, 56, 1>(a, p, st);
    return kv_fail(KANVIT_EINVAL, "internal: bf16 register forward dispatch");
}

// ---- bf16 matrix-core forward ----------------------------------------------------------------------
struct FwdBf16Plan {
    bool ok;
    int ic, nt, nsh, kc, kcp, nch;
    size_t lds, ws_bytes;
};

FwdBf16Plan plan_fwd_bf16(const kanvit_layer_desc* d) {
    FwdBf16Plan p{};
    const int gp = gp_of(d);
    p.nt = d->O <= 32 ? 1 : (d->O <= 64 ? 2 : 4);
    const int nshare = d->groups / d->x_group_mod;
    const bool shared_fam = kv_share_ok(d->family, d->flags);
    p.nsh = (shared_fam && nshare == 3 && p.nt <= 2) ? 3 : 1;
    if (kv_config().bf16_nsh) p.nsh = (kv_config().bf16_nsh == 3 && p.nsh == 3) ? 3 : 1;   // tuning knob
    const int icmax = kv_config().bf16_ic ? kv_config().bf16_ic : 64;                    // tuning knob
    const int rbf = d->family == KANVIT_RBF ? 2 : 1;
    for (int ic = 16; ic >= 8; ic >>= 1) {        // largest power-of-two chunk (<= 16: register-staged loads) dividing I
        if (d->I % ic || ic > icmax) continue;
        const int kc = ic * gp, kcp = (kc + 15) & ~15;
        if ((kcp / 8) * 32 * p.nt > 8 * NPROD) continue;          // W vectors per thread and group <= WQ
        const size_t xarea = sizeof(float) * 2 * (size_t)BM * (ic | 1) * rbf;
        const size_t opnd = 2 * (sizeof(float) * (size_t)kcp * AS + (size_t)kcp * 32 * p.nt * p.nsh * 2);
        const size_t otile = sizeof(float) * (size_t)BM * (32 * p.nt * p.nsh + 4);
        const size_t lds = xarea + (opnd > otile ? opnd : otile);
        if (lds > 160 * 1024) continue;
        p.ic = ic; p.kc = kc; p.kcp = kcp; p.nch = d->I / ic; p.lds = lds;
        p.ok = (d->O % (32 * p.nt) == 0) && ((long long)BM * d->ldx < (1LL << 30)) && ((long long)BM * d->ldy < (1LL << 30)) &&
               ((long long)BM * d->ldu < (1LL << 30));
        p.ws_bytes = (size_t)d->groups * p.nch * kcp * d->O * 2;
        return p;
    }
    p.ok = false;
    return p;
}

template <int FAM, int NT, int NSH>
int launch_fwd_bf16(const LayerArgs& a, const FwdBf16Plan& p, hipStream_t st) {
    KV_ALLOW_LDS(160 * 1024, kan_fwd_bf16_kernel<FAM, NT, NSH>);
    dim3 grid((unsigned)((a.groups / NSH) * (a.O / (32 * NT))), (unsigned)((a.M + BM - 1) / BM), 1);
    hipLaunchKernelGGL((kan_fwd_bf16_kernel<FAM, NT, NSH>), grid, dim3(NTHR), p.lds, st, a);
    KV_LAUNCH_CHECK("kan_fwd_bf16_kernel");
    return 0;
}

template <int FAM>
int dispatch_fwd_bf16(LayerArgs& a, const FwdBf16Plan& p, void* ws, hipStream_t st) {
    unsigned short* wb = (unsigned short*)ws;
    const long long total = (long long)a.groups * p.nch * (p.kcp / 8) * a.O;
    hipLaunchKernelGGL(kan_pack_w_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.w, wb, a.K, a.O, p.kc,
                       p.kcp, p.nch, total);
    KV_LAUNCH_CHECK("kan_pack_w_fwd_kernel");
    a.wb = wb;
    a.IC = p.ic;
    if (p.nsh == 3) {
        if constexpr (kv_shared_basis<FAM>()) {
            if (p.nt == 1) return launch_fwd_bf16<FAM, 1, 3>(a, p, st);
            return launch_fwd_bf16<FAM, 2, 3>(a, p, st);
        }
    }
    if (p.nt == 1) return launch_fwd_bf16<FAM, 1, 1>(a, p, st);
    if (p.nt == 2) return launch_fwd_bf16<FAM, 2, 1>(a, p, st);
    return launch_fwd_bf16<FAM, 4, 1>(a, p, st);
}

// ---- register-operand forward (fp32 exact) --------------------------------------------------------
template <int FAM, int NT, int NSH, int ICH, int GPC = 0>
int launch_fwd_reg(const LayerArgs& a, size_t lds, hipStream_t st) {
    KV_ALLOW_LDS(160 * 1024, (kan_fwd_reg_kernel<FAM, NT, NSH, ICH, GPC>));
    dim3 grid((unsigned)((a.groups / NSH) * (a.O / (32 * NT))), (unsigned)((a.M + BM - 1) / BM), 1);
    hipLaunchKernelGGL((kan_fwd_reg_kernel<FAM, NT, NSH, ICH, GPC>), grid, dim3(256), lds, st, a);
    KV_LAUNCH_CHECK("kan_fwd_reg_kernel");
    return 0;
}

template <int FAM, int NT, int NSH>
int launch_fwd_reg_ich(const LayerArgs& a, int ich, size_t lds, hipStream_t st) {
    // compile-time GP instantiations (pipelined chunk body): the basis sizes the reference's call sites build
    if (!kv_config().no_pipe) {
        if (ich == 4) {
            if constexpr (FAM == KV_LINEAR) { if (a.GP == 1) return launch_fwd_reg<FAM, NT, NSH, 4, 1>(a, lds, st); }
            if constexpr (FAM == KV_CHEBY) { if (a.GP == 5) return launch_fwd_reg<FAM, NT, NSH, 4, 5>(a, lds, st); }
            if constexpr (FAM == KV_BSPLINE || FAM == KV_RBF) { if (a.GP == 9) return launch_fwd_reg<FAM, NT, NSH, 4, 9>(a, lds, st); }
            if constexpr (FAM == KV_SINE) { if (a.GP == 4) return launch_fwd_reg<FAM, NT, NSH, 4, 4>(a, lds, st); }
        }
        if (ich == 1) {
            if constexpr (FAM == KV_SINE) { if (a.GP == 28) return launch_fwd_reg<FAM, NT, NSH, 1, 28>(a, lds, st); }
            if constexpr (FAM == KV_FOURIER) { if (a.GP == 56) return launch_fwd_reg<FAM, NT, NSH, 1, 56>(a, lds, st); }
        }
    }
    if (ich == 4) return launch_fwd_reg<FAM, NT, NSH, 4>(a, lds, st);
    if (ich == 2) return launch_fwd_reg<FAM, NT, NSH, 2>(a, lds, st);
    return launch_fwd_reg<FAM, NT, NSH, 1>(a, lds, st);
}

// returns 1 when the shape is not covered (caller falls back to the LDS-tile kernel), 0 on success, < 0 on error
template <int FAM>
int try_fwd_reg(const LayerArgs& a, hipStream_t st) {
    if (kv_config().no_reg) return 1;
    if (FAM == KV_BSPLINE && !((a.flags & KANVIT_FLAG_UNIFORM_KNOTS) && a.order == 3)) return 1;
    if (FAM == KV_RBF && !kv_rbf_reg_ok(a.flags, a.G)) return 1;
    const int nt = a.O <= 32 ? 1 : (a.O <= 64 ? 2 : 4);
    if (a.O % (32 * nt)) return 1;
    const int nshare = a.groups / a.xmod;
    // q|k|v sharing one basis evaluation (NSH = 3) triples the MFMA chain of every wave; when the launch has fewer
    // work-groups than CUs (the small geometries: 50 row tiles x 2 heads) the chain length IS the kernel time, so each
    // projection gets its own work-groups there and re-evaluates the basis
    const bool share3 = kv_shared_basis<FAM>() && kv_share_ok(FAM, a.flags) && nshare == 3 && nt <= 2 &&
                        ((a.M + BM - 1) / BM) * a.xmod >= N_CU;
    const int nsh = share3 ? 3 : 1;
    if ((a.O & 3) || (a.ldy & 3) || ((uintptr_t)a.y & 15) || ((uintptr_t)a.w & 15) || (a.bias && ((uintptr_t)a.bias & 15))) return 1;
    const int wrow = 32 * nt * nsh, wrs = 256 / (8 * nt);
    for (int ich = 4; ich >= 1; ich >>= 1) {
        const int ic = 2 * ich, kc = ic * a.GP;
        if (a.I % ic) continue;
        if (a.pg && (((a.pg_W / a.pg_n) % ic) || (ich == 4 && (a.pg_W & 3)))) continue;     // a chunk is ic consecutive pixels of one line
        if (ich == 4 && ((a.ldx & 3) || (a.I & 3) || ((uintptr_t)a.x & 15) ||
                         (FAM == KV_RBF && a.u && ((a.ldu & 3) || ((uintptr_t)a.u & 15)))))
            continue;
        if ((kc + wrs - 1) / wrs > (share3 ? 4 : 8)) continue;                   // W passes held in registers
        if ((long long)kc * a.O >= (1LL << 30)) continue;
        size_t lds = sizeof(float) * 2 * (size_t)kc * wrow;
        if (lds < sizeof(float) * 4 * 32 * 36) lds = sizeof(float) * 4 * 32 * 36;   // epilogue patches alias the W buffers
        if (lds > 160 * 1024) continue;
        if (share3) {
            if constexpr (kv_shared_basis<FAM>()) {
                if (nt == 1) return launch_fwd_reg_ich<FAM, 1, 3>(a, ich, lds, st);
                return launch_fwd_reg_ich<FAM, 2, 3>(a, ich, lds, st);
            }
        }
        if (nt == 1) return launch_fwd_reg_ich<FAM, 1, 1>(a, ich, lds, st);
        if (nt == 2) return launch_fwd_reg_ich<FAM, 2, 1>(a, ich, lds, st);
        return launch_fwd_reg_ich<FAM, 4, 1>(a, ich, lds, st);
    }
    return 1;
}

template <int FAM>
int dispatch_fwd(LayerArgs& a, hipStream_t st) {
    {
        const int rc = try_fwd_reg<FAM>(a, st);      // register-operand kernel when the shape allows it
        if (rc <= 0) return rc;
        if (a.ln) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: KANVIT_FLAG_FUSED_LN needs the register kernel (alignment / shape)");
    }
    const int nt = a.O <= 32 ? 1 : (a.O <= 64 ? 2 : 4);
    const int nshare = a.groups / a.xmod;
    const bool share3 = kv_shared_basis<FAM>() && kv_share_ok(FAM, a.flags) && nshare == 3 && nt <= 2;
    const int nsh = share3 ? 3 : 1;
    // largest feature chunk whose two operand buffers fit the 160 KiB LDS (cap 80 columns)
    int ic = 80 / a.GP;
    if (ic < 1) ic = 1;
    if (ic > a.I) ic = a.I;
    while (ic > 1 && fwd_lds<FAM>(ic, a.GP, nt, nsh) > 160 * 1024) --ic;
    if (fwd_lds<FAM>(ic, a.GP, nt, nsh) > 160 * 1024)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: %d generated columns per feature with O=%d does not fit the LDS", a.GP, a.O);
    // fast path: power-of-two chunk dividing I, whole column tiles, 32-bit tile-local offsets
    int icf = 1;
    while (icf * 2 <= ic) icf *= 2;
    const bool fast = (icf >= 8) && (a.I % icf == 0) && (a.O % (32 * nt) == 0) &&
                      ((long long)BM * a.ldx < (1LL << 30)) && ((long long)BM * a.ldy < (1LL << 30)) &&
                      ((long long)BM * a.ldu < (1LL << 30)) && ((long long)a.K * a.O < (1LL << 30)) && !kv_config().no_fast;
    a.IC = fast ? icf : ic;
    if (share3) {
        if constexpr (kv_shared_basis<FAM>()) {
            if (nt == 1) return launch_fwd_sel<FAM, 1, 3>(a, fast, st);
            return launch_fwd_sel<FAM, 2, 3>(a, fast, st);
        }
    }
    if (nt == 1) return launch_fwd_sel<FAM, 1, 1>(a, fast, st);
    if (nt == 2) return launch_fwd_sel<FAM, 2, 1>(a, fast, st);
    return launch_fwd_sel<FAM, 4, 1>(a, fast, st);
}

// ---- backward input ------------------------------------------------------------------------------
template <int FAM>
size_t bwd_input_lds(int ic, int gp, int G, int nshare, int bf_O = 0) {
    const int kct = 32 * ((ic * gp + 31) / 32);
    const size_t ops = bf_O ? ((size_t)BM * (bf_O + 8) / 2 + (size_t)(bf_O / 8) * kct * 4)
                            : ((size_t)BIN_NC * AS + (size_t)BIN_NC * (kct + 1));
    return sizeof(float) * ((size_t)BM * (ic | 1) * (FAM == KV_RBF ? 8 : 4) + (FAM == KV_SINE ? (size_t)nshare * 4 * G : 0) +
                            (size_t)kct * AS + 2 * ops);
}

template <int FAM, int KT, bool SHARED, bool BF>
int launch_bwd_input(const LayerArgs& a, hipStream_t st) {
    const size_t lds = bwd_input_lds<FAM>(a.IC, a.GP, a.G, a.groups / a.xmod, BF ? a.O : 0);
    KV_ALLOW_LDS(160 * 1024, (kan_bwd_input_kernel<FAM, KT, SHARED, BF>));
    dim3 grid((unsigned)a.xmod, (unsigned)((a.M + BM - 1) / BM), 1);
    hipLaunchKernelGGL((kan_bwd_input_kernel<FAM, KT, SHARED, BF>), grid, dim3(NTHR), lds, st, a);
    KV_LAUNCH_CHECK("kan_bwd_input_kernel");
    return 0;
}

template <int FAM, bool SHARED>
int launch_bwd_input_kt(const LayerArgs& a, int kt, bool bf, hipStream_t st) {
    if (bf) {
        if (kt == 1) return launch_bwd_input<FAM, 1, SHARED, true>(a, st);
        if (kt == 2) return launch_bwd_input<FAM, 2, SHARED, true>(a, st);
        return launch_bwd_input<FAM, 3, SHARED, true>(a, st);
    }
    if (kt == 1) return launch_bwd_input<FAM, 1, SHARED, false>(a, st);
    if (kt == 2) return launch_bwd_input<FAM, 2, SHARED, false>(a, st);
    return launch_bwd_input<FAM, 3, SHARED, false>(a, st);
}

// chunking of the input-gradient kernel (shared by the workspace query and the launch)
template <int FAM>
int bwd_input_ic(int I, int gp, int G, int nshare, int bf_O) {
    int ic = 96 / gp;
    if (ic < 1) ic = 1;
    if (ic > I) ic = I;
    while (ic > 1 && bwd_input_lds<FAM>(ic, gp, G, nshare, bf_O) > 160 * 1024) --ic;
    return bwd_input_lds<FAM>(ic, gp, G, nshare, bf_O) > 160 * 1024 ? 0 : ic;
}

bool bwd_input_bf16_ok(const kanvit_layer_desc* d);

template <int FAM>
size_t bwd_input_ws(const kanvit_layer_desc* d) {
    const int gp = gp_of(d), nshare = d->groups / d->x_group_mod;
    const int ic = bwd_input_ic<FAM>(d->I, gp, d->G, nshare, d->O);
    if (!ic) return 0;
    const int kct = 32 * ((ic * gp + 31) / 32), nci = (d->I + ic - 1) / ic;
    return (size_t)d->groups * nci * (d->O / 8) * kct * 16;
}

// ---- register-form input gradient (fp32 exact) ---------------------------------------------------
template <int FAM, int GP, int KT>
int launch_bwd_input_reg(const LayerArgs& a, hipStream_t st) {
    constexpr int FPH = (16 * KT) / GP, IC = 2 * FPH;
    const int nshare = a.groups / a.xmod;
    if (a.I % IC || a.O % 32) return 1;
    if ((a.ldx & 3) || (a.ldy & 3) || (a.O & 3) || (FPH & 3 ? false : ((a.I & 3) != 0)) || ((uintptr_t)a.x & 15) ||
        ((uintptr_t)a.dx & 15) || ((uintptr_t)a.dy & 15) || ((uintptr_t)a.w & 15))
        return 1;
    if ((long long)IC * GP * a.O >= (1LL << 30)) return 1;
    constexpr int KCT_ = 32 * KT, HOFF_ = (KCT_ % 64 == 32) ? KCT_ : KCT_ + 32, WS2_ = ((HOFF_ + KCT_ + 13) / 16) * 16 + 2;
    const size_t lds = sizeof(float) * (2 * 16 * WS2_ + (FAM == KV_SINE ? (size_t)nshare * 4 * GP : 0));
    if (FAM == KV_SINE && !a.dparam) return 1;
    const bool shared = kv_shared_basis<FAM>() && kv_share_ok(FAM, a.flags) && nshare > 1;
    dim3 grid((unsigned)a.xmod, (unsigned)((a.M + BM - 1) / BM), 1);
    if (shared) {
        if constexpr (kv_shared_basis<FAM>()) {
            hipLaunchKernelGGL((kan_bwd_input_reg_kernel<FAM, GP, KT, true>), grid, dim3(256), lds, st, a);
            KV_LAUNCH_CHECK("kan_bwd_input_reg_kernel");
            return 0;
        }
    }
    hipLaunchKernelGGL((kan_bwd_input_reg_kernel<FAM, GP, KT, false>), grid, dim3(256), lds, st, a);
    KV_LAUNCH_CHECK("kan_bwd_input_reg_kernel");
    return 0;
}

// returns 1 when not covered (fall back to the LDS-tile kernel)
template <int FAM>
int try_bwd_input_reg(const LayerArgs& a, hipStream_t st) {
    if (kv_config().no_reg) return 1;
    if constexpr (FAM == KV_LINEAR) { if (a.GP == 1) return launch_bwd_input_reg<FAM, 1, 2>(a, st); }
    if constexpr (FAM == KV_CHEBY) { if (a.GP == 5) return launch_bwd_input_reg<FAM, 5, 5>(a, st); }
    if constexpr (FAM == KV_BSPLINE) {
        if (a.GP == 9 && (a.flags & KANVIT_FLAG_UNIFORM_KNOTS) && a.order == 3) return launch_bwd_input_reg<FAM, 9, 5>(a, st);
    }
    if constexpr (FAM == KV_RBF) { if (a.GP == 9 && a.has_base && kv_rbf_reg_ok(a.flags, a.G)) return launch_bwd_input_reg<FAM, 9, 5>(a, st); }
    if constexpr (FAM == KV_FOURIER) { if (a.GP == 56) return launch_bwd_input_reg<FAM, 56, 7>(a, st); }
    if constexpr (FAM == KV_SINE) {   // attention.py:140 builds the per-head sine mappings with grid_size = 4; 5 is the layer's default
        if (a.GP == 4) return launch_bwd_input_reg<FAM, 4, 4>(a, st);
        if (a.GP == 5) return launch_bwd_input_reg<FAM, 5, 5>(a, st);
        if (a.GP == 28) return launch_bwd_input_reg<FAM, 28, 7>(a, st);      // the G = 28 patch embedding (model.py:72)
    }
    return 1;
}

// ---- register-form input gradient on the bf16 matrix cores -----------------------------------------
struct BwdRegBf16Plan {
    bool ok;
    int gp, kt, fph, nci;
    int vcols;            // > 0: one wide layer (groups = 1, O = 64*vcols) contracted 64 columns at a time into the same accumulators
    size_t lds, ws_bytes;
};

BwdRegBf16Plan plan_bwd_input_reg_bf16(const kanvit_layer_desc* d) {
    BwdRegBf16Plan p{};
    if (kv_config().no_reg || kv_config().no_bf16 || !(d->flags & KANVIT_FLAG_BF16_MFMA)) return p;
    p.gp = gp_of(d);
    const int fam = d->family;
    if (fam == KANVIT_LINEAR && p.gp == 1) p.kt = 2;
    else if (fam == KANVIT_CHEBY && p.gp == 5) p.kt = 5;
    else if (fam == KANVIT_BSPLINE && p.gp == 9 && (d->flags & KANVIT_FLAG_UNIFORM_KNOTS) && d->spline_order == 3) p.kt = 5;
    else if (fam == KANVIT_RBF && p.gp == 9 && d->has_base && kv_rbf_reg_ok(d->flags, d->G)) p.kt = 5;
    else if (fam == KANVIT_SINE && p.gp == 4) p.kt = 4;        // the per-head mappings (attention.py:140)
    else if (fam == KANVIT_SINE && p.gp == 28) p.kt = 7;       // the G = 28 patch embedding (model.py:72)
    else return p;
    p.fph = 16 * p.kt / p.gp;
    const int ic = 2 * p.fph;
    const bool wide = d->groups == 1 && d->x_group_mod == 1 && d->O > 64 && d->O % 64 == 0 && d->O <= 64 * 64;
    if (d->I % ic || !(d->O == 32 || d->O == 64 || wide)) return p;
    if ((d->ldx & 3) || (d->ldy & 3) || (d->I & 3) || (fam == KANVIT_RBF && (d->ldu & 3))) return p;
    p.nci = d->I / ic;
    p.vcols = wide ? d->O / 64 : 0;
    const int oc = wide ? 64 : d->O;                 // columns per step
    p.lds = (size_t)2 * (oc / 16) * 2 * 32 * p.kt * 16;
    if (fam == KANVIT_SINE) p.lds += sizeof(float) * (size_t)(wide ? 1 : d->groups / d->x_group_mod) * 4 * p.gp;
    p.ws_bytes = (size_t)d->groups * p.nci * (d->O / 16) * 2 * 32 * p.kt * 16;
    p.ok = true;
    return p;
}

bool bwd_input_bf16_ok(const kanvit_layer_desc* d) {
    const bool wide = d->groups == 1 && d->x_group_mod == 1 && d->O > 64 && d->O % 64 == 0 && d->O <= 64 * 64;      // register kernel only
    // SINE has no bf16 register kernel (its d loss / d freq partials), and the bf16 LDS-tile kernel measures SLOWER than the exact
    // fp32 register kernel (0.81 vs 0.38 ms on the ViT-B q|k|v launch): the flag allows bf16, it does not require it
    if (d->family == KANVIT_SINE && !kv_config().no_reg && !plan_bwd_input_reg_bf16(d).ok) return false;
    return (d->flags & KANVIT_FLAG_BF16_MFMA) && (d->O == 16 || d->O == 32 || d->O == 64 || wide) && (d->ldy % 4 == 0) && !kv_config().no_bf16;
}

template <int FAM, int GP, int KT>
int launch_bwd_input_reg_bf16(LayerArgs& a, const BwdRegBf16Plan& p, hipStream_t st) {
    const long long total = (long long)a.groups * p.nci * (a.O / 16) * 2 * 32 * KT;
    if (p.vcols) {
        // one wide layer (the patch embedding: O = 384 / 768): its 64-column chunks are contracted one per step into the SAME
        // accumulators -- exactly the SHARED schedule with the chunks in the role of the groups that share x and the basis
        hipLaunchKernelGGL(kan_pack_w_bwd_reg_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.w,
                           const_cast<unsigned short*>(a.wb2), a.K, 64, GP, p.fph, 32 * KT, p.nci, total, (long long)a.O, 64LL);
        KV_LAUNCH_CHECK("kan_pack_w_bwd_reg_kernel");
        LayerArgs v = a;
        v.groups = p.vcols;
        v.xmod = 1;
        v.O = 64;
        v.vcols = 1;
        dim3 vgrid(1, (unsigned)((a.M + BM - 1) / BM), 1);
        hipLaunchKernelGGL((kan_bwd_input_reg_bf16_kernel<FAM, GP, KT, true>), vgrid, dim3(256), p.lds, st, v);
        KV_LAUNCH_CHECK("kan_bwd_input_reg_bf16_kernel");
        return 0;
    }
    hipLaunchKernelGGL(kan_pack_w_bwd_reg_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.w,
                       const_cast<unsigned short*>(a.wb2), a.K, a.O, GP, p.fph, 32 * KT, p.nci, total, (long long)a.O, (long long)a.K * a.O);
    KV_LAUNCH_CHECK("kan_pack_w_bwd_reg_kernel");
    const int nshare = a.groups / a.xmod;
    const bool shared = kv_shared_basis<FAM>() && kv_share_ok(FAM, a.flags) && nshare > 1;
    dim3 grid((unsigned)a.xmod, (unsigned)((a.M + BM - 1) / BM), 1);
    if (shared) {
        if constexpr (kv_shared_basis<FAM>()) {
            hipLaunchKernelGGL((kan_bwd_input_reg_bf16_kernel<FAM, GP, KT, true>), grid, dim3(256), p.lds, st, a);
            KV_LAUNCH_CHECK("kan_bwd_input_reg_bf16_kernel");
            return 0;
        }
    }
    hipLaunchKernelGGL((kan_bwd_input_reg_bf16_kernel<FAM, GP, KT, false>), grid, dim3(256), p.lds, st, a);
    KV_LAUNCH_CHECK("kan_bwd_input_reg_bf16_kernel");
    return 0;
}

template <int FAM>
int dispatch_bwd_input_reg_bf16(LayerArgs& a, const BwdRegBf16Plan& p, hipStream_t st) {
    if constexpr (FAM == KV_LINEAR) return launch_bwd_input_reg_bf16<FAM, 1, 2>(a, p, st);
    if constexpr (FAM == KV_CHEBY) return launch_bwd_input_reg_bf16<FAM, 5, 5>(a, p, st);
    if constexpr (FAM == KV_BSPLINE) return launch_bwd_input_reg_bf16<FAM, 9, 5>(a, p, st);
    if constexpr (FAM == KV_RBF) return launch_bwd_input_reg_bf16<FAM, 9, 5>(a, p, st);
    if constexpr (FAM == KV_SINE) {
        if (!a.dparam) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: SINE needs dparam");
        return p.gp == 28 ? launch_bwd_input_reg_bf16<FAM, 28, 7>(a, p, st) : launch_bwd_input_reg_bf16<FAM, 4, 4>(a, p, st);
    }
    return kv_fail(KANVIT_EINVAL, "internal: bf16 register input-gradient dispatch");
}

template <int FAM>
int dispatch_bwd_input(LayerArgs& a, hipStream_t st) {
    if ((long long)BM * a.ldx >= (1LL << 30) || (long long)BM * a.ldy >= (1LL << 30) || (long long)BM * a.ldu >= (1LL << 30) ||
        (long long)a.K * a.O >= (1LL << 30))
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: row strides / weight slab too large for 32-bit tile offsets");
    const int nshare = a.groups / a.xmod;
    bool bf = a.wb2 != nullptr;                         // set by the entry point when the bf16 path applies
    if (!bf) {
        const int rc = try_bwd_input_reg<FAM>(a, st);   // register-form kernel when the shape allows it
        if (rc <= 0) return rc;
        if (a.ln) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: KANVIT_FLAG_FUSED_LN needs the register kernel (alignment / shape)");
    }
    int ic = bwd_input_ic<FAM>(a.I, a.GP, a.G, nshare, bf ? a.O : 0);
    if (!ic && bf) {
        bf = false;
        ic = bwd_input_ic<FAM>(a.I, a.GP, a.G, nshare, 0);
    }
    if (!ic) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: tile does not fit the LDS");
    a.IC = ic;
    const int kt = (ic * a.GP + 31) / 32;
    if (bf) {
        const int kct = 32 * kt, nci = (a.I + ic - 1) / ic;
        const long long total = (long long)a.groups * nci * (a.O / 8) * kct;
        hipLaunchKernelGGL(kan_pack_w_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.w,
                           const_cast<unsigned short*>(a.wb2), a.K, a.O, ic * a.GP, kct, nci, total);
        KV_LAUNCH_CHECK("kan_pack_w_bwd_kernel");
    }
    if constexpr (kv_shared_basis<FAM>()) {
        if (kv_share_ok(FAM, a.flags) && nshare > 1) return launch_bwd_input_kt<FAM, true>(a, kt, bf, st);
    }
    return launch_bwd_input_kt<FAM, false>(a, kt, bf, st);
}

// ---- backward weight -----------------------------------------------------------------------------
struct BwPlan {
    int ic, nfchunks, nchunks_n, msplit, nsh;
    long long rows_per_split;
};

BwPlan plan_bwd_weight(const kanvit_layer_desc* d) {
    BwPlan p;
    const int gp = gp_of(d);
    const int nshare = d->groups / d->x_group_mod;
    const bool shared_fam = kv_share_ok(d->family, d->flags);
    p.nsh = (shared_fam && nshare == 3) ? 3 : 1;
    // NSH = 3: 6 column tiles per k tile, at most 8 tiles per wave -> KT <= 5 (KC <= 160)
    const int kcmax = p.nsh == 3 ? 160 : BW_KC_MAX;
    int ic = 64;
    while (ic > 1 && (ic * gp > kcmax || ic / 2 >= d->I)) ic >>= 1;
    p.ic = ic;
    p.nfchunks = (d->I + ic - 1) / ic;
    p.nchunks_n = (d->O + 32 * BW_NT - 1) / (32 * BW_NT);
    const long long base = (long long)p.nfchunks * p.nchunks_n * (d->groups / p.nsh);
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

template <int FAM, int NSH, bool BF>
int launch_bwd_weight_n(const LayerArgs& a, const BwPlan& p, hipStream_t st) {
    const int ICP = a.IC | 1;
    const int KT = (a.IC * a.GP + 31) / 32;
    const size_t lds = sizeof(float) * 2 * ((size_t)BW_ROWS * ICP * (FAM == KV_RBF ? 2 : 1) + (size_t)BW_ROWS * 32 * BW_NT * NSH +
                                            (size_t)KT * 32 * BW_AS);
    if (lds > 160 * 1024) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: tile does not fit the LDS");
    if (KT * BW_NT * NSH > 4 * ((NSH == 1) ? BW_TPW : 8))
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: internal tiling error");
    KV_ALLOW_LDS(160 * 1024, kan_bwd_weight_kernel<FAM, NSH, BF>);
    dim3 grid((unsigned)p.nfchunks, (unsigned)p.msplit, (unsigned)((a.groups / NSH) * p.nchunks_n));
    hipLaunchKernelGGL((kan_bwd_weight_kernel<FAM, NSH, BF>), grid, dim3(NTHR), lds, st, a);
    KV_LAUNCH_CHECK("kan_bwd_weight_kernel");
    return 0;
}

template <int FAM>
int launch_bwd_weight(const LayerArgs& a, const BwPlan& p, bool bf, hipStream_t st) {
    if ((long long)BW_ROWS * a.ldy >= (1LL << 30))
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: ldy too large for 32-bit tile offsets");
    if (p.nsh == 3) {
        if constexpr (kv_shared_basis<FAM>())
            return bf ? launch_bwd_weight_n<FAM, 3, true>(a, p, st) : launch_bwd_weight_n<FAM, 3, false>(a, p, st);
    }
    return bf ? launch_bwd_weight_n<FAM, 1, true>(a, p, st) : launch_bwd_weight_n<FAM, 1, false>(a, p, st);
}

// ---- register-form (streaming) weight gradient -----------------------------------------------------
struct BwRegPlan {
    bool ok;
    int gp, nt, nfb, nos, tiles_per_bg, nbg, shared, slabs, njc;
    long long rows_per_slab;
    size_t ws_bytes;
};

BwRegPlan plan_bwd_weight_reg(const kanvit_layer_desc* d) {
    BwRegPlan p{};
    if (kv_config().no_reg || kv_config().no_reg_bw) return p;
    p.njc = 1;
    p.gp = gp_of(d);
    const int fam = d->family;
    if (fam == KANVIT_LINEAR && p.gp == 1) p.nt = 6;
    else if (fam == KANVIT_CHEBY && p.gp == 5) p.nt = 3;
    // BSPLINE (GP = 9 -> one column tile per wave): every column-tile wave re-evaluates the spline basis, and measured
    // it loses to the LDS-tile kernel (2.46 vs 1.15 ms on the ViT-B q|k|v launch) -- opt-in only until that is fixed
    else if (fam == KANVIT_BSPLINE && p.gp == 9 && (d->flags & KANVIT_FLAG_UNIFORM_KNOTS) && d->spline_order == 3 &&
             kv_config().reg_bw_bspline) p.nt = 2;
    else if (fam == KANVIT_RBF && p.gp == 9 && d->has_base && kv_rbf_reg_ok(d->flags, d->G)) p.nt = 2;      // (windows of 3 measured slower for both: the basis is re-evaluated per window)
    else if (fam == KANVIT_SINE && (p.gp == 4 || p.gp == 5)) p.nt = 2;
    else if (fam == KANVIT_SINE && p.gp == 28) { p.nt = 4; p.njc = 7; }          // windows of 4 basis functions
    else if (fam == KANVIT_FOURIER && p.gp == 56) { p.nt = 4; p.njc = 14; }
    else return p;
    if (d->I % 32 || d->O % 32 || d->M < 256) return p;
    const int nshare = d->groups / d->x_group_mod;
    p.shared = (kv_share_ok(fam, d->flags) && nshare > 1) ? 1 : 0;
    p.nbg = p.shared ? d->x_group_mod : d->groups;
    p.tiles_per_bg = (p.shared ? nshare : 1) * (d->O / 32);
    p.nfb = d->I / 32;
    p.nos = (p.tiles_per_bg + p.nt - 1) / p.nt;
    // Small launches (the T / C geometries: 6400 rows, 2 heads): even at the shortest slab (64 tokens) the wave units cannot
    // fill the chip, and a wave's MFMA chain (tokens x GP x NOT) IS the kernel time.  One column tile per wave instead of
    // three: three times the waves, a third of the chain each; the basis is re-evaluated per wave (cheap against the chain).
    if (fam == KANVIT_CHEBY && p.nt == 3 && (long long)p.nbg * p.nfb * p.nos * (d->M / 64) < 4LL * N_CU) {
        p.nt = 1;
        p.nos = p.tiles_per_bg;
    }
    // one live wave per SIMD (the accumulator block fills the register file): size the slab count so that the live waves
    // (work-groups whose 2x2 wave grid is only partly populated retire their idle waves at once) cover the chip r times
    const long long units = (long long)p.nbg * p.nfb * p.nos * p.njc;
    long long r = 1;
    while (4LL * N_CU * r < units) ++r;
    long long S = 4LL * N_CU * r / units;
    const long long smax = d->M / 64;             // at least 64 tokens per slab (small M: parallelism beats slab traffic)
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    if (S > 65535) S = 65535;
    const bool ln = fam == KANVIT_RBF && (d->flags & KANVIT_FLAG_FUSED_LN);
    if (ln && S * 4096 < d->M) S = (d->M + 4095) / 4096;      // fused LayerNorm: a wave's (mean, rstd) strip is 8 bytes per slab row of LDS
    if (S > 65535) return p;
    long long rps = (d->M + S - 1) / S;
    rps = (rps + 15) / 16 * 16;
    if (ln && rps > 4096) return p;
    p.rows_per_slab = rps;
    p.slabs = (int)((d->M + rps - 1) / rps);
    if (units > (1LL << 30)) return p;
    {       // 32-bit in-slab element offsets (see the kernel)
        long long ld = d->ldx > d->ldy ? d->ldx : d->ldy;
        if (d->ldu > ld) ld = d->ldu;
        if (rps * ld + ld >= (1LL << 29)) return p;
    }
    p.ws_bytes = p.slabs > 1 ? sizeof(float) * (size_t)p.slabs * d->groups * ((size_t)d->I * p.gp) * d->O : 0;
    p.ok = true;
    return p;
}

template <int FAM, int GP, int NOT, int JC = GP>
int launch_bwd_weight_reg(LayerArgs& a, const BwRegPlan& p, bool bf, hipStream_t st) {
    const long long units = (long long)p.nbg * p.nfb * p.nos * p.njc;
    dim3 grid((unsigned)((units * p.slabs + 3) / 4), 1, 1);
    const size_t lds = a.ln ? (size_t)4 * p.rows_per_slab * sizeof(float2) : 0;      // four wave-private (mean, rstd) strips
    if (lds > 64 * 1024) {
        if (bf) KV_ALLOW_LDS(160 * 1024, (kan_bwd_weight_reg_kernel<FAM, GP, NOT, true, JC>));
        else KV_ALLOW_LDS(160 * 1024, (kan_bwd_weight_reg_kernel<FAM, GP, NOT, false, JC>));
    }
    if (bf)
        hipLaunchKernelGGL((kan_bwd_weight_reg_kernel<FAM, GP, NOT, true, JC>), grid, dim3(256), lds, st, a, p.nfb, p.nos, p.tiles_per_bg, p.shared, p.nbg);
    else
        hipLaunchKernelGGL((kan_bwd_weight_reg_kernel<FAM, GP, NOT, false, JC>), grid, dim3(256), lds, st, a, p.nfb, p.nos, p.tiles_per_bg, p.shared, p.nbg);
    KV_LAUNCH_CHECK("kan_bwd_weight_reg_kernel");
    return 0;
}

int dispatch_bwd_weight_reg(int family, LayerArgs& a, const BwRegPlan& p, bool bf, hipStream_t st) {
    switch (family) {
        case KANVIT_LINEAR: return launch_bwd_weight_reg<KV_LINEAR, 1, 6>(a, p, bf, st);
        case KANVIT_CHEBY: return p.nt == 1 ? launch_bwd_weight_reg<KV_CHEBY, 5, 1>(a, p, bf, st) : launch_bwd_weight_reg<KV_CHEBY, 5, 3>(a, p, bf, st);
        case KANVIT_BSPLINE: return launch_bwd_weight_reg<KV_BSPLINE, 9, 2>(a, p, bf, st);
        case KANVIT_RBF: return launch_bwd_weight_reg<KV_RBF, 9, 2>(a, p, bf, st);
        case KANVIT_SINE:
            if (a.GP == 28) return launch_bwd_weight_reg<KV_SINE, 28, 4, 4>(a, p, bf, st);
            return a.GP == 4 ? launch_bwd_weight_reg<KV_SINE, 4, 2>(a, p, bf, st) : launch_bwd_weight_reg<KV_SINE, 5, 2>(a, p, bf, st);
        case KANVIT_FOURIER: return launch_bwd_weight_reg<KV_FOURIER, 56, 4, 4>(a, p, bf, st);
        default: return kv_fail(KANVIT_EINVAL, "internal: register weight-gradient dispatch");
    }
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

static KvTinyArgs tiny_args(const kanvit_layer_desc* d) {
    KvTinyArgs t{};
    t.M = d->M; t.ldx = d->ldx; t.ldy = d->ldy; t.bp_stride = d->bparam_stride;
    t.family = d->family; t.I = d->I; t.O = d->O; t.groups = d->groups; t.xmod = d->x_group_mod; t.G = d->G;
    t.GP = gp_of(d); t.K = d->I * t.GP; t.order = d->spline_order; t.nk = d->G + d->spline_order + 1;
    t.has_base = d->has_base; t.flags = d->flags;
    return t;
}

thread_local char g_kanvit_err[512] = "";

// ---- run-time switches: read once, reported, never consulted through getenv on the launch path ----
static KvConfig g_kv_config;
static int g_kv_config_state = 0;      // 0 = not loaded
static void kv_config_load() {
    KvConfig c{};
    auto flag = [](const char* n) { const char* v = getenv(n); return (v && *v && !(v[0] == '0' && !v[1])) ? 1 : 0; };
    auto num = [](const char* n) { const char* v = getenv(n); return v ? atoi(v) : 0; };
    c.no_reg = flag("KANVIT_NO_REG");
    c.no_reg_bw = flag("KANVIT_NO_REG_BW");
    c.reg_bw_bspline = flag("KANVIT_REG_BW_BSPLINE");
    c.no_fast = flag("KANVIT_NO_FAST");
    c.no_pipe = flag("KANVIT_NO_PIPE");
    c.no_ws = flag("KANVIT_NO_WS");
    c.no_bf16 = flag("KANVIT_NO_BF16");
    c.no_fused_ln = flag("KANVIT_NO_FUSED_LN");
    c.no_tiny = flag("KANVIT_NO_TINY");
    c.attn_v1 = flag("KANVIT_ATTN_V1");
    c.attn_v2 = flag("KANVIT_ATTN_V2");
    c.attn_no_ds = flag("KANVIT_ATTN_NO_DS");
    c.attn_grid = num("KANVIT_ATTN_GRID");
    c.ff_grid = num("KANVIT_FF_GRID");
    c.bf16_nsh = num("KANVIT_BF16_NSH");
    c.bf16_ic = num("KANVIT_BF16_IC");
    snprintf(c.text, sizeof(c.text),
             "no_reg=%d no_reg_bw=%d reg_bw_bspline=%d no_fast=%d no_pipe=%d no_ws=%d no_bf16=%d no_fused_ln=%d no_tiny=%d attn_v1=%d attn_v2=%d attn_no_ds=%d attn_grid=%d bf16_nsh=%d bf16_ic=%d ff_grid=%d",
             c.no_reg, c.no_reg_bw, c.reg_bw_bspline, c.no_fast, c.no_pipe, c.no_ws, c.no_bf16, c.no_fused_ln, c.no_tiny, c.attn_v1, c.attn_v2, c.attn_no_ds, c.attn_grid, c.bf16_nsh, c.bf16_ic, c.ff_grid);
    g_kv_config = c;
    __atomic_store_n(&g_kv_config_state, 1, __ATOMIC_RELEASE);
}
const KvConfig& kv_config() {
    if (!__atomic_load_n(&g_kv_config_state, __ATOMIC_ACQUIRE)) kv_config_load();     // idempotent: a race loads the same values twice
    return g_kv_config;
}

extern "C" {

const char* kanvit_last_error(void) { return g_kanvit_err; }
const char* kanvit_config(void) { return kv_config().text; }
int kanvit_config_reload(void) {
    kv_config_load();
    return 0;
}
int kanvit_abi_version(void) { return KANVIT_ABI_VERSION; }
int kanvit_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return kv_fail(KANVIT_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

size_t kanvit_layer_fwd_workspace(const kanvit_layer_desc* d) {
    if (!d || !(d->flags & KANVIT_FLAG_BF16_MFMA) || gp_of(d) < 1 || d->groups < 1 || d->x_group_mod < 1 || d->I < 1 || d->O < 1)
        return 0;
    const FwdRegBf16Plan pr = plan_fwd_reg_bf16(d);
    const FwdBf16Plan p = plan_fwd_bf16(d);          // fallback when the register kernel's alignment checks fail at launch
    const size_t a1 = pr.ok ? pr.ws_bytes : 0, a2 = p.ok ? p.ws_bytes : 0;
    return a1 > a2 ? a1 : a2;
}

int kanvit_layer_fwd(const kanvit_layer_desc* d, const float* x, const float* u, const float* w, const float* bparams,
                     const float* bias, float* y, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = validate(d, "kanvit_layer_fwd")) return rc;
    if (d->M == 0) return 0;
    if (!x || !w || !y) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: null x/w/y");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: family %d needs bparams", d->family);
    if (d->family == KANVIT_RBF && u && d->ldu < (int64_t)d->groups * d->I)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: ldu < groups*I");
    if (d->M == 0) return 0;
    if (kv_tiny_ok(d)) {              // tiny per-head layers (I, O <= 16): vector-pipe kernels, csrc/kan_tiny.hip
        KvTinyArgs t = tiny_args(d);
        t.x = x; t.w = w; t.bp = bparams; t.bias = bias; t.y = y;
        return kv_tiny_fwd(t, (hipStream_t)stream);
    }
    LayerArgs a = base_args(d);
    a.x = x;
    a.u = u;
    a.w = w;
    a.bp = bparams;
    a.bias = bias;
    a.y = y;
    if (a.ln) {                       // the u slot carries the statistics buffer [M][x_group_mod][2] (written here)
        if (!u || ((uintptr_t)u & 7)) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: KANVIT_FLAG_FUSED_LN needs the (8-byte aligned) statistics buffer in the u argument");
        a.stats = const_cast<float*>(u);
        a.u = nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    if ((d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16) {
        const FwdRegBf16Plan pr = plan_fwd_reg_bf16(d);
        if (pr.ok && !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)(u ? u : x) | (uintptr_t)(bias ? bias : x)) & 15)) {
            if (!workspace || workspace_bytes < pr.ws_bytes || ((uintptr_t)workspace & 15))
                return kv_fail(KANVIT_ENOMEM, "kanvit_layer_fwd: workspace %zu bytes < required %zu (or not 16-byte aligned)",
                               workspace_bytes, pr.ws_bytes);
#define KV_CALL(F) dispatch_fwd_reg_bf16<F>(a, pr, workspace, st)
            KV_FAMILY_SWITCH(d->family, KV_CALL)
#undef KV_CALL
        }
        const FwdBf16Plan p = plan_fwd_bf16(d);
        if (p.ok && !a.ln) {
            if (!workspace || workspace_bytes < p.ws_bytes || ((uintptr_t)workspace & 15))
                return kv_fail(KANVIT_ENOMEM, "kanvit_layer_fwd: workspace %zu bytes < required %zu (or not 16-byte aligned)",
                               workspace_bytes, p.ws_bytes);
#define KV_CALL(F) dispatch_fwd_bf16<F>(a, p, workspace, st)
            KV_FAMILY_SWITCH(d->family, KV_CALL)
#undef KV_CALL
        }
    }
#define KV_CALL(F) dispatch_fwd<F>(a, st)
    KV_FAMILY_SWITCH(d->family, KV_CALL)
#undef KV_CALL
}

/* 1 when the three register kernels that can form the FastKAN LayerNorm in-kernel cover this layer (pure host function) */
int kanvit_layer_ln_fusable(const kanvit_layer_desc* d) {
    if (!d || d->family != KANVIT_RBF || !d->has_base || !kv_rbf_reg_ok(d->flags, d->G) || d->groups < 1 || d->x_group_mod < 1) return 0;
    if (d->I % 32 || d->O % 32 || d->M < 256 || (d->ldx & 3) || (d->ldy & 3)) return 0;
    if (d->O > 64 && d->O % 128) return 0;                 // forward column tiling: 32, 64 or multiples of 128
    {                                                      // kanvit_layer_ln_bwd's lane-group layout
        const int ns = d->groups / d->x_group_mod;
        if (d->groups % d->x_group_mod || (ns != 1 && ns != 3) || d->I > (ns == 3 ? 512 : 1024)) return 0;
    }
    if (kv_config().no_reg || kv_config().no_reg_bw || kv_config().no_fused_ln) return 0;
    kanvit_layer_desc e = *d;
    e.flags |= KANVIT_FLAG_FUSED_LN;
    if (!plan_bwd_weight_reg(&e).ok) return 0;
    if ((d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16 && !plan_fwd_reg_bf16(&e).ok) return 0;
    return 1;
}

// ---- fused patch embedding (model.py:111-126 patchify, :144-152 class token + position embedding) ----
static int patch_validate(const kanvit_layer_desc* d, const kanvit_patch_desc* p, const char* who) {
    if (!p) return kv_fail(KANVIT_EINVAL, "%s: null patch descriptor", who);
    if (p->C < 1 || p->H < 1 || p->W < 1 || p->n_patches < 1 || p->H % p->n_patches || p->W % p->n_patches)
        return kv_fail(KANVIT_EINVAL, "%s: image %dx%dx%d is not divisible into %d x %d patches", who, p->C, p->H, p->W, p->n_patches,
                       p->n_patches);
    if (p->prepend_rows != 0 && p->prepend_rows != 1) return kv_fail(KANVIT_EINVAL, "%s: prepend_rows must be 0 or 1", who);
    const long long P = (long long)p->n_patches * p->n_patches, I = (long long)p->C * (p->H / p->n_patches) * (p->W / p->n_patches);
    if (d->groups != 1 || d->x_group_mod != 1) return kv_fail(KANVIT_EINVAL, "%s: one layer per launch (groups = 1)", who);
    if (d->I != I) return kv_fail(KANVIT_EINVAL, "%s: I=%d but a patch has %lld pixels", who, d->I, I);
    if (d->M % P) return kv_fail(KANVIT_EINVAL, "%s: M=%lld is not a whole number of images (%lld patches each)", who, (long long)d->M, P);
    if ((long long)p->C * p->H * p->W >= (1LL << 30)) return kv_fail(KANVIT_EINVAL, "%s: image too large for 32-bit offsets", who);
    if (d->family == KANVIT_RBF) return kv_fail(KANVIT_EINVAL, "%s: FastKAN needs its LayerNorm'ed input u (use kanvit_layer_fwd)", who);
    return 0;
}

static void patch_args(LayerArgs& a, const kanvit_patch_desc* p, const float* cls, const float* pos) {
    a.pg = 1;
    a.pg_C = p->C;
    a.pg_H = p->H;
    a.pg_W = p->W;
    a.pg_n = p->n_patches;
    a.pg_pre = p->prepend_rows;
    a.cls = cls;
    a.pos = pos;
    a.ldx = a.I;                         // unused by the gather; keeps the alignment checks of the dispatcher meaningful
}

int kanvit_patch_embed_fwd(const kanvit_layer_desc* d, const kanvit_patch_desc* p, const float* images, const float* w,
                           const float* bparams, const float* bias, const float* cls, const float* pos, float* y, void* stream) {
    if (int rc = validate(d, "kanvit_patch_embed_fwd")) return rc;
    if (int rc = patch_validate(d, p, "kanvit_patch_embed_fwd")) return rc;
    if (d->M == 0) return 0;
    if (!images || !w || !y) return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: null images/w/y");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: family %d needs bparams", d->family);
    if (p->prepend_rows && !cls) return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: prepend_rows = 1 needs the class token");
    if (((uintptr_t)images | (uintptr_t)(cls ? cls : w) | (uintptr_t)(pos ? pos : w)) & 15)
        return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: images / cls / pos must be 16-byte aligned");
    LayerArgs a = base_args(d);
    a.x = images;
    a.w = w;
    a.bp = bparams;
    a.bias = bias;
    a.y = y;
    patch_args(a, p, cls, pos);
    hipStream_t st = (hipStream_t)stream;
    int rc = 1;
    if (!kv_config().no_reg) {
        switch (d->family) {
            case KANVIT_LINEAR: rc = try_fwd_reg<KV_LINEAR>(a, st); break;
            case KANVIT_CHEBY: rc = try_fwd_reg<KV_CHEBY>(a, st); break;
            case KANVIT_BSPLINE: rc = try_fwd_reg<KV_BSPLINE>(a, st); break;
            case KANVIT_SINE: rc = try_fwd_reg<KV_SINE>(a, st); break;
            case KANVIT_FOURIER: rc = try_fwd_reg<KV_FOURIER>(a, st); break;
            default: break;
        }
    }
    if (rc == 1)
        return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: shape not covered by the fused kernel (O %% 32, patch width %% chunk, "
                                      "basis size); use patchify + kanvit_layer_fwd");
    return rc;
}

int64_t kanvit_layer_dparam_tiles(const kanvit_layer_desc* d) {
    if (!d || d->family != KANVIT_SINE) return 0;
    return (d->M + BM - 1) / BM;
}

size_t kanvit_layer_bwd_input_workspace(const kanvit_layer_desc* d) {
    if (!d || gp_of(d) < 1 || d->groups < 1 || d->x_group_mod < 1 || d->I < 1 || d->O < 1 || !bwd_input_bf16_ok(d)) return 0;
    size_t reg_ws = 0;
    {
        const BwdRegBf16Plan pr = plan_bwd_input_reg_bf16(d);
        if (pr.ok) reg_ws = pr.ws_bytes;
    }
    size_t lds_ws = 0;
    switch (d->family) {
        case KANVIT_LINEAR: lds_ws = bwd_input_ws<KV_LINEAR>(d); break;
        case KANVIT_CHEBY: lds_ws = bwd_input_ws<KV_CHEBY>(d); break;
        case KANVIT_BSPLINE: lds_ws = bwd_input_ws<KV_BSPLINE>(d); break;
        case KANVIT_RBF: lds_ws = bwd_input_ws<KV_RBF>(d); break;
        case KANVIT_SINE: lds_ws = bwd_input_ws<KV_SINE>(d); break;
        case KANVIT_FOURIER: lds_ws = bwd_input_ws<KV_FOURIER>(d); break;
        default: break;
    }
    return reg_ws > lds_ws ? reg_ws : lds_ws;
}

int kanvit_layer_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,
                           const float* bparams, const float* dy, float* dx, float* du, float* dparam, void* workspace,
                           size_t workspace_bytes, void* stream) {
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
    if (kv_tiny_ok(d)) {
        KvTinyArgs t = tiny_args(d);
        t.x = x; t.w = w; t.bp = bparams; t.dy = dy; t.dx = dx;
        return kv_tiny_bwd_input(t, (hipStream_t)stream);
    }
    LayerArgs a = base_args(d);
    a.x = x;
    a.u = u;
    a.w = w;
    a.bp = bparams;
    a.dy = dy;
    a.dx = dx;
    a.du = du;
    a.dparam = dparam;
    a.wb2 = nullptr;
    if (a.ln) {
        if (!u) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: KANVIT_FLAG_FUSED_LN needs the statistics buffer in the u argument");
        a.stats = const_cast<float*>(u);
        a.u = nullptr;
    }
    if (bwd_input_bf16_ok(d) && (((uintptr_t)dy & 15) == 0)) {
        const size_t need = kanvit_layer_bwd_input_workspace(d);
        if (need) {
            if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15))
                return kv_fail(KANVIT_ENOMEM, "kanvit_layer_bwd_input: workspace %zu bytes < required %zu (or not 16-byte aligned)",
                               workspace_bytes, need);
            a.wb2 = (const unsigned short*)workspace;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (a.wb2) {
        const BwdRegBf16Plan pr = plan_bwd_input_reg_bf16(d);
        if (pr.ok && !(((uintptr_t)x | (uintptr_t)dx | (uintptr_t)(u ? u : x) | (uintptr_t)(du ? du : dx)) & 15)) {
#define KV_CALL(F) dispatch_bwd_input_reg_bf16<F>(a, pr, st)
            KV_FAMILY_SWITCH(d->family, KV_CALL)
#undef KV_CALL
        }
        if (a.ln || d->O > 64 || d->family == KANVIT_SINE) a.wb2 = nullptr;     // no LayerNorm fusion / no wide layers in the LDS-tile bf16 kernel: the exact register kernel runs instead
    }
#define KV_CALL(F) dispatch_bwd_input<F>(a, st)
    KV_FAMILY_SWITCH(d->family, KV_CALL)
#undef KV_CALL
}

size_t kanvit_layer_bwd_weight_workspace(const kanvit_layer_desc* d) {
    if (!d || gp_of(d) < 1 || d->groups < 1 || d->I < 1 || d->O < 1) return 0;
    if (kv_tiny_ok(d)) {
        const int s = kv_tiny_slabs(d);
        return s > 1 ? sizeof(float) * (size_t)s * d->groups * ((size_t)d->I * gp_of(d)) * d->O : 0;
    }
    const BwRegPlan pr = plan_bwd_weight_reg(d);
    if (pr.ok) return pr.ws_bytes;
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
    const size_t need = kanvit_layer_bwd_weight_workspace(d);
    if (need > 0 && (!workspace || workspace_bytes < need))
        return kv_fail(KANVIT_ENOMEM, "kanvit_layer_bwd_weight: workspace %zu bytes < required %zu", workspace_bytes, need);
    if (kv_tiny_ok(d)) {
        hipStream_t st = (hipStream_t)stream;
        KvTinyArgs t = tiny_args(d);
        t.x = x; t.bp = bparams; t.dy = dy;
        t.slabs = kv_tiny_slabs(d);
        t.rows_per_slab = ((d->M + t.slabs - 1) / t.slabs + 63) / 64 * 64;
        t.slabs = (int)((d->M + t.rows_per_slab - 1) / t.rows_per_slab);
        t.slab = t.slabs > 1 ? (float*)workspace : dw;
        if (int rc = kv_tiny_bwd_weight(t, st)) return rc;
        if (t.slabs > 1) {
            const long long total = (long long)d->groups * t.K * d->O;
            long long nb = (total + 255) / 256;
            if (nb > 8 * N_CU) nb = 8 * N_CU;
            hipLaunchKernelGGL(kan_slab_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float*)workspace, dw, total, t.slabs);
            KV_LAUNCH_CHECK("kan_slab_reduce_kernel");
        }
        return 0;
    }
    const BwPlan p = plan_bwd_weight(d);
    LayerArgs a = base_args(d);
    a.x = x;
    a.u = u;
    a.bp = bparams;
    a.dy = dy;
    if (a.ln) {
        if (!u) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: KANVIT_FLAG_FUSED_LN needs the statistics buffer in the u argument");
        a.stats = const_cast<float*>(u);
        a.u = nullptr;
    }
    {
        const BwRegPlan pr = plan_bwd_weight_reg(d);
        if (!pr.ok && a.ln) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: KANVIT_FLAG_FUSED_LN needs the register kernel (shape)");
        if (pr.ok) {
            hipStream_t st = (hipStream_t)stream;
            const bool bf = (d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16;
            a.rows_per_split = pr.rows_per_slab;
            a.msplit = pr.slabs;
            a.slab = (pr.slabs > 1) ? (float*)workspace : dw;
            if (int rc = dispatch_bwd_weight_reg(d->family, a, pr, bf, st)) return rc;
            if (pr.slabs > 1) {
                const long long total = (long long)d->groups * a.K * d->O;
                long long nb = (total + 255) / 256;
                if (nb > 8 * N_CU) nb = 8 * N_CU;
                hipLaunchKernelGGL(kan_slab_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float*)workspace, dw, total,
                                   pr.slabs);
                KV_LAUNCH_CHECK("kan_slab_reduce_kernel");
            }
            return 0;
        }
    }
    a.IC = p.ic;
    a.msplit = p.msplit;
    a.nchunks_n = p.nchunks_n;
    a.rows_per_split = p.rows_per_split;
    a.slab = (p.msplit > 1) ? (float*)workspace : dw;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const bool bf = (d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16;
    switch (d->family) {
        case KANVIT_LINEAR: rc = launch_bwd_weight<KV_LINEAR>(a, p, bf, st); break;
        case KANVIT_CHEBY: rc = launch_bwd_weight<KV_CHEBY>(a, p, bf, st); break;
        case KANVIT_BSPLINE: rc = launch_bwd_weight<KV_BSPLINE>(a, p, bf, st); break;
        case KANVIT_RBF: rc = launch_bwd_weight<KV_RBF>(a, p, bf, st); break;
        case KANVIT_SINE: rc = launch_bwd_weight<KV_SINE>(a, p, bf, st); break;
        case KANVIT_FOURIER: rc = launch_bwd_weight<KV_FOURIER>(a, p, bf, st); break;
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
                            const float* bp, const float* bias, float* y, void* ws, size_t wsb, void* s) {         \
        if (!d || d->family != FAMID) return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_fwd: descriptor family mismatch"); \
        return kanvit_layer_fwd(d, x, u, w, bp, bias, y, ws, wsb, s);                                               \
    }                                                                                                               \
    int kanvit_##name##_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,      \
                                  const float* bp, const float* dy, float* dx, float* du, float* dp, void* ws,     \
                                  size_t wsb, void* s) {                                                            \
        if (!d || d->family != FAMID) return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_bwd_input: descriptor family mismatch"); \
        return kanvit_layer_bwd_input(d, x, u, w, bp, dy, dx, du, dp, ws, wsb, s);                                  \
    }                                                                                                               \
    int kanvit_##name##_bwd_weight(const kanvit_layer_desc* d, const float* x, const float* u, const float* bp,    \
                                   const float* dy, float* dw, void* ws, size_t wsb, void* s) {                    \
        if (!d || d->family != FAMID) return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_bwd_weight: descriptor family mismatch"); \
        return kanvit_layer_bwd_weight(d, x, u, bp, dy, dw, ws, wsb, s);                                            \
    }                                                                                                               \
    int kanvit_##name##_qkv_fwd(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,        \
                                const float* bp, const float* bias, float* y, void* ws, size_t wsb, void* s) {     \
        if (!d || d->family != FAMID || d->groups != 3 * d->x_group_mod)                                            \
            return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_qkv_fwd: need family match and groups == 3*x_group_mod"); \
        return kanvit_layer_fwd(d, x, u, w, bp, bias, y, ws, wsb, s);                                               \
    }                                                                                                               \
    int kanvit_##name##_qkv_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,  \
                                      const float* bp, const float* dy, float* dx, float* du, float* dp, void* ws, \
                                      size_t wsb, void* s) {                                                        \
        if (!d || d->family != FAMID || d->groups != 3 * d->x_group_mod)                                            \
            return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_qkv_bwd_input: need family match and groups == 3*x_group_mod"); \
        return kanvit_layer_bwd_input(d, x, u, w, bp, dy, dx, du, dp, ws, wsb, s);                                  \
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
