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
#include "kan_layer_common.h"

namespace {

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

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
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

template <int FAM>
int tile_fwd(LayerArgs& a, hipStream_t st) {
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

template <int FAM>
size_t bwd_input_ws(const kanvit_layer_desc* d) {
    const int gp = gp_of(d), nshare = d->groups / d->x_group_mod;
    const int ic = bwd_input_ic<FAM>(d->I, gp, d->G, nshare, d->O);
    if (!ic) return 0;
    const int kct = 32 * ((ic * gp + 31) / 32), nci = (d->I + ic - 1) / ic;
    return (size_t)d->groups * nci * (d->O / 8) * kct * 16;
}

template <int FAM>
int tile_bwd_input(LayerArgs& a, hipStream_t st) {
    const int nshare = a.groups / a.xmod;
    bool bf = a.wb2 != nullptr;                         // set by the entry point when the bf16 path applies
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


}  // namespace

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

int kv_tile_fwd(int family, LayerArgs& a, hipStream_t st) {
#define KV_CALL(F) tile_fwd<F>(a, st)
    KV_FAMILY_SWITCH(family, KV_CALL)
#undef KV_CALL
}
int kv_tile_fwd_bf16(int family, LayerArgs& a, const FwdBf16Plan& p, void* ws, hipStream_t st) {
#define KV_CALL(F) dispatch_fwd_bf16<F>(a, p, ws, st)
    KV_FAMILY_SWITCH(family, KV_CALL)
#undef KV_CALL
}
int kv_tile_bwd_input(int family, LayerArgs& a, hipStream_t st) {
#define KV_CALL(F) tile_bwd_input<F>(a, st)
    KV_FAMILY_SWITCH(family, KV_CALL)
#undef KV_CALL
}
int kv_tile_bwd_weight(int family, const LayerArgs& a, const BwPlan& p, bool bf, hipStream_t st) {
#define KV_CALL(F) launch_bwd_weight<F>(a, p, bf, st)
    KV_FAMILY_SWITCH(family, KV_CALL)
#undef KV_CALL
}
size_t kv_tile_bwd_input_ws(const kanvit_layer_desc* d) {
    switch (d->family) {
        case KANVIT_LINEAR: return bwd_input_ws<KV_LINEAR>(d);
        case KANVIT_CHEBY: return bwd_input_ws<KV_CHEBY>(d);
        case KANVIT_BSPLINE: return bwd_input_ws<KV_BSPLINE>(d);
        case KANVIT_RBF: return bwd_input_ws<KV_RBF>(d);
        case KANVIT_SINE: return bwd_input_ws<KV_SINE>(d);
        case KANVIT_FOURIER: return bwd_input_ws<KV_FOURIER>(d);
        default: return 0;
    }
}
