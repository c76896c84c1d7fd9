// Register-form input gradient of the fused KAN layers, exact fp32 (split out of kan_layer.hip; see kan_layer_common.h).
#include "kan_layer_common.h"

namespace {

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
    const int nshare = a.groups / a.xmod;
    const int nci = a.I / IC, ncn = a.O / 32;
    // launch tail (kv_tail_first_tile; never set for SINE, whose d freq partials are per row tile): grid rows >= tail_y0 are row
    // tile tail_y0 + t / nci restricted to the ONE feature chunk t % nci -- dx columns are independent, so nothing is summed twice
    int by = (int)blockIdx.y, ci0 = 0, ci1 = nci;
    if (by >= a.tail_y0) {
        const int t = by - a.tail_y0, tl = t / nci;
        ci0 = t - tl * nci;
        ci1 = ci0 + 1;
        by = a.tail_y0 + tl;
    }
    const long long m0 = (long long)by * BM;
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
        for (int q = 0; q < WQ; ++q)      // unconditional (a predicated load is a branch around it and a full vmcnt(0) wait at its use):
            wreg[q] = *reinterpret_cast<const f32x4*>(src + (kofs[q] >= 0 ? kofs[q] : 0) + nofs[q]);      // padding rows re-read row 0, zeroed in store_w
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
                    dst[(n & 15) * WS2 + (n >> 4) * HOFF + kr] = kofs[q] >= 0 ? wreg[q][e] : 0.0f;
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
    const int T = (ci1 - ci0) * spc;
    f32x16 acc[KT];
    float dxacc[FPH];
    float xv[FPH];

    float lnp[RBF ? 2 * FPH : 1];               // FastKAN: (gamma | beta) of the chunk's features, or u (see the loop)
    float2 ln_st = {0.0f, 1.0f};                  // KANVIT_FLAG_FUSED_LN: (mean, rstd) of this lane's row and x slice -- the same for every step
    if constexpr (RBF) {
        if (a.ln) ln_st = *reinterpret_cast<const float2*>(a.stats + (grow * a.xmod + gx) * 2);
    }
    int ci = ci0, p = 0, cn = 0;
    load_w(ci0, gx, 0);
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
        // FastKAN: what the chain rule at the END of this (feature chunk, group) needs besides x -- gamma / beta or u -- is requested
        // here, BEFORE the next step's W / dY prefetch.  Memory returns in order: requested inside the chain rule these loads sat
        // behind the prefetch, and every chain rule began by waiting for it.
        if constexpr (RBF) {
            if (cn == 0) {
                const int g = p * a.xmod + gx;
                if (a.ln) {
                    const float* gb = a.bp + (long long)g * a.bp_stride + a.G + ci * IC + hf * FPH;
#pragma unroll
                    for (int j = 0; j < FPH; ++j) {
                        lnp[j] = gb[j];
                        lnp[FPH + j] = gb[a.I + j];
                    }
                } else {
                    const float* urow = a.u ? a.u + grow * a.ldu + (long long)g * a.I + hf * FPH + ci * IC : xrow + ci * IC;
#pragma unroll
                    for (int j = 0; j < FPH; ++j) lnp[j] = urow[j];
                }
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
#pragma unroll
                    for (int j = 0; j < FPH; ++j) uvv[j] = (xv[j] - ln_st.x) * ln_st.y * lnp[j] + lnp[FPH + j];
                } else {
#pragma unroll
                    for (int j = 0; j < FPH; ++j) uvv[j] = lnp[j];
                }
            }
            float dfq[SINE ? GP : 1];
            if constexpr (SINE) {
#pragma unroll
                for (int g_ = 0; g_ < GP; ++g_) dfq[g_] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < FPH; ++j) {
                BasisDGen<FAM, kv_gc(FAM, GP)> gen;
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
            a.dparam[((long long)by * a.groups + (pp * a.xmod + gx)) * a.G + gg] = v;
        }
    }
}

// ---- register-form input gradient (fp32 exact) ---------------------------------------------------
template <int FAM, int GP, int KT>
int launch_bwd_input_reg(const LayerArgs& a0, hipStream_t st) {
    constexpr int FPH = (16 * KT) / GP, IC = 2 * FPH;
    const int nshare = a0.groups / a0.xmod;
    LayerArgs a = a0;
    if (a.I % IC || a.O % 32) return 1;
    if ((a.ldx & 3) || (a.ldy & 3) || (a.O & 3) || (FPH & 3 ? false : ((a.I & 3) != 0)) || ((uintptr_t)a.x & 15) ||
        ((uintptr_t)a.dx & 15) || ((uintptr_t)a.dy & 15) || ((uintptr_t)a.w & 15))
        return 1;
    if ((long long)IC * GP * a.O >= (1LL << 30)) return 1;
    constexpr int KCT_ = 32 * KT, HOFF_ = (KCT_ % 64 == 32) ? KCT_ : KCT_ + 32, WS2_ = ((HOFF_ + KCT_ + 13) / 16) * 16 + 2;
    const size_t lds = sizeof(float) * (2 * 16 * WS2_ + (FAM == KV_SINE ? (size_t)nshare * 4 * GP : 0));
    if (FAM == KV_SINE && !a.dparam) return 1;
    const bool shared = kv_shared_basis<FAM>() && kv_share_ok(FAM, a.flags) && nshare > 1;
    const long long tiles = (a0.M + BM - 1) / BM;
    const int nci = a.I / IC;
    if (FAM != KV_SINE && nci > 1) a.tail_y0 = kv_tail_first_tile(tiles, a.xmod);
    const long long t1 = a.tail_y0 < tiles ? a.tail_y0 : tiles;
    dim3 grid((unsigned)a.xmod, (unsigned)(t1 + nci * (tiles - t1)), 1);
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
        if (a.GP == 9 && a.has_base && (a.flags & KANVIT_FLAG_UNIFORM_KNOTS) && a.order == 3) return launch_bwd_input_reg<FAM, 9, 5>(a, st);
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


}  // namespace

int kv_try_bwd_input_reg(int family, const LayerArgs& a, hipStream_t st) {
#define KV_CALL(F) try_bwd_input_reg<F>(a, st)
    KV_FAMILY_SWITCH(family, KV_CALL)
#undef KV_CALL
}
