// Register-form input gradient on the bf16 matrix cores (split out of kan_layer.hip; see kan_layer_common.h).
#include "kan_layer_common.h"

namespace {

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
            wreg[q] = *reinterpret_cast<const u32x4*>(src + (long long)(v < NV ? v : 0) * 8);      // unconditional (predicated loads serialise: a branch + vmcnt wait each); lanes past NV re-read vector 0, never stored
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

    float lnp[RBF ? 2 * FPH : 1];               // FastKAN: (gamma | beta) of the step's features, or u (see the loop)
    float2 ln_st = {0.0f, 1.0f};                  // KANVIT_FLAG_FUSED_LN: (mean, rstd) of this lane's row and x slice
    if constexpr (RBF) {
        if (a.ln) ln_st = *reinterpret_cast<const float2*>(a.stats + (grow * a.xmod + gx) * 2);
    }
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
        // FastKAN: the chain rule's gamma / beta (or u) are requested BEFORE the next step's W / dY prefetch -- memory returns in
        // order, and requested inside the chain rule they waited behind the whole prefetch every step
        if constexpr (RBF) {
            const int g = a.vcols ? gx : p * a.xmod + gx;
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
// The same kernel with the dY of its row tile RESIDENT (round 4), for the per-head layers (O = 64, at most three groups sharing x).
// The kernel above asks for the dY columns of step t + 1 during step t: one step is 20 MFMAs of 32 cycles per wave, the memory latency
// several times that, and a work-group lives for nci * nshare = 6 .. 12 steps -- it spent its life waiting (q|k|v of ViT-B ChebyKAN:
// 0.142 ms for 390 MB, 10.7 % matrix-pipe busy).  A row's dY is the same in every feature chunk ci, so it is loaded ONCE, all groups
// in one burst in the prologue, rounded to bf16 (16 registers per group) and kept; W (L2-resident) arrives by LDS-DMA in a ring of
// three step images, two steps ahead (no staging registers, no ds_write, one barrier per step); the next chunk's x a chunk ahead;
// the other work-group of the CU covers the prologue.  Same operands, same roundings, same order of the sums as the kernel above: bitwise the same dx.
// =============================================================================================
template <int FAM, int GP, int KT, int NSH, bool SHARED>
__global__ __launch_bounds__(256, 2) void kan_bwd_input_res_bf16_kernel(const LayerArgs a) {
    static_assert(FAM != KV_SINE, "SineKAN keeps the streaming kernel (its d freq partials)");
    static_assert(NSH == 1 || NSH == 3, "one group, or q|k|v");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KCT = 32 * KT;
    constexpr int FPH = (16 * KT) / GP;
    constexpr int IC = 2 * FPH;
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr int NKS = 4;                        // O = 64 (host-checked)
    constexpr int WSZ = NKS * 2 * KCT * 8;        // bf16 elements per W buffer
    constexpr int NV = NKS * 2 * KCT;             // 16-byte vectors per W buffer
    constexpr int WQ = NV / 256;                  // LDS-DMA instructions per thread and buffer
    static_assert(NV % 256 == 0, "whole DMA instructions (KT * 256 vectors per buffer)");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hf = lane >> 5;
    const int gx = blockIdx.x;
    const int nci = a.I / IC;
    // launch tail (kv_tail_first_tile, as kan_bwd_input_reg_kernel): grid rows >= tail_y0 are row tile tail_y0 + t / nci restricted to the ONE feature
    // chunk t % nci -- dx columns are independent, nothing is summed twice; such a piece still loads its rows' whole dY
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
    unsigned short* W_s = reinterpret_cast<unsigned short*>(smem);     // [3][NKS][2][KCT][8]: a ring of three step images
    // dx / du leave through a wave-private LDS strip [32 rows][IC + 4] as whole row segments (IC floats per row: a 128-byte line for
    // ChebyKAN) -- straight from the registers a store instruction touches 32 rows with two 16-byte pieces each (see kan_fwd_ws_bf16_kernel)
    constexpr int STW = IC + 4, PPR = IC / 4, RPI = 64 / PPR;
    static_assert(FPH % 4 == 0 && 64 % PPR == 0 && 32 % RPI == 0, "whole 16-byte pieces, whole rows per store instruction");
    float* strip = reinterpret_cast<float*>(W_s + (size_t)3 * WSZ) + wave * (32 * STW);
    // (three groups that share only x run a chain rule per group in a rolled loop at the register limit: with the strip code that instantiation
    //  spilled 59 registers and FastKAN's launch went 301 -> 419 us; it keeps the direct stores)
    constexpr bool STRIP = SHARED || NSH == 1;
    auto store_tile = [&](const float (&v)[FPH], float* dst, long long ld) __attribute__((always_inline)) {      // dst: (row m0 + 32 wave, first column of the tile)
#pragma unroll
        for (int j4 = 0; j4 < FPH / 4; ++j4)
            *reinterpret_cast<f32x4*>(strip + l31 * STW + hf * FPH + 4 * j4) = f32x4{v[4 * j4], v[4 * j4 + 1], v[4 * j4 + 2], v[4 * j4 + 3]};
        const int pc = lane % PPR, rs = lane / PPR;
#pragma unroll
        for (int k = 0; k < 32 / RPI; ++k) {
            const int rr = rs + RPI * k;
            const f32x4 t = *reinterpret_cast<const f32x4*>(strip + rr * STW + 4 * pc);
            if (wave * 32 + rr < mrem) *reinterpret_cast<f32x4*>(dst + (long long)rr * ld + 4 * pc) = t;
        }
    };
    const float* xrow = a.x + grow * a.ldx + (long long)gx * a.I + hf * FPH;
    float* dxrow = a.dx + grow * a.ldx + (long long)gx * a.I + hf * FPH;
    const float* dyrow = a.dy + grow * a.ldy + hf * 8;
    const int T = (ci1 - ci0) * NSH;

    // W of step t (feature chunk t / NSH, group t % NSH) -> ring slot t % 3, by LDS-DMA: the packed image is copied as it stands, 16 bytes
    // per lane and instruction, WQ instructions per thread; no register, no ds_write.  Issued two steps ahead (see the loop).
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto fill_w = [&](int t) __attribute__((always_inline)) {
        const int cl = t / NSH, ci = ci0 + cl, g = (t - cl * NSH) * a.xmod + gx;
        const unsigned short* src = a.wb2 + (((long long)g * nci + ci) * NV) * 8 + (long long)lane * 8;      // this lane's 16 bytes of a 1 KiB piece
        unsigned short* dst = W_s + (size_t)(t % 3) * WSZ;
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int piece = wave + 4 * q;           // uniform: 64 consecutive vectors
            __builtin_amdgcn_global_load_lds((glb_ptr)(src + (long long)piece * 512), (lds_ptr)(dst + piece * 512), 16, 0, 0);
        }
    };
    float xv[FPH], xn[FPH];
    auto load_x = [&](int ci, float (&dst)[FPH]) __attribute__((always_inline)) {
        if constexpr (FPH % 4 == 0) {
#pragma unroll
            for (int j4 = 0; j4 < FPH / 4; ++j4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + ci * IC + 4 * j4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[4 * j4 + e] = v[e];
            }
        } else {
#pragma unroll
            for (int j = 0; j < FPH; ++j) dst[j] = xrow[ci * IC + j];
        }
    };

    // ---- prologue: ALL of this row's dY (NSH groups x 64 columns; this lane: its 8 of every 16), x of chunk 0, then W of steps 0 and 1 ----
    // STRIP: the same bytes as whole row segments -- lane = (row rs + RPD k, 16-byte piece pc) of a [32 rows][TW columns] tile, RPD rows of TW * 4
    // contiguous bytes per instruction -- turned into the operand layout (lane = row, 8 of every 16 columns) by a pass through the
    // wave's LDS strip.  Straight into the operand layout a load instruction touches 32 rows with two 16-byte pieces each, a 128-byte
    // line is fetched by four instructions of each of four waves, and the L1 does not hold them: 1.25 GB of L2 requests for the 390 MB
    // this launch moves (TCC_REQ, round 4).
    constexpr int TW = IC >= 32 ? 32 : 16, PPD = TW / 4, RPD = 64 / PPD, NTD = 64 / TW, STD = TW + 4;      // NTD tiles per group
    f32x4 raw[NSH][NKS][2];
    f32x4 craw[STRIP ? NSH : 1][STRIP ? NTD : 1][STRIP ? 32 / RPD : 1];
    if constexpr (STRIP) {
        const int pc = lane % PPD, rs = lane / PPD;
#pragma unroll
        for (int p = 0; p < NSH; ++p)
#pragma unroll
            for (int h2 = 0; h2 < NTD; ++h2)
#pragma unroll
                for (int k = 0; k < 32 / RPD; ++k) {
                    const int rr = wave * 32 + rs + RPD * k;
                    craw[p][h2][k] = *reinterpret_cast<const f32x4*>(a.dy + (m0 + (rr < mrem ? rr : 0)) * a.ldy + (long long)(p * a.xmod + gx) * a.O + TW * h2 + 4 * pc);
                }
    } else {
#pragma unroll
        for (int p = 0; p < NSH; ++p) {
            const float* src = dyrow + (long long)(p * a.xmod + gx) * a.O;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                raw[p][ks][0] = *reinterpret_cast<const f32x4*>(src + 16 * ks);
                raw[p][ks][1] = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4);
            }
        }
    }
    load_x(ci0, xn);
    float2 ln_st = {0.0f, 1.0f};                  // KANVIT_FLAG_FUSED_LN: (mean, rstd) of this lane's row and x slice
    if constexpr (RBF) {
        if (a.ln) ln_st = *reinterpret_cast<const float2*>(a.stats + (grow * a.xmod + gx) * 2);
    }
    fill_w(0);
    if (T > 1) fill_w(1);
    bf16x8_t dyres[NSH][NKS];
    if constexpr (STRIP) {
        const int pc = lane % PPD, rs = lane / PPD;
#pragma unroll
        for (int p = 0; p < NSH; ++p)
#pragma unroll
            for (int h2 = 0; h2 < NTD; ++h2) {
#pragma unroll
                for (int k = 0; k < 32 / RPD; ++k) *reinterpret_cast<f32x4*>(strip + (rs + RPD * k) * STD + 4 * pc) = craw[p][h2][k];
#pragma unroll
                for (int ksl = 0; ksl < TW / 16; ++ksl) {
                    raw[p][h2 * (TW / 16) + ksl][0] = *reinterpret_cast<const f32x4*>(strip + l31 * STD + 16 * ksl + 8 * hf);
                    raw[p][h2 * (TW / 16) + ksl][1] = *reinterpret_cast<const f32x4*>(strip + l31 * STD + 16 * ksl + 8 * hf + 4);
                }
            }
    }
#pragma unroll
    for (int p = 0; p < NSH; ++p)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const u32x4 u = {kv_pack_bf16(raw[p][ks][0][0], raw[p][ks][0][1]), kv_pack_bf16(raw[p][ks][0][2], raw[p][ks][0][3]),
                             kv_pack_bf16(raw[p][ks][1][0], raw[p][ks][1][1]), kv_pack_bf16(raw[p][ks][1][2], raw[p][ks][1][3])};
            dyres[p][ks] = __builtin_bit_cast(bf16x8_t, u);
        }

    constexpr int UNR = SHARED ? NSH : 1;
    f32x16 acc[KT];
    float dxacc[FPH];
    float lnp[RBF ? 2 * FPH : 1];
    for (int ci = ci0; ci < ci1; ++ci) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kt][r] = 0.0f;
#pragma unroll
        for (int j = 0; j < FPH; ++j) {
            dxacc[j] = 0.0f;
            xv[j] = xn[j];
        }
        if (ci + 1 < ci1) load_x(ci + 1, xn);
        // groups that share the basis (one chain rule per chunk): the group loop is unrolled, dyres[p] a register name; groups that only
        // share x (a chain rule per group -- three copies of it unrolled spilled 46 - 98 registers): a rolled loop that selects its dY
#pragma unroll UNR
        for (int p = 0; p < NSH; ++p) {
            const int t = (ci - ci0) * NSH + p;
            const int g = p * a.xmod + gx;
            bf16x8_t dyb[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if constexpr (SHARED || NSH == 1) {
                    dyb[ks] = dyres[p < NSH ? p : 0][ks];
                } else {
                    const u32x4 v0 = __builtin_bit_cast(u32x4, dyres[0][ks]), v1 = __builtin_bit_cast(u32x4, dyres[NSH > 1 ? 1 : 0][ks]),
                                v2 = __builtin_bit_cast(u32x4, dyres[NSH > 2 ? 2 : 0][ks]);
                    u32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = p == 0 ? v0[e] : (p == 1 ? v1[e] : v2[e]);
                    dyb[ks] = __builtin_bit_cast(bf16x8_t, v);
                }
            }
            if constexpr (RBF) {                  // the chain rule's gamma / beta (or u): requested before the W prefetch (in-order returns)
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
            // W of this step has landed for this thread (one younger image may still be in flight), then for the work-group; behind the
            // barrier every wave has also left step t - 1, whose ring slot takes the image of step t + 2.  One barrier per step.
            if (t + 1 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WQ) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 2 < T) fill_w(t + 2);
            const unsigned short* wp = W_s + (size_t)(t % 3) * WSZ + ((size_t)hf * KCT + l31) * 8;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const bf16x8_t a8 = *reinterpret_cast<const bf16x8_t*>(wp + ((size_t)(2 * ks) * KCT + kt * 32) * 8);
                    acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, dyb[ks], acc[kt], 0, 0, 0);
                }
            if (!SHARED || p == NSH - 1) {
                const BasisArgs b = make_basis(a, g);
                float duv[RBF ? FPH : 1];
                float uvv[RBF ? FPH : 1];
                if constexpr (RBF) {
                    if (a.ln) {
#pragma unroll
                        for (int j = 0; j < FPH; ++j) uvv[j] = (xv[j] - ln_st.x) * ln_st.y * lnp[j] + lnp[FPH + j];
                    } else {
#pragma unroll
                        for (int j = 0; j < FPH; ++j) uvv[j] = lnp[j];
                    }
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
                        if (RBF && g_ < GP - 1) usum += v * d;
                        else dsum += v * d;
                    }
                    dxacc[j] += dsum;
                    if constexpr (RBF) duv[j] = usum;
                }
                if constexpr (STRIP) {
                    if constexpr (RBF) {
                        if (a.du) store_tile(duv, a.du + (m0 + wave * 32) * a.ldu + (long long)g * a.I + ci * IC, a.ldu);
                    }
                    if (p == NSH - 1) store_tile(dxacc, a.dx + (m0 + wave * 32) * a.ldx + (long long)gx * a.I + ci * IC, a.ldx);
                } else {
                    if constexpr (RBF) {
                        if (a.du && row_ok) {
                            float* durow = a.du + grow * a.ldu + (long long)g * a.I + hf * FPH + ci * IC;
#pragma unroll
                            for (int j = 0; j < FPH; ++j) durow[j] = duv[j];
                        }
                    }
                    if (p == NSH - 1 && row_ok) {
#pragma unroll
                        for (int j4 = 0; j4 < FPH / 4; ++j4) {
                            const f32x4 v = {dxacc[4 * j4], dxacc[4 * j4 + 1], dxacc[4 * j4 + 2], dxacc[4 * j4 + 3]};
                            *reinterpret_cast<f32x4*>(dxrow + ci * IC + 4 * j4) = v;
                        }
                    }
                }
                if (!SHARED && p < NSH - 1) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.0f;
                }
            }
        }
    }
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
    if constexpr (FAM != KV_SINE) {
        if (a.O == 64 && (nshare == 1 || nshare == 3) && !kv_config().bi_no_res) {      // the per-head layers: dY resident (see the kernel)
            const long long tiles = (a.M + BM - 1) / BM;
            if (p.nci > 1) a.tail_y0 = kv_tail_first_tile(tiles, a.xmod);
            const long long t1 = a.tail_y0 < tiles ? a.tail_y0 : tiles;
            grid.y = (unsigned)(t1 + (long long)p.nci * (tiles - t1));
            constexpr int IC_ = 2 * ((16 * KT) / GP);
            const size_t lds3 = (size_t)3 * 4 * 2 * 32 * KT * 16 + sizeof(float) * 4 * 32 * (IC_ + 4);      // a ring of three W step images + four store strips
            if (nshare == 1) {
                KV_ALLOW_LDS(160 * 1024, (kan_bwd_input_res_bf16_kernel<FAM, GP, KT, 1, false>));
                hipLaunchKernelGGL((kan_bwd_input_res_bf16_kernel<FAM, GP, KT, 1, false>), grid, dim3(256), lds3, st, a);
                KV_LAUNCH_CHECK("kan_bwd_input_res_bf16_kernel");
                return 0;
            }
            if (shared) {
                if constexpr (kv_shared_basis<FAM>()) {
                    KV_ALLOW_LDS(160 * 1024, (kan_bwd_input_res_bf16_kernel<FAM, GP, KT, 3, true>));
                    hipLaunchKernelGGL((kan_bwd_input_res_bf16_kernel<FAM, GP, KT, 3, true>), grid, dim3(256), lds3, st, a);
                    KV_LAUNCH_CHECK("kan_bwd_input_res_bf16_kernel");
                    return 0;
                }
            } else {
                KV_ALLOW_LDS(160 * 1024, (kan_bwd_input_res_bf16_kernel<FAM, GP, KT, 3, false>));
                hipLaunchKernelGGL((kan_bwd_input_res_bf16_kernel<FAM, GP, KT, 3, false>), grid, dim3(256), lds3, st, a);
                KV_LAUNCH_CHECK("kan_bwd_input_res_bf16_kernel");
                return 0;
            }
        }
    }
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


}  // namespace

BwdRegBf16Plan plan_bwd_input_reg_bf16(const kanvit_layer_desc* d) {
    BwdRegBf16Plan p{};
    if (kv_config().no_reg || kv_config().no_bf16 || !(d->flags & KANVIT_FLAG_BF16_MFMA)) return p;
    p.gp = gp_of(d);
    const int fam = d->family;
    if (fam == KANVIT_LINEAR && p.gp == 1) p.kt = 2;
    else if (fam == KANVIT_CHEBY && p.gp == 5) p.kt = 5;
    else if (fam == KANVIT_BSPLINE && p.gp == 9 && d->has_base && (d->flags & KANVIT_FLAG_UNIFORM_KNOTS) && d->spline_order == 3) p.kt = 5;
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

int kv_bwd_input_reg_bf16(int family, LayerArgs& a, const BwdRegBf16Plan& p, hipStream_t st) {
#define KV_CALL(F) dispatch_bwd_input_reg_bf16<F>(a, p, st)
    KV_FAMILY_SWITCH(family, KV_CALL)
#undef KV_CALL
}
