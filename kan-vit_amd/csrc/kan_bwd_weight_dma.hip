// Streaming weight gradient, LDS-DMA form (round 4).  Same contraction and wave unit as kan_bwd_weight_reg_kernel
//   dW[g][i*GP + j][o] = sum_m Phi_j(x[m][i]) * dY[m][g*O + o]           (models/cheby.py:36-48 backward)
// -- a wave owns [32 features x GP values] x [NOT column tiles of 32] of dW in GP*NOT*16 accumulators and streams its rows -- but
// the operands no longer travel through a register ring.  Why: in the register form hipcc turns the ring into "load into
// temporaries, copy into the slot", and the copy waits for the loads just issued (s_waitcnt vmcnt(0) per block: a prefetch distance
// of zero; the bf16 launch ran at the memory latency, 3 450 cycles per block of 480 MFMA cycles) -- and every dword load carries a
// 64-bit address chain (130 of the 390 vector instructions of a bf16 block).  Here:
//   * a block is 16 rows: the wave's x tile [16][32] and its dY tiles [16][32*NOT] land in a wave-private LDS ring of four slots by
//     global_load_lds_dwordx4 (8 instructions per block instead of 32 dword loads; uniform base + per-lane offsets formed once);
//     three blocks are in flight while one is contracted, the only wait is an explicit s_waitcnt vmcnt(24);
//   * the MFMA operands are ds_read_b32 at immediate offsets (k index i of lane half h = row 2i + h of the block: both halves of a
//     read fall into different banks); no register is live across blocks except the accumulators;
//   * a work-group is FOUR row ranges of ONE wave unit: the four partial sums are added through the LDS (the ring is dead by then)
//     in a fixed order, (s0 + s1) + (s2 + s3), so a launch writes a quarter of the slab partials the register form wrote
//     (q|k|v at ViT-B: 21 slabs -> 5; 124 MB of slab traffic -> 30) and the ordered reduce has a quarter of the work.
// (Measured and not kept: the two-window B-spline schedule of the bf16 mode in this form -- 257 against 259 us on the ViT-B q|k|v launch: that
//  kernel is bound by the evaluation of the spline pieces, not by addresses or the ring.)
// fp32: v_mfma_f32_32x32x2_f32, 8 steps per block; bf16 flag: v_mfma_f32_32x32x16_bf16, one MFMA per (value, tile) and block.
// grid = 8 XCDs x cap slots (>= units * slabs work-groups, see the mapping in the kernel), 256 threads, 128 KiB LDS.
#include "kan_layer_common.h"

namespace {

typedef const __attribute__((address_space(1))) void* kvd_glb_ptr;
typedef __attribute__((address_space(3))) void* kvd_lds_ptr;

#ifndef KVD_NSLOT_N
#define KVD_NSLOT_N 4
#endif
constexpr int KVD_NSLOT = KVD_NSLOT_N;      // ring slots per wave
constexpr int KVD_BLK = 16;       // rows per block
#ifndef KVD_ABLATE
#define KVD_ABLATE 0      // diagnostic builds (tools/build_variant.sh): 1 = no contraction (fills and waits only), 2 = no fills
#endif

template <int N>
__device__ __forceinline__ void kvd_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int FAM, int GP, int NOT, bool BF>
__global__ __launch_bounds__(256) void kan_bwd_weight_dma_kernel(const LayerArgs a, int nfb, int nos, int tiles_per_bg, int shared,
                                                                 int nbg, int sub_rows, int cap, int placed) {
    constexpr int XF = KVD_BLK * 32;                    // floats of a block's x tile
    constexpr int YW = 32 * NOT;                        // floats of a dY row of the block
    constexpr int SLOTF = XF + KVD_BLK * YW;            // floats of a ring slot
    constexpr int NDX = XF / 256;                       // DMA instructions (64 lanes x 16 bytes) per x tile: 2
    constexpr int NDY = KVD_BLK * YW / 256;             // ... per dY tile set: 6 for three tiles
    constexpr int NDMA = NDX + NDY;
    static_assert(XF % 256 == 0 && (KVD_BLK * YW) % 256 == 0, "whole DMA instructions");
    static_assert(NDMA * (KVD_NSLOT - 1) < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Work-group -> (slab, wave unit), XCD-aware.  The nfb * nos units of one (basis group, slab) read the same rows: the same x blocks
    // (across column-tile sets) and the same dY tiles (across feature blocks).  Work-groups go round-robin over the 8 XCDs, each with its
    // own L2 -- numbered naively, the units of a sharing set land on different XCDs and every one of them misses on its own (measured:
    // 620 MB through the fabric for 310 MB of operands, the launch bound by exactly that).  So XCD x = blockIdx % 8 takes the
    // contiguous range [x * cap, (x + 1) * cap) of the (set, member) order, cap a multiple of the set size; the slots past the
    // last set idle (16 of 256 work-groups for q|k|v at ViT-B).
    // (sets larger than an XCD's 32 CUs -- one wide layer -- are not placed: the order is then plain (slab, unit), unit fastest)
    const int gsz = nfb * nos;
    const int L = (int)(blockIdx.x & 7) * cap + (int)(blockIdx.x >> 3);
    int bg, slab, fb, os;
    if (placed) {
        const int set = L / gsz, mem = L - set * gsz;
        if (set >= nbg * a.msplit) return;
        bg = set % nbg, slab = set / nbg;
        fb = mem % nfb, os = mem / nfb;
    } else {
        const int units = gsz * nbg;
        if (L >= units * a.msplit) return;
        const int u = L % units;
        slab = L / units;
        fb = u % nfb, os = (u / nfb) % nos, bg = u / gsz;
    }
    // rows of this wave: the wave-th quarter of the work-group's slab
    const long long s0 = (long long)slab * a.rows_per_split;
    long long s1 = s0 + a.rows_per_split;
    if (s1 > a.M) s1 = a.M;
    const long long ms = s0 + (long long)wave * sub_rows;
    long long me = ms + sub_rows;
    if (me > s1) me = s1;
    const int len = __builtin_amdgcn_readfirstlane(me > ms ? (int)(me - ms) : 0);
    const int nblk = (len + KVD_BLK - 1) / KVD_BLK;
    const int otpg = a.O / 32;
    const int gx = bg % a.xmod;

    // this wave unit's column tiles (as kan_bwd_weight_reg_kernel)
    int tg[NOT], tcol[NOT];
#pragma unroll
    for (int i = 0; i < NOT; ++i) {
        const int tt = os * NOT + i;
        if (tt < tiles_per_bg) {
            const int p = tt / otpg;
            tg[i] = shared ? p * a.xmod + bg : bg;
            tcol[i] = tg[i] * a.O + (tt - p * otpg) * 32;
        } else {            // past the last tile: the first tile again (no branch around the MFMAs), never stored
            tg[i] = -1;
            tcol[i] = (shared ? ((os * NOT) / otpg) * a.xmod + bg : bg) * a.O + ((os * NOT) % otpg) * 32;
        }
    }
    const BasisArgs b = make_basis(a, bg);
    BasisGenP<FAM, GP, -1> proto;
    proto.prepare(b, fb * 32 + l31, 0);

    // ---- DMA addressing: uniform 64-bit bases (first row of this wave) + per-lane byte offsets inside a block, formed once ----
    const int ldx32 = (int)a.ldx, ldy32 = (int)a.ldy;
    const char* xbase = reinterpret_cast<const char*>(a.x + ms * a.ldx + (long long)gx * a.I + fb * 32);      // uniform
    const char* dybase = reinterpret_cast<const char*>(a.dy + ms * a.ldy);                                     // uniform
    int xrow[NDX], xcol[NDX], yrow[NDY], ycol[NDY];      // row of the block / float offset inside the row, of the 16 bytes this lane moves
#pragma unroll
    for (int k = 0; k < NDX; ++k) {
        const int c = 64 * k + lane;
        xrow[k] = c >> 3;
        xcol[k] = (c & 7) * 4;
    }
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
        const int c = 64 * k + lane;
        const int row = c / (8 * NOT), rem = c - row * (8 * NOT), tile = rem >> 3;
        int tc = tcol[0];
#pragma unroll
        for (int i = 1; i < NOT; ++i) tc = tile == i ? tcol[i] : tc;
        yrow[k] = row;
        ycol[k] = tc + (rem & 7) * 4;
    }
    unsigned xo[NDX], yo[NDY];
#pragma unroll
    for (int k = 0; k < NDX; ++k) xo[k] = 4u * (unsigned)(xrow[k] * ldx32 + xcol[k]);
#pragma unroll
    for (int k = 0; k < NDY; ++k) yo[k] = 4u * (unsigned)(yrow[k] * ldy32 + ycol[k]);
    float* ring = smem + wave * (KVD_NSLOT * SLOTF);

    // one block into one slot.  The last block of a row range that is not a multiple of 16 re-reads its last row for the missing ones
    // (the consumer zeroes their dY): a uniform branch, the offsets of every other block are the ones formed above.
    auto issue = [&](int blk, int slot) __attribute__((always_inline)) {
        if (KVD_ABLATE & 2) return;
        float* dst = ring + slot * SLOTF;
        const bool ragged = (blk + 1) * KVD_BLK > len;
        if (!ragged) {
            const char* xb = xbase + (long long)blk * (KVD_BLK * 4) * ldx32;
            const char* yb = dybase + (long long)blk * (KVD_BLK * 4) * ldy32;
#pragma unroll
            for (int k = 0; k < NDX; ++k) __builtin_amdgcn_global_load_lds((kvd_glb_ptr)(xb + xo[k]), (kvd_lds_ptr)(dst + 256 * k), 16, 0, 0);
#pragma unroll
            for (int k = 0; k < NDY; ++k) __builtin_amdgcn_global_load_lds((kvd_glb_ptr)(yb + yo[k]), (kvd_lds_ptr)(dst + XF + 256 * k), 16, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < NDX; ++k) {
                int r = blk * KVD_BLK + xrow[k];
                r = r < len ? r : len - 1;
                __builtin_amdgcn_global_load_lds((kvd_glb_ptr)(xbase + 4u * (unsigned)(r * ldx32 + xcol[k])), (kvd_lds_ptr)(dst + 256 * k), 16, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < NDY; ++k) {
                int r = blk * KVD_BLK + yrow[k];
                r = r < len ? r : len - 1;
                __builtin_amdgcn_global_load_lds((kvd_glb_ptr)(dybase + 4u * (unsigned)(r * ldy32 + ycol[k])), (kvd_lds_ptr)(dst + XF + 256 * k), 16, 0, 0);
            }
        }
    };

    f32x16 acc[GP][NOT];
#pragma unroll
    for (int j = 0; j < GP; ++j)
#pragma unroll
        for (int i = 0; i < NOT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

    // lane (l31, hf) reads row 2i + hf of the block, i = 0..7: x[row][l31] and dY[row][32 t + l31]
    const float* rdx = ring + hf * 32 + l31;
    const float* rdy = ring + XF + hf * YW + l31;

    auto step = [&](int blk, auto slotc) __attribute__((always_inline)) {
        constexpr int S = decltype(slotc)::value;
        // block blk lives in slot blk % 4.  Refill the slot of block blk - 1 (its reads are long complete: their values went through
        // the MFMAs of that block) with block blk + 3, then wait until block blk itself has landed: 8 instructions per block in
        // flight behind it.  Near the end of the row range fewer blocks are behind.
        const int behind = nblk - 1 - blk;           // uniform
        if (behind >= KVD_NSLOT - 1) {
            issue(blk + KVD_NSLOT - 1, (S + KVD_NSLOT - 1) % KVD_NSLOT);
            kvd_wait_vm<NDMA * (KVD_NSLOT - 1)>();
        } else if (KVD_NSLOT > 4 && behind == 3) {
            kvd_wait_vm<NDMA * 3>();
        } else if (behind == 2) {
            kvd_wait_vm<NDMA * 2>();
        } else if (behind == 1) {
            kvd_wait_vm<NDMA>();
        } else {
            kvd_wait_vm<0>();
        }
        if (KVD_ABLATE & 1) return;
        float cx[8], cdy[8][NOT];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            cx[i] = rdx[S * SLOTF + i * 64];
#pragma unroll
            for (int t = 0; t < NOT; ++t) cdy[i][t] = rdy[S * SLOTF + i * (2 * YW) + 32 * t];
        }
        if (behind == 0) {      // the last block: rows past the end contribute nothing
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = blk * KVD_BLK + 2 * i + hf < len;
#pragma unroll
                for (int t = 0; t < NOT; ++t) cdy[i][t] = ok ? cdy[i][t] : 0.0f;
            }
        }
        if constexpr (!BF) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                BasisGenP<FAM, GP, -1> gen = proto;
                gen.init(cx[i], 0.0f);
#pragma unroll
                for (int j = 0; j < GP; ++j) {
                    const float av = gen.next(j);
#pragma unroll
                    for (int t = 0; t < NOT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, cdy[i][t], acc[j][t], 0, 0, 0);
                }
            }
        } else {
            unsigned af[GP][4];
#pragma unroll
            for (int ep = 0; ep < 4; ++ep) {
                BasisGenP<FAM, GP, -1> g0_ = proto, g1_ = proto;
                g0_.init(cx[2 * ep], 0.0f);
                g1_.init(cx[2 * ep + 1], 0.0f);
#pragma unroll
                for (int j = 0; j < GP; ++j) af[j][ep] = kv_pack_bf16(g0_.next(j), g1_.next(j));
            }
            bf16x8_t bfr[NOT];
#pragma unroll
            for (int t = 0; t < NOT; ++t) {
                const u32x4 u4 = {kv_pack_bf16(cdy[0][t], cdy[1][t]), kv_pack_bf16(cdy[2][t], cdy[3][t]),
                                  kv_pack_bf16(cdy[4][t], cdy[5][t]), kv_pack_bf16(cdy[6][t], cdy[7][t])};
                bfr[t] = __builtin_bit_cast(bf16x8_t, u4);
            }
#pragma unroll
            for (int j = 0; j < GP; ++j) {
                const u32x4 a4 = {af[j][0], af[j][1], af[j][2], af[j][3]};
                const bf16x8_t afr = __builtin_bit_cast(bf16x8_t, a4);
#pragma unroll
                for (int t = 0; t < NOT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[t], acc[j][t], 0, 0, 0);
            }
        }
    };

#pragma unroll
    for (int p = 0; p < KVD_NSLOT - 1; ++p)
        if (p < nblk) issue(p, p);
    for (int blk0 = 0; blk0 < nblk; blk0 += KVD_NSLOT) {
        step(blk0, std::integral_constant<int, 0>{});
        if (blk0 + 1 < nblk) step(blk0 + 1, std::integral_constant<int, 1>{});
        if (blk0 + 2 < nblk) step(blk0 + 2, std::integral_constant<int, 2>{});
        if (blk0 + 3 < nblk) step(blk0 + 3, std::integral_constant<int, 3>{});
        if constexpr (KVD_NSLOT > 4) {
            if (blk0 + 4 < nblk) step(blk0 + 4, std::integral_constant<int, 4>{});
        }
    }
    static_assert(KVD_NSLOT == 4 || KVD_NSLOT == 5, "the step sequence above is written out for four or five slots");

    // ---- the four partial sums of the work-group, through the LDS: images [accumulator register][lane], two at a time ----
    constexpr int NREG = GP * NOT * 16;
    static_assert(2 * NREG * 64 <= 4 * KVD_NSLOT * SLOTF, "two accumulator images fit the rings");
    float* img = smem + (wave >> 1) * (NREG * 64) + lane;
    auto put = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < GP; ++j)
#pragma unroll
            for (int t = 0; t < NOT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) img[((j * NOT + t) * 16 + r) * 64] = acc[j][t][r];
    };
    auto add = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < GP; ++j)
#pragma unroll
            for (int t = 0; t < NOT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][t][r] += img[((j * NOT + t) * 16 + r) * 64];
    };
    __syncthreads();                  // every wave has left its ring
    if (wave & 1) put();              // waves 1, 3 -> images 0, 1
    __syncthreads();
    if (!(wave & 1)) add();           // wave 0 += wave 1, wave 2 += wave 3
    __syncthreads();
    if (wave == 2) {
        img = smem + lane;
        put();
    }
    __syncthreads();
    if (wave != 0) return;
    img = smem + lane;
    add();

    // dW partial of this slab: row k = (fb*32 + acc row)*GP + j, 32 contiguous columns per row
    float* base = a.slab + (long long)slab * ((long long)a.groups * a.K * a.O);
#pragma unroll
    for (int i = 0; i < NOT; ++i) {
        if (tg[i] < 0) continue;
        const int tt = os * NOT + i;
        const int col0 = (tt % otpg) * 32;
        float* gb = base + (long long)tg[i] * a.K * a.O + col0 + l31;
#pragma unroll
        for (int j = 0; j < GP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int fr = fb * 32 + kv_acc_row(r, hf);
                gb[((long long)fr * GP + j) * a.O] = acc[j][i][r];
            }
    }
}

template <int FAM, int GP, int NOT, bool BF>
int launch_dma(LayerArgs& a, const BwRegPlan& p, hipStream_t st) {
    const long long gsz = (long long)p.nfb * p.nos, sets = (long long)p.nbg * p.slabs;
    const long long per_xcd = (sets * gsz + 7) / 8;
    const int placed = gsz <= 32 ? 1 : 0;
    const long long cap = placed ? (per_xcd + gsz - 1) / gsz * gsz : per_xcd;       // slots of an XCD: whole sharing sets (see the kernel)
    dim3 grid((unsigned)(8 * cap), 1, 1);
    constexpr size_t lds = sizeof(float) * 4 * KVD_NSLOT * (KVD_BLK * 32 * (1 + NOT));
    KV_ALLOW_LDS(160 * 1024, (kan_bwd_weight_dma_kernel<FAM, GP, NOT, BF>));
    hipLaunchKernelGGL((kan_bwd_weight_dma_kernel<FAM, GP, NOT, BF>), grid, dim3(256), lds, st, a, p.nfb, p.nos, p.tiles_per_bg, p.shared, p.nbg,
                       (int)(p.rows_per_slab / 4), (int)cap, placed);
    KV_LAUNCH_CHECK("kan_bwd_weight_dma_kernel");
    return 0;
}

}  // namespace

// 16-byte DMA pieces: the rows of x and dY this launch reads must start on 16-byte boundaries
bool kv_bwd_weight_dma_aligned(const LayerArgs& a) {
    return !(((uintptr_t)a.x | (uintptr_t)a.dy) & 15) && a.ldx % 4 == 0 && a.ldy % 4 == 0 && a.I % 4 == 0 && a.O % 4 == 0;
}

int kv_bwd_weight_dma(int family, LayerArgs& a, const BwRegPlan& p, bool bf, hipStream_t st) {
    if (family != KANVIT_CHEBY || p.gp != 5 || p.nt != 3 || a.pg) return kv_fail(KANVIT_EINVAL, "internal: LDS-DMA weight-gradient dispatch");
    return bf ? launch_dma<KV_CHEBY, 5, 3, true>(a, p, st) : launch_dma<KV_CHEBY, 5, 3, false>(a, p, st);
}
