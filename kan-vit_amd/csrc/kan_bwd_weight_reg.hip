// Streaming register-form weight gradient + the ordered slab reduction (split out of kan_layer.hip; see kan_layer_common.h).
#include "kan_layer_common.h"

namespace {

template <int N, int I = 0, typename F>
__device__ __forceinline__ void kv_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        kv_static_for<N, I + 1>(f);
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
// PG (kanvit_patch_embed_bwd_weight): the rows of x are patches of an NCHW image batch and dY carries the class-token rows
// (PatchWalk, kan_layer_common.h) -- no transient [B*P, I] patch matrix, no copy of dY without its class-token rows.
template <int FAM, int GP, int NOT, bool BF, int JC = GP, bool PG = false>
__global__ __launch_bounds__(256) void kan_bwd_weight_reg_kernel(const LayerArgs a, int nfb, int nos, int tiles_per_bg,
                                                                 int shared, int nbg) {
    static_assert(!(PG && FAM == KV_RBF), "FastKAN's patch embedding reads u = LayerNorm(x): no gather form");
    static_assert(!(PG && BF), "the gather form is exact fp32 (bf16 MFMA phases are too short to hide the row walker: launch_bwd_weight_reg)");
    constexpr int NJC = (GP + JC - 1) / JC;       // wide bases (G = 28) are contracted in NJC windows of JC basis functions,
                                                  // each its own wave unit (every window regenerates only its own values)
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr int TS = BF ? 8 : 1;            // tokens per lane per step
    constexpr int UB = BF ? 1 : 4;            // steps per prefetch block
    // blocks in flight.  bf16 mode: three -- two for the G = 28 windows (256 accumulators: a third ring slot cost them 14 - 20 spilled registers
    // inside the loop; with two, 1 - 2, and the patch-matrix passes of the bf16 Sine+Fourier config went 2.48 / 2.62 -> 2.42 / 2.51 ms).  The
    // B-spline instantiation keeps three although it spills 32 registers with them: two measured 267 against 257 us on the ViT-B q|k|v launch.
    constexpr int PD = BF ? (GP >= 28 ? 2 : 3) : 2;
    constexpr int NTOK = TS * UB;             // tokens per lane per block
    // The wave index is uniform by construction; telling the compiler (readfirstlane) keeps everything derived from it -- unit
    // and slab numbers, tile columns, the slab's base pointers -- in scalar registers instead of one vector register each.
    // Applied where the register file is the limit (B-spline / FastKAN, the G = 28 windows: it removes their scratch spills,
    // FastKAN q|k|v 968 -> 945 us); the Chebyshev / linear / narrow-sine instantiations measured 7 % SLOWER with it (373 -> 401 us).
    constexpr bool SCALAR_WAVE = FAM == KV_BSPLINE || FAM == KV_RBF || GP >= 28 || PG;      // (PG: the row walker lives on the scalar unit only under wave-uniform control flow)
    const int lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int wave = SCALAR_WAVE ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
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
    // B-spline / FastKAN windows (GP = 9 contracted as windows of JC basis slots, each its own wave unit): the
    // window start is made a COMPILE-TIME constant by a wave-uniform switch over three copies of the body -- the selection of
    // a window's values (kv_bsel4 lane masks, kv_sel8) then folds, and the Gaussians / spline pieces the window does not need
    // are never computed.  Every other instantiation runs the body once with the run-time window start j0.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto run = [&](auto j0c) {
    constexpr int J0C = decltype(j0c)::value;
    const int g0 = bg;                 // basis parameters: identical for every group of a shared launch
    const BasisArgs b = make_basis(a, g0);
    BasisGenP<FAM, JC, J0C> proto;          // knots / centres / frequencies / phases of this lane's feature, loaded once
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
    // patch gather: psx / psdy hold the x / dY offsets of the rows of the NEXT block to be requested (blocks are requested in order);
    // rows past the slab take the slab's first row (their dY is zeroed when the block leaves the ring).  The walker that fills them is
    // stepped row by row BETWEEN the MFMA groups of the block being contracted (walk_fill below): run as one cluster in front of the
    // loads, its ~200 dependent scalar instructions per block sat in front of an idle matrix pipe (one wave per SIMD: +17 % on the
    // ChebyKAN launch).
    constexpr int TPB = 2 * UB;                // rows of a block (gather form, fp32): lane half hf takes row 2t + hf
    PatchWalk walk;
    int pg_f = 0, pg_x0 = 0, pg_dy0 = 0, pg_len = 0, pg_wtok = 0;
    int psx[PG ? TPB : 1], psdy[PG ? TPB : 1];
    const int pg_hm = -hf;                       // all ones in the upper lane half
    if constexpr (PG) {
        walk.init(a, __builtin_amdgcn_readfirstlane((int)ms));
        pg_x0 = walk.xoff;
        pg_dy0 = walk.dyoff;
        pg_len = __builtin_amdgcn_readfirstlane(len);
        pg_f = kv_patch_feature_offset(a, f);
    }
    auto walk_fill = [&](int k) {                // row k of the next block: the walker's row, then one step
        if constexpr (PG) {
            const bool in = pg_wtok < pg_len;
            psx[k] = in ? walk.xoff : pg_x0;
            psdy[k] = in ? walk.dyoff : pg_dy0;
            walk.step();
            ++pg_wtok;
        }
    };
    auto load_block = [&](int q, int blk) {
        if constexpr (PG) {
#pragma unroll
            for (int t = 0; t < NTOK; ++t) {
                // this lane's row is k0 (lower lane half) or k1 (upper): base + (difference & lane mask) -- written as a select, the
                // pair becomes a dynamically indexed read of psx[] and a chain of seven v_cndmask per value
                const int k0 = 2 * t, k1 = 2 * t + 1;
                const int xo = psx[k0] + ((psx[k1] - psx[k0]) & pg_hm), dyr = psdy[k0] + ((psdy[k1] - psdy[k0]) & pg_hm);
                rx[q][t] = a.x[xo + pg_f];
#pragma unroll
                for (int i = 0; i < NOT; ++i) rdy[q][t][i] = a.dy[dyr + dyo[i]];
            }
            return;
        }
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
    // wave-uniform by construction, and the compiler must know it: with a per-lane block count the guards below are exec-masked regions,
    // the refill of a ring slot lands in temporaries that are copied into the slot at the region's end -- behind an s_waitcnt vmcnt(0)
    // on the loads just issued, i.e. a prefetch distance of zero (found in the ISA, round 4: the bf16 launch ran at the memory latency,
    // 3 450 cycles per block of 480 MFMA cycles)
    const int nblk_v = (len + tok_per_blk - 1) / tok_per_blk;
    const int nblk = PG ? nblk_v : __builtin_amdgcn_readfirstlane(nblk_v);      // (the patch-gather instantiations are scalar already; left exactly as they were: below)
    if constexpr (PG) {
#pragma unroll
        for (int k = 0; k < TPB; ++k) walk_fill(k);
    }
    if constexpr (PG) {
    // The patch-gather form keeps the guarded ring of round 3 as it stood: its control flow is scalar throughout (SCALAR_WAVE), its row
    // walker is interleaved with the MFMA groups, and every variant of the unguarded ring measured 5 - 8 % SLOWER on its launches
    // (SineKAN G = 28: 8.40 -> 8.8 - 9.1 ms per pass, with and without the scheduling fences).
#pragma unroll
    for (int q = 0; q < PD; ++q)
        if (q < nblk) {
            load_block(q, q);
            if constexpr (PG) {
#pragma unroll
                for (int k = 0; k < TPB; ++k) walk_fill(k);
            }
        }

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
                        walk_fill(2 * t);                      // (patch gather) two rows of the next block per MFMA group
                        walk_fill(2 * t + 1);
                        BasisGenP<FAM, JC, J0C> gen = proto;
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
                        BasisGenP<FAM, JC, J0C> g0_ = proto, g1_ = proto;
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

    } else {
    // The ring is filled and refilled UNCONDITIONALLY (rows past the slab are clamped to its last row; their dY is zeroed when the block
    // leaves the ring) and the loop runs whole groups of PD blocks without a guard; only the last nblk % PD blocks are guarded, and they
    // refill nothing.  hipcc's s_waitcnt placement is path-insensitive: with a guarded prologue ("if (q < nblk)") there is a path on
    // which slot 0 holds the YOUNGEST loads, so every block waited for all but the last few loads in flight -- vmcnt(28) .. vmcnt(0)
    // where vmcnt(63) is meant -- and the prefetch ring was one block deep at best.
#pragma unroll
    for (int q = 0; q < PD; ++q) load_block(q, q);

    auto body = [&](auto qc, int blk, auto refill) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        constexpr bool REFILL = decltype(refill)::value;
        constexpr bool FENCE = true;
        // (scheduling fences: left free, the scheduler hoists the refills of two blocks to the top of the loop and saves the slots it is about
        // to overwrite by copies -- each copy a wait on loads that should stay in flight)
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
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
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
        if constexpr (REFILL) load_block(q, blk + PD);
        if constexpr (!BF) {
#pragma unroll
            for (int t = 0; t < NTOK; ++t) {
                BasisGenP<FAM, JC, J0C> gen = proto;
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
                BasisGenP<FAM, JC, J0C> g0_ = proto, g1_ = proto;
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
    };
    int blk0 = 0;
    for (; blk0 + PD <= nblk; blk0 += PD) {
        body(std::integral_constant<int, 0>{}, blk0, std::true_type{});
        body(std::integral_constant<int, 1>{}, blk0 + 1, std::true_type{});
        if constexpr (PD > 2) body(std::integral_constant<int, 2>{}, blk0 + 2, std::true_type{});
    }
    static_assert(PD == 2 || PD == 3, "the group above is written out for two or three ring slots");
    if (blk0 < nblk) body(std::integral_constant<int, 0>{}, blk0, std::false_type{});
    if constexpr (PD > 2) {
        if (blk0 + 1 < nblk) body(std::integral_constant<int, 1>{}, blk0 + 1, std::false_type{});
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
                if (GP % JC == 0 || j0 + j < GP) gb[((long long)fr * GP + j0 + j) * a.O] = acc[j][i][r];      // the last window of 9 = 5 + 4 carries one idle slot
            }
    }
    };
    if constexpr ((FAM == KV_BSPLINE || FAM == KV_RBF) && NJC > 1) {
        static_assert(NJC <= 3, "window switch covers three windows");
        if (jc == 0) run(std::integral_constant<int, 0>{});
        else if (NJC == 2 || jc == 1) run(std::integral_constant<int, JC>{});
        else if constexpr (NJC == 3) run(std::integral_constant<int, 2 * JC>{});
    } else if constexpr (FAM == KV_BSPLINE || FAM == KV_RBF) {
        // one window holding all GP = 9 values (FastKAN, also its bf16 mode): the window start is the constant 0 -- with a runtime
        // start every value is a runtime-indexed select out of the eight Gaussians plus a branch around its own copy of silu
        // (43 VALU instructions per MFMA in the bf16 launch, profiles/r03_sq_pmc_fast_vits_bf16.md)
        static_assert(GP == 9, "compile-time windows assume G = 8 + the silu column (host-checked)");
        run(std::integral_constant<int, 0>{});
    } else {
        run(std::integral_constant<int, -1>{});
    }
}

// =============================================================================================
// The same streaming weight gradient on 16-ROW tiles (exact fp32, v_mfma_f32_16x16x4_f32), for the GP = 9 B-spline basis.
// On 32x32x2 tiles a [32 features x JC values] x [NOT tiles of 32] block costs JC*NOT*16 accumulators and 240 is the most the
// allocator places, so 9 values need 5 + 4 windows with an idle tenth slot and 15 MFMAs per basis evaluation.  A 16x16 tile
// costs 4 registers: [16 features x 3 values] x [12 tiles of 16] = 144 accumulators -- three windows of exactly three
// values, 36 MFMAs (of 32 cycles) per basis evaluation, and room for two waves per SIMD.
//   lane l: feature f = fb*16 + (l & 15) (A operand, one generated value per lane and MFMA), column (l & 15) of each
//   16-column tile (B operand, one dY value per lane and tile), token 4*step + (l >> 4) for BOTH -- four tokens per step;
//   D[row = feature 4*(l >> 4) + r][col = l & 15] in register r of the tile's accumulator.
// Everything else as above: wave units (basis group, column-tile set, window, feature block) flattened over the grid,
// compile-time window start, scalar wave index, operands prefetched PD blocks ahead, slabs + ordered reduce.
// =============================================================================================
// (No patch-gather form: with the row walker of kan_bwd_weight_reg_kernel this kernel spilled 19 scalar registers into vector lanes and ran
// the ViT-B efficient-KAN patch embedding in 2.98 instead of 2.72-2.80 ms -- more than the two copies the gather removes; measured twice, removed.)
// (Measured and removed, round 4: ChebyKAN's q|k|v launch on this kernel -- all five values x 12 tiles of 16 columns in 240 accumulators,
// one wave per SIMD, 60 MFMAs of 32 cycles per evaluation of tanh + the recurrence instead of 15 of 64: 379 against 368 us for the 32-row
// form on one box.  The ISA says why: the 32-row loop carries 54 vector instructions per four tokens and the 16-row loop 67 -- the basis
// evaluation is a minority of them, and the twelve dY streams add more ring moves and selects than the halved evaluations remove.  Five
// values x SIX tiles of 16 at two waves per SIMD -- the same work per evaluation as the 32-row form, a partner wave to cover the
// latencies -- was measured too: 372-374 against 360-364 us, same box.)
template <int FAM, int GP, int JC, int NC>
__global__ __launch_bounds__(256, 2) void kan_bwd_weight_reg16_kernel(const LayerArgs a, int nfb, int nos, int tiles_per_bg, int shared, int nbg) {
    constexpr int NJC = (GP + JC - 1) / JC;
    static_assert(GP % JC == 0 && (NJC == 3 || NJC == 1), "whole windows: three of three values (B-spline) or all nine (FastKAN)");
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr int UB = 1;                     // steps (of 4 tokens) per prefetch block
    constexpr int PD = NC >= 12 ? 3 : 6;      // blocks in flight (twelve dY streams: four measured the same as three, at 255 registers)
    const int lane = threadIdx.x & 63, l15 = lane & 15, tq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int units = nfb * nos * nbg * NJC;
    const long long gw = (long long)blockIdx.x * 4 + wave;       // global wave index over (slab, unit), unit fastest
    if (gw >= (long long)units * a.msplit) return;
    const int u = (int)(gw % units), slab = (int)(gw / units);
    const int fb = u % nfb, os = (u / nfb) % nos, jc = (u / (nfb * nos)) % NJC, bg = u / (nfb * nos * NJC);
    const long long ms = (long long)slab * a.rows_per_split;
    long long me = ms + a.rows_per_split;
    if (me > a.M) me = a.M;
    const int len = (int)(me - ms);
    if (len <= 0) return;
    const int otpg = a.O / 16;                // 16-column tiles per group
    const int f = fb * 16 + l15;
    const int gx = bg % a.xmod;
    int tg[NC];                               // group of column tile i (or -1 past the last tile: recomputes tile 0, never stored)
    int dyo[NC];                              // column of this lane in tile i, relative to a dY row
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int tt = os * NC + i;
        const int ts = tt < tiles_per_bg ? tt : os * NC;
        const int pj = ts / otpg;
        const int g = shared ? pj * a.xmod + bg : bg;
        tg[i] = tt < tiles_per_bg ? g : -1;
        dyo[i] = g * a.O + (ts - pj * otpg) * 16 + l15;
    }
    const float* xbase = a.x + ms * a.ldx + (long long)gx * a.I;      // wave-uniform
    const float* dybase = a.dy + ms * a.ldy;                          // wave-uniform
    const int ldx32 = (int)a.ldx, ldy32 = (int)a.ldy;
    const BasisArgs b = make_basis(a, bg);
    // FastKAN: the spline path reads u = LayerNorm(x) -- a separate tensor, or (KANVIT_FLAG_FUSED_LN) rebuilt from x, the saved
    // (mean, rstd) of the token's x slice (one 8-byte load per token, the same address in all 16 lanes of a token) and this
    // lane's gamma / beta
    const bool ln = RBF && a.ln;
    const float* ubase = (RBF && !ln && a.u) ? a.u + ms * a.ldu + (long long)bg * a.I : xbase;
    const int ldu32 = (RBF && !ln && a.u) ? (int)a.ldu : ldx32;
    const float* stbase = ln ? a.stats + (ms * a.xmod + gx) * 2 : nullptr;
    const int ldst = 2 * a.xmod;
    float ln_g = 1.0f, ln_b = 0.0f;
    if constexpr (RBF) {
        if (ln) {
            ln_g = b.bp[a.G + f];
            ln_b = b.bp[a.G + a.I + f];
        }
    }

    auto run = [&](auto j0c) {
        constexpr int J0C = decltype(j0c)::value;
        BasisGenP<FAM, JC, J0C> proto;
        proto.prepare(b, f, J0C);
        f32x4 acc[JC][NC];
#pragma unroll
        for (int j = 0; j < JC; ++j)
#pragma unroll
            for (int i = 0; i < NC; ++i) acc[j][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float rx[PD][UB], rdy[PD][UB][NC];
        float2 ru[RBF ? PD : 1][RBF ? UB : 1];      // FastKAN: u (in .x), or the token's (mean, rstd)
        auto tok_of = [&](int blk, int t) -> int { return 4 * (blk * UB + t) + tq; };
        auto load_block = [&](int q, int blk) {
#pragma unroll
            for (int t = 0; t < UB; ++t) {
                int tk = tok_of(blk, t);
                if (tk > len - 1) tk = len - 1;
                rx[q][t] = xbase[tk * ldx32 + f];
                if constexpr (RBF) {
                    if (ln) ru[q][t] = *reinterpret_cast<const float2*>(stbase + tk * ldst);
                    else ru[q][t].x = ubase[tk * ldu32 + f];
                }
                const int dyr = tk * ldy32;
#pragma unroll
                for (int i = 0; i < NC; ++i) rdy[q][t][i] = dybase[dyr + dyo[i]];
            }
        };
        const int nblk = (len + 4 * UB - 1) / (4 * UB);
        // A block is copied out of the ring and its slot refilled BEFORE its MFMAs (prefetch distance PD blocks).  Reading the ring
        // registers directly and refilling after the MFMAs saves NC + 1 moves per step but shortens the distance to PD - 1 blocks:
        // measured slower (767 -> 812 us on the ViT-B q|k|v launch) -- at two waves per SIMD this kernel lives on its prefetch depth.
        // The ring is filled and refilled UNCONDITIONALLY (rows past the slab are clamped to its last row) and the loop runs whole
        // groups of PD blocks without a guard -- as kan_bwd_weight_reg_kernel, and for the same reason: with guards, hipcc's
        // path-insensitive s_waitcnt placement waited for every load in flight (vmcnt(12), vmcnt(0) per block) instead of the oldest block.
#pragma unroll
        for (int q = 0; q < PD; ++q) load_block(q, q);
        auto body = [&](auto qc, int blk, auto refill) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            constexpr bool REFILL = decltype(refill)::value;
            __builtin_amdgcn_sched_barrier(0);
            float cx[UB], cu[UB], cdy[UB][NC];
            bool ok[UB];
#pragma unroll
            for (int t = 0; t < UB; ++t) {
                ok[t] = tok_of(blk, t) < len;
                cx[t] = rx[q][t];
                cu[t] = 0.0f;
                if constexpr (RBF) cu[t] = ln ? (rx[q][t] - ru[q][t].x) * ru[q][t].y * ln_g + ln_b : ru[q][t].x;
#pragma unroll
                for (int i = 0; i < NC; ++i) cdy[t][i] = rdy[q][t][i];
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (REFILL) load_block(q, blk + PD);
#pragma unroll
            for (int t = 0; t < UB; ++t) {
                BasisGenP<FAM, JC, J0C> gen = proto;
                gen.init(cx[t], cu[t]);
#pragma unroll
                for (int j = 0; j < JC; ++j) {
                    const float av = ok[t] ? gen.next(j) : 0.0f;          // rows past the slab contribute nothing (JC selects, not NC)
#pragma unroll
                    for (int i = 0; i < NC; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cdy[t][i], acc[j][i], 0, 0, 0);
                }
            }
        };
        int blk0 = 0;
        for (; blk0 + PD <= nblk; blk0 += PD) kv_static_for<PD>([&](auto qc) __attribute__((always_inline)) { body(qc, blk0 + decltype(qc)::value, std::true_type{}); });
        kv_static_for<PD - 1>([&](auto qc) __attribute__((always_inline)) {
            if (blk0 + decltype(qc)::value < nblk) body(qc, blk0 + decltype(qc)::value, std::false_type{});
        });
        // dW partial of this slab: row k = (fb*16 + 4*tq + r)*GP + J0C + j, 16 contiguous columns per row and tile
        float* base = a.slab + (long long)slab * ((long long)a.groups * a.K * a.O);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (tg[i] < 0) continue;
            float* gb = base + (long long)tg[i] * a.K * a.O + (dyo[i] - tg[i] * a.O);
#pragma unroll
            for (int j = 0; j < JC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) gb[((long long)(fb * 16 + 4 * tq + r) * GP + J0C + j) * a.O] = acc[j][i][r];
        }
    };
    if (NJC == 1 || jc == 0) run(std::integral_constant<int, 0>{});
    else if constexpr (NJC == 3) {
        if (jc == 1) run(std::integral_constant<int, JC>{});
        else run(std::integral_constant<int, 2 * JC>{});
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

template <int FAM, int GP, int NOT, bool BF, int JC, bool PG>
int launch_bwd_weight_reg_one(LayerArgs& a, const BwRegPlan& p, hipStream_t st) {
    const long long units = (long long)p.nbg * p.nfb * p.nos * p.njc;
    dim3 grid((unsigned)((units * p.slabs + 3) / 4), 1, 1);
    const size_t lds = a.ln ? (size_t)4 * p.rows_per_slab * sizeof(float2) : 0;      // four wave-private (mean, rstd) strips
    if (lds > 64 * 1024) KV_ALLOW_LDS(160 * 1024, (kan_bwd_weight_reg_kernel<FAM, GP, NOT, BF, JC, PG>));
    hipLaunchKernelGGL((kan_bwd_weight_reg_kernel<FAM, GP, NOT, BF, JC, PG>), grid, dim3(256), lds, st, a, p.nfb, p.nos, p.tiles_per_bg, p.shared, p.nbg);
    KV_LAUNCH_CHECK("kan_bwd_weight_reg_kernel");
    return 0;
}

// HAS_PG: the instantiation also exists in its patch-gather form (the patch-embedding layers the model builds: kv_bwd_weight_reg_pg_ok)
template <int FAM, int GP, int NOT, int JC = GP, bool HAS_BF = true, bool HAS_PG = false>
int launch_bwd_weight_reg(LayerArgs& a, const BwRegPlan& p, bool bf, hipStream_t st) {
    if (bf && !HAS_BF) return kv_fail(KANVIT_EINVAL, "internal: this register weight-gradient instantiation has no bf16 form");
    if (a.pg && !HAS_PG) return kv_fail(KANVIT_EINVAL, "internal: this register weight-gradient instantiation has no patch-gather form");
    if constexpr (HAS_PG) {
        if (a.pg) {      // exact fp32 only: on the bf16 matrix cores the MFMA phases are too short to hide the scalar row walker (SineKAN G = 28:
                         // 2.76 -> 3.15 ms per pass, ChebyKAN 0.37 -> 0.50 ms -- more than the patch-matrix copies cost), kv_bwd_weight_reg_pg_ok
            if (bf) return kv_fail(KANVIT_EINVAL, "internal: the patch-gather weight gradient is an exact-fp32 form");
            return launch_bwd_weight_reg_one<FAM, GP, NOT, false, JC, true>(a, p, st);
        }
    }
    if constexpr (HAS_BF) {
        if (bf) return launch_bwd_weight_reg_one<FAM, GP, NOT, true, JC, false>(a, p, st);
    }
    return launch_bwd_weight_reg_one<FAM, GP, NOT, false, JC, false>(a, p, st);
}

template <int FAM, int GP, int JC, int NC>
int launch_bwd_weight_reg16(LayerArgs& a, const BwRegPlan& p, hipStream_t st) {
    const long long units = (long long)p.nbg * p.nfb * p.nos * p.njc;
    dim3 grid((unsigned)((units * p.slabs + 3) / 4), 1, 1);
    if (a.pg) return kv_fail(KANVIT_EINVAL, "internal: the 16-row weight-gradient kernel has no patch-gather form");
    hipLaunchKernelGGL((kan_bwd_weight_reg16_kernel<FAM, GP, JC, NC>), grid, dim3(256), 0, st, a, p.nfb, p.nos, p.tiles_per_bg, p.shared, p.nbg);
    KV_LAUNCH_CHECK("kan_bwd_weight_reg16_kernel");
    return 0;
}

int dispatch_bwd_weight_reg(int family, LayerArgs& a, const BwRegPlan& p, bool bf, hipStream_t st) {
    if (p.t16) {
        if (bf || (family != KANVIT_BSPLINE && family != KANVIT_RBF)) return kv_fail(KANVIT_EINVAL, "internal: 16-row weight-gradient dispatch");
        if (family == KANVIT_RBF) return launch_bwd_weight_reg16<KV_RBF, 9, 9, 4>(a, p, st);
        return p.nt == 12 ? launch_bwd_weight_reg16<KV_BSPLINE, 9, 3, 12>(a, p, st) : launch_bwd_weight_reg16<KV_BSPLINE, 9, 3, 4>(a, p, st);
    }
    switch (family) {
        case KANVIT_LINEAR: return launch_bwd_weight_reg<KV_LINEAR, 1, 6>(a, p, bf, st);
        case KANVIT_CHEBY: return p.nt == 1 ? launch_bwd_weight_reg<KV_CHEBY, 5, 1, 5, true, true>(a, p, bf, st) : launch_bwd_weight_reg<KV_CHEBY, 5, 3, 5, true, true>(a, p, bf, st);
        case KANVIT_BSPLINE:      // two windows of five basis slots (exact fp32 when the 16-row kernel does not apply; bf16 mode)
            return p.nt == 3 ? launch_bwd_weight_reg<KV_BSPLINE, 9, 3, 5, true>(a, p, bf, st) : launch_bwd_weight_reg<KV_BSPLINE, 9, 2, 5, true>(a, p, bf, st);
        case KANVIT_RBF: return launch_bwd_weight_reg<KV_RBF, 9, 2>(a, p, bf, st);
        case KANVIT_SINE:
            if (a.flags & KANVIT_FLAG_SINE_DFREQ) {      // the x * cos operand (d loss / d freq through a weight-gradient pass; kanvit.h)
                if (a.GP == 28) return launch_bwd_weight_reg<KV_SINE_DF, 28, 4, 4, true, true>(a, p, bf, st);
                return a.GP == 4 ? launch_bwd_weight_reg<KV_SINE_DF, 4, 2>(a, p, bf, st) : launch_bwd_weight_reg<KV_SINE_DF, 5, 2>(a, p, bf, st);
            }
            if (a.GP == 28) return launch_bwd_weight_reg<KV_SINE, 28, 4, 4, true, true>(a, p, bf, st);
            return a.GP == 4 ? launch_bwd_weight_reg<KV_SINE, 4, 2>(a, p, bf, st) : launch_bwd_weight_reg<KV_SINE, 5, 2>(a, p, bf, st);
        case KANVIT_FOURIER: return launch_bwd_weight_reg<KV_FOURIER, 56, 4, 4, true, true>(a, p, bf, st);
        default: return kv_fail(KANVIT_EINVAL, "internal: register weight-gradient dispatch");
    }
}


}  // namespace

BwRegPlan plan_bwd_weight_reg(const kanvit_layer_desc* d) {
    BwRegPlan p{};
    if (kv_config().no_reg || kv_config().no_reg_bw) return p;
    p.njc = 1;
    p.gp = gp_of(d);
    const int fam = d->family;
    if (fam == KANVIT_LINEAR && p.gp == 1) p.nt = 6;
    else if (fam == KANVIT_CHEBY && p.gp == 5) p.nt = 3;
    // BSPLINE (GP = 9: 8 cubic bases + silu): two windows of FIVE basis slots (0..4 | 5..7, silu, one idle slot), each window
    // its own wave unit that contracts its values against three column tiles -- the Chebyshev schedule (240 accumulators, 15
    // MFMAs per basis evaluation), a window evaluating only its own values (compile-time window start).  The idle slot
    // costs 10 % of the MFMAs; 3 x 6 tiles (no idle slot, 288 accumulators) and round 2's 9 x 2 both spill accumulators
    // inside the token loop (the allocator cannot place more than 256 of them) and lose to the LDS-tile kernel.
    // (bf16 mode runs the same two-window schedule on v_mfma_f32_32x32x16_bf16, see below)
    else if (fam == KANVIT_BSPLINE && p.gp == 9 && (d->flags & KANVIT_FLAG_UNIFORM_KNOTS) && d->spline_order == 3 && d->has_base &&
             !((d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16 && kv_config().bs_bw_bf16 == 1)) { p.nt = 3; p.njc = 2; }
    else if (fam == KANVIT_RBF && p.gp == 9 && d->has_base && kv_rbf_reg_ok(d->flags, d->G)) p.nt = 2;      // (windows of 3 measured slower for both: the basis is re-evaluated per window)
    else if (fam == KANVIT_SINE && (p.gp == 4 || p.gp == 5)) p.nt = 2;
    // SINE G = 28: windows of 4 basis functions x 4 column tiles.  (Windows of 2 x 6 tiles -- 12 MFMAs per pair of sines instead
    // of 16 per four, 7.4 -> 4.4 VALU instructions per MFMA -- measured SLOWER, 8.6 -> 13.1 ms: twice the wave units re-read dY.)
    else if (fam == KANVIT_SINE && p.gp == 28) { p.nt = 4; p.njc = 7; }
    else if (fam == KANVIT_FOURIER && p.gp == 56) { p.nt = 4; p.njc = 14; }
    else return p;
    if (d->I % 32 || d->O % 32 || d->M < 256) return p;
    const int nshare = d->groups / d->x_group_mod;
    // B-spline (exact fp32): the 16-row-tile kernel -- three windows of three values x 12 (or 4) column tiles of 16, no idle
    // slot, two waves per SIMD -- when the 16-column tiles of a basis group divide by 12 or 4 (q|k|v of a 64-wide head: 12)
    // FastKAN (exact fp32): the same kernel with all nine values x 4 column tiles of 16 per wave (its q, k, v do not share u):
    // 36 MFMAs per evaluation of the eight Gaussians + silu instead of 18
    bool bf16_mode = (d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16;
    // B-splines under the bf16 flag: the two-window 32-row kernel on the bf16 matrix cores (measured at ViT-B: q|k|v 0.26 ms against 0.70
    // for the exact 16-row kernel and 0.75 for the LDS-tile bf16 kernel, the patch embedding 1.13 / 3.0 / 3.8 ms; efficient-KAN bf16 step
    // 39.4 -> 30.7 ms).  KANVIT_BSPLINE_BW_BF16 = 2 keeps the exact kernel under the flag, = 1 the LDS-tile bf16 kernel (A/B).
    if (fam == KANVIT_BSPLINE && bf16_mode && kv_config().bs_bw_bf16 == 2) bf16_mode = false;
    p.bf = bf16_mode ? 1 : 0;
    if (((fam == KANVIT_BSPLINE && p.njc == 2) || fam == KANVIT_RBF) && !bf16_mode && !kv_config().bw_no_t16) {
        const int sh = (kv_share_ok(fam, d->flags) && nshare > 1) ? 1 : 0;
        const int t16 = (sh ? nshare : 1) * (d->O / 16);
        const int nc = (fam != KANVIT_RBF && t16 % 12 == 0) ? 12 : ((t16 % 4 == 0) ? 4 : 0);      // else: the 32-row kernel below
        if (nc) {
            p.t16 = 1;
            p.shared = sh;
            p.nbg = sh ? d->x_group_mod : d->groups;
            p.tiles_per_bg = t16;
            p.nfb = d->I / 16;
            p.nt = nc;
            p.njc = fam == KANVIT_RBF ? 1 : 3;
            p.nos = t16 / nc;
            const long long units = (long long)p.nbg * p.nfb * p.nos * p.njc;
            const long long slots = 8LL * N_CU;                   // two waves per SIMD
            long long r = 1;
            while (slots * r < units) ++r;
            // the round count that wastes the fewest slots among r, r + 1 (e.g. 576 units: 3 slabs fill 84 % of one round, 7 slabs 98 % of two)
            long long S = slots * r / units, S2 = slots * (r + 1) / units;
            if (S2 * units * r > S * units * (r + 1) && S2 <= d->M / 64) { S = S2; }
            const long long smax = d->M / 64;
            if (S > smax) S = smax;
            if (S < 1) S = 1;
            if (S > 65535) return BwRegPlan{};
            long long rps = (d->M + S - 1) / S;
            rps = (rps + 15) / 16 * 16;
            p.rows_per_slab = rps;
            p.slabs = (int)((d->M + rps - 1) / rps);
            long long ld = d->ldx > d->ldy ? d->ldx : d->ldy;
            if (d->ldu > ld) ld = d->ldu;
            if (rps * ld + ld >= (1LL << 29) || units > (1LL << 30)) return BwRegPlan{};
            p.ws_bytes = p.slabs > 1 ? sizeof(float) * (size_t)p.slabs * d->groups * ((size_t)d->I * p.gp) * d->O : 0;
            p.ok = true;
            return p;
        }
    }
    p.shared = (kv_share_ok(fam, d->flags) && nshare > 1) ? 1 : 0;
    p.nbg = p.shared ? d->x_group_mod : d->groups;
    p.tiles_per_bg = (p.shared ? nshare : 1) * (d->O / 32);
    p.nfb = d->I / 32;
    if (fam == KANVIT_BSPLINE && p.tiles_per_bg <= 2) p.nt = 2;      // one unshared narrow layer: no idle column tile
    p.nos = (p.tiles_per_bg + p.nt - 1) / p.nt;
    // Small launches (the T / C geometries: 6400 rows, 2 heads): even at the shortest slab (64 tokens) the wave units cannot
    // fill the chip, and a wave's MFMA chain (tokens x GP x NOT) IS the kernel time.  One column tile per wave instead of
    // three: three times the waves, a third of the chain each; the basis is re-evaluated per wave (cheap against the chain).
    if (fam == KANVIT_CHEBY && p.nt == 3 && (long long)p.nbg * p.nfb * p.nos * (d->M / 64) < 4LL * N_CU && !kv_config().bw_dma_force) {
        p.nt = 1;
        p.nos = p.tiles_per_bg;
    }
    // ChebyKAN, three column tiles per wave: the LDS-DMA form (kan_bwd_weight_dma.hip) -- every bf16-mode launch, and the exact fp32
    // launches with more than one group (q|k|v; a one-group fp32 layer keeps the register form, whose sums the patch-gather kernel
    // reproduces bit for bit).  One work-group per CU (128 KiB of rings), four row ranges each: the slab count is the number of
    // work-groups per wave unit that fills the CUs r times; taken only when that fills at least 85 % of the last round.
    if (fam == KANVIT_CHEBY && p.nt == 3 && !kv_config().bw_no_dma && (bf16_mode || d->groups > 1) && d->ldx % 4 == 0 && d->ldy % 4 == 0) {
        const long long units = (long long)p.nbg * p.nfb * p.nos;
        long long r = 1;
        while ((long long)N_CU * r < units) ++r;
        long long S = (long long)N_CU * r / units;
        const bool force = kv_config().bw_dma_force != 0;        // (parity tests: small shapes through this kernel, down to 16 rows per wave)
        const long long smax = force ? (d->M / 64 > 0 ? d->M / 64 : 1) : d->M / 256;            // at least 64 rows per wave
        if (S > smax) S = smax;
        if (S >= 1 && S <= 65535 && (force || units * S * 100 >= 85LL * N_CU * r)) {
            long long rps = (d->M + S - 1) / S;
            rps = (rps + 63) / 64 * 64;
            const long long ld = d->ldx > d->ldy ? d->ldx : d->ldy;
            if (rps / 4 * ld + ld < (1LL << 29) && units * S < (1LL << 31)) {
                p.dma = 1;
                p.rows_per_slab = rps;
                p.slabs = (int)((d->M + rps - 1) / rps);
                p.ws_bytes = p.slabs > 1 ? sizeof(float) * (size_t)p.slabs * d->groups * ((size_t)d->I * p.gp) * d->O : 0;
                p.ok = true;
                return p;
            }
        }
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

// (Measured and removed, round 3: an XCD-aware order of the work-groups -- every XCD one contiguous range of the (slab, unit) order, so
// that waves sharing a dY tile or an x feature block meet in one L2 -- changed no weight-gradient launch by more than the run-to-run
// noise (ChebyKAN / B-spline / FastKAN / SineKAN G = 28, fp32 and bf16): these kernels wait on their prefetch depth, not on L2 misses.)
int kv_bwd_weight_reg(int family, LayerArgs& a, const BwRegPlan& p, bool bf, hipStream_t st) {
    // (rows of x / dY off the 16-byte grid -- a view into the middle of a tensor: the register form runs the same plan, correct and slower)
    if (p.dma && kv_bwd_weight_dma_aligned(a)) return kv_bwd_weight_dma(family, a, p, bf, st);
    return dispatch_bwd_weight_reg(family, a, p, bf, st);
}

// the plans whose kernels exist in the patch-gather form (dispatch_bwd_weight_reg's HAS_PG instantiations): the patch-embedding
// layers VisionTransformer builds (model.py:67-80: ChebyKAN degree 4, SineKAN / FourierKAN at grid 28)
bool kv_bwd_weight_reg_pg_ok(const kanvit_layer_desc* d, const BwRegPlan& p) {
    if (!p.ok || d->groups != 1 || p.bf) return false;      // (bf16 mode keeps the patch matrix: see launch_bwd_weight_reg)
    // B-splines are NOT among them: their kernels are at the register limit, and the walker's scalar state spilled (19 scalar registers
    // in the 16-row kernel, 16 vector registers in the 32-row one): 2.97 against 2.80 ms for the ViT-B patch embedding, 1.82 against
    // 1.13 ms in bf16 mode -- more than the patch-matrix copy costs.  efficient-KAN's patch embedding keeps the transient patch matrix.
    if (p.t16) return false;
    switch (d->family) {
        case KANVIT_CHEBY: return p.gp == 5 && (p.nt == 3 || p.nt == 1);
        case KANVIT_SINE: return p.gp == 28;
        case KANVIT_FOURIER: return p.gp == 56;
        default: return false;
    }
}

// dw[e] = sum over the `slabs` partial slabs (each `total` floats), in slab order
int kv_slab_reduce(const float* slab, float* dw, long long total, int slabs, hipStream_t st) {
    long long nb = (total + 255) / 256;
    if (nb > 8 * N_CU) nb = 8 * N_CU;
    hipLaunchKernelGGL(kan_slab_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, slab, dw, total, slabs);
    KV_LAUNCH_CHECK("kan_slab_reduce_kernel");
    return 0;
}
