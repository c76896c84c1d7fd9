// Register-form forward of the fused KAN layers, exact fp32 (split out of kan_layer.hip; see kan_layer_common.h).
#include "kan_layer_common.h"

namespace {

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
// The body takes its block coordinates as arguments: the kernel below hands the LAST row tiles of a shared-basis (NSH = 3) launch
// to the NSH = 1 body, one projection per work-group (kv_tail_first_tile).
template <int FAM, int NT, int NSH, int ICH, int GPC>
__device__ __forceinline__ void kan_fwd_reg_body(const LayerArgs& a, float* __restrict__ smem, const int bx, const int by) {
    constexpr int BN = 32 * NT;
    constexpr int WROW = NSH * BN;
    constexpr int V4 = BN / 4;
    constexpr int IC = 2 * ICH;
    constexpr bool RBF = (FAM == KV_RBF);
    constexpr int TS = 36;                        // staging patch row stride (floats): 16-byte aligned, conflict free
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int ntn = a.O / BN;
    const int gs = bx / ntn;
    const int n0 = (bx - gs * ntn) * BN;
    const int nsets = a.groups / NSH;
    const long long m0 = (long long)by * BM;
    const int GP = GPC > 0 ? GPC : a.GP, KC = IC * GP;      // k rows per chunk (even); compile-time in the GPC instantiations (host: a.GP == GPC)
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
            for (int q = 0; q < WQ; ++q)      // unconditional: a predicated load is a branch around it plus an s_waitcnt vmcnt(1) in front of every
                if (q * WRS < KC)             // load (two in flight instead of all of them); rows past KC re-read row 0 and are never stored
                    wreg[p][q] = *reinterpret_cast<const f32x4*>(src + (koff[q] >= 0 ? koff[q] : 0));
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

    // SINE with a compile-time basis size: the phases p[i][g] of the chunk's features travel one chunk ahead like x.  Read inside
    // the basis evaluation they are issued AFTER the next chunk's W / x prefetch, and memory returns in order: every chunk then
    // began by waiting for its own prefetch (s_waitcnt vmcnt(3) behind eight 16-byte loads).
    constexpr bool SINE_PH = (FAM == KV_SINE && GPC > 0);
    constexpr int NPH = SINE_PH ? ICH * GPC : 1;
    float phv[NPH];
    auto load_ph = [&](int c) {
        if constexpr (SINE_PH) {
            const float* ps = b.bp + GPC + (long long)(c * IC + hf * ICH) * GPC;      // [ICH features][GPC phases], contiguous
            if constexpr ((ICH * GPC) % 4 == 0) {
#pragma unroll
                for (int e4 = 0; e4 < NPH / 4; ++e4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ps + 4 * e4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) phv[4 * e4 + e] = v[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < NPH; ++e) phv[e] = ps[e];
            }
        }
    };

    load_w(0);
    load_x(0);
    load_ph(0);
    store_w(0);
    __syncthreads();

    for (int c = 0; c < nch; ++c) {
        float xc[ICH], uc[ICH], phc[NPH];
#pragma unroll
        for (int e = 0; e < ICH; ++e) {
            xc[e] = xv[e];
            uc[e] = RBF ? uv[e] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < NPH; ++e) phc[e] = phv[e];
        if (c + 1 < nch) {                        // prefetch the next chunk; lands while this chunk's MFMAs run
            load_w(c + 1);
            load_x(c + 1);
            load_ph(c + 1);
        }
        const float* wp = W_s + (c & 1) * WSZ + hf * WROW + l31;
        if constexpr (GPC > 0) {
            constexpr int VH = ICH * GPC;               // k-steps of this chunk (one generated value per lane and step)
            constexpr int NTT = NSH * NT;
            float phi[VH];
#pragma unroll
            for (int j = 0; j < ICH; ++j) {
                BasisGen<FAM, kv_gc(FAM, GPC)> gen;       // compile-time G: one silu per feature, no branch per value (host-checked has_base)
                gen.init(b, xc[j], uc[j], c * IC + hf * ICH + j);
#pragma unroll
                for (int g = 0; g < GPC; ++g) {
                    if constexpr (SINE_PH) phi[j * GPC + g] = kv_sin(__fadd_rn(__fmul_rn(xc[j], b.bp[g]), phc[j * GPC + g]));      // BasisGen<KV_SINE>::next with the phase from registers
                    else phi[j * GPC + g] = gen.next(g);
                }
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

// TAIL (NSH = 3 only): grid rows [tail_y0, ...) are the sub-divided end of the launch -- row tile tail_y0 + t / 3, projection t % 3,
// run by the one-projection body (it re-evaluates the basis; same k order per output, so the results are bitwise the shared ones).
template <int FAM, int NT, int NSH, int ICH, int GPC = 0, bool TAIL = false>
__global__ __launch_bounds__(256, GPC ? 2 : 1) void kan_fwd_reg_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (TAIL) {
        static_assert(NSH == 3, "the tail hands single projections of a shared-basis launch to the NSH = 1 body");
        const int ty = (int)blockIdx.y - a.tail_y0;
        if (ty >= 0) {
            const int tile = ty / 3, proj = ty - 3 * tile;
            kan_fwd_reg_body<FAM, NT, 1, ICH, GPC>(a, smem, proj * (int)gridDim.x + (int)blockIdx.x, a.tail_y0 + tile);
            return;
        }
    }
    kan_fwd_reg_body<FAM, NT, NSH, ICH, GPC>(a, smem, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- register-operand forward (fp32 exact) --------------------------------------------------------
template <int FAM, int NT, int NSH, int ICH, int GPC = 0>
int launch_fwd_reg(const LayerArgs& a0, size_t lds, hipStream_t st) {
    const unsigned gx = (unsigned)((a0.groups / NSH) * (a0.O / (32 * NT)));
    const long long tiles = (a0.M + BM - 1) / BM;
    if constexpr (NSH == 3 && GPC > 0) {
        const int t1 = kv_tail_first_tile(tiles, (int)gx);
        if (t1 < tiles) {
            LayerArgs a = a0;
            a.tail_y0 = t1;
            KV_ALLOW_LDS(160 * 1024, (kan_fwd_reg_kernel<FAM, NT, NSH, ICH, GPC, true>));
            dim3 grid(gx, (unsigned)(t1 + 3 * (tiles - t1)), 1);
            hipLaunchKernelGGL((kan_fwd_reg_kernel<FAM, NT, NSH, ICH, GPC, true>), grid, dim3(256), lds, st, a);
            KV_LAUNCH_CHECK("kan_fwd_reg_kernel");
            return 0;
        }
    }
    const LayerArgs& a = a0;
    KV_ALLOW_LDS(160 * 1024, (kan_fwd_reg_kernel<FAM, NT, NSH, ICH, GPC>));
    dim3 grid(gx, (unsigned)tiles, 1);
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
            if constexpr (FAM == KV_BSPLINE || FAM == KV_RBF) { if (a.GP == 9 && a.has_base) return launch_fwd_reg<FAM, NT, NSH, 4, 9>(a, lds, st); }
            if constexpr (FAM == KV_SINE) { if (a.GP == 4) return launch_fwd_reg<FAM, NT, NSH, 4, 4>(a, lds, st); }
        }
        if (ich == 2) {     // GP = 9 (B-spline, FastKAN): eight features x 9 rows x three projections overflow the W staging registers
            if constexpr (FAM == KV_BSPLINE || FAM == KV_RBF) { if (a.GP == 9 && a.has_base) return launch_fwd_reg<FAM, NT, NSH, 2, 9>(a, lds, st); }
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
    // (Measured and rejected, round 3: EIGHT column tiles per generated value for SineKAN's G = 28 patch embedding -- it halves the
    // sine evaluations per MFMA, 5.5 -> 2.9 VALU instructions, but its 114 KB of W per work-group leave one wave per SIMD:
    // 10.06 -> 10.49 ms.)
    int nt = a.O <= 32 ? 1 : (a.O <= 64 ? 2 : 4);
    if (a.O % (32 * nt)) return 1;
    const int nshare = a.groups / a.xmod;
    // Launches that cannot fill the chip (the small geometries' patch embedding: 2048 rows x 64 columns = 16 work-groups of two column
    // tiles): a wave's MFMA chain IS the kernel time there, so one column tile per work-group -- twice (four times) the work-groups, half
    // (a quarter of) the chain each; the basis is re-evaluated per column tile, the k order of every output is unchanged (bitwise equal).
    if (nt > 1 && ((a.M + BM - 1) / BM) * (long long)a.groups * (a.O / (32 * nt)) < N_CU) nt = 1;
    // q|k|v sharing one basis evaluation (NSH = 3) triples the MFMA chain of every wave; when the launch has fewer
    // work-groups than CUs (the small geometries: 50 row tiles x 2 heads) the chain length IS the kernel time, so each
    // projection gets its own work-groups there and re-evaluates the basis
    const bool share3 = kv_shared_basis<FAM>() && kv_share_ok(FAM, a.flags) && nshare == 3 && nt <= 2 &&
                        ((a.M + BM - 1) / BM) * a.xmod >= N_CU;
    const int nsh = share3 ? 3 : 1;
    if ((a.O & 3) || (a.ldy & 3) || ((uintptr_t)a.y & 15) || ((uintptr_t)a.w & 15) || (a.bias && ((uintptr_t)a.bias & 15))) return 1;
    if (FAM == KV_SINE && (((uintptr_t)a.bp & 15) || (a.bp_stride & 3))) return 1;      // the phase rows are prefetched as 16-byte vectors
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


}  // namespace

int kv_try_fwd_reg(int family, const LayerArgs& a, hipStream_t st) {
#define KV_CALL(F) try_fwd_reg<F>(a, st)
    KV_FAMILY_SWITCH(family, KV_CALL)
#undef KV_CALL
}
