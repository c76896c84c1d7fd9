// Register-form and W-stationary forward on the bf16 matrix cores (split out of kan_layer.hip; see kan_layer_common.h).
#include "kan_layer_common.h"

namespace {

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

// PG (kanvit_patch_embed_fwd_ws): rows gathered from the NCHW images, position embedding added and class-token rows written
// in the epilogue -- the prologue / epilogue of kan_fwd_reg_kernel's patch form on the bf16 matrix cores.
template <int FAM, int GP, int NT, int NSH, int ICH, bool PG = false>
__global__ __launch_bounds__(256) void kan_fwd_reg_bf16_kernel(const LayerArgs a) {
    static_assert(!(PG && (FAM == KV_RBF || NSH != 1)), "patch gather: one plain layer, no LayerNorm'ed second input");
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

    // patch gather: the lane's ICH features of a chunk are ICH consecutive pixels of one image line (host-checked: the chunk
    // divides the patch width); chunks are visited in order, so the position inside the patch advances incrementally
    int pg_ix = 0, pg_iy = 0, pg_off = 0, pg_pw = 0, pg_ph = 0;
    if constexpr (PG) {
        const int P = a.pg_n * a.pg_n;
        pg_ph = a.pg_H / a.pg_n;
        pg_pw = a.pg_W / a.pg_n;
        const long long m = m0 + (row_ok ? row : 0);
        const long long smp = m / P;
        const int pidx = (int)(m - smp * P);
        const int py = pidx / a.pg_n, px = pidx - py * a.pg_n;
        xrow = a.x + ((smp * a.pg_C) * a.pg_H + (long long)py * pg_ph) * a.pg_W + px * pg_pw;      // patch origin in channel 0
        pg_ix = pg_off = hf * ICH;
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
                if (q * 256 < NV)             // unconditional inside a live pass (a predicated load is a branch + a vmcnt wait per load); lanes past NV re-read vector 0
                    wreg[p][q] = *reinterpret_cast<const u32x4*>(src + (v < NV ? ((long long)kr * a.O + n0 + n) * 8 : (long long)n0 * 8));
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
        const float* xs = PG ? xrow + pg_off : xrow + c * IC;
        if constexpr (PG) {                       // next chunk: IC pixels further along the line, then next line, then next channel
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
        if constexpr (ICH % 4 == 0) {
#pragma unroll
            for (int j4 = 0; j4 < ICH / 4; ++j4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xs + 4 * j4);
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
                xv[e] = xs[e];
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

    // SINE: the phases of the chunk's features travel one chunk ahead like x (read inside the basis evaluation they are requested
    // after the next chunk's prefetch and, memory returning in order, wait for all of it -- see kan_fwd_reg_kernel)
    constexpr bool SINE_PH = (FAM == KV_SINE);
    constexpr int NPH = SINE_PH ? VH : 1;
    float phv[NPH];
    auto load_ph = [&](int c) {
        if constexpr (SINE_PH) {
            const float* ps = b.bp + GP + (long long)(c * IC + hf * ICH) * GP;      // [ICH features][GP phases], contiguous (host: 16-byte aligned)
#pragma unroll
            for (int e4 = 0; e4 < NPH / 4; ++e4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ps + 4 * e4);
#pragma unroll
                for (int e = 0; e < 4; ++e) phv[4 * e4 + e] = v[e];
            }
        }
    };
    static_assert(!SINE_PH || VH % 4 == 0, "phase rows are read as 16-byte vectors");

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
        if (c + 1 < nch) {
            load_w(c + 1);
            load_x(c + 1);
            load_ph(c + 1);
        }
        const unsigned short* wp = W_s + (size_t)(c & 1) * WSZ + ((size_t)hf * WROW + l31) * 8;
        BasisGen<FAM, kv_gc(FAM, GP)> gen;
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
                    if constexpr (SINE_PH) {
                        av[e] = kv_sin(__fadd_rn(__fmul_rn(xc[j], b.bp[g]), phc[vi]));      // BasisGen<KV_SINE>::next with the phase from registers
                    } else {
                        if (g == 0) gen.init(b, xc[j], uc[j], c * IC + hf * ICH + j);
                        av[e] = gen.next(g);
                    }
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
    // patch form: the rows this lane STORES (er + 8q of the wave's strip) sit pg_pre rows per image further down, get their
    // position-embedding row added, and the first patch of an image also writes the image's class-token row cls + pos[0]
    long long pg_yrow[PG ? 4 : 1];
    int pg_pidx[PG ? 4 : 1];
    if constexpr (PG) {
        const int P = a.pg_n * a.pg_n;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long m = m0 + wave * 32 + er + q * 8;
            const long long smp = m / P;
            pg_pidx[q] = (int)(m - smp * P);
            pg_yrow[q] = m + (smp + 1) * a.pg_pre;
        }
    }
#pragma unroll
    for (int t = 0; t < NSH * NT; ++t) {
        const int p = t / NT, nt = t - p * NT;
        const int g = (NSH == 1) ? gs : p * nsets + gs;
#pragma unroll
        for (int r = 0; r < 16; ++r) T_w[kv_acc_row(r, hf) * TS + l31] = acc[t][r];
        f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
        const int col = n0 + nt * 32 + ec;
        if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + (long long)g * a.O + col);
        float* yt = a.y + (m0 + wave * 32) * a.ldy + (long long)g * a.O + col;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = er + q * 8;
            if (wave * 32 + rr < mrem) {
                f32x4 v = *reinterpret_cast<const f32x4*>(T_w + rr * TS + ec);
                v += bv;
                if constexpr (PG) {
                    if (a.pos) v += *reinterpret_cast<const f32x4*>(a.pos + (long long)(pg_pidx[q] + a.pg_pre) * a.O + col);
                    float* yp = a.y + pg_yrow[q] * a.ldy + col;
                    *reinterpret_cast<f32x4*>(yp) = v;
                    if (a.pg_pre && a.cls && pg_pidx[q] == 0) {
                        f32x4 cv = *reinterpret_cast<const f32x4*>(a.cls + col);
                        if (a.pos) cv += *reinterpret_cast<const f32x4*>(a.pos + col);
                        *reinterpret_cast<f32x4*>(yp - a.ldy) = cv;
                    }
                } else {
                    *reinterpret_cast<f32x4*>(yt + (long long)rr * a.ldy) = v;
                }
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
//   * (round 4) the stores: an accumulator quad is 16 bytes of the lane's own row, so a store instruction used to touch 32 rows with
//     32 bytes each -- four instructions, 128 write transactions per 128-byte line set; without its stores the ViT-B q|k|v launch
//     ran in 59 instead of 100 us.  With ST a tile passes through a wave-private LDS strip ([32 rows][36 floats]: b128 writes and
//     reads conflict free) and leaves as four instructions of eight whole 128-byte lines each.
template <int FAM, int GP, int NT, int NSH, int ICH, int NCH, bool ST = false>
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
    constexpr int STW = 36;                       // row stride of the store strip (floats)
    float* strip = bias_s + WROW + (ST ? wave * (32 * STW) : 0);      // ST: [32 rows][STW] of this wave

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
            BasisGen<FAM, kv_gc(FAM, GP)> gen;
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
            if constexpr (ST) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = t0 + tt;
                    if (t < NTT) {
                        const int p = t / NT, nt = t - p * NT;
                        const int g = (NSH == 1) ? gs : p * nsets + gs;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<f32x4*>(strip + l31 * STW + 8 * q + 4 * hf) =
                                tt == 0 ? f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]}
                                        : f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
                        const int pc = lane & 7;                           // 16-byte piece of the 128-byte row segment
                        f32x4 bv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        if (has_bias) bv = *reinterpret_cast<const f32x4*>(bias_s + t * 32 + 4 * pc);
                        const long long r0 = tile * ROWS + wave * 32 + (lane >> 3);
                        float* yp = a.y + r0 * a.ldy + (long long)g * a.O + n0 + nt * 32 + 4 * pc;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            f32x4 v = *reinterpret_cast<const f32x4*>(strip + (8 * k + (lane >> 3)) * STW + 4 * pc);
                            v += bv;
                            if (FULL || r0 + 8 * k < a.M) *reinterpret_cast<f32x4*>(yp + (long long)(8 * k) * a.ldy) = v;
                        }
                    }
                }
            } else if (st_ok) {
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

template <int FAM, int GP, int NT, int NSH, int ICH>
int launch_fwd_reg_bf16(const LayerArgs& a, const FwdRegBf16Plan& p, hipStream_t st) {
    // W-stationary persistent form when the whole weight image of a column set fits the LDS and there are enough row tiles
    // (instantiated for I = 64 per group: 4 chunks of 16 features -- the per-head q|k|v launches of ViT-B/S)
    if constexpr (ICH == 8 && (FAM == KV_LINEAR || FAM == KV_CHEBY)) {     // the families whose basis fragments fit the register file
        const size_t wlds = (size_t)p.nch * p.vs * 2 * 32 * NT * NSH * 16 + sizeof(float) * 32 * NT * NSH;
        const int gx = (a.groups / NSH) * (a.O / (32 * NT));
        if (wlds <= 150 * 1024 && p.nch == 4 && a.M >= 4096 && gx <= N_CU && !((uintptr_t)a.y & 15) && !kv_config().no_ws && !a.pg) {
            const long long ntiles = (a.M + KV_WS_THREADS / 2 - 1) / (KV_WS_THREADS / 2);
            long long py = N_CU / gx;             // one work-group per CU (the image fills the LDS)
            if (py > ntiles) py = ntiles;
            if (py < 1) py = 1;
            const size_t slds = wlds + sizeof(float) * (KV_WS_THREADS / 64) * 32 * 36;       // + the store strips (see the kernel), when they fit
            if (slds <= 160 * 1024 && !kv_config().ws_no_strip) {
                KV_ALLOW_LDS(160 * 1024, (kan_fwd_ws_bf16_kernel<FAM, GP, NT, NSH, ICH, 4, true>));
                hipLaunchKernelGGL((kan_fwd_ws_bf16_kernel<FAM, GP, NT, NSH, ICH, 4, true>), dim3((unsigned)gx, (unsigned)py, 1), dim3(KV_WS_THREADS), slds, st, a);
            } else {
                KV_ALLOW_LDS(160 * 1024, (kan_fwd_ws_bf16_kernel<FAM, GP, NT, NSH, ICH, 4>));
                hipLaunchKernelGGL((kan_fwd_ws_bf16_kernel<FAM, GP, NT, NSH, ICH, 4>), dim3((unsigned)gx, (unsigned)py, 1), dim3(KV_WS_THREADS), wlds, st, a);
            }
            KV_LAUNCH_CHECK("kan_fwd_ws_bf16_kernel");
            return 0;
        }
    }
    dim3 grid((unsigned)((a.groups / NSH) * (a.O / (32 * NT))), (unsigned)((a.M + BM - 1) / BM), 1);
    if (a.pg) {         // fused patch embedding: one wide layer (NT = 4), the families whose patch embedding the model builds
        if constexpr (NT == 4 && NSH == 1 && FAM != KV_RBF && FAM != KV_LINEAR) {
            KV_ALLOW_LDS(160 * 1024, (kan_fwd_reg_bf16_kernel<FAM, GP, NT, NSH, ICH, true>));
            hipLaunchKernelGGL((kan_fwd_reg_bf16_kernel<FAM, GP, NT, NSH, ICH, true>), grid, dim3(256), p.lds, st, a);
            KV_LAUNCH_CHECK("kan_fwd_reg_bf16_kernel (patch gather)");
            return 0;
        }
        return 1;       // not covered
    }
    KV_ALLOW_LDS(160 * 1024, (kan_fwd_reg_bf16_kernel<FAM, GP, NT, NSH, ICH>));
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


}  // namespace

FwdRegBf16Plan plan_fwd_reg_bf16(const kanvit_layer_desc* d) {
    FwdRegBf16Plan p{};
    if (kv_config().no_reg) return p;
    p.gp = gp_of(d);
    const int fam = d->family;
    const bool gp_ok = (fam == KANVIT_LINEAR && p.gp == 1) || (fam == KANVIT_CHEBY && p.gp == 5) ||
                       (fam == KANVIT_BSPLINE && p.gp == 9 && d->has_base && (d->flags & KANVIT_FLAG_UNIFORM_KNOTS) && d->spline_order == 3) ||
                       (fam == KANVIT_RBF && p.gp == 9 && d->has_base && kv_rbf_reg_ok(d->flags, d->G)) || (fam == KANVIT_SINE && (p.gp == 4 || p.gp == 28)) ||
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
    if (fam == KANVIT_SINE && (d->bparam_stride & 3)) return p;      // phase rows are prefetched as 16-byte vectors (the pointer itself is checked at launch)
    if ((p.vs * 2 * 32 * p.nt + 255) / 256 > 12) return p;
    p.ws_bytes = (size_t)d->groups * p.nch * p.vs * 2 * d->O * 16;
    p.ok = true;
    return p;
}

int kv_fwd_reg_bf16(int family, LayerArgs& a, const FwdRegBf16Plan& p, void* ws, hipStream_t st) {
#define KV_CALL(F) dispatch_fwd_reg_bf16<F>(a, p, ws, st)
    KV_FAMILY_SWITCH(family, KV_CALL)
#undef KV_CALL
}
