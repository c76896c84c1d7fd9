// Multi-head attention core on 16-ROW MFMA tiles (v_mfma_f32_16x16x4_f32, exact fp32), round 4.
//
// Replaces attention.py:199-200 (softmax(q k^T / sqrt(dh)) v inside MSA's python double loop) and utils.py:137-295
// (FlashAttentionFunction forward / recompute backward) for the shape every 224x224 configuration launches: D = 64,
// 64 < N <= 204 (ViT-B/16, ViT-S/16: N = 197), self-attention without a mask.
//
// Why another form.  The fourth form (attention.hip) cuts a head into 32-row tiles: 197 rows pad to 224 (1.29x the
// algorithmic MFMA flops) and seven tiles on four SIMDs are two rounds where 1.75 would do -- a kernel of that shape with a
// 100 % busy matrix pipe reaches 0.68 of the fp32 peak, and it ran at 0.51 / 0.46.  Here
//   * tiles are 16 rows (13 x 16 = 208 padded rows, 1.11x).  Same cycles per flop: 32 per 16x16x4 against 64 per 32x32x2;
//   * a work-group is TWELVE waves (three per SIMD, <= 168 registers): twelve whole tiles, one per wave, and the thirteenth tile
//     (rows 192..N-1) is cut into four KEY quarters handled by waves 0-3 -- the first four waves of a work-group sit on four
//     different SIMDs (MI355X_MICROARCH.md, LDS section: waves go to SIMDs cyclically), so every SIMD carries 3.25 tiles;
//   * the accumulator-as-operand orientation of the older forms is kept (S^T = K.Q^T puts the key in the accumulator register
//     index, so P feeds O^T = V^T.P^T from registers), in its 16x16x4 shape: register r of lane group g = l >> 4 of score tile j
//     is key 16 j + 4 g + r, which is exactly the k index (lane group) of k-step r of the second product;
//   * the head-dimension index of the second product's OUTPUT rows is permuted, row m' of d-tile dt <-> d = 4 m' + dt, so the
//     V operand of all four d-tiles of a k-step is ONE ds_read_b128 (V[key][4 m' .. 4 m' + 3]) and the four accumulators
//     (dt = 0..3) of register r' are four consecutive d: float4 stores straight from registers;
//   * images are the fourth form's: unpadded [row][64] floats filled by LDS-DMA (global_load_lds_dwordx4, the swizzle in the
//     per-lane SOURCE address), three in a ring (K0 V0 K1 V1 ...), one barrier per phase.  The 16-byte slot p of row R holds
//     logical slot p ^ f(R), f(4a + b) = 4b + a: conflict free for both fragment reads (tools/lds_swizzle_check.py);
//   * every wave issues its share of the fills (a thirteenth loader wave would put four waves on one SIMD: 128 registers).
#include "attention_common.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

template <int N, typename F, int... I>
__device__ __forceinline__ void a16_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void a16_static_for(F&& f) { a16_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

typedef __attribute__((address_space(3))) void* a16_lds_ptr;
typedef const __attribute__((address_space(1))) void* a16_glb_ptr;

constexpr int A16_D = 64;
constexpr int A16_NW = 12;                       // waves per work-group
constexpr int A16_THREADS = 64 * A16_NW;
constexpr int A16_MAXT = 13;                     // 16-row tiles of a head: N <= 208 (the LDS limits N to 204, see a16_lds_bytes)
constexpr int A16_SCR = 1024 + 64 + 64 + 16;     // floats behind the images: Q rows of the cut tile, 4 x 16 maxima, 4 x 16 sums, counter

__host__ __device__ constexpr int a16_rows(int N) { return (N + 3) & ~3; }      // image rows: whole 1-KiB pieces (4 rows of 256 bytes)
inline size_t a16_lds_bytes(int N) { return sizeof(float) * ((size_t)3 * a16_rows(N) * A16_D + A16_SCR); }

__device__ __forceinline__ int a16_f(int row) { return 4 * (row & 3) + ((row >> 2) & 3); }

// rows [row0, row0 + 4 * npieces) of a [.][64] fp32 matrix -> swizzled LDS image at img, by LDS-DMA; this wave issues pieces
// first, first + step, ...  Lane l of piece p lands in physical slot 64 p + l = row 4 p + (l >> 4), slot l & 15, and fetches the
// logical slot (l & 15) ^ f(row).  step % 4 == 0, so f(row) is the same for all pieces of a wave.  Rows past N - 1 repeat row N - 1.
__device__ __forceinline__ void a16_fill(float* __restrict__ img, const float* __restrict__ src, int stride_n, int row0, int npieces, int N,
                                         int first, int step, int lane) {
    // the lane index goes through an empty asm: otherwise hipcc forms every piece's per-lane 64-bit source offset once, at kernel
    // entry, keeps them across the head loop and spills them (their reloads then wait for the fills in flight)
    asm volatile("" : "+v"(lane));
    const int lq = lane >> 4;
    for (int p = first; p < npieces; p += step) {
        int row = row0 + 4 * p + lq;
        const int ls = (lane & 15) ^ a16_f(4 * p + lq);
        row = row < N ? row : N - 1;
        __builtin_amdgcn_global_load_lds((a16_glb_ptr)(src + row * stride_n + 4 * ls), (a16_lds_ptr)(img + p * 256), 16, 0, 0);
    }
}

__device__ __forceinline__ f32x4 a16_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// max / sum over the four lane groups (lanes m, m + 16, m + 32, m + 48 hold the same query)
__device__ __forceinline__ float a16_gmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float a16_gsum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// =============================================================================================
// forward
// =============================================================================================
// NKT = the number of 16-row tiles of a head, exactly (host: (N + 15) / 16): every tile loop unrolls without a run-time guard
template <int NKT, bool CAUSAL>
__global__ __launch_bounds__(A16_THREADS) void attn16_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = A16_D;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, g = lane >> 4;
    const int N = a.N, nbh = a.B * a.H;
    constexpr int nt = NKT;                                        // 16-row tiles of a head (keys and queries alike)
    const int R = a16_rows(N), IMG = R * D;                       // floats per image
    const float sc2 = a.scale * KV_LOG2E;
    float* const scr = smem + 3 * IMG;
    float* const q12_s = scr;                                      // [16][64] swizzled: the query rows of the cut tile
    float* const mx4_s = scr + 1024;                               // [4][16]
    float* const sum4_s = scr + 1088;                              // [4][16]
    unsigned* const cnt_s = reinterpret_cast<unsigned*>(scr + 1152);
    const bool has_tile = wave < nt && wave < A16_NW;              // this wave's whole query tile: tile `wave`
    constexpr bool cut = nt > A16_NW;                              // a thirteenth tile exists: key quarters on waves 0-3
    const bool cutw = cut && wave < 4;
    const int npieces = R >> 2;
    const int qrow = 16 * wave + m;
    const bool qok = has_tile && qrow < N;
    const int crow = 16 * A16_NW + m;                              // this lane's query of the cut tile
    // key-tile range of this wave's quarter of the cut tile
    const int kb0 = cutw ? (wave * nt) >> 2 : 0, kb1 = cutw ? ((wave + 1) * nt) >> 2 : 0;

    // per-lane LDS offsets (floats, relative to an image) of the two fragment reads of a 16-row tile at row 0:
    //   rows-as-A (K):  row m, logical slots 4 g + i  -> koff[i];   cols-as-A (V): row 4 g + r, logical slot m -> voff[r]
    // They are re-formed from an opaque copy of the lane index at the start of every phase: as loop invariants hipcc keeps all
    // eight across the head loop, runs out of registers, spills them -- and a scratch reload waits for the fill just issued.
    int koff[4], voff[4];
    auto make_koff = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int mm = l & 15, gg = l >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) koff[i] = mm * D + 4 * ((4 * gg + i) ^ a16_f(mm));
    };
    auto make_voff = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int mm = l & 15, gg = l >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) voff[r] = (4 * gg + r) * D + 4 * (mm ^ a16_f(4 * gg + r));
    };
    // The ragged last tile reads up to 16 * nt - R rows past its image: the head of the next image, or (third image) of the cut
    // tile's query rows.  Both hold finite numbers -- the whole LDS is zeroed once at kernel entry and only ever receives
    // operand rows, finite partial sums and small counters -- the scores of those rows are masked by SELECT and their
    // probabilities are exactly 0, so nothing is clamped and every fragment address is a register plus an immediate.
    constexpr int jl = nt - 1;

    float qf[16];
    auto load_q = [&](int bh) {          // this lane's 16 head-dimension values 16 g .. 16 g + 15 of its query row
        const int bi = bh / a.H, hi = bh - bi * a.H;
        int l = lane;                        // (opaque: the per-lane offset is re-formed per head instead of being hoisted and spilled)
        asm volatile("" : "+v"(l));
        const int qr = 16 * wave + (l & 15);
        const float* qp = a.q + bi * a.qsb + hi * a.qsh + (qr < N ? qr : 0) * (int)a.qsn + 16 * (l >> 4);      // 32-bit row offsets: a16_shape_ok
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(qp + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[4 * i + e] = u[e];
        }
    };
    // fills: image `item` <- rows of src; with_q: also the cut tile's query rows (4 pieces, waves 4-7)
    auto fill = [&](int item, const float* base, long long sn) {
        a16_fill(smem + (item % 3) * IMG, base, (int)sn, 0, npieces, N, wave, A16_NW, lane);
    };
    auto fill_q12 = [&](int bh) {
        if (cut && wave >= 4 && wave < 8) {
            const int bi = bh / a.H, hi = bh - bi * a.H;
            a16_fill(q12_s, a.q + bi * a.qsb + hi * a.qsh, (int)a.qsn, 16 * A16_NW, 4, N, wave - 4, 4, lane);
        }
    };

    f32x4 oacc[4];
    float o_inv = 0.0f, o_lse = 0.0f;
    int o_bh = -1;                       // head whose output tile waits in oacc (stored after the next phase's barrier)
    auto store_o = [&]() {               // the whole tile's output: 4 consecutive d per (r', lane group)
        if (o_bh >= 0) {
            if (qok) {
                const int bi = o_bh / a.H, hi = o_bh - bi * a.H;
                int l = lane;
                asm volatile("" : "+v"(l));
                float* op = a.out + bi * a.osb + hi * a.osh + (16 * wave + (l & 15)) * (int)a.osn + 16 * (l >> 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4 v = {oacc[0][r] * o_inv, oacc[1][r] * o_inv, oacc[2][r] * o_inv, oacc[3][r] * o_inv};
                    *reinterpret_cast<f32x4*>(op + 4 * r) = v;
                }
                if (l < 16 && a.lse) a.lse[(long long)o_bh * N + 16 * wave + l] = o_lse;      // (a hoisted per-lane lse pointer is spilled, and its reload waits for the fill just issued)
            }
            o_bh = -1;
        }
    };

    for (int e = tid * 4; e < 3 * IMG + A16_SCR; e += A16_THREADS * 4) *reinterpret_cast<f32x4*>(smem + e) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};      // (also the counter)
    __syncthreads();
    int item = 0;
    unsigned cut_done = 0;                                           // heads of the cut tile finished (4 arrivals each)
    int bh = blockIdx.x;
    if (bh < nbh) {
        const int bi = bh / a.H, hi = bh - bi * a.H;
        fill(0, a.k + bi * a.ksb + hi * a.ksh, a.ksn);
        fill_q12(bh);
        if (has_tile) load_q(bh);
    }
    for (; bh < nbh; bh += gridDim.x) {
        const int bi = bh / a.H, hi = bh - bi * a.H;
        // ---------------- phase S: scores and softmax from the K image (item); the V fill in flight ----------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(qf[e]));      // hipcc's own wait for the Q registers lands here, not behind the next fill
        __builtin_amdgcn_s_barrier();
        fill(item + 1, a.v + bi * a.vsb + hi * a.vsh, a.vsn);
        store_o();
        const float* K_s = smem + (item % 3) * IMG;
        make_koff();
        f32x4 sacc[NKT];
        float inv = 0.0f, lse_v = 0.0f;
        if (has_tile) {
#pragma unroll
            for (int j = 0; j < NKT; ++j) sacc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            // Two key tiles at a time: their MFMAs alternate, so no instruction waits for the accumulator of its predecessor (the
            // 16x16x4 shape returns its result after 40 cycles and issues every 32).  A granule = one 16-byte K read per tile of
            // the pair = 8 MFMAs; the reads of granule t + 2 are issued in front of the MFMAs of granule t (ring of three).
            constexpr int NG = ((NKT + 1) / 2) * 4;
            f32x4 fa[3], fb[3];
            auto rd = [&](auto tc, f32x4& xa, f32x4& xb) {
                constexpr int t = decltype(tc)::value, j = 2 * (t >> 2), i = t & 3;
                xa = *reinterpret_cast<const f32x4*>(K_s + koff[i] + j * 16 * D);
                if constexpr (j + 1 < NKT) xb = *reinterpret_cast<const f32x4*>(K_s + koff[i] + (j + 1) * 16 * D);
            };
            rd(std::integral_constant<int, 0>{}, fa[0], fb[0]);
            rd(std::integral_constant<int, 1>{}, fa[1], fb[1]);
            a16_static_for<NG>([&](auto tc) {
                constexpr int t = decltype(tc)::value, j = 2 * (t >> 2), i = t & 3;
                if constexpr (t + 2 < NG) rd(std::integral_constant<int, t + 2>{}, fa[(t + 2) % 3], fb[(t + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sacc[j] = a16_mfma(fa[t % 3][e], qf[4 * i + e], sacc[j]);
                    if constexpr (j + 1 < NKT) sacc[j + 1] = a16_mfma(fb[t % 3][e], qf[4 * i + e], sacc[j + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            const int lim = CAUSAL ? (qrow + 1 < N ? qrow + 1 : N) : N;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if (CAUSAL || j == jl) {              // compile time: the ragged last tile, or every tile under the causal mask
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[j][r] = (16 * j + 4 * g + r >= lim) ? -INFINITY : sacc[j][r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[j][r]);
            }
            mx = a16_gmax(mx);
            const float mxs = (mx == -INFINITY) ? 0.0f : mx * sc2;
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sacc[j][r] * sc2 - mxs);      // arguments <= 0
                    sacc[j][r] = p;
                    sum += p;
                }
            }
            sum = a16_gsum(sum);
            inv = 1.0f / sum;
            lse_v = mx * a.scale + logf(sum);
        }
        // the cut tile: this wave's key quarter [kb0, kb1) -- scores now, the maximum over the quarter to the exchange slots
        f32x4 cacc[4];
        if (cutw) {
            float cq[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(q12_s + koff[i]);      // row m of the [16][64] image, this lane group's 16 values
#pragma unroll
                for (int e = 0; e < 4; ++e) cq[4 * i + e] = u[e];
            }
            const int clim = CAUSAL ? (crow + 1 < N ? crow + 1 : N) : N;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) cacc[jj] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            // a quarter is three or four key tiles (13 = 3 + 3 + 3 + 4): tiles 0 and 1 as an interleaved pair, then tile 2 alone or
            // with tile 3 -- the same two-accumulator interleave as above
            const float* Kq = K_s + kb0 * 16 * D;
            auto cut_s = [&](auto j0c, auto twoc) {
                constexpr int j0 = decltype(j0c)::value;
                constexpr bool two = decltype(twoc)::value;
                f32x4 ka[4], kb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ka[i] = *reinterpret_cast<const f32x4*>(Kq + koff[i] + j0 * 16 * D);
                    if constexpr (two) kb[i] = *reinterpret_cast<const f32x4*>(Kq + koff[i] + (j0 + 1) * 16 * D);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cacc[j0] = a16_mfma(ka[i][e], cq[4 * i + e], cacc[j0]);
                        if constexpr (two) cacc[j0 + 1] = a16_mfma(kb[i][e], cq[4 * i + e], cacc[j0 + 1]);
                    }
            };
            cut_s(std::integral_constant<int, 0>{}, std::true_type{});
            if (kb1 - kb0 == 4) cut_s(std::integral_constant<int, 2>{}, std::true_type{});
            else cut_s(std::integral_constant<int, 2>{}, std::false_type{});
            float mx = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = kb0 + jj;
                if (CAUSAL || jj >= 2) {              // (the ragged last tile is the third or fourth of its quarter; an absent fourth is all masked)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cacc[jj][r] = (j >= kb1 || 16 * j + 4 * g + r >= clim) ? -INFINITY : cacc[jj][r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, cacc[jj][r]);
            }
            mx = a16_gmax(mx);
            if (g == 0) mx4_s[wave * 16 + m] = mx;
        }
        ++item;
        // ---------------- phase P.V from the V image (item); the next head's K fill and Q rows in flight ----------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int bhn = bh + gridDim.x;
        if (bhn < nbh) {
            const int bn = bhn / a.H, hn = bhn - bn * a.H;
            fill(item + 1, a.k + bn * a.ksb + hn * a.ksh, a.ksn);
            fill_q12(bhn);
            if (has_tile) load_q(bhn);
        }
        const float* V_s = smem + (item % 3) * IMG;
        make_voff();
        float* const part_s = smem + ((item + 2) % 3) * IMG;          // the K image of this head is dead: exchange buffer of the cut tile
        if (cutw) {
            // global maximum of the cut tile's queries, probabilities of this quarter, partial O^T over its keys
            const float mx = fmaxf(fmaxf(mx4_s[m], mx4_s[16 + m]), fmaxf(mx4_s[32 + m], mx4_s[48 + m]));
            const float mxs = (mx == -INFINITY) ? 0.0f : mx * sc2;
            float sum = 0.0f;
            f32x4 pacc[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) pacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(cacc[jj][r] * sc2 - mxs);      // (an absent fourth tile: exp2(-inf) = 0)
                    cacc[jj][r] = p;
                    sum += p;
                }
            const float* Vq = V_s + kb0 * 16 * D;
            auto cut_pv = [&](auto ngc) {            // granules (tile jj, k-step r) of this quarter, reads two granules ahead
                constexpr int NGC = decltype(ngc)::value;
                f32x4 fv[3];
                fv[0] = *reinterpret_cast<const f32x4*>(Vq + voff[0]);
                fv[1] = *reinterpret_cast<const f32x4*>(Vq + voff[1]);
                a16_static_for<NGC>([&](auto tc) {
                    constexpr int t = decltype(tc)::value, jj = t >> 2, r = t & 3;
                    if constexpr (t + 2 < NGC) fv[(t + 2) % 3] = *reinterpret_cast<const f32x4*>(Vq + voff[(t + 2) & 3] + ((t + 2) >> 2) * 16 * D);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) pacc[dt] = a16_mfma(fv[t % 3][dt], cacc[jj][r], pacc[dt]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            if (kb1 - kb0 == 4) cut_pv(std::integral_constant<int, 16>{});
            else cut_pv(std::integral_constant<int, 12>{});
            sum = a16_gsum(sum);
            if (g == 0) sum4_s[wave * 16 + m] = sum;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 v = {pacc[0][r], pacc[1][r], pacc[2][r], pacc[3][r]};
                *reinterpret_cast<f32x4*>(part_s + ((wave * 4 + r) * 64 + lane) * 4) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (workgroup scope: hipcc puts s_waitcnt vmcnt(0) in front of an agent-scope atomic -- here that is a wait for this wave's
            // share of the next head's fill, issued a moment ago)
            if (lane == 0) __hip_atomic_fetch_add(cnt_s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (has_tile) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            // O^T[d = 4 m' + dt][query] += V[key][4 m' + dt] P[key][query]: k-step r of key tile j is keys 16 j + 4 g + r = accumulator
            // register r of the score tile; one 16-byte V read feeds the four d-tiles
            // granule = one V read = four MFMAs (the four d-tiles); the read of granule t + 2 in front of the MFMAs of granule t
            constexpr int NGV = 4 * NKT;
            f32x4 fv[3];
            auto rdv = [&](auto tc, f32x4& x) {
                constexpr int t = decltype(tc)::value, j = t >> 2, r = t & 3;
                x = *reinterpret_cast<const f32x4*>(V_s + voff[r] + j * 16 * D);
            };
            rdv(std::integral_constant<int, 0>{}, fv[0]);
            rdv(std::integral_constant<int, 1>{}, fv[1]);
            a16_static_for<NGV>([&](auto tc) {
                constexpr int t = decltype(tc)::value, j = t >> 2, r = t & 3;
                if constexpr (t + 2 < NGV) rdv(std::integral_constant<int, t + 2>{}, fv[(t + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[dt] = a16_mfma(fv[t % 3][dt], sacc[j][r], oacc[dt]);
                __builtin_amdgcn_sched_barrier(0);
            });
            o_bh = bh;
            o_inv = inv;
            o_lse = lse_v;
        }
        if (cutw) {
            // wave q finishes register row r = q of the cut tile: the four partials in quarter order, the total sum, the store
            ++cut_done;
            while (*reinterpret_cast<volatile unsigned*>(cnt_s) < 4u * cut_done) __builtin_amdgcn_s_sleep(1);
            f32x4 t = *reinterpret_cast<const f32x4*>(part_s + ((0 * 4 + wave) * 64 + lane) * 4);
#pragma unroll
            for (int q = 1; q < 4; ++q) t += *reinterpret_cast<const f32x4*>(part_s + ((q * 4 + wave) * 64 + lane) * 4);
            const float mx = fmaxf(fmaxf(mx4_s[m], mx4_s[16 + m]), fmaxf(mx4_s[32 + m], mx4_s[48 + m]));
            const float sum = ((sum4_s[m] + sum4_s[16 + m]) + sum4_s[32 + m]) + sum4_s[48 + m];
            const float cinv = 1.0f / sum;
            int l = lane;
            asm volatile("" : "+v"(l));
            const int cl = 16 * A16_NW + (l & 15);
            if (cl < N) {
                *reinterpret_cast<f32x4*>(a.out + bi * a.osb + hi * a.osh + cl * (int)a.osn + 16 * (l >> 4) + 4 * wave) = t * cinv;
                if (wave == 0 && l < 16 && a.lse) a.lse[(long long)bh * N + cl] = mx * a.scale + logf(sum);
            }
        }
        ++item;
    }
    store_o();
}

template <int NKT, bool CAUSAL>
int launch_fwd16c(const AttnArgs& a, hipStream_t st) {
    const size_t lds = a16_lds_bytes(a.N);
    KV_ALLOW_LDS(160 * 1024, (attn16_fwd_kernel<NKT, CAUSAL>));
    const int nbh = a.B * a.H;
    const int gmax = kv_config().attn_grid > 0 ? kv_config().attn_grid : KV_N_CU;      // one work-group per CU (the three images fill its LDS)
    hipLaunchKernelGGL((attn16_fwd_kernel<NKT, CAUSAL>), dim3((unsigned)(nbh < gmax ? nbh : gmax)), dim3(A16_THREADS), lds, st, a);
    KV_LAUNCH_CHECK("attn16_fwd_kernel");
    return 0;
}

template <int NKT>
int launch_fwd16(const AttnArgs& a, hipStream_t st) { return launch_fwd16c<NKT, false>(a, st); }      // (the causal instantiation of 13 tiles spills: causal launches keep the fourth form)

bool a16_shape_ok(const AttnArgs& a) {
    if (a.D != A16_D || a.N <= 64 || a16_lds_bytes(a.N) > 160 * 1024 || (a.N + 15) / 16 > A16_MAXT || !a.vec) return false;
    // 32-bit row offsets inside a head's operand (row * stride elements)
    for (long long sn : {a.qsn, a.ksn, a.vsn, a.osn})
        if (sn * (long long)(a.N + 1) >= (1LL << 31)) return false;
    return true;
}

}  // namespace

int kv_attn16_fwd(const AttnArgs& a, hipStream_t st) {
    if (!a16_shape_ok(a) || a.causal || kv_config().attn_v4) return 1;
    if (((uintptr_t)a.out | (uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) % 16) return 1;
    switch ((a.N + 15) / 16) {
        case 5: return launch_fwd16<5>(a, st);
        case 6: return launch_fwd16<6>(a, st);
        case 7: return launch_fwd16<7>(a, st);
        case 8: return launch_fwd16<8>(a, st);
        case 9: return launch_fwd16<9>(a, st);
        case 10: return launch_fwd16<10>(a, st);
        case 11: return launch_fwd16<11>(a, st);
        case 12: return launch_fwd16<12>(a, st);
        case 13: return launch_fwd16<13>(a, st);
        default: return 1;
    }
}

bool kv_attn16_bwd_ok(const kanvit_attn_desc*) { return false; }
int kv_attn16_bwd(const AttnArgs&, hipStream_t) { return 1; }
