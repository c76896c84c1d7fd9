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

// Counters in LDS, polled.  The pointer is cast to the LDS address space: through a generic pointer hipcc emits a FLAT load, and a
// flat load is followed by s_waitcnt vmcnt(0) -- every poll would drain the LDS-DMA fills the wave has in flight (measured on the
// first version of the backward: the 270 us of hand-offs and the 280 us of MFMAs did not overlap at all).
typedef __attribute__((address_space(3))) unsigned a16_lds_u32;
__device__ __forceinline__ void a16_wait(const unsigned* c, unsigned target) {
    const volatile a16_lds_u32* p = (const volatile a16_lds_u32*)c;
    while (*p < target) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void a16_signal(unsigned* c, int lane, unsigned n = 1u) {
    if (lane == 0) __hip_atomic_fetch_add((a16_lds_u32*)c, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ f32x4 a16_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// bf16 matrix-core mode (KANVIT_FLAG_BF16_MFMA): v_mfma_f32_16x16x32_bf16 -- lane (m = l & 15, g = l >> 4) supplies the eight k-slots
// 8 g .. 8 g + 7 of row / column m; operands are rounded to bf16 (v_cvt_pk_bf16_f32), products accumulate in fp32, all I/O stays fp32.
typedef __bf16 a16_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 a16_bf2 __attribute__((ext_vector_type(2)));
typedef unsigned a16_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned a16_pk(float lo, float hi) {
    a16_bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ a16_bf8 a16_pack8(float f0, float f1, float f2, float f3, float f4, float f5, float f6, float f7) {
    const a16_u4 u = {a16_pk(f0, f1), a16_pk(f2, f3), a16_pk(f4, f5), a16_pk(f6, f7)};
    return __builtin_bit_cast(a16_bf8, u);
}
__device__ __forceinline__ f32x4 a16_mfma_bf(a16_bf8 a, a16_bf8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// v_mfma_f32_16x16x16_bf16: k = 16, lane group g supplies k-slots 4 g .. 4 g + 3 (the products of the backward that contract over the 16
// queries of a step or the 16 keys of a tile)
typedef short a16_s4 __attribute__((ext_vector_type(4)));
typedef unsigned a16_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ a16_s4 a16_pack4(float f0, float f1, float f2, float f3) {
    const a16_u2 u = {a16_pk(f0, f1), a16_pk(f2, f3)};
    return __builtin_bit_cast(a16_s4, u);
}
__device__ __forceinline__ f32x4 a16_mfma_bf4(a16_s4 a, a16_s4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

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
// BF: the two products on the bf16 matrix cores.  Same images, fills, phases, fragment reads and accumulator layouts; a score tile is two
// MFMAs (k = 64 head-dimension values, lane group g holds d = 16 g + 8 c + j of MFMA c) instead of sixteen, and P.V contracts two key
// tiles per MFMA (k-slot j of lane group g is key 4 g + j of the first tile, 4 g + j - 4 of the second: the accumulator registers of
// the two score tiles as they stand), the four d-tiles' V operands gathered from eight 16-byte reads.
template <int NKT, bool CAUSAL, bool BF = false>
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

    // (Measured, no effect: s_setprio 1 for waves 0-3, which carry a quarter of the cut tile on top of their own.)
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
        const float* K_s = smem + (item % 3) * IMG;
        make_koff();
        f32x4 sacc[NKT];
        a16_bf8 pbs[BF ? (NKT + 1) / 2 : 1];      // BF: the probabilities of key tiles 2 t, 2 t + 1 as the B operand of their P.V MFMA (packed here: half the registers across the barrier)
        float inv = 0.0f, lse_v = 0.0f;
        if (has_tile) {
#pragma unroll
            for (int j = 0; j < NKT; ++j) sacc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (BF) {
                const a16_bf8 qb0 = a16_pack8(qf[0], qf[1], qf[2], qf[3], qf[4], qf[5], qf[6], qf[7]);
                const a16_bf8 qb1 = a16_pack8(qf[8], qf[9], qf[10], qf[11], qf[12], qf[13], qf[14], qf[15]);
#pragma unroll
                for (int j = 0; j < NKT; ++j) {
                    f32x4 ka[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) ka[i] = *reinterpret_cast<const f32x4*>(K_s + koff[i] + j * 16 * D);
                    sacc[j] = a16_mfma_bf(a16_pack8(ka[0][0], ka[0][1], ka[0][2], ka[0][3], ka[1][0], ka[1][1], ka[1][2], ka[1][3]), qb0, sacc[j]);
                    sacc[j] = a16_mfma_bf(a16_pack8(ka[2][0], ka[2][1], ka[2][2], ka[2][3], ka[3][0], ka[3][1], ka[3][2], ka[3][3]), qb1, sacc[j]);
                    if (j & 1) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
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
            }
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
            if constexpr (BF) {
#pragma unroll
                for (int t = 0; t < (NKT + 1) / 2; ++t) {
                    constexpr int JL = NKT - 1;
                    const bool pair = 2 * t + 1 < NKT;
                    const int j1 = pair ? 2 * t + 1 : JL;
                    pbs[t] = a16_pack8(sacc[2 * t][0], sacc[2 * t][1], sacc[2 * t][2], sacc[2 * t][3], pair ? sacc[j1][0] : 0.0f, pair ? sacc[j1][1] : 0.0f,
                                       pair ? sacc[j1][2] : 0.0f, pair ? sacc[j1][3] : 0.0f);
                }
            }
        }
        store_o();      // the previous head's tile: behind this phase's barrier and fill (in front of the barrier the stores sat in every wave's
                        // vmcnt(0)), and behind the score products -- the matrix pipe starts right after the barrier
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
            if constexpr (BF) {
                const a16_bf8 qb0 = a16_pack8(cq[0], cq[1], cq[2], cq[3], cq[4], cq[5], cq[6], cq[7]);
                const a16_bf8 qb1 = a16_pack8(cq[8], cq[9], cq[10], cq[11], cq[12], cq[13], cq[14], cq[15]);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {       // (an absent fourth tile reads the next quarter's first tile: its scores are masked below)
                    if (jj < 3 || kb1 - kb0 == 4) {
                        f32x4 ka[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) ka[i] = *reinterpret_cast<const f32x4*>(Kq + koff[i] + jj * 16 * D);
                        cacc[jj] = a16_mfma_bf(a16_pack8(ka[0][0], ka[0][1], ka[0][2], ka[0][3], ka[1][0], ka[1][1], ka[1][2], ka[1][3]), qb0, cacc[jj]);
                        cacc[jj] = a16_mfma_bf(a16_pack8(ka[2][0], ka[2][1], ka[2][2], ka[2][3], ka[3][0], ka[3][1], ka[3][2], ka[3][3]), qb1, cacc[jj]);
                    }
                }
            } else {
                cut_s(std::integral_constant<int, 0>{}, std::true_type{});
                if (kb1 - kb0 == 4) cut_s(std::integral_constant<int, 2>{}, std::true_type{});
                else cut_s(std::integral_constant<int, 2>{}, std::false_type{});
            }
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
            if constexpr (BF) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {          // key tiles 2 t, 2 t + 1 of the quarter (an absent fourth: probabilities 0, finite V rows of the next quarter)
                    f32x4 v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = *reinterpret_cast<const f32x4*>(Vq + voff[r] + (2 * t) * 16 * D);
                        v[4 + r] = *reinterpret_cast<const f32x4*>(Vq + voff[r] + (2 * t + 1) * 16 * D);
                    }
                    const a16_bf8 pb = a16_pack8(cacc[2 * t][0], cacc[2 * t][1], cacc[2 * t][2], cacc[2 * t][3], cacc[2 * t + 1][0], cacc[2 * t + 1][1],
                                                 cacc[2 * t + 1][2], cacc[2 * t + 1][3]);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        pacc[dt] = a16_mfma_bf(a16_pack8(v[0][dt], v[1][dt], v[2][dt], v[3][dt], v[4][dt], v[5][dt], v[6][dt], v[7][dt]), pb, pacc[dt]);
                }
            } else {
                if (kb1 - kb0 == 4) cut_pv(std::integral_constant<int, 16>{});
                else cut_pv(std::integral_constant<int, 12>{});
            }
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
            a16_signal(cnt_s, lane);
        }
        if (has_tile) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            // O^T[d = 4 m' + dt][query] += V[key][4 m' + dt] P[key][query]: k-step r of key tile j is keys 16 j + 4 g + r = accumulator
            // register r of the score tile; one 16-byte V read feeds the four d-tiles
            if constexpr (BF) {
#pragma unroll
                for (int t = 0; t < (NKT + 1) / 2; ++t) {      // key tiles 2 t and 2 t + 1 in one MFMA per d-tile (a missing last partner: probabilities 0)
                    constexpr int JL = NKT - 1;
                    const int j1 = (2 * t + 1 < NKT) ? 2 * t + 1 : JL;
                    f32x4 v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = *reinterpret_cast<const f32x4*>(V_s + voff[r] + (2 * t) * 16 * D);
                        v[4 + r] = *reinterpret_cast<const f32x4*>(V_s + voff[r] + j1 * 16 * D);
                    }
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        oacc[dt] = a16_mfma_bf(a16_pack8(v[0][dt], v[1][dt], v[2][dt], v[3][dt], v[4][dt], v[5][dt], v[6][dt], v[7][dt]), pbs[t], oacc[dt]);
                    __builtin_amdgcn_sched_barrier(0);      // (without it hipcc hoists the reads of several pairs and spills)
                }
            } else {
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
            }
            o_bh = bh;
            o_inv = inv;
            o_lse = lse_v;
        }
        if (cutw) {
            // wave q finishes register row r = q of the cut tile: the four partials in quarter order, the total sum, the store
            ++cut_done;
            a16_wait(cnt_s, 4u * cut_done);
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

template <int NKT, bool CAUSAL, bool BF>
int launch_fwd16c(const AttnArgs& a, hipStream_t st) {
    const size_t lds = a16_lds_bytes(a.N);
    KV_ALLOW_LDS(160 * 1024, (attn16_fwd_kernel<NKT, CAUSAL, BF>));
    const int nbh = a.B * a.H;
    const int gmax = kv_config().attn_grid > 0 ? kv_config().attn_grid : KV_N_CU;      // one work-group per CU (the three images fill its LDS)
    hipLaunchKernelGGL((attn16_fwd_kernel<NKT, CAUSAL, BF>), dim3((unsigned)(nbh < gmax ? nbh : gmax)), dim3(A16_THREADS), lds, st, a);
    KV_LAUNCH_CHECK("attn16_fwd_kernel");
    return 0;
}

template <int NKT>
int launch_fwd16(const AttnArgs& a, hipStream_t st) { return launch_fwd16c<NKT, false, false>(a, st); }      // (the causal instantiation of 13 tiles spills: causal launches keep the fourth form)

bool a16_shape_ok(const AttnArgs& a) {
    if (a.D != A16_D || a.N <= 64 || a16_lds_bytes(a.N) > 160 * 1024 || (a.N + 15) / 16 > A16_MAXT || !a.vec) return false;
    // 32-bit row offsets inside a head's operand (row * stride elements)
    for (long long sn : {a.qsn, a.ksn, a.vsn, a.osn})
        if (sn * (long long)(a.N + 1) >= (1LL << 31)) return false;
    return true;
}


// =============================================================================================
// backward: ONE kernel, all five products, no dS hand-off through HBM
// =============================================================================================
// dV = P^T dO, dK = dS^T Q, dQ = dS K with P = exp(S scale - lse), dS = P (dP - delta) scale, S = Q K^T, dP = dO V^T, delta = rowsum(dO O)
// (utils.py:229-295).  The fourth form ran the first four products in a key-stationary kernel, stored dS ([B H][224][224] fp32:
// 308 MB written and read back at ViT-B) and formed dQ in a second kernel: 1.92x the algorithmic traffic, 0.46 of the fp32 peak.
// Here, for 192 < N <= 204 (13 tiles of 16; ViT-B/16 and ViT-S/16: N = 197):
//   * a work-group (8 waves, two per SIMD, one per CU, persistent over heads) is KEY-stationary: waves 0-4 own two key tiles each
//     (tiles 2w, 2w + 1), waves 5-7 own tiles 10, 11, 12.  K and V rows of a wave's keys are its MFMA B operands, in registers
//     for the whole head; dK^T and dV^T accumulate in registers (key on the lane: float4 stores at the end of the head).
//   * Orientation S = Q.K^T with the QUERY in the accumulator register index: register r of lane group g is query 4 g + r, the
//     k index of k-step r of the products that contract over queries -- P and dS feed dV^T = dO^T.P and dK^T = Q^T.dS from
//     registers, exactly as P feeds O^T in the forward.
//   * The query side STREAMS: a step is one 16-query tile, the same for all waves; its Q, dO and O rows and lse values arrive by
//     LDS-DMA as a 12 KB slice, four steps ahead, in a ring of six (across head boundaries: nothing is fetched "at" a boundary).
//   * dQ contracts over KEYS, the lane index of dS: each dS tile crosses the LDS once ([16 q][17], 1 KB) and a "unit" -- 16 MFMAs,
//     dQ^T[d][q] += K^T[d][key].dS^T[key][q] for one (query tile, key tile) pair -- can run on ANY wave.  That is what balances
//     the SIMDs: thirteen key tiles are 4 + 3 + 3 + 3 on the four SIMDs (waves w and w + 4 share one), and the 13 units of a step
//     all go to waves 5, 6, 7 -- the ones whose SIMD has only three key tiles: 256 / 261 / 261 / 261 MFMAs per step.
//     Unit wave u = w - 5 serves key tiles u, u + 3, u + 6, u + 9 (wave 6 also tile 12) and keeps THOSE tiles' K rows in registers,
//     in the operand layout of the unit product.  Waves 6, 7 hand their partial sums to wave 5, which adds them in a fixed order
//     and stores dQ: deterministic, no float atomics.
//   * The unit waves run TWO steps behind the dS tiles they consume (ring of four), so they never wait for a producer, and they
//     carry the whole fill protocol (issue, confirm, delta): on the two-tile waves -- which bound the step -- that bookkeeping was
//     2-3 k cycles per step.  (First version, measured with phase stamps: units one step behind on a ring of three, fills on
//     waves 0-4, the K tiles of the units in a 51 KB LDS image that left no room for deeper rings: every wave waited 2-5 k cycles
//     per step for some other wave.)
//   * No barrier after the prologue: every hand-off is a monotonic LDS counter (slice landed / delta formed / slice consumed /
//     dS tiles written / read / partial written / read) that the waiter polls.
#ifndef B16_ABLATE
#define B16_ABLATE 0          // diagnostic builds only (tools/build_variant.sh abl -DB16_ABLATE=n): 1 no S / dP products, 2 no dV / dK products, 4 no unit products,
                              // 16 no slice fills after the prologue, 32 no waits on counters (timing only: the results are wrong)
#endif
#ifdef KANVIT_CLOCK_PROBE
#ifndef B16_CLK_WAVE
#define B16_CLK_WAVE 0
#endif
// diagnostic build only (tools/clock_probe16.py): phase stamps of one wave of the middle work-group, written to a buffer nothing reads
__device__ unsigned long long g_b16_clk[16];
#define B16_CLK_BEGIN() unsigned long long b16_t0 = 0, b16_r0 = 0, b16_tp = 0, b16_ph[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    const bool b16_stamp = (blockIdx.x == gridDim.x / 2) && wave == B16_CLK_WAVE && lane == 0;                      \
    if (b16_stamp) { b16_t0 = b16_tp = __builtin_amdgcn_s_memtime(); b16_r0 = __builtin_amdgcn_s_memrealtime(); }
#define B16_CLK_PHASE(i) if (b16_stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long b16_now = __builtin_amdgcn_s_memtime(); b16_ph[i] += b16_now - b16_tp; b16_tp = b16_now; __builtin_amdgcn_sched_barrier(0); }
#define B16_CLK_END() if (b16_stamp) { g_b16_clk[0] = __builtin_amdgcn_s_memtime() - b16_t0; g_b16_clk[1] = __builtin_amdgcn_s_memrealtime() - b16_r0; \
    for (int b16_i = 0; b16_i < 14; ++b16_i) g_b16_clk[2 + b16_i] = b16_ph[b16_i]; }
#else
#define B16_CLK_BEGIN()
#define B16_CLK_PHASE(i)
#define B16_CLK_END()
#endif
constexpr int B16_NW = 8;
constexpr int B16_THREADS = 64 * B16_NW;
constexpr int B16_NT = 13;                        // key tiles = query tiles = steps per head
constexpr int B16_SL = 3 * 1024 + 64 + 16;        // slice: Q, dO, O images [16][64], lse (64-float landing zone of a dword DMA), delta
constexpr int B16_NSL = 6;                        // slice ring: issued 4 steps ahead (of the unit waves' clock), landed 3 ahead
constexpr int B16_AHEAD = 4;
constexpr int B16_DST = 16 * 17;                  // one dS tile, [16 queries][17]
constexpr int B16_DSS = B16_NT * B16_DST;
constexpr int B16_NDS = 4;                        // dS ring: the units run B16_LAG steps behind
constexpr int B16_LAG = 2;
constexpr int B16_PART = 2 * 1024;                // partial dQ tiles of waves 6 and 7
constexpr int B16_NPT = 2;
constexpr int B16_NCNT = 32;
enum { BC_READY = 0, BC_DELTA = 6, BC_DONE = 12, BC_DSW = 18, BC_DSR = 22, BC_PW = 26, BC_PR = 28 };
constexpr int b16_lds_floats() { return B16_NDS * B16_DSS + B16_NPT * B16_PART + B16_NSL * B16_SL + B16_NCNT; }
inline size_t b16_lds_bytes(int) { return sizeof(float) * (size_t)b16_lds_floats(); }

__device__ __forceinline__ void b16_wait(const unsigned* c, unsigned target) {
    if constexpr (!(B16_ABLATE & 32)) a16_wait(c, target);
}
__device__ __forceinline__ void b16_signal(unsigned* c, int lane, unsigned n = 1u) { a16_signal(c, lane, n); }

// NTL = key tiles of this wave (2: waves 0-4; 1: waves 5-7, which also run the fills and the dQ units)
// BF: the five products on the bf16 matrix cores (KANVIT_FLAG_BF16_MFMA): S and dP are two v_mfma_f32_16x16x32_bf16 per tile, the three
// products that contract over 16 queries / 16 keys one v_mfma_f32_16x16x16_bf16 per d-tile; Q, K, V, dO, P and dS are rounded to bf16 as
// operands, everything accumulates in fp32, delta and the dS tiles in the LDS stay fp32.  Same slices, rings, counters and waves.
template <int NTL, bool BF>
__device__ __forceinline__ void attn16_bwd_body(const AttnArgs& a, float* __restrict__ smem, const int lane, const int wave) {
    constexpr int D = A16_D;
    constexpr bool UNITS = (NTL == 1);
    const int N = a.N, nbh = a.B * a.H;
    float* const dsr_s = smem;
    float* const part_s = dsr_s + B16_NDS * B16_DSS;
    float* const slices = part_s + B16_NPT * B16_PART;
    unsigned* const cnt = reinterpret_cast<unsigned*>(slices + B16_NSL * B16_SL);
    const float sc2 = a.scale * KV_LOG2E;
    const int nh = (nbh - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // heads of this work-group
    const int T = B16_NT * nh;                                        // steps
    const int kt0 = UNITS ? 10 + (wave - 5) : 2 * wave;               // first key tile of this wave
    const int u = wave - 5;                                           // unit index (UNITS only)

    auto head_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };
    auto slot6 = [&](int X) { return X % B16_NSL; };

    // ---- fills (waves 0-4) ----
    // slice X = step X = (head X / 13, query tile X % 13): 13 pieces -- Q rows (0-3), dO rows (4-7), O rows (8-11), lse (12);
    // wave w < 5 issues pieces w, w + 5, w + 10
    auto issue_slice = [&](int X) {
        if constexpr (!UNITS) {
            int l = lane;
            asm volatile("" : "+v"(l));
            const int k = X / B16_NT, sq = X - k * B16_NT;
            const int bh = head_of(k), bi = bh / a.H, hi = bh - bi * a.H;
            float* sl = slices + slot6(X) * B16_SL;
            const int lq = l >> 4;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int p = wave + 5 * c;
                if (p < 12) {
                    const int which = p >> 2, pp = p & 3;            // 0 Q, 1 dO, 2 O; piece pp = rows 4 pp .. 4 pp + 3 of the tile
                    int row = 16 * sq + 4 * pp + lq;
                    const int ls = (l & 15) ^ a16_f(4 * pp + lq);
                    row = row < N ? row : N - 1;
                    const float* src = which == 0 ? a.q + bi * a.qsb + hi * a.qsh + row * (int)a.qsn
                                                  : (which == 1 ? a.d_o : a.o) + bi * a.osb + hi * a.osh + row * (int)a.osn;
                    __builtin_amdgcn_global_load_lds((a16_glb_ptr)(src + 4 * ls), (a16_lds_ptr)(sl + which * 1024 + pp * 256), 16, 0, 0);
                } else if (p == 12) {
                    int row = 16 * sq + l;
                    row = row < N ? row : N - 1;
                    __builtin_amdgcn_global_load_lds((a16_glb_ptr)(a.lse_in + (long long)bh * N + row), (a16_lds_ptr)(sl + 3072), 4, 0, 0);
                }
            }
        }
    };
    // delta[q] = sum_d dO[q][d] O[q][d] of a landed slice: 4 lanes per row, the same physical slots of both images
    auto form_delta = [&](int X) {
        int l = lane;
        asm volatile("" : "+v"(l));
        float* sl = slices + slot6(X) * B16_SL;
        const int row = l >> 2, part = l & 3;
        float dl = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jr = (j + row) & 3;       // (rows are 256 bytes apart: without the rotation the four rows of a 16-lane read group share their banks)
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sl + 1024 + row * D + 16 * part + 4 * jr);
            const f32x4 o4 = *reinterpret_cast<const f32x4*>(sl + 2048 + row * D + 16 * part + 4 * jr);
            dl += d4[0] * o4[0] + d4[1] * o4[1] + d4[2] * o4[2] + d4[3] * o4[3];
        }
        dl += __shfl_xor(dl, 1);
        dl += __shfl_xor(dl, 2);
        if (part == 0) sl[3072 + 64 + row] = dl;
    };

    // ---- this wave's K / V rows (B operands of S and dP): row = key 16 kt + m, values d = 16 g .. 16 g + 15 ----
    float kf[NTL][16], vf[NTL][16];
    a16_bf8 kfb[NTL][2], vfb[NTL][2];          // BF: the same rows as MFMA operands, d = 16 g + 8 c + j in element j of [c]
    auto load_kv = [&](int k) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int bh = head_of(k), bi = bh / a.H, hi = bh - bi * a.H;
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            int key = 16 * (kt0 + t) + (l & 15);
            key = key < N ? key : N - 1;
            const float* kp = a.k + bi * a.ksb + hi * a.ksh + key * (int)a.ksn + 16 * (l >> 4);
            const float* vp = a.v + bi * a.vsb + hi * a.vsh + key * (int)a.vsn + 16 * (l >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 k4 = *reinterpret_cast<const f32x4*>(kp + 4 * i);
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(vp + 4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kf[t][4 * i + e] = k4[e];
                    vf[t][4 * i + e] = v4[e];
                }
            }
            if constexpr (BF) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    kfb[t][c] = a16_pack8(kf[t][8 * c], kf[t][8 * c + 1], kf[t][8 * c + 2], kf[t][8 * c + 3], kf[t][8 * c + 4], kf[t][8 * c + 5], kf[t][8 * c + 6], kf[t][8 * c + 7]);
                    vfb[t][c] = a16_pack8(vf[t][8 * c], vf[t][8 * c + 1], vf[t][8 * c + 2], vf[t][8 * c + 3], vf[t][8 * c + 4], vf[t][8 * c + 5], vf[t][8 * c + 6], vf[t][8 * c + 7]);
                }
            }
        }
    };
    // ---- unit wave: the K rows of the key tiles it serves, as A operands of the unit product: ku[it][4 ks + dt] =
    //      K[key 16 kt + 4 g + ks][d = 4 m' + dt] (one 16-byte load per k-step) ----
    auto unit_tile = [&](int it) -> int { return it < 4 ? u + 3 * it : (u == 1 ? B16_NT - 1 : B16_NT); };
    float ku[UNITS ? 5 : 1][16];
    a16_s4 kub[UNITS ? 5 : 1][4];              // BF: [it][dt] = the four k-steps ks of d-tile dt
    auto load_ku = [&](int k) {
        if constexpr (UNITS) {
            int l = lane;
            asm volatile("" : "+v"(l));
            const int bh = head_of(k), bi = bh / a.H, hi = bh - bi * a.H;
            const float* kb = a.k + bi * a.ksb + hi * a.ksh + 4 * (l & 15);
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int kt = unit_tile(it);
                if (kt < B16_NT) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        int key = 16 * kt + 4 * (l >> 4) + ks;
                        key = key < N ? key : N - 1;
                        const f32x4 k4 = *reinterpret_cast<const f32x4*>(kb + key * (int)a.ksn);
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) ku[it][4 * ks + dt] = k4[dt];
                    }
                    if constexpr (BF) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) kub[it][dt] = a16_pack4(ku[it][dt], ku[it][4 + dt], ku[it][8 + dt], ku[it][12 + dt]);
                    }
                }
            }
        }
    };

    f32x4 dk[NTL][4], dv[NTL][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dk[t][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                dv[t][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
    };
    auto store_dkv = [&](int k) {          // lane = key: registers (dt = 0..3, r') are d = 16 g + 4 r' + dt
        int l = lane;
        asm volatile("" : "+v"(l));
        const int bh = head_of(k), bi = bh / a.H, hi = bh - bi * a.H;
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const int key = 16 * (kt0 + t) + (l & 15);
            if (key < N) {
                float* kp = a.dk + bi * a.ksb + hi * a.ksh + key * (int)a.ksn + 16 * (l >> 4);
                float* vp = a.dv + bi * a.vsb + hi * a.vsh + key * (int)a.vsn + 16 * (l >> 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    *reinterpret_cast<f32x4*>(kp + 4 * r) = f32x4{dk[t][0][r], dk[t][1][r], dk[t][2][r], dk[t][3][r]};
                    *reinterpret_cast<f32x4*>(vp + 4 * r) = f32x4{dv[t][0][r], dv[t][1][r], dv[t][2][r], dv[t][3][r]};
                }
            }
        }
    };

    // ---------------- prologue (the only barriers): slices 0 .. AHEAD - 1 land, delta of 0 .. AHEAD - 2 ----------------
    if constexpr (!UNITS) {
        for (int X = 0; X < B16_AHEAD && X < T; ++X) issue_slice(X);
    } else {
        load_ku(0);
    }
    load_kv(0);
    zero_acc();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) form_delta(0);
    if (threadIdx.x == 0) {
        for (int X = 0; X < B16_AHEAD; ++X) cnt[BC_READY + X] = 5u;
        cnt[BC_DELTA + 0] = 1u;
    }
    __syncthreads();

    // A wave's step is two bursts of MFMAs (S / dP, then dV / dK) and everything else is latency: LDS round trips for counters,
    // fragments and dS tiles.  ALL fragment reads of a step are issued up front (the second products' operands do not depend on P);
    // a signal is never preceded by a wait (the LDS executes a wave's instructions in order, so a flag cannot overtake the data it
    // publishes).
    auto peek = [&](const unsigned* c) -> unsigned { return *(const volatile a16_lds_u32*)c; };

    // Wave 5 sums the three partial dQ tiles of a step and stores them one step after its own units of that step, at the top of its
    // next step: summed right behind the units it waited for waves 6 / 7.  Its own partial stays in registers meanwhile.
    f32x4 dqp[UNITS ? 4 : 1];
    auto reduce_dq = [&](int Yp) {        // step Yp = (head Yp / 13, query tile Yp % 13)
        if constexpr (UNITS) {
            const int pslot = Yp & 1, ky = Yp / B16_NT, sy = Yp - ky * B16_NT;
            const float* pp = part_s + pslot * B16_PART;
            b16_wait(cnt + BC_PW + pslot, 2u * (unsigned)((Yp >> 1) + 1));
            int l = lane;
            asm volatile("" : "+v"(l));
            const int bh = head_of(ky), bi = bh / a.H, hi = bh - bi * a.H;
            const int q = 16 * sy + (l & 15);
            float* qp = a.dq + bi * a.qsb + hi * a.qsh + q * (int)a.qsn + 16 * (l >> 4);
            f32x4 p6[4], p7[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p6[r] = *reinterpret_cast<const f32x4*>(pp + (r * 64 + l) * 4);
                p7[r] = *reinterpret_cast<const f32x4*>(pp + 1024 + (r * 64 + l) * 4);
            }
            b16_signal(cnt + BC_PR + pslot, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 t = dqp[r];
                t += p6[r];
                t += p7[r];
                if (q < N) *reinterpret_cast<f32x4*>(qp + 4 * r) = t;
            }
        }
    };

    // ---------------- steps ----------------
    int k = 0, s = 0;                     // head / query tile of step G
    B16_CLK_BEGIN()
    for (int G = 0; G < T + B16_LAG; ++G) {
        B16_CLK_PHASE(7)
        if constexpr (UNITS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // ku / kf / vf of a new head, the dQ stores
            B16_CLK_PHASE(11)
            // ---- dQ: wave 5 sums and stores step G - LAG - 1; then the units of step G - LAG ----
            if (u == 0 && G >= B16_LAG + 1) reduce_dq(G - B16_LAG - 1);
            if (G >= B16_LAG) {
                const int Y = G - B16_LAG, dslot = Y % B16_NDS, pslot = Y & 1;
                const int ky = Y / B16_NT, sy = Y - ky * B16_NT;
                b16_wait(cnt + BC_DSW + dslot, (unsigned)B16_NT * (unsigned)(Y / B16_NDS + 1));      // (written two steps ago)
                B16_CLK_PHASE(8)
                int l = lane;
                asm volatile("" : "+v"(l));
                const int mm = l & 15, gg = l >> 4;
                f32x4 dq[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                const float* dsb = dsr_s + dslot * B16_DSS + mm * 17 + 4 * gg;
                float b[5][4];
#pragma unroll
                for (int it = 0; it < 5; ++it)
                    if (unit_tile(it) < B16_NT) {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) b[it][ks] = dsb[unit_tile(it) * B16_DST + ks];      // dS[q m''][key 4 g + ks]
                    }
#pragma unroll
                for (int it = 0; it < 5; ++it)
                    if (unit_tile(it) < B16_NT) {
                        if constexpr (BF) {
                            const a16_s4 bb = a16_pack4(b[it][0], b[it][1], b[it][2], b[it][3]);
#pragma unroll
                            for (int dt = 0; dt < 4; ++dt) dq[dt] = a16_mfma_bf4(kub[it][dt], bb, dq[dt]);
                        } else {
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                                for (int dt = 0; dt < 4; ++dt) {
                                    if constexpr (!(B16_ABLATE & 4)) dq[dt] = a16_mfma(ku[it][4 * ks + dt], b[it][ks], dq[dt]);
                                    else dq[dt][0] += ku[it][4 * ks + dt] * b[it][ks];
                                }
                        }
                    }
                B16_CLK_PHASE(9)
                b16_signal(cnt + BC_DSR + dslot, lane);                          // (issued behind the reads: executes behind them)
                if (sy == B16_NT - 1 && ky + 1 < nh) load_ku(ky + 1);           // the units of a head are done: the next head's K rows
                B16_CLK_PHASE(13)
                if (u != 0) {
                    float* pp = part_s + pslot * B16_PART;
                    b16_wait(cnt + BC_PR + pslot, (unsigned)(Y >> 1));           // wave 5 is through the partials of step Y - 2
                    B16_CLK_PHASE(6)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<f32x4*>(pp + (u - 1) * 1024 + (r * 64 + l) * 4) = f32x4{dq[0][r], dq[1][r], dq[2][r], dq[3][r]};
                    b16_signal(cnt + BC_PW + pslot, lane);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dqp[r] = f32x4{dq[0][r], dq[1][r], dq[2][r], dq[3][r]};      // summed and stored a step from now (reduce_dq)
                }
            }
        }
        if (G >= T) continue;
        // ---- delta of slice G + 1 (signalled at the end of step G - 1), by one of the fill-issuing waves: a whole step ahead of its use ----
        if constexpr (!UNITS) {
            if (G + 1 < T && wave == (G + 1) % 5) {
                b16_wait(cnt + BC_READY + slot6(G + 1), 5u * (unsigned)((G + 1) / B16_NSL + 1));
                form_delta(G + 1);
                b16_signal(cnt + BC_DELTA + slot6(G + 1), lane);
            }
        }
        B16_CLK_PHASE(0)
        // ---- the key-stationary step: query tile s of head k against this wave's key tiles ----
        {
            const int slot = slot6(G), dslot = G % B16_NDS;
            const float* sl = slices + slot * B16_SL;
            b16_wait(cnt + BC_READY + slot, 5u * (unsigned)(G / B16_NSL + 1));      // (signalled steps ago)
            B16_CLK_PHASE(1)
            int l = lane;
            asm volatile("" : "+v"(l));
            const int mm = l & 15, gg = l >> 4;
            int ko[4], vo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ko[i] = mm * D + 4 * ((4 * gg + i) ^ a16_f(mm));
#pragma unroll
            for (int r = 0; r < 4; ++r) vo[r] = (4 * gg + r) * D + 4 * (mm ^ a16_f(4 * gg + r));
            // every fragment of the step, now: rows-as-A of Q and dO (first products), the row constants, cols-as-A of dO and Q (second
            // products: they do not depend on P, so their latency hides under the first burst)
            f32x4 fq[4], fd[4], fo[4], fq2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fq[i] = *reinterpret_cast<const f32x4*>(sl + ko[i]);
                fd[i] = *reinterpret_cast<const f32x4*>(sl + 1024 + ko[i]);
            }
            const f32x4 ls4 = *reinterpret_cast<const f32x4*>(sl + 3072 + 4 * gg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                fo[r] = *reinterpret_cast<const f32x4*>(sl + 1024 + vo[r]);
                fq2[r] = *reinterpret_cast<const f32x4*>(sl + vo[r]);
            }
            const unsigned dsr_seen = peek(cnt + BC_DSR + dslot);     // (for the dS write at the end of the step)
            // S[q][key] and dP[q][key]: A = Q / dO rows of the slice (lane = query row), B = this lane's K / V row
            f32x4 sS[NTL], sP[NTL];
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                sS[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                sP[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            if constexpr (BF) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const a16_bf8 aq = a16_pack8(fq[2 * c][0], fq[2 * c][1], fq[2 * c][2], fq[2 * c][3], fq[2 * c + 1][0], fq[2 * c + 1][1], fq[2 * c + 1][2], fq[2 * c + 1][3]);
                    const a16_bf8 ad = a16_pack8(fd[2 * c][0], fd[2 * c][1], fd[2 * c][2], fd[2 * c][3], fd[2 * c + 1][0], fd[2 * c + 1][1], fd[2 * c + 1][2], fd[2 * c + 1][3]);
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {
                        sS[t] = a16_mfma_bf(aq, kfb[t][c], sS[t]);
                        sP[t] = a16_mfma_bf(ad, vfb[t][c], sP[t]);
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {
                        if constexpr (!(B16_ABLATE & 1)) {
                            sS[t] = a16_mfma(fq[i][e], kf[t][4 * i + e], sS[t]);
                            sP[t] = a16_mfma(fd[i][e], vf[t][4 * i + e], sP[t]);
                        } else if (i == 0 && e == 0) {
                            sS[t][0] += fq[0][0] * kf[t][0] + fq[1][1] + fq[2][2] + fq[3][3];
                            sP[t][0] += fd[0][0] * vf[t][0] + fd[1][1] + fd[2][2] + fd[3][3];
                        }
                    }
            B16_CLK_PHASE(2)
            const bool last_s = (s == B16_NT - 1);
            if (last_s && k + 1 < nh) load_kv(k + 1);                 // the K / V registers are dead until the next head: its rows travel under the second products
            // P and dS on the accumulator registers: register r is query 16 s + 4 g + r
            b16_wait(cnt + BC_DELTA + slot, (unsigned)(G / B16_NSL + 1));
            const f32x4 dl4 = *reinterpret_cast<const f32x4*>(sl + 3072 + 64 + 4 * gg);
            float pr[NTL][4], ds[NTL][4];
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(sS[t][r] * sc2 - ls4[r] * KV_LOG2E);
                    const bool dead = (last_s && 16 * s + 4 * gg + r >= N) || (kt0 + t == B16_NT - 1 && 16 * (kt0 + t) + mm >= N);
                    p = dead ? 0.0f : p;
                    pr[t][r] = p;
                    ds[t][r] = p * (sP[t][r] - dl4[r]) * a.scale;
                }
            B16_CLK_PHASE(3)
            // dV^T[d][key] += dO^T[d][q] P[q][key], dK^T[d][key] += Q^T[d][q] dS[q][key]: k-step r is queries 4 g + r; one 16-byte read
            // (row 4 g + r, slot m') feeds the four d-tiles of every key tile of this wave
            if constexpr (BF) {
                a16_s4 pb[NTL], db[NTL];
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    pb[t] = a16_pack4(pr[t][0], pr[t][1], pr[t][2], pr[t][3]);
                    db[t] = a16_pack4(ds[t][0], ds[t][1], ds[t][2], ds[t][3]);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const a16_s4 ao = a16_pack4(fo[0][dt], fo[1][dt], fo[2][dt], fo[3][dt]);
                    const a16_s4 aq2 = a16_pack4(fq2[0][dt], fq2[1][dt], fq2[2][dt], fq2[3][dt]);
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {
                        dv[t][dt] = a16_mfma_bf4(ao, pb[t], dv[t][dt]);
                        dk[t][dt] = a16_mfma_bf4(aq2, db[t], dk[t][dt]);
                    }
                }
            } else
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {
                        if constexpr (!(B16_ABLATE & 2)) {
                            dv[t][dt] = a16_mfma(fo[r][dt], pr[t][r], dv[t][dt]);
                            dk[t][dt] = a16_mfma(fq2[r][dt], ds[t][r], dk[t][dt]);
                        } else {
                            dv[t][dt][0] += fo[r][dt] * pr[t][r];
                            dk[t][dt][0] += fq2[r][dt] * ds[t][r];
                        }
                    }
            B16_CLK_PHASE(4)
            b16_signal(cnt + BC_DONE + slot, lane);                   // (behind every read of the slice)
            // the dS tiles of this step, [q][17]: the unit waves read them two steps from now
            if (G >= B16_NDS && dsr_seen < 3u * (unsigned)(G / B16_NDS)) b16_wait(cnt + BC_DSR + dslot, 3u * (unsigned)(G / B16_NDS));
            B16_CLK_PHASE(10)
            float* dsw = dsr_s + dslot * B16_DSS + (4 * gg) * 17 + mm;
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dsw[(kt0 + t) * B16_DST + r * 17] = ds[t][r];
            b16_signal(cnt + BC_DSW + dslot, lane, (unsigned)NTL);
            if (last_s) {
                store_dkv(k);
                zero_acc();
            }
        }
        B16_CLK_PHASE(5)
        // ---- the fills, at the END of the step (waves 0-4): slice G + AHEAD - 1 (issued a step ago) has landed; slice G + AHEAD goes into
        //      the slot of slice G + AHEAD - 6, which every wave -- the unit waves, two steps behind, included -- has left ----
        if constexpr (!UNITS) {
            const int XL = G + B16_AHEAD - 1, XI = G + B16_AHEAD;
            if (XL >= B16_AHEAD && XL < T) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                b16_signal(cnt + BC_READY + slot6(XL), lane);
            }
            B16_CLK_PHASE(11)
            if (XI < T && !(B16_ABLATE & 16)) {
                if (XI >= B16_NSL) b16_wait(cnt + BC_DONE + slot6(XI), (unsigned)B16_NW * (unsigned)(XI / B16_NSL));      // every wave is through step XI - 6
                B16_CLK_PHASE(12)
                issue_slice(XI);
            }
        }
        B16_CLK_PHASE(6)
        if (++s == B16_NT) { s = 0; ++k; }
    }
    if constexpr (UNITS) {
        if (u == 0) reduce_dq(T - 1);
    }
    B16_CLK_END()
}

template <bool BF>
__global__ __launch_bounds__(B16_THREADS) void attn16_bwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid * 4; e < b16_lds_floats(); e += B16_THREADS * 4) *reinterpret_cast<f32x4*>(smem + e) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
    if (wave < 5) attn16_bwd_body<2, BF>(a, smem, lane, wave);
    else attn16_bwd_body<1, BF>(a, smem, lane, wave);
}

bool b16_shape_ok(const AttnArgs& a) {
    return a16_shape_ok(a) && (a.N + 15) / 16 == B16_NT && b16_lds_bytes(a.N) <= 160 * 1024 && !a.causal;
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

#ifdef KANVIT_CLOCK_PROBE
extern "C" __attribute__((visibility("default"))) int kanvit_debug_clock16(unsigned long long* out) {      // diagnostic build only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_b16_clk), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -5;
}
#endif

// the same forward with its two products on the bf16 matrix cores (KANVIT_FLAG_BF16_MFMA), 13 tiles only (N = 193 .. 204)
int kv_attn16_fwd_bf16(const AttnArgs& a, hipStream_t st) {
    if (!a16_shape_ok(a) || a.causal || kv_config().attn_v4 || (a.N + 15) / 16 != A16_MAXT) return 1;
    if (((uintptr_t)a.out | (uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) % 16) return 1;
    return launch_fwd16c<A16_MAXT, false, true>(a, st);
}

bool kv_attn16_bwd_ok(const kanvit_attn_desc* d) {
    if (!d || kv_config().attn_v4 || kv_config().attn_v1 || kv_config().attn_v2 || kv_config().attn_v3 || kv_config().attn_no_ds) return false;
    if ((d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16) return false;
    return d->D == A16_D && !d->causal && (d->N + 15) / 16 == B16_NT && b16_lds_bytes(d->N) <= 160 * 1024;
}

template <bool BF>
static int launch_bwd16(const AttnArgs& a, hipStream_t st) {
    if (!b16_shape_ok(a) || kv_config().attn_v4) return 1;
    if (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.o | (uintptr_t)a.d_o | (uintptr_t)a.dq | (uintptr_t)a.dk | (uintptr_t)a.dv) % 16) return 1;
    if ((uintptr_t)a.lse_in % 4) return 1;
    const size_t lds = b16_lds_bytes(a.N);
    KV_ALLOW_LDS(160 * 1024, attn16_bwd_kernel<BF>);
    const int nbh = a.B * a.H;
    const int gmax = kv_config().attn_grid > 0 ? kv_config().attn_grid : KV_N_CU;
    hipLaunchKernelGGL(attn16_bwd_kernel<BF>, dim3((unsigned)(nbh < gmax ? nbh : gmax)), dim3(B16_THREADS), lds, st, a);
    KV_LAUNCH_CHECK("attn16_bwd_kernel");
    return 0;
}
int kv_attn16_bwd(const AttnArgs& a, hipStream_t st) { return launch_bwd16<false>(a, st); }
int kv_attn16_bwd_bf16(const AttnArgs& a, hipStream_t st) { return launch_bwd16<true>(a, st); }
