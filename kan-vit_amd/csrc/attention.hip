// Multi-head attention core for gfx950, fp32-exact MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces attention.py:199-200 (softmax(q k^T / sqrt(dh)) v inside MSA's python double loop) and
// utils.py:137-295 (FlashAttentionFunction forward / recompute backward).  Sequence lengths on
// this path are 17 / 50 / 197, so a whole head fits one workgroup:
// no online rescaling, no cross-workgroup reductions, bitwise run-to-run reproducible.
//
// One workgroup = one (batch, head), 4 waves; a wave owns 32-query (or 32-key) tiles.
// Layout trick used throughout (cdna guide section 3, "accumulator tile as the next MFMA's
// operand", fp32 form): the first product is oriented so the reduction index of the SECOND
// product lands in the accumulator's register index.  For the 32x32x2 shape the k-step r of the
// second product then takes B[k = lane>>5][col = lane&31] = accumulator register r directly,
// with the other operand fetched from LDS row kv_acc_row(r, lane>>5) -- no LDS round trip and
// no cross-lane traffic for P, dS.
//
// LDS tiles are [rows][KS] with KS = 32*DT + 1 (odd): conflict free both when lanes walk rows
// (A operand of q.k^T) and when lanes walk columns (A operand of the second products).  Columns
// D..32*DT-1 and rows N..NP-1 are zero.
#include "attention_common.h"

#include <type_traits>

#include <stdlib.h>

namespace {

constexpr int ATHR = 256;
constexpr float LOG2E = 1.4426950408889634f;

// ---- bf16 matrix-core mode (KANVIT_FLAG_BF16_MFMA): operands gathered from the fp32 LDS tiles / accumulators, rounded to
// bf16, contracted by v_mfma_f32_32x32x16_bf16 (16 k per instruction instead of 2).  k-slot e of lane half h is
//   * the (8h+e)-th of 16 consecutive head-dim positions when the operand comes from LDS rows, or
//   * accumulator register 8s+e, i.e. tile row 16s + 8(e>>2) + 4h + (e&3), when a score tile feeds the second product
//     (cdna guide section 3, "accumulator tile as the next MFMA's operand"); the LDS operand then reads the SAME rows.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned kv_pk(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16x8_t kv_pack8(const float (&f)[8]) {
    const u32x4 u = {kv_pk(f[0], f[1]), kv_pk(f[2], f[3]), kv_pk(f[4], f[5]), kv_pk(f[6], f[7])};
    return __builtin_bit_cast(bf16x8_t, u);
}
// 8 consecutive floats of one LDS row
__device__ __forceinline__ bf16x8_t kv_row8(const float* __restrict__ p) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = p[e];
    return kv_pack8(f);
}
// one column of the 8 tile rows that accumulator registers 8s..8s+7 of lane half hf stand for; p -> (row 16s + 4hf, col)
__device__ __forceinline__ bf16x8_t kv_col8(const float* __restrict__ p, int stride) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = p[e * stride];
        f[4 + e] = p[(8 + e) * stride];
    }
    return kv_pack8(f);
}
__device__ __forceinline__ bf16x8_t kv_acc8(const f32x16& a, int s) {
    const u32x4 u = {kv_pk(a[8 * s], a[8 * s + 1]), kv_pk(a[8 * s + 2], a[8 * s + 3]), kv_pk(a[8 * s + 4], a[8 * s + 5]),
                     kv_pk(a[8 * s + 6], a[8 * s + 7])};
    return __builtin_bit_cast(bf16x8_t, u);
}



// Fill dst[rows][KS] from src rows row0.. (row stride stride_n), zero-padding rows >= N and columns
// >= D.  VEC path (D, strides and base 16-byte aligned): float4 global loads, four passes issued
// before the first LDS store so their latencies overlap; scalar path for odd head sizes.
template <int DT>
__device__ __forceinline__ void load_tile(float* __restrict__ dst, const float* __restrict__ src, long long stride_n,
                                          int row0, int rows, int N, int D, int tid, int nthr, bool vec) {
    constexpr int W = 32 * DT;
    constexpr int KS = W + 1;
    if (vec) {
        constexpr int W4 = W / 4;                 // float4 per padded row (8 or 16)
        const int c4 = (tid & (W4 - 1)) * 4, r0 = tid / W4, rp = nthr / W4;
        const bool cok = c4 < D;
        const float* sp = src + c4;
        for (int rb = 0; rb < rows; rb += 4 * rp) {
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rb + q * rp + r0;
                const int n = row0 + r;
                f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
                if (cok && r < rows && n < N) t = *reinterpret_cast<const f32x4*>(sp + (long long)n * stride_n);
                v[q] = t;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rb + q * rp + r0;
                if (r < rows) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[r * KS + c4 + e] = v[q][e];
                }
            }
        }
    } else {
        for (int idx = tid; idx < rows * W; idx += nthr) {
            const int r = idx / W, c = idx - r * W;
            const int n = row0 + r;
            dst[r * KS + c] = (n < N && c < D) ? src[(long long)n * stride_n + c] : 0.0f;
        }
    }
}

// One wave writes its [32][KS] staging tile to 32 consecutive global rows (float4 when aligned).
template <int DT>
__device__ __forceinline__ void store_tile(float* __restrict__ dstg, long long stride_n, int row0, int N, int D,
                                           const float* __restrict__ T, int lane, bool vec) {
    constexpr int W = 32 * DT;
    constexpr int KS = W + 1;
    if (vec) {
        constexpr int W4 = W / 4;
        const int c4 = (lane & (W4 - 1)) * 4, r0 = lane / W4, rp = 64 / W4;
        if (c4 < D) {
#pragma unroll 4
            for (int r = r0; r < 32; r += rp) {
                const int n = row0 + r;
                if (n < N) {
                    f32x4 v = {T[r * KS + c4], T[r * KS + c4 + 1], T[r * KS + c4 + 2], T[r * KS + c4 + 3]};
                    *reinterpret_cast<f32x4*>(dstg + (long long)n * stride_n + c4) = v;
                }
            }
        }
    } else {
        for (int idx = lane; idx < 32 * D; idx += 64) {
            const int r = idx / D, c = idx - r * D;
            const int n = row0 + r;
            if (n < N) dstg[(long long)n * stride_n + c] = T[r * KS + c];
        }
    }
}

// =============================================================================================
// forward: grid B*H.  NKT = compile-time bound on the number of 32-key tiles (registers).
// =============================================================================================
template <int DT, int NKT, bool BF>
__global__ __launch_bounds__(ATHR) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 32 * DT + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, D = a.D, nkt = a.nkt, NP = nkt * 32;
    float* K_s = smem;                   // [NP][KS]
    float* V_s = K_s + NP * KS;          // [NP][KS]
    float* Q_w = V_s + NP * KS + wave * 32 * KS;   // per wave [32][KS]; doubles as the O staging tile

    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    float* ob = a.out + bi * a.osb + hi * a.osh;

    load_tile<DT>(K_s, kb, a.ksn, 0, NP, N, D, tid, ATHR, a.vec);
    load_tile<DT>(V_s, vb, a.vsn, 0, NP, N, D, tid, ATHR, a.vec);

    const float sc2 = a.scale * LOG2E;
    const int niter = (nkt + 3) / 4;
    for (int it = 0; it < niter; ++it) {
        const int qt = it * 4 + wave;        // tiles past nkt run on zero rows and store nothing
        load_tile<DT>(Q_w, qb, a.qsn, qt * 32, 32, (qt < nkt) ? N : 0, D, lane, 64, a.vec);
        __syncthreads();                     // also covers the K/V fill on the first trip

        float qf[BF ? 1 : 16 * DT];
        bf16x8_t qb[BF ? 2 * DT : 1];
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < 2 * DT; ++ks) qb[ks] = kv_row8(Q_w + l31 * KS + 16 * ks + 8 * hf);
        } else {
#pragma unroll
            for (int s = 0; s < 16 * DT; ++s) qf[s] = Q_w[l31 * KS + 2 * s + hf];
        }

        // S^T tiles: rows = keys, cols = queries
        f32x16 sacc[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[j][r] = 0.0f;
            if (j < nkt) {
                if constexpr (BF) {
                    const float* kp = K_s + (j * 32 + l31) * KS + 8 * hf;
#pragma unroll
                    for (int ks = 0; ks < 2 * DT; ++ks)
                        sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv_row8(kp + 16 * ks), qb[ks], sacc[j], 0, 0, 0);
                } else {
                    const float* kp = K_s + (j * 32 + l31) * KS + hf;
#pragma unroll
                    for (int s = 0; s < 16 * DT; ++s)
                        sacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s], qf[s], sacc[j], 0, 0, 0);
                }
            }
        }
        // softmax over keys: registers of this lane + the partner lane in the other half
        const int qrow = qt * 32 + l31;
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            if (j < nkt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = j * 32 + kv_acc_row(r, hf);
                    const bool dead = (key >= N) || (a.causal && key > qrow);
                    const float sv = dead ? -INFINITY : sacc[j][r];
                    sacc[j][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mxs = (mx == -INFINITY) ? 0.0f : mx * sc2;
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            if (j < nkt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = exp2f(sacc[j][r] * sc2 - mxs);
                    sacc[j][r] = p;
                    sum += p;
                }
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;

        // O^T[d][query] = sum_key V[key][d] * P[key][query]
        f32x16 oacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.0f;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            if (j < nkt) {
                if constexpr (BF) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[j][r] *= inv;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8_t pb = kv_acc8(sacc[j], s2);
                        const float* vp = V_s + (j * 32 + 16 * s2 + 4 * hf) * KS + l31;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
                            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv_col8(vp + dt * 32, KS), pb, oacc[dt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = sacc[j][r] * inv;
                        const float* vp = V_s + (j * 32 + kv_acc_row(r, hf)) * KS + l31;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
                            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[dt * 32], pv, oacc[dt], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                     // everyone is done reading Q_w as Q
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Q_w[l31 * KS + dt * 32 + kv_acc_row(r, hf)] = oacc[dt][r];
        __syncthreads();
        if (qt < nkt) {
            store_tile<DT>(ob, a.osn, qt * 32, N, D, Q_w, lane, a.vec && ((uintptr_t)ob % 16 == 0));
            if (hf == 0 && qrow < N && a.lse) a.lse[(long long)bh * N + qrow] = mx * a.scale + logf(sum);
        }
        __syncthreads();
    }
}

// =============================================================================================
// forward, second form (D == 32*DT, 16-byte aligned operands).  Same mathematics and tile orientation as attn_fwd_kernel;
// what changes is where the operands live:
//   * Q never touches the LDS: the B operand of S^T = K.Q^T is (d, query = lane), i.e. a piece of the lane's OWN query
//     row, loaded as float4 from global.  The contraction order over d is free, so in fp32 lane half h takes
//     d = 16*DT*h + s at k-step s (one contiguous half row) instead of d = 2s + h.
//   * O is stored straight from the accumulators: registers 4q..4q+3 of O^T are 4 consecutive d of the lane's query row
//     (one float4 per (dt, q)); no staging tile, so the only barrier of the kernel is the one after the K/V fill.
//   * 1/sum is applied to the 32*DT output values, not to the 16*nkt probabilities.
//   * bf16 mode keeps K and V in the LDS AS bf16, in the order the matrix core wants them: K rows [key][D + 8] (the
//     A fragment of S^T is one ds_read_b128), V transposed [d][NP + 8] with the 32 keys of a tile stored in accumulator
//     order (slot 16*s2 + 8*h + e <-> key 16*s2 + 8*(e>>2) + 4*h + (e&3)), so the A fragment of O^T = V^T.P is one
//     ds_read_b128 too.  62 KiB for N = 197: two work-groups per CU, one loading while the other computes.
// fp32: 8 waves, one query tile each (2 waves per SIMD hide each other's LDS latency); bf16: 4 waves, 2 trips.
// =============================================================================================
__device__ __forceinline__ int kv_key_slot(int ko) {      // position of key offset ko (0..31) inside its tile, bf16 V^T image
    const int w = ko & 15;
    return (ko & 16) + 8 * ((w >> 2) & 1) + 4 * (w >> 3) + (w & 3);
}

template <int DT, int NKT, bool BF>
__global__ __launch_bounds__(BF ? 256 : 512, 2) void attn_fwd2_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 32 * DT;
    constexpr int NW = BF ? 4 : 8;
    constexpr int NTHR = NW * 64;
    constexpr int KS = D + 1;          // fp32 row stride (floats)
    constexpr int KB = D + 8;          // bf16 K row stride (elements)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, nkt = a.nkt, NP = nkt * 32;
    const int VB = NP + 8;             // bf16 V^T row stride (elements)
    float* K_s = smem;                 // fp32: [NP][KS]
    float* V_s = K_s + NP * KS;        // fp32: [NP][KS]
    unsigned short* Kb = reinterpret_cast<unsigned short*>(smem);   // bf16: [NP][KB]
    unsigned short* Vt = Kb + NP * KB;                              // bf16: [D][VB]

    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    float* ob = a.out + bi * a.osb + hi * a.osh;

    if constexpr (!BF) {
        load_tile<DT>(K_s, kb, a.ksn, 0, NP, N, D, tid, NTHR, true);
        load_tile<DT>(V_s, vb, a.vsn, 0, NP, N, D, tid, NTHR, true);
    } else {
        constexpr int C4 = D / 4;                       // float4 per row
        const int c4 = (tid % C4) * 4, r0 = tid / C4;
        constexpr int RP = NTHR / C4;                   // rows per pass
        // ALL of the head's K and V rows are requested before anything is converted: 2 * NKT * 32 / RP float4 per thread in flight
        // (28 for N = 197) -- one memory round trip for the fill instead of one per 64 rows (seven of them, each exposed: the
        // work-group has nothing else to do until the barrier below)
        constexpr int KQ = (NKT * 32 + RP - 1) / RP;    // K rows per thread
        constexpr int VQ = (NKT * 16 + RP - 1) / RP;    // V row PAIRS per thread
        f32x4 kx[KQ], v0[VQ], v1[VQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int key = q * RP + r0;
            f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
            if (key < N) t = *reinterpret_cast<const f32x4*>(kb + (long long)key * a.ksn + c4);
            kx[q] = t;
        }
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
            const int key = 2 * (q * RP + r0);
            f32x4 t0 = {0.0f, 0.0f, 0.0f, 0.0f}, t1 = t0;
            if (key < N) t0 = *reinterpret_cast<const f32x4*>(vb + (long long)key * a.vsn + c4);
            if (key + 1 < N) t1 = *reinterpret_cast<const f32x4*>(vb + (long long)(key + 1) * a.vsn + c4);
            v0[q] = t0;
            v1[q] = t1;
        }
        // K: [key][d] rows, 4 floats -> 4 bf16 (8 bytes)
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int key = q * RP + r0;
            if (key < NP) {
                unsigned* dst = reinterpret_cast<unsigned*>(Kb + key * KB + c4);
                dst[0] = kv_pk(kx[q][0], kx[q][1]);
                dst[1] = kv_pk(kx[q][2], kx[q][3]);
            }
        }
        // V: key pairs (2m, 2m+1) are adjacent slots of the transposed image
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
            const int key = 2 * (q * RP + r0);
            if (key < NP) {
                const int pos = (key & ~31) + kv_key_slot(key & 31);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    *reinterpret_cast<unsigned*>(Vt + (c4 + e) * VB + pos) = kv_pk(v0[q][e], v1[q][e]);
            }
        }
    }
    __syncthreads();

    const float sc2 = a.scale * LOG2E;
    const int niter = (nkt + NW - 1) / NW;
    for (int it = 0; it < niter; ++it) {
        const int qt = it * NW + wave;
        if (qt >= nkt) break;                  // no barriers below: a wave without a tile is done
        const int qrow = qt * 32 + l31;
        const bool qok = qrow < N;
        const float* qp = qb + (long long)(qok ? qrow : 0) * a.qsn;

        float qf[BF ? 1 : 16 * DT];
        bf16x8_t qfb[BF ? 2 * DT : 1];
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < 2 * DT; ++ks) {
                f32x4 u0 = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 8 * hf);
                f32x4 u1 = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 8 * hf + 4);
                if (!qok) { u0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; u1 = u0; }
                const u32x4 u = {kv_pk(u0[0], u0[1]), kv_pk(u0[2], u0[3]), kv_pk(u1[0], u1[1]), kv_pk(u1[2], u1[3])};
                qfb[ks] = __builtin_bit_cast(bf16x8_t, u);
            }
        } else {
#pragma unroll
            for (int s4 = 0; s4 < 4 * DT; ++s4) {
                f32x4 u0 = *reinterpret_cast<const f32x4*>(qp + 16 * DT * hf + 4 * s4);
                if (!qok) u0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[4 * s4 + e] = u0[e];
            }
        }

        f32x16 sacc[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[j][r] = 0.0f;
            if (j < nkt) {
                if constexpr (BF) {
                    const unsigned short* kp = Kb + (j * 32 + l31) * KB + 8 * hf;
#pragma unroll
                    for (int ks = 0; ks < 2 * DT; ++ks)
                        sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(kp + 16 * ks), qfb[ks],
                                                                          sacc[j], 0, 0, 0);
                } else {
                    const float* kp = K_s + (j * 32 + l31) * KS + 16 * DT * hf;
#pragma unroll
                    for (int s = 0; s < 16 * DT; ++s)
                        sacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[s], qf[s], sacc[j], 0, 0, 0);
                }
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            if (j < nkt) {
                if (j == nkt - 1 || a.causal) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = j * 32 + kv_acc_row(r, hf);
                        const bool dead = (key >= N) || (a.causal && key > qrow);
                        sacc[j][r] = dead ? -INFINITY : sacc[j][r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[j][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mxs = (mx == -INFINITY) ? 0.0f : mx * sc2;
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            if (j < nkt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = exp2f(sacc[j][r] * sc2 - mxs);
                    sacc[j][r] = p;
                    sum += p;
                }
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;

        f32x16 oacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.0f;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            if (j < nkt) {
                if constexpr (BF) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8_t pb = kv_acc8(sacc[j], s2);
                        const unsigned short* vp = Vt + l31 * VB + j * 32 + 16 * s2 + 8 * hf;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
                            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(vp + dt * 32 * VB), pb,
                                                                               oacc[dt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float* vp = V_s + (j * 32 + kv_acc_row(r, hf)) * KS + l31;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
                            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[dt * 32], sacc[j][r], oacc[dt], 0, 0, 0);
                    }
                }
            }
        }
        if (qok) {
            float* op = ob + (long long)qrow * a.osn + 4 * hf;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {oacc[dt][4 * q] * inv, oacc[dt][4 * q + 1] * inv, oacc[dt][4 * q + 2] * inv, oacc[dt][4 * q + 3] * inv};
                    *reinterpret_cast<f32x4*>(op + dt * 32 + 8 * q) = v;
                }
            if (hf == 0 && a.lse) a.lse[(long long)bh * N + qrow] = mx * a.scale + logf(sum);
        }
    }
}

// =============================================================================================
// forward, third form (exact fp32, D == 32*DT, 16-byte aligned operands): attn_fwd2_kernel's mathematics and tile
// orientation with
//   * K and V rows 16-byte aligned in the LDS (row stride D + 4 floats: conflict-free ds_read_b128 with lane = row and
//     ds_read2_b32 with lane = column; the fill is one ds_write_b128 per float4 instead of four ds_write_b32), S^T = K.Q^T
//     reading FOUR k-steps of K per LDS instruction;
//   * the ragged last key tile contributing only its valid keys to O^T (the other probabilities are exactly zero: for
//     N = 197, 4 of 16 k-steps), masks applied to that tile only (or to every tile under the causal mask), v_exp_f32
//     directly (arguments are <= 0) instead of the range-checked exp2f;
//   * work-groups that walk the (batch, head) pairs with stride gridDim.x (one launch-resident round).
// What the in-kernel phase stamps of the diagnostic build (tools/clock_probe.py, -DKANVIT_CLOCK_PROBE) show for one head at
// ViT-B size (~90 k cycles on a CU, 2.3 GHz held): the two MFMA phases are half of it, a fifth is WAITING for the K / V rows
// (nothing else is resident on the CU: a head's fp32 K and V fill the LDS), 7 % waiting for each tile's Q rows, 10 %
// barriers and LDS fill, 6 % the exponentials.  Hiding the fill behind MFMA work needs 28-56 staging registers per lane
// next to the 16*NKT scores: at 256 registers hipcc parks them in scratch (which needs the data, so it waits), any
// exec-masked or run-time-conditional staging load draws s_waitcnt vmcnt(0), and the 4-wave / 512-register form drowns in
// AGPR <-> VGPR copies of the accumulators (all three measured) -- that is the open item.
// =============================================================================================
constexpr int kv_pad4(int d) { return d + 4; }
// persistent kernels: as many work-groups as are resident at once (LDS-limited), never more than heads
inline int kv_persistent_grid(int nbh, size_t lds_bytes) {
    const int per_cu = lds_bytes * 2 <= 160 * 1024 ? 2 : 1;
    const int g = kv_config().attn_grid > 0 ? kv_config().attn_grid : KV_N_CU * per_cu;      // KANVIT_ATTN_GRID: tuning knob
    return nbh < g ? nbh : g;
}

#ifdef KANVIT_CLOCK_PROBE
#ifndef KV_CLK_TID
#define KV_CLK_TID 0          // first lane of the stamped wave (-DKV_CLK_TID=256: wave 4, the younger partner of wave 0 on its SIMD)
#endif
// Diagnostic build only (never shipped: tools/README.md): one work-group in the middle of the grid stamps the shader clock
// and the 100 MHz real-time counter around its lifetime and around the phases of its wave 0, so the clock the chip holds
// DURING this kernel inside a real training step and the share of each phase can be read (MI355X_MICROARCH.md, "DVFS
// give-back" item 6; cdna_hip_programming.md section 7 "In-kernel stamps").  The stamps go to a buffer nothing reads.
__device__ unsigned long long g_kv_clk[16];
#define KV_CLK_BEGIN()                                                            \
    unsigned long long kv_t0 = 0, kv_r0 = 0, kv_tp = 0, kv_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};                  \
    const bool kv_stamp = (blockIdx.x == gridDim.x / 2) && threadIdx.x == KV_CLK_TID;      \
    if (kv_stamp) { kv_t0 = kv_tp = __builtin_amdgcn_s_memtime(); kv_r0 = __builtin_amdgcn_s_memrealtime(); }
#define KV_CLK_PHASE(i)                                                           \
    if (kv_stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long kv_now = __builtin_amdgcn_s_memtime(); kv_ph[i] += kv_now - kv_tp; kv_tp = kv_now; __builtin_amdgcn_sched_barrier(0); }
#define KV_CLK_END()                                                              \
    if (kv_stamp) { g_kv_clk[0] = __builtin_amdgcn_s_memtime() - kv_t0; g_kv_clk[1] = __builtin_amdgcn_s_memrealtime() - kv_r0; \
                    for (int kv_i = 0; kv_i < 8; ++kv_i) g_kv_clk[2 + kv_i] = kv_ph[kv_i]; }
#else
#define KV_CLK_BEGIN()
#define KV_CLK_PHASE(i)
#define KV_CLK_END()
#endif

template <int DT, int NTHR>
__device__ __forceinline__ void fill_rows_f32(float* __restrict__ dst, const float* __restrict__ src, long long stride_n, int NP, int N,
                                              int tid) {
    // dst[NP][32*DT + 4] <- src rows (float4 global loads, four passes in flight, ds_write_b128); rows >= N are zero
    constexpr int D = 32 * DT, C4 = D / 4, RP = NTHR / C4, KS = kv_pad4(D);
    const int c4 = (tid % C4) * 4, r0 = tid / C4;
    for (int rb = 0; rb < NP; rb += 4 * RP) {
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = rb + q * RP + r0;
            f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
            if (n < N) t = *reinterpret_cast<const f32x4*>(src + (long long)n * stride_n + c4);
            v[q] = t;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = rb + q * RP + r0;
            if (n < NP) *reinterpret_cast<f32x4*>(dst + n * KS + c4) = v[q];
        }
    }
}

constexpr int KV_A3_THREADS = 512;     // third-form attention kernels: 8 waves (two per SIMD)

template <int DT, int NKT>
__global__ __launch_bounds__(KV_A3_THREADS, 2) void attn_fwd3_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 32 * DT, NTHR = KV_A3_THREADS, NW = NTHR / 64, KS = kv_pad4(D);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int N = a.N, nkt = a.nkt, NP = nkt * 32, nbh = a.B * a.H;
    float* K_s = smem;                 // [NP][KS]
    float* V_s = K_s + NP * KS;        // [NP][KS]
    KV_CLK_BEGIN()
    const float sc2 = a.scale * LOG2E;
    for (int bh = blockIdx.x; bh < nbh; bh += gridDim.x) {       // persistent over (batch, head) pairs
        const int bi = bh / a.H, hi = bh - bi * a.H;
        const float* qb = a.q + bi * a.qsb + hi * a.qsh;
        float* ob = a.out + bi * a.osb + hi * a.osh;
        __syncthreads();                   // every wave is done with the previous head's tiles
        fill_rows_f32<DT, NTHR>(K_s, a.k + bi * a.ksb + hi * a.ksh, a.ksn, NP, N, tid);
        fill_rows_f32<DT, NTHR>(V_s, a.v + bi * a.vsb + hi * a.vsh, a.vsn, NP, N, tid);
        __syncthreads();
        KV_CLK_PHASE(0)

        for (int qt = wave; qt < nkt; qt += NW) {     // no barriers below: a wave without a tile is done
            __builtin_amdgcn_sched_barrier(0);       // phases are fenced: nothing of the next tile is hoisted into this one's registers
            const int qrow = qt * 32 + l31;
            const bool qok = qrow < N;
            // Q rows as B-operand fragments: lane half hf holds d = 16*DT*hf + s (the contraction order over d is free); rows
            // past N read row 0 (their outputs are never stored)
            const float* qp = qb + (long long)(qok ? qrow : 0) * a.qsn + 16 * DT * hf;
            float qf[16 * DT];
#pragma unroll
            for (int s4 = 0; s4 < 4 * DT; ++s4) {
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(qp + 4 * s4);
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[4 * s4 + e] = u0[e];
            }
            KV_CLK_PHASE(2)
            // ---- S^T tiles (rows = keys, columns = queries): four k-steps of K per ds_read_b128
            f32x16 sacc[NKT];
            constexpr int NG = 4 * DT;
            const float* kp = K_s + l31 * KS + 16 * DT * hf;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[j][r] = 0.0f;
                if (j < nkt) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const f32x4 ka = *reinterpret_cast<const f32x4*>(kp + j * 32 * KS + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            sacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qf[4 * g + e], sacc[j], 0, 0, 0);
                    }
                }
            }
            KV_CLK_PHASE(3)
            // ---- exact softmax over the keys held in this lane and its partner lane (other half).  Keys at or beyond `lim`
            //      are dead: N, or the query's own position + 1 under the causal mask (utils.py:177-180)
            __builtin_amdgcn_sched_barrier(0);
            const int lim = a.causal ? (qrow + 1 < N ? qrow + 1 : N) : N;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if (j < nkt) {
                    if ((j + 1) * 32 > N || a.causal) {      // wave-uniform: only the ragged last tile, or every tile when causal
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = j * 32 + kv_acc_row(r, hf);
                            sacc[j][r] = (key >= lim) ? -INFINITY : sacc[j][r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[j][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mxs = (mx == -INFINITY) ? 0.0f : mx * sc2;
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if (j < nkt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(sacc[j][r] * sc2 - mxs);      // v_exp_f32: arguments <= 0, results in [0, 1]
                        sacc[j][r] = p;
                        sum += p;
                    }
                }
            }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            KV_CLK_PHASE(4)

            // ---- O^T[d][query] = sum_key V[key][d] P[key][query]: k-step r of key tile j <-> key j*32 + kv_acc_row(r, hf)
            //      (both d tiles of a key: one ds_read2_b32); the ragged last key tile contributes only its valid keys
            __builtin_amdgcn_sched_barrier(0);
            f32x16 oacc[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.0f;
            const float* vp = V_s + (4 * hf) * KS + l31;
            const int nq_last = (N - (nkt - 1) * 32 + 7) >> 3;      // quads of the last key tile that hold a valid key (1..4)
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if (j < nkt) {
                    const int nq = (j == nkt - 1) ? nq_last : 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q < nq) {
                            const float* vq = vp + (j * 32 + 8 * q) * KS;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int dt = 0; dt < DT; ++dt)
                                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vq[e * KS + dt * 32], sacc[j][4 * q + e], oacc[dt], 0, 0, 0);
                        }
                    }
                }
            }
            KV_CLK_PHASE(5)
            if (qok) {
                float* op = ob + (long long)qrow * a.osn + 4 * hf;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {oacc[dt][4 * q] * inv, oacc[dt][4 * q + 1] * inv, oacc[dt][4 * q + 2] * inv, oacc[dt][4 * q + 3] * inv};
                        *reinterpret_cast<f32x4*>(op + dt * 32 + 8 * q) = v;
                    }
                if (hf == 0 && a.lse) a.lse[(long long)bh * N + qrow] = mx * a.scale + logf(sum);
            }
            KV_CLK_PHASE(6)
        }
        KV_CLK_PHASE(7)
    }   // heads
    KV_CLK_END()
}

// =============================================================================================
// forward, fourth form (exact fp32, D == 32*DT, N > 64, 16-byte aligned operands): the third form's mathematics with the
// operand fills taken OFF the critical path.  In the third form a head's K and V (2 x 61 KB with padded rows) fill the LDS,
// one work-group owns the CU, and nothing overlaps the fill of the next head: a fifth of the time is spent waiting for K / V
// rows.  Here
//   * the images are UNPADDED rows ([row][D] floats, 51 KB for 200 rows of 64) whose 16-byte slots are XOR-swizzled inside
//     each 256-byte line (slot p of line l holds logical slot p ^ (l & 15)): the same conflict-free ds_read_b128 (lane = row)
//     and ds_read_b32 (lane = column) as the padded image, in 84 % of the space -- THREE images fit the 160 KiB;
//   * the fills are LDS-DMA (global_load_lds_dwordx4: no staging registers, one wave-instruction = 1 KiB of the image; the
//     swizzle lives in the per-lane SOURCE address, cdna guide section 5 rule 21), issued one phase AHEAD into a ring of
//     three buffers: item i of the sequence K0 V0 K1 V1 ... lives in buffer i % 3, its fill is issued at the start of phase
//     i - 1 and lands while that phase's MFMAs run; each phase starts with s_waitcnt vmcnt(0) + ONE raw s_barrier (all
//     pieces of item i have landed; every wave is done with phase i - 1, so buffer (i + 1) % 3 = (i - 2) % 3 is free);
//   * the Q rows of the NEXT head are requested (plain loads into the dead Q registers) at the start of the P.V phase, so
//     no plain load is ever consumed while a fill is in flight (hipcc would drain the whole queue there).
// Rows >= N of an image repeat row N - 1 (finite values; their probabilities are exactly 0); the score tiles of the ragged
// last key tile read up to 27 rows past the image (whatever the next buffer holds) and mask them by SELECT before any
// arithmetic.  One work-group (8 waves: wave w owns query tile w) per CU, persistent over the (batch, head) pairs.
// =============================================================================================
typedef __attribute__((address_space(3))) void* kv_lds_ptr;
typedef const __attribute__((address_space(1))) void* kv_glb_ptr;

// rows [0, rows) of a [.][32*DT] fp32 matrix -> swizzled LDS image, by LDS-DMA; rows % (64 / (8*DT)) == 0
template <int DT>
__device__ __forceinline__ void kv_fill_glds(float* __restrict__ img, const float* __restrict__ src, long long stride_n, int rows, int N,
                                             int first, int step, int lane) {
    constexpr int SPR = 8 * DT;                   // 16-byte slots per row
    const int npieces = rows * SPR / 64;
    for (int p = first; p < npieces; p += step) {         // this wave's pieces: first, first + step, ... (wave-uniform)
        const int phys = p * 64 + lane, line = phys >> 4;
        const int logical = (phys & ~15) | ((phys & 15) ^ (line & 15));      // the swizzle is an involution
        int row = logical / SPR;
        const int sl = logical - row * SPR;
        row = row < N ? row : N - 1;
        __builtin_amdgcn_global_load_lds((kv_glb_ptr)(src + (long long)row * stride_n + 4 * sl), (kv_lds_ptr)(img + p * 256), 16, 0, 0);
    }
}

constexpr int kv_a4_rows(int N) { return (N + 7) & ~7; }                   // image rows (whole 1-KiB pieces for D = 32 and 64)
inline size_t kv_a4_lds(int N, int D) { return sizeof(float) * ((size_t)3 * kv_a4_rows(N) + 32) * D; }     // + the over-read of the last score tile

template <int DT, int NKT>
__global__ __launch_bounds__(KV_A3_THREADS, 2) void attn_fwd4_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 32 * DT, NW = KV_A3_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hf = lane >> 5;
    const int N = a.N, nkt = a.nkt, nbh = a.B * a.H;
    const int rows = kv_a4_rows(N), BUF = rows * D;                   // floats per image
    const float sc2 = a.scale * LOG2E;
    const int qt = wave;                                               // this wave's query tile
    const bool has_tile = qt < nkt;
    // Waves without a query tile (wave 7 at N = 197) are the LOADERS: they issue every LDS-DMA piece (~180 cycles of issue
    // each, 13 per head and wave when all eight share them) and leave the computing waves' instruction streams alone.
    const int nload = NW - nkt;
    const bool loader = nload > 0 ? !has_tile : true;
    const int lw = nload > 0 ? wave - nkt : wave, lstep = nload > 0 ? nload : NW;
    const int qrow = qt * 32 + l31;
    const bool qok = has_tile && qrow < N;
    // per-lane LDS offsets (floats) of the fragment reads in the swizzled image:
    //   K (lane = key row l31 of a tile, half hf takes d = 16*DT*hf + 4g + e): slot 4*DT*hf + g of line (row) -- row & 15 == l31 & 15
    //   V (lane = column d = l31 + 32*dt of key row 8q + 4hf + e): slot (l31 >> 2) + 8*dt; (row & 15) = 8*(q & 1) + 4hf + e
    static_assert(DT == 2, "the offsets below assume 256-byte rows (D = 64)");
    int koff[4 * DT];
#pragma unroll
    for (int g = 0; g < 4 * DT; ++g) koff[g] = l31 * D + 4 * ((8 * hf + g) ^ (l31 & 15));
    int voff[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) voff[e] = (4 * hf) * D + 4 * ((l31 >> 2) ^ (4 * hf + e)) + (l31 & 3);

    float qf[16 * DT];
    auto load_q = [&](int bh) {          // this lane's half row of Q (rows past N read row 0: their outputs are never stored)
        const int bi = bh / a.H, hi = bh - bi * a.H;
        const float* qp = a.q + bi * a.qsb + hi * a.qsh + (long long)(qok ? qrow : 0) * a.qsn + 16 * DT * hf;
#pragma unroll
        for (int s4 = 0; s4 < 4 * DT; ++s4) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(qp + 4 * s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[4 * s4 + e] = u0[e];
        }
    };
    auto fill = [&](int item, const float* base, long long sn) {
        if (loader) kv_fill_glds<DT>(smem + (item % 3) * BUF, base, sn, rows, N, lw, lstep, lane);
    };
    // the output tile of a head is stored AFTER the next phase's barrier, behind the next fill: issued at the end of its own
    // phase, the stores would sit in front of the next phase's s_waitcnt vmcnt(0) and every wave would wait for them to drain
    f32x16 oacc[DT];
    float o_inv = 0.0f, o_lse = 0.0f;
    float* o_ptr = nullptr;
    float* lse_ptr = nullptr;
    auto store_o = [&]() {
        if (o_ptr) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {oacc[dt][4 * q] * o_inv, oacc[dt][4 * q + 1] * o_inv, oacc[dt][4 * q + 2] * o_inv, oacc[dt][4 * q + 3] * o_inv};
                    *reinterpret_cast<f32x4*>(o_ptr + dt * 32 + 8 * q) = v;
                }
            if (lse_ptr) *lse_ptr = o_lse;
            o_ptr = nullptr;
        }
    };

    int item = 0;                                                      // K of head t is item 2t, V item 2t + 1
    int bh = blockIdx.x;
    KV_CLK_BEGIN()
    if (bh < nbh) {
        const int bi = bh / a.H, hi = bh - bi * a.H;
        fill(0, a.k + bi * a.ksb + hi * a.ksh, a.ksn);
        if (has_tile) load_q(bh);
    }
    for (; bh < nbh; bh += gridDim.x) {
        const int bi = bh / a.H, hi = bh - bi * a.H;
        // ---------------- phase S: scores and softmax from the K image (item), V fill in flight ----------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // my pieces of K (and my Q rows) have arrived
        // hipcc does not see that wait: it would wait for the Q registers itself at their first use -- AFTER the V fill below
        // has been issued, i.e. for the whole fill (vmcnt retires in order).  An empty statement that "uses" them here makes its
        // own wait land in front of the barrier, where nothing is in flight any more.
#pragma unroll
        for (int e = 0; e < 16 * DT; ++e) asm volatile("" : "+v"(qf[e]));
        __builtin_amdgcn_s_barrier();                                  // ... and everybody's; the previous P.V phase is over
        KV_CLK_PHASE(0)
        fill(item + 1, a.v + bi * a.vsb + hi * a.vsh, a.vsn);
        store_o();                                                     // the previous head's output tile
        KV_CLK_PHASE(1)
        const float* K_s = smem + (item % 3) * BUF;
        f32x16 sacc[NKT];
        float inv = 0.0f, lse_v = 0.0f;
        if (has_tile) {
            constexpr int NG = 4 * DT;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[j][r] = 0.0f;
                if (j < nkt) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const f32x4 ka = *reinterpret_cast<const f32x4*>(K_s + j * 32 * D + koff[g]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            sacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qf[4 * g + e], sacc[j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            KV_CLK_PHASE(2)
            const int lim = a.causal ? (qrow + 1 < N ? qrow + 1 : N) : N;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if (j < nkt) {
                    if ((j + 1) * 32 > N || a.causal) {      // wave-uniform: only the ragged last tile, or every tile when causal
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = j * 32 + kv_acc_row(r, hf);
                            sacc[j][r] = (key >= lim) ? -INFINITY : sacc[j][r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[j][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mxs = (mx == -INFINITY) ? 0.0f : mx * sc2;
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if (j < nkt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(sacc[j][r] * sc2 - mxs);      // v_exp_f32: arguments <= 0, results in [0, 1]
                        sacc[j][r] = p;
                        sum += p;
                    }
                }
            }
            sum += __shfl_xor(sum, 32);
            inv = 1.0f / sum;
            lse_v = mx * a.scale + logf(sum);
        }
        ++item;
        KV_CLK_PHASE(3)
        // ---------------- phase P.V from the V image (item); the next head's K fill and Q rows in flight ----------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        KV_CLK_PHASE(4)
        const int bhn = bh + gridDim.x;
        if (bhn < nbh) {
            const int bn = bhn / a.H, hn = bhn - bn * a.H;
            fill(item + 1, a.k + bn * a.ksb + hn * a.ksh, a.ksn);
            if (has_tile) load_q(bhn);
        }
        KV_CLK_PHASE(5)
        if (has_tile) {
            const float* V_s = smem + (item % 3) * BUF;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.0f;
            // O^T[d][query] = sum_key V[key][d] P[key][query], in quads of four k-steps (keys 8q + 4hf + e of key tile j).  The V
            // values of quad i + 1 are read while the MFMAs of quad i issue (a read -> wait -> two MFMAs chain exposes the LDS
            // latency on every pair: 154 cycles per MFMA measured instead of the 128 two waves of a SIMD can reach); the read
            // ahead is unconditional -- past the last valid quad it reads rows that exist in the LDS and are never used.
            const int nquads = (nkt - 1) * 4 + ((N - (nkt - 1) * 32 + 7) >> 3);      // quads that hold a valid key
            float va[2][4][DT];
            auto read_quad = [&](int qi, float (&dst)[4][DT]) {
                const int jq = qi >> 2, qq = qi & 3;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)          // key row 32*jq + 8*qq + 4hf + e, column l31 + 32*dt: slot bit 3 = dt ^ (qq & 1)
                        dst[e][dt] = V_s[voff[e] + (jq * 32 + 8 * qq + e) * D + 32 * (dt ^ (qq & 1))];
            };
            read_quad(0, va[0]);
#pragma unroll
            for (int qi = 0; qi < NKT * 4; ++qi) {
                if (qi + 1 < NKT * 4) read_quad(qi + 1, va[(qi + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (qi < nquads) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
                            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[qi & 1][e][dt], sacc[qi >> 2][4 * (qi & 3) + e], oacc[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            KV_CLK_PHASE(6)
            if (qok) {
                o_ptr = a.out + bi * a.osb + hi * a.osh + (long long)qrow * a.osn + 4 * hf;
                lse_ptr = (hf == 0 && a.lse) ? a.lse + (long long)bh * N + qrow : nullptr;
                o_inv = inv;
                o_lse = lse_v;
            }
        }
        ++item;
        KV_CLK_PHASE(7)
    }
    store_o();
    KV_CLK_END()
}

// =============================================================================================
// backward helper: delta[b,h,n] = sum_d dO * O   (utils.py:286, "D")
// =============================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnArgs a) {
    const int sub = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long long rows = (long long)a.B * a.H * a.N;
    float s = 0.0f;
    if (row < rows) {
        const int n = (int)(row % a.N);
        const long long bh = row / a.N;
        const int hi = (int)(bh % a.H);
        const long long bi = bh / a.H;
        const float* op = a.o + bi * a.osb + hi * a.osh + (long long)n * a.osn;
        const float* dp = a.d_o + bi * a.osb + hi * a.osh + (long long)n * a.osn;
        if (a.vec && (a.D & 3) == 0 && (((uintptr_t)op | (uintptr_t)dp) & 15) == 0) {      // 16 lanes x 16 bytes = one 64-float row per trip
            for (int c = sub * 4; c < a.D; c += 64) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(op + c), y = *reinterpret_cast<const f32x4*>(dp + c);
                s += (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]);
            }
        } else {
            for (int c = sub; c < a.D; c += 16) s += op[c] * dp[c];
        }
    }
    s += __shfl_xor(s, 8);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 1);
    if (row < rows && sub == 0) a.delta[row] = s;
}

// =============================================================================================
// backward dK, dV: key-stationary.  A wave owns a 32-key tile (K, V fragments in registers,
// dK^T / dV^T in accumulators) and sweeps all query tiles; S and dP are computed with the key
// on the lane so they feed the second products as B operands straight from registers.
// =============================================================================================
template <int DT, bool BF>
__global__ __launch_bounds__(ATHR) void attn_bwd_kv_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 32 * DT + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, D = a.D, nkt = a.nkt, NP = nkt * 32;
    float* Q_s = smem;                     // [NP][KS]
    float* dO_s = Q_s + NP * KS;           // [NP][KS]
    float* lse_s = dO_s + NP * KS;         // [NP]
    float* dl_s = lse_s + NP;              // [NP]
    float* T_w = dl_s + NP + wave * 32 * KS;   // per wave staging [32][KS]

    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    const float* dob = a.d_o + bi * a.osb + hi * a.osh;
    float* dkb = a.dk + bi * a.ksb + hi * a.ksh;
    float* dvb = a.dv + bi * a.vsb + hi * a.vsh;

    load_tile<DT>(Q_s, qb, a.qsn, 0, NP, N, D, tid, ATHR, a.vec);
    load_tile<DT>(dO_s, dob, a.osn, 0, NP, N, D, tid, ATHR, a.vec);
    for (int n = tid; n < NP; n += ATHR) {
        lse_s[n] = (n < N) ? a.lse_in[(long long)bh * N + n] * LOG2E : INFINITY;   // exp2(-inf) = 0 on pad rows
        dl_s[n] = (n < N) ? a.delta_in[(long long)bh * N + n] : 0.0f;
    }
    const float sc2 = a.scale * LOG2E;

    const int niter = (nkt + 3) / 4;
    for (int it = 0; it < niter; ++it) {
        const int jt = it * 4 + wave;
        const int key = jt * 32 + l31;
        const bool key_ok = (jt < nkt) && (key < N);
        // K and V fragments of this wave's keys (B operands): stage through the wave's tile
        float kf[BF ? 1 : 16 * DT], vf[BF ? 1 : 16 * DT];
        bf16x8_t kbf[BF ? 2 * DT : 1], vbf[BF ? 2 * DT : 1];
        load_tile<DT>(T_w, kb, a.ksn, jt * 32, 32, (jt < nkt) ? N : 0, D, lane, 64, a.vec);
        __syncthreads();
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < 2 * DT; ++ks) kbf[ks] = kv_row8(T_w + l31 * KS + 16 * ks + 8 * hf);
        } else {
#pragma unroll
            for (int s = 0; s < 16 * DT; ++s) kf[s] = T_w[l31 * KS + 2 * s + hf];
        }
        __syncthreads();
        load_tile<DT>(T_w, vb, a.vsn, jt * 32, 32, (jt < nkt) ? N : 0, D, lane, 64, a.vec);
        __syncthreads();
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < 2 * DT; ++ks) vbf[ks] = kv_row8(T_w + l31 * KS + 16 * ks + 8 * hf);
        } else {
#pragma unroll
            for (int s = 0; s < 16 * DT; ++s) vf[s] = T_w[l31 * KS + 2 * s + hf];
        }

        f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dkacc[dt][r] = 0.0f;
                dvacc[dt][r] = 0.0f;
            }

        for (int qt = 0; qt < nkt; ++qt) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.0f;
                pacc[r] = 0.0f;
            }
            if constexpr (BF) {
                const float* qp = Q_s + (qt * 32 + l31) * KS + 8 * hf;
                const float* dp = dO_s + (qt * 32 + l31) * KS + 8 * hf;
#pragma unroll
                for (int ks = 0; ks < 2 * DT; ++ks) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv_row8(qp + 16 * ks), kbf[ks], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv_row8(dp + 16 * ks), vbf[ks], pacc, 0, 0, 0);
                }
            } else {
                const float* qp = Q_s + (qt * 32 + l31) * KS + hf;
                const float* dp = dO_s + (qt * 32 + l31) * KS + hf;
#pragma unroll
                for (int s = 0; s < 16 * DT; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[2 * s], kf[s], sacc, 0, 0, 0);    // S[q][key]
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[2 * s], vf[s], pacc, 0, 0, 0);    // dP[q][key]
                }
            }
            // p = exp(s*scale - lse), ds = p * scale * (dp - delta)        (utils.py:278-287)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = qt * 32 + kv_acc_row(r, hf);
                float p = exp2f(sacc[r] * sc2 - lse_s[qrow]);
                if (!key_ok || (a.causal && key > qrow)) p = 0.0f;
                sacc[r] = p;
                pacc[r] = p * a.scale * (pacc[r] - dl_s[qrow]);
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
            if constexpr (BF) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t pb = kv_acc8(sacc, s2), db = kv_acc8(pacc, s2);
                    const int row = (qt * 32 + 16 * s2 + 4 * hf) * KS + l31;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv_col8(dO_s + row + dt * 32, KS), pb, dvacc[dt], 0, 0, 0);
                        dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv_col8(Q_s + row + dt * 32, KS), db, dkacc[dt], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (qt * 32 + kv_acc_row(r, hf)) * KS + l31;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dO_s[row + dt * 32], sacc[r], dvacc[dt], 0, 0, 0);
                        dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Q_s[row + dt * 32], pacc[r], dkacc[dt], 0, 0, 0);
                    }
                }
            }
        }
        // write dK, dV tiles through the wave's staging tile (rows = keys, coalesced along d)
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) T_w[l31 * KS + dt * 32 + kv_acc_row(r, hf)] = dkacc[dt][r];
        __syncthreads();
        if (jt < nkt) store_tile<DT>(dkb, a.ksn, jt * 32, N, D, T_w, lane, a.vec && ((uintptr_t)dkb % 16 == 0));
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) T_w[l31 * KS + dt * 32 + kv_acc_row(r, hf)] = dvacc[dt][r];
        __syncthreads();
        if (jt < nkt) store_tile<DT>(dvb, a.vsn, jt * 32, N, D, T_w, lane, a.vec && ((uintptr_t)dvb % 16 == 0));
        __syncthreads();
    }
}

// =============================================================================================
// backward dQ: query-stationary mirror of the forward kernel.
// =============================================================================================
template <int DT, bool BF>
__global__ __launch_bounds__(ATHR) void attn_bwd_q_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 32 * DT + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, D = a.D, nkt = a.nkt, NP = nkt * 32;
    float* K_s = smem;
    float* V_s = K_s + NP * KS;
    float* T_w = V_s + NP * KS + wave * 32 * KS;

    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    const float* dob = a.d_o + bi * a.osb + hi * a.osh;
    float* dqb = a.dq + bi * a.qsb + hi * a.qsh;

    load_tile<DT>(K_s, kb, a.ksn, 0, NP, N, D, tid, ATHR, a.vec);
    load_tile<DT>(V_s, vb, a.vsn, 0, NP, N, D, tid, ATHR, a.vec);
    const float sc2 = a.scale * LOG2E;

    const int niter = (nkt + 3) / 4;
    for (int it = 0; it < niter; ++it) {
        const int qt = it * 4 + wave;
        const int qrow = qt * 32 + l31;
        const bool q_ok = (qt < nkt) && (qrow < N);
        float qf[BF ? 1 : 16 * DT], dof[BF ? 1 : 16 * DT];
        bf16x8_t qbf[BF ? 2 * DT : 1], dobf[BF ? 2 * DT : 1];
        load_tile<DT>(T_w, qb, a.qsn, qt * 32, 32, (qt < nkt) ? N : 0, D, lane, 64, a.vec);
        __syncthreads();
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < 2 * DT; ++ks) qbf[ks] = kv_row8(T_w + l31 * KS + 16 * ks + 8 * hf);
        } else {
#pragma unroll
            for (int s = 0; s < 16 * DT; ++s) qf[s] = T_w[l31 * KS + 2 * s + hf];
        }
        __syncthreads();
        load_tile<DT>(T_w, dob, a.osn, qt * 32, 32, (qt < nkt) ? N : 0, D, lane, 64, a.vec);
        __syncthreads();
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < 2 * DT; ++ks) dobf[ks] = kv_row8(T_w + l31 * KS + 16 * ks + 8 * hf);
        } else {
#pragma unroll
            for (int s = 0; s < 16 * DT; ++s) dof[s] = T_w[l31 * KS + 2 * s + hf];
        }
        const float lse2 = q_ok ? a.lse_in[(long long)bh * N + qrow] * LOG2E : INFINITY;
        const float dl = q_ok ? a.delta_in[(long long)bh * N + qrow] : 0.0f;

        f32x16 dqacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.0f;

        for (int j = 0; j < nkt; ++j) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.0f;
                pacc[r] = 0.0f;
            }
            if constexpr (BF) {
                const float* kp = K_s + (j * 32 + l31) * KS + 8 * hf;
                const float* vp = V_s + (j * 32 + l31) * KS + 8 * hf;
#pragma unroll
                for (int ks = 0; ks < 2 * DT; ++ks) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv_row8(kp + 16 * ks), qbf[ks], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv_row8(vp + 16 * ks), dobf[ks], pacc, 0, 0, 0);
                }
            } else {
                const float* kp = K_s + (j * 32 + l31) * KS + hf;
                const float* vp = V_s + (j * 32 + l31) * KS + hf;
#pragma unroll
                for (int s = 0; s < 16 * DT; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s], qf[s], sacc, 0, 0, 0);    // S^T[key][q]
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[2 * s], dof[s], pacc, 0, 0, 0);   // dP^T[key][q]
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = j * 32 + kv_acc_row(r, hf);
                float p = exp2f(sacc[r] * sc2 - lse2);
                if (key >= N || (a.causal && key > qrow)) p = 0.0f;
                pacc[r] = p * a.scale * (pacc[r] - dl);                                          // dS^T
            }
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
            if constexpr (BF) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t db = kv_acc8(pacc, s2);
                    const float* kr = K_s + (j * 32 + 16 * s2 + 4 * hf) * KS + l31;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv_col8(kr + dt * 32, KS), db, dqacc[dt], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* kr = K_s + (j * 32 + kv_acc_row(r, hf)) * KS + l31;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[dt * 32], pacc[r], dqacc[dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) T_w[l31 * KS + dt * 32 + kv_acc_row(r, hf)] = dqacc[dt][r];
        __syncthreads();
        if (qt < nkt) store_tile<DT>(dqb, a.qsn, qt * 32, N, D, T_w, lane, a.vec && ((uintptr_t)dqb % 16 == 0));
        __syncthreads();
    }
}

// =============================================================================================
// backward, second form (D == 32*DT, aligned operands): the same two kernels with the operand placement of attn_fwd2_kernel.
//   * the stationary tile's fragments (K, V rows of the wave's keys; Q, dO rows of the wave's queries) are B operands
//     indexed (d, lane) -> loaded from the lane's own global row, no staging, no barriers;
//   * dK^T / dV^T / dQ^T accumulators hold 4 consecutive d of the lane's own row per register quad -> float4 stores;
//   * bf16 mode keeps bf16 images of the swept operand in the LDS: a row image [n][D + 8] for the first products and a
//     transposed image [d][NP + 8] (tile rows in accumulator order, kv_key_slot) for the second products, so every A
//     fragment is one ds_read_b128;
//   * 8 waves, one tile each: one barrier per kernel (after the LDS fill).
// =============================================================================================
// Fill bf16 images of src[n][D] (fp32, row stride `stride`): rowimg[n][D + 8] and/or timg[d][NP + 8]; rows >= N are zero.
template <int DT, int NTHR, bool NATURAL = false>      // NATURAL: the transposed image keeps the rows of a tile in natural order (operand read from memory, not from accumulators)
__device__ __forceinline__ void fill_bf16_images(unsigned short* __restrict__ rowimg, unsigned short* __restrict__ timg,
                                                 const float* __restrict__ src, long long stride, int NP, int N, int tid) {
    constexpr int D = 32 * DT, C4 = D / 4, RP = NTHR / C4, KB = D + 8;
    const int VB = NP + 8;
    const int c4 = (tid % C4) * 4, r0 = tid / C4;
    const int NPAIR = NP / 2;
    for (int pb = 0; pb < NPAIR; pb += 4 * RP) {
        f32x4 v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = 2 * (pb + q * RP + r0);
            f32x4 t0 = {0.0f, 0.0f, 0.0f, 0.0f}, t1 = t0;
            if (n < N) t0 = *reinterpret_cast<const f32x4*>(src + (long long)n * stride + c4);
            if (n + 1 < N) t1 = *reinterpret_cast<const f32x4*>(src + (long long)(n + 1) * stride + c4);
            v0[q] = t0;
            v1[q] = t1;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = 2 * (pb + q * RP + r0);
            if (n < NP) {
                if (rowimg) {
                    unsigned* d0 = reinterpret_cast<unsigned*>(rowimg + n * KB + c4);
                    d0[0] = kv_pk(v0[q][0], v0[q][1]);
                    d0[1] = kv_pk(v0[q][2], v0[q][3]);
                    unsigned* d1 = reinterpret_cast<unsigned*>(rowimg + (n + 1) * KB + c4);
                    d1[0] = kv_pk(v1[q][0], v1[q][1]);
                    d1[1] = kv_pk(v1[q][2], v1[q][3]);
                }
                if (timg) {
                    const int pos = NATURAL ? n : (n & ~31) + kv_key_slot(n & 31);
#pragma unroll
                    for (int e = 0; e < 4; ++e) *reinterpret_cast<unsigned*>(timg + (c4 + e) * VB + pos) = kv_pk(v0[q][e], v1[q][e]);
                }
            }
        }
    }
}

// B-operand fragments of one global row: bf16: 2*DT fragments of 8 consecutive d (16*ks + 8*hf ..); fp32: the half row
// d = 16*DT*hf + s.  ok == false gives zeros.
template <int DT, bool BF>
__device__ __forceinline__ void row_frags(const float* __restrict__ rowp, bool ok, int hf, float (&f)[BF ? 1 : 16 * DT],
                                          bf16x8_t (&fb)[BF ? 2 * DT : 1]) {
    if constexpr (BF) {
#pragma unroll
        for (int ks = 0; ks < 2 * DT; ++ks) {
            f32x4 u0 = {0.0f, 0.0f, 0.0f, 0.0f}, u1 = u0;
            if (ok) {
                u0 = *reinterpret_cast<const f32x4*>(rowp + 16 * ks + 8 * hf);
                u1 = *reinterpret_cast<const f32x4*>(rowp + 16 * ks + 8 * hf + 4);
            }
            const u32x4 u = {kv_pk(u0[0], u0[1]), kv_pk(u0[2], u0[3]), kv_pk(u1[0], u1[1]), kv_pk(u1[2], u1[3])};
            fb[ks] = __builtin_bit_cast(bf16x8_t, u);
        }
    } else {
#pragma unroll
        for (int s4 = 0; s4 < 4 * DT; ++s4) {
            f32x4 u0 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (ok) u0 = *reinterpret_cast<const f32x4*>(rowp + 16 * DT * hf + 4 * s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * s4 + e] = u0[e];
        }
    }
}

// accumulator [d][row = lane] -> the lane's global row (registers 4q..4q+3 = d = 32*dt + 8q + 4hf + 0..3)
template <int DT>
__device__ __forceinline__ void store_acc_rows(float* __restrict__ rowp, const f32x16 (&acc)[DT], int hf, float mul) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[dt][4 * q] * mul, acc[dt][4 * q + 1] * mul, acc[dt][4 * q + 2] * mul, acc[dt][4 * q + 3] * mul};
            *reinterpret_cast<f32x4*>(rowp + dt * 32 + 8 * q + 4 * hf) = v;
        }
}

template <int DT, bool BF, bool DSOUT>
__global__ __launch_bounds__(512) void attn_bwd_kv2_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 32 * DT, KS = D + 1, KB = D + 8, NTHR = 512, NW = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, nkt = a.nkt, NP = nkt * 32, VB = NP + 8;
    // fp32: Q_s[NP][KS] dO_s[NP][KS] | bf16: Qr[NP][KB] dOr[NP][KB] Qt[D][VB] dOt[D][VB]; then lse_s[NP] dl_s[NP]
    float* Q_s = smem;
    float* dO_s = Q_s + NP * KS;
    unsigned short* Qr = reinterpret_cast<unsigned short*>(smem);
    unsigned short* dOr = Qr + NP * KB;
    unsigned short* Qt = dOr + NP * KB;
    unsigned short* dOt = Qt + D * VB;
    float* lse_s = BF ? reinterpret_cast<float*>(dOt + D * VB) : dO_s + NP * KS;
    float* dl_s = lse_s + NP;

    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    const float* dob = a.d_o + bi * a.osb + hi * a.osh;
    float* dkb = a.dk + bi * a.ksb + hi * a.ksh;
    float* dvb = a.dv + bi * a.vsb + hi * a.vsh;

    if constexpr (BF) {
        fill_bf16_images<DT, NTHR>(Qr, Qt, qb, a.qsn, NP, N, tid);
        fill_bf16_images<DT, NTHR>(dOr, dOt, dob, a.osn, NP, N, tid);
    } else {
        load_tile<DT>(Q_s, qb, a.qsn, 0, NP, N, D, tid, NTHR, true);
        load_tile<DT>(dO_s, dob, a.osn, 0, NP, N, D, tid, NTHR, true);
    }
    for (int n = tid; n < NP; n += NTHR) {
        lse_s[n] = (n < N) ? a.lse_in[(long long)bh * N + n] * LOG2E : INFINITY;   // exp2(-inf) = 0 on pad rows
        dl_s[n] = (n < N) ? a.delta_in[(long long)bh * N + n] : 0.0f;
    }
    __syncthreads();
    const float sc2 = a.scale * LOG2E;

    for (int jt = wave; jt < nkt; jt += NW) {
        const int key = jt * 32 + l31;
        const bool key_ok = key < N;
        float kf[BF ? 1 : 16 * DT], vf[BF ? 1 : 16 * DT];
        bf16x8_t kbf[BF ? 2 * DT : 1], vbf[BF ? 2 * DT : 1];
        row_frags<DT, BF>(kb + (long long)(key_ok ? key : 0) * a.ksn, key_ok, hf, kf, kbf);
        row_frags<DT, BF>(vb + (long long)(key_ok ? key : 0) * a.vsn, key_ok, hf, vf, vbf);

        f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dkacc[dt][r] = 0.0f;
                dvacc[dt][r] = 0.0f;
            }

        for (int qt = 0; qt < nkt; ++qt) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.0f;
                pacc[r] = 0.0f;
            }
            if constexpr (BF) {
                const unsigned short* qp = Qr + (qt * 32 + l31) * KB + 8 * hf;
                const unsigned short* dp = dOr + (qt * 32 + l31) * KB + 8 * hf;
#pragma unroll
                for (int ks = 0; ks < 2 * DT; ++ks) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(qp + 16 * ks), kbf[ks], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(dp + 16 * ks), vbf[ks], pacc, 0, 0, 0);
                }
            } else {
                const float* qp = Q_s + (qt * 32 + l31) * KS + 16 * DT * hf;
                const float* dp = dO_s + (qt * 32 + l31) * KS + 16 * DT * hf;
#pragma unroll
                for (int s = 0; s < 16 * DT; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[s], kf[s], sacc, 0, 0, 0);    // S[q][key]
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[s], vf[s], pacc, 0, 0, 0);    // dP[q][key]
                }
            }
            // p = exp(s*scale - lse), ds = p * scale * (dp - delta)        (utils.py:278-287)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = qt * 32 + kv_acc_row(r, hf);
                float p = exp2f(sacc[r] * sc2 - lse_s[qrow]);
                if (!key_ok || (a.causal && key > qrow)) p = 0.0f;
                sacc[r] = p;
                pacc[r] = p * a.scale * (pacc[r] - dl_s[qrow]);
            }
            if constexpr (DSOUT) {       // dS[q][key] for the dQ kernel: row = q (register), 32 consecutive keys per lane half
                if constexpr (BF) {      // bf16 mode: the hand-off is the bf16 operand the dQ product consumes (attn_bwd_dq_bf16_kernel)
                    unsigned short* dsp = reinterpret_cast<unsigned short*>(a.ds) + ((long long)bh * NP + qt * 32) * NP + jt * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dsp[(long long)kv_acc_row(r, hf) * NP] = (unsigned short)(kv_pk(pacc[r], 0.0f) & 0xffffu);
                } else {
                    float* dsp = a.ds + ((long long)bh * NP + qt * 32) * NP + jt * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dsp[(long long)kv_acc_row(r, hf) * NP] = pacc[r];
                }
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
            if constexpr (BF) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t pb = kv_acc8(sacc, s2), db = kv_acc8(pacc, s2);
                    const int off = l31 * VB + qt * 32 + 16 * s2 + 8 * hf;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(dOt + off + dt * 32 * VB), pb,
                                                                            dvacc[dt], 0, 0, 0);
                        dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(Qt + off + dt * 32 * VB), db,
                                                                            dkacc[dt], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (qt * 32 + kv_acc_row(r, hf)) * KS + l31;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dO_s[row + dt * 32], sacc[r], dvacc[dt], 0, 0, 0);
                        dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Q_s[row + dt * 32], pacc[r], dkacc[dt], 0, 0, 0);
                    }
                }
            }
        }
        if (key_ok) {
            store_acc_rows<DT>(dkb + (long long)key * a.ksn, dkacc, hf, 1.0f);
            store_acc_rows<DT>(dvb + (long long)key * a.vsn, dvacc, hf, 1.0f);
        }
    }
}

template <int DT, bool BF>
__global__ __launch_bounds__(512) void attn_bwd_q2_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 32 * DT, KS = D + 1, KB = D + 8, NTHR = 512, NW = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, nkt = a.nkt, NP = nkt * 32, VB = NP + 8;
    float* K_s = smem;                    // fp32: K_s[NP][KS] V_s[NP][KS]
    float* V_s = K_s + NP * KS;
    unsigned short* Kr = reinterpret_cast<unsigned short*>(smem);   // bf16: Kr[NP][KB] Vr[NP][KB] Kt[D][VB]
    unsigned short* Vr = Kr + NP * KB;
    unsigned short* Kt = Vr + NP * KB;

    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    const float* dob = a.d_o + bi * a.osb + hi * a.osh;
    float* dqb = a.dq + bi * a.qsb + hi * a.qsh;

    if constexpr (BF) {
        fill_bf16_images<DT, NTHR>(Kr, Kt, kb, a.ksn, NP, N, tid);
        fill_bf16_images<DT, NTHR>(Vr, nullptr, vb, a.vsn, NP, N, tid);
    } else {
        load_tile<DT>(K_s, kb, a.ksn, 0, NP, N, D, tid, NTHR, true);
        load_tile<DT>(V_s, vb, a.vsn, 0, NP, N, D, tid, NTHR, true);
    }
    __syncthreads();
    const float sc2 = a.scale * LOG2E;

    for (int qt = wave; qt < nkt; qt += NW) {
        const int qrow = qt * 32 + l31;
        const bool q_ok = qrow < N;
        float qf[BF ? 1 : 16 * DT], dof[BF ? 1 : 16 * DT];
        bf16x8_t qbf[BF ? 2 * DT : 1], dobf[BF ? 2 * DT : 1];
        row_frags<DT, BF>(qb + (long long)(q_ok ? qrow : 0) * a.qsn, q_ok, hf, qf, qbf);
        row_frags<DT, BF>(dob + (long long)(q_ok ? qrow : 0) * a.osn, q_ok, hf, dof, dobf);
        const float lse2 = q_ok ? a.lse_in[(long long)bh * N + qrow] * LOG2E : INFINITY;
        const float dl = q_ok ? a.delta_in[(long long)bh * N + qrow] : 0.0f;

        f32x16 dqacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.0f;

        for (int j = 0; j < nkt; ++j) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.0f;
                pacc[r] = 0.0f;
            }
            if constexpr (BF) {
                const unsigned short* kp = Kr + (j * 32 + l31) * KB + 8 * hf;
                const unsigned short* vp = Vr + (j * 32 + l31) * KB + 8 * hf;
#pragma unroll
                for (int ks = 0; ks < 2 * DT; ++ks) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(kp + 16 * ks), qbf[ks], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(vp + 16 * ks), dobf[ks], pacc, 0, 0, 0);
                }
            } else {
                const float* kp = K_s + (j * 32 + l31) * KS + 16 * DT * hf;
                const float* vp = V_s + (j * 32 + l31) * KS + 16 * DT * hf;
#pragma unroll
                for (int s = 0; s < 16 * DT; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[s], qf[s], sacc, 0, 0, 0);    // S^T[key][q]
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[s], dof[s], pacc, 0, 0, 0);   // dP^T[key][q]
                }
            }
            const bool edge = (j == nkt - 1) || a.causal;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = exp2f(sacc[r] * sc2 - lse2);
                if (edge) {
                    const int key = j * 32 + kv_acc_row(r, hf);
                    if (key >= N || (a.causal && key > qrow)) p = 0.0f;
                }
                pacc[r] = p * a.scale * (pacc[r] - dl);                                          // dS^T
            }
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
            if constexpr (BF) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t db = kv_acc8(pacc, s2);
                    const unsigned short* kr = Kt + l31 * VB + j * 32 + 16 * s2 + 8 * hf;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(kr + dt * 32 * VB), db, dqacc[dt],
                                                                            0, 0, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* kr = K_s + (j * 32 + kv_acc_row(r, hf)) * KS + l31;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[dt * 32], pacc[r], dqacc[dt], 0, 0, 0);
                }
            }
        }
        if (q_ok) store_acc_rows<DT>(dqb + (long long)qrow * a.qsn, dqacc, hf, 1.0f);
    }
}

// dQ from the stored dS, bf16 mode: dQ^T[d][q] = sum_key K^T[d][key] dS[q][key] with dS already bf16 in memory (written by
// attn_bwd_kv2_kernel<DT, true, true>) -- one product and no exponentials instead of attn_bwd_q2_kernel's three products, two
// row images and 16*nkt exp2 per lane.  K^T is a bf16 image [d][NP + 8] in natural key order (A operand: one ds_read_b128 per
// k-step); the B operand is 8 consecutive keys of the lane's OWN dS row, one 16-byte global load per k-step, all 2*nkt of them
// requested before the first product.  8 waves, one query tile each, one barrier (after the fill).
template <int DT, int NKT>
__global__ __launch_bounds__(512) void attn_bwd_dq_bf16_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 32 * DT, NTHR = 512, NW = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, nkt = a.nkt, NP = nkt * 32, VB = NP + 8;
    unsigned short* Kt = reinterpret_cast<unsigned short*>(smem);      // [D][VB]
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    float* dqb = a.dq + bi * a.qsb + hi * a.qsh;
    const unsigned short* dsb = reinterpret_cast<const unsigned short*>(a.ds) + (long long)bh * NP * NP;

    // this wave's dS fragments first (they do not depend on the fill): rows >= N of a tile were written as zeros by the kv kernel
    const int qt = wave;
    const int qrow = qt * 32 + l31;
    bf16x8_t dsf[2 * NKT];
    if (qt < nkt) {
        const unsigned short* dsr = dsb + (long long)qrow * NP + 8 * hf;
#pragma unroll
        for (int f = 0; f < 2 * NKT; ++f)
            if (f < 2 * nkt) dsf[f] = *reinterpret_cast<const bf16x8_t*>(dsr + 16 * f);
    }
    fill_bf16_images<DT, NTHR, true>(nullptr, Kt, kb, a.ksn, NP, N, tid);
    __syncthreads();
    if (qt >= nkt) return;

    f32x16 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.0f;
    const unsigned short* kr = Kt + l31 * VB + 8 * hf;
#pragma unroll
    for (int f = 0; f < 2 * NKT; ++f) {
        if (f < 2 * nkt) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(kr + dt * 32 * VB + 16 * f), dsf[f],
                                                                    dqacc[dt], 0, 0, 0);
        }
    }
    if (qrow < N) store_acc_rows<DT>(dqb + (long long)qrow * a.qsn, dqacc, hf, 1.0f);
}

// dQ from the stored dS (fp32 path): dQ^T[d][q] = sum_key K^T[d][key] dS[q][key] -- one product instead of the three of
// attn_bwd_q2_kernel (S and dP are not recomputed, no exponentials).  K in LDS (A operand, lanes walk d); the B operand is 16
// consecutive keys of the lane's own dS row per key tile (lane half h takes keys 16h + s), float4 loads one tile ahead.
template <int DT>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 32 * DT, KS = D + 1, NTHR = 512, NW = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, nkt = a.nkt, NP = nkt * 32;
    float* K_s = smem;                    // [NP][KS]
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    float* dqb = a.dq + bi * a.qsb + hi * a.qsh;
    load_tile<DT>(K_s, kb, a.ksn, 0, NP, N, D, tid, NTHR, true);
    __syncthreads();
    for (int qt = wave; qt < nkt; qt += NW) {
        const int qrow = qt * 32 + l31;
        const float* dsrow = a.ds + ((long long)bh * NP + qrow) * NP + 16 * hf;
        f32x16 dqacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.0f;
        f32x4 nxt[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) nxt[e] = *reinterpret_cast<const f32x4*>(dsrow + 4 * e);
        for (int j = 0; j < nkt; ++j) {
            float dsv[16];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) dsv[4 * e + c] = nxt[e][c];
            if (j + 1 < nkt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) nxt[e] = *reinterpret_cast<const f32x4*>(dsrow + (j + 1) * 32 + 4 * e);
            }
            const float* kr = K_s + (j * 32 + 16 * hf) * KS + l31;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[s2 * KS + dt * 32], dsv[s2], dqacc[dt], 0, 0, 0);
            }
        }
        if (qrow < N) store_acc_rows<DT>(dqb + (long long)qrow * a.qsn, dqacc, hf, 1.0f);
    }
}

// =============================================================================================
// backward, third form (exact fp32, D == 32*DT, aligned operands): the two kernels of the dS hand-off path (key-stationary
// dK / dV kernel that also stores dS, then dQ = dS.K as one plain product) with the LDS traffic of attn_fwd3_kernel:
// 16-byte aligned rows, ds_read_b128 for the row-walking operands (four k-steps per instruction), ds_read2_b32 for the
// column-walking ones, every fragment read one group of MFMAs ahead into a second register set, and the ragged last tile
// contributing only its valid rows to the products that contract over it.  rowsum(dO*O) (utils.py:286) is formed in the
// prologue of the first kernel from the dO tile it has just staged (no separate pass over dO and O).
// =============================================================================================
template <int DT, int NKT>
__global__ __launch_bounds__(512, 2) void attn_bwd_kv3_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 32 * DT, KS = kv_pad4(D), NTHR = 512, NW = 8, NG = 4 * DT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int N = a.N, nkt = a.nkt, NP = nkt * 32, nbh = a.B * a.H;
    float* Q_s = smem;                     // [NP][KS]
    float* dO_s = Q_s + NP * KS;           // [NP][KS]
    float* ld_s = dO_s + NP * KS;          // [NP][2]: lse * log2(e) (+inf on pad rows: p = exp2(-inf) = 0), rowsum(dO*O)
    for (int bh = blockIdx.x; bh < nbh; bh += gridDim.x) {       // persistent over (batch, head) pairs
    const int bi = bh / a.H, hi = bh - bi * a.H;
    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    const float* ob = a.o + bi * a.osb + hi * a.osh;
    const float* dob = a.d_o + bi * a.osb + hi * a.osh;
    float* dkb = a.dk + bi * a.ksb + hi * a.ksh;
    float* dvb = a.dv + bi * a.vsb + hi * a.vsh;

    __syncthreads();                       // every wave is done with the previous head's tiles
    fill_rows_f32<DT, NTHR>(Q_s, qb, a.qsn, NP, N, tid);
    fill_rows_f32<DT, NTHR>(dO_s, dob, a.osn, NP, N, tid);
    __syncthreads();
    for (int n = tid; n < NP; n += NTHR) {
        float dl = 0.0f, l2 = INFINITY;
        if (n < N) {
            const float* orow = ob + (long long)n * a.osn;
            const float* drow = dO_s + n * KS;
#pragma unroll
            for (int c4 = 0; c4 < D / 4; ++c4) {
                const f32x4 o4 = *reinterpret_cast<const f32x4*>(orow + 4 * c4);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(drow + 4 * c4);
                dl += o4[0] * d4[0] + o4[1] * d4[1] + o4[2] * d4[2] + o4[3] * d4[3];
            }
            l2 = a.lse_in[(long long)bh * N + n] * LOG2E;
        }
        ld_s[2 * n] = l2;
        ld_s[2 * n + 1] = dl;
    }
    __syncthreads();
    const float sc2 = a.scale * LOG2E;
    const int nq_last = (N - (nkt - 1) * 32 + 7) >> 3;      // quads of k-steps of the last QUERY tile that hold a valid row

    for (int jt = wave; jt < nkt; jt += NW) {
        const int key = jt * 32 + l31;
        const bool key_ok = key < N;
        float kf[16 * DT], vf[16 * DT];
        bf16x8_t unused0[1], unused1[1];
        row_frags<DT, false>(kb + (long long)(key_ok ? key : 0) * a.ksn, key_ok, hf, kf, unused0);
        row_frags<DT, false>(vb + (long long)(key_ok ? key : 0) * a.vsn, key_ok, hf, vf, unused1);

        f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dkacc[dt][r] = 0.0f;
                dvacc[dt][r] = 0.0f;
            }

        const float* qrp = Q_s + l31 * KS + 16 * DT * hf;       // row-walking reads (first products)
        const float* drp = dO_s + l31 * KS + 16 * DT * hf;
        const float* qcp = Q_s + (4 * hf) * KS + l31;           // column-walking reads (second products)
        const float* dcp = dO_s + (4 * hf) * KS + l31;
        for (int qt = 0; qt < nkt; ++qt) {
            // ---- S[q][key] = Q.K^T and dP[q][key] = dO.V^T, two independent accumulation chains interleaved
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.0f;
                pacc[r] = 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f32x4 qa = *reinterpret_cast<const f32x4*>(qrp + qt * 32 * KS + 4 * g);
                const f32x4 da = *reinterpret_cast<const f32x4*>(drp + qt * 32 * KS + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[e], kf[4 * g + e], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(da[e], vf[4 * g + e], pacc, 0, 0, 0);
                }
            }
            const int nq = (qt == nkt - 1) ? nq_last : 4;
            // ---- p = exp(s*scale - lse), ds = p * scale * (dp - delta)        (utils.py:278-287)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = qt * 32 + kv_acc_row(r, hf);
                const float2 ld = *reinterpret_cast<const float2*>(ld_s + 2 * qrow);
                float p = __builtin_amdgcn_exp2f(sacc[r] * sc2 - ld.x);      // v_exp_f32: exp2(-inf) = 0 on pad rows
                if (!key_ok || (a.causal && key > qrow)) p = 0.0f;
                sacc[r] = p;
                pacc[r] = p * a.scale * (pacc[r] - ld.y);
            }
            {       // dS[q][key] for the dQ kernel: row = q (register), 32 consecutive keys per lane half
                float* dsp = a.ds + ((long long)bh * NP + qt * 32) * NP + jt * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) dsp[(long long)kv_acc_row(r, hf) * NP] = pacc[r];
            }
            // ---- dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]: k-step r <-> query row
            //      qt*32 + kv_acc_row(r, hf); the ragged last query tile contributes only its valid rows (p = ds = 0 elsewhere)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nq) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = (qt * 32 + 8 * q + e) * KS;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dcp[row + dt * 32], sacc[4 * q + e], dvacc[dt], 0, 0, 0);
                            dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qcp[row + dt * 32], pacc[4 * q + e], dkacc[dt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (key_ok) {
            store_acc_rows<DT>(dkb + (long long)key * a.ksn, dkacc, hf, 1.0f);
            store_acc_rows<DT>(dvb + (long long)key * a.vsn, dvacc, hf, 1.0f);
        }
    }
    }   // heads
}

// =============================================================================================
// backward, fourth form of the key-stationary kernel (exact fp32, D = 64, 64 < N <= 208): attn_bwd_kv3_kernel's mathematics
// with NOTHING fetched at the head boundary.  The third form spends a fifth of a head there -- two synchronous 61 KB fills
// (Q, dO), rowsum(dO*O) with one row per thread, the K / V rows of every wave -- and, the work-groups running in step, the
// whole chip asks for its 250 KB per CU at the same moment (phase stamps: 35 k of 179 k cycles per head).  Here
//   * Q and dO are unpadded XOR-swizzled 51 KB images (attn_fwd4_kernel's layout) filled by LDS-DMA into a ring of three
//     buffers by the LOADER wave (wave 7: seven key tiles on eight waves), which also forms the (lse * log2 e, rowsum(dO*O))
//     table of the next head from global memory;
//   * head t computes from Q_t in buffer 2t % 3 and dO_t in buffer (2t + 1) % 3.  Q_{t+1} goes to the free third buffer.
//     dO_{t+1} goes INTO THE Q_t BUFFER, query tile by query tile: the rows of tile qt are dead once every wave has finished
//     its second products on them -- one barrier per query tile tells the loader so -- and by the end of the head only the
//     ragged last tile's pieces are still in flight;
//   * the K / V rows of the next head are requested into the (dead) fragment registers right after the LAST first products
//     of a head, and land under its last second products.
// The loader spreads its work over the query-tile intervals (a seventh of the Q image and of the table, then the barrier,
// then eight dO pieces), so it never keeps the computing waves waiting.  The column-walking operands of the second products
// (dO^T, Q^T) are read one half quad of k-steps ahead into a second register set.  dS goes to the workspace for
// attn_bwd_dq3_kernel exactly as before (a dQ product inside this kernel would need the K tiles AND a cross-wave reduction
// buffer in an LDS that is full).
// =============================================================================================
inline size_t kv_b4_lds(int N, int D) {
    const int np = (N + 31) / 32 * 32, rows = kv_a4_rows(N);
    return sizeof(float) * (((size_t)3 * rows + (np - rows)) * D + (size_t)2 * np * 2 + 8);      // images + over-read of the ragged tile + 2 x (lse, delta)[NP] + tile counters
}

// pieces [p0, p1) of a swizzled image (see kv_fill_glds), issued by ONE wave
template <int DT>
__device__ __forceinline__ void kv_fill_glds_range(float* __restrict__ img, const float* __restrict__ src, long long stride_n, int N, int p0,
                                                   int p1, int lane) {
    constexpr int SPR = 8 * DT;
    for (int p = p0; p < p1; ++p) {
        const int phys = p * 64 + lane, line = phys >> 4;
        const int logical = (phys & ~15) | ((phys & 15) ^ (line & 15));
        int row = logical / SPR;
        const int sl = logical - row * SPR;
        row = row < N ? row : N - 1;
        __builtin_amdgcn_global_load_lds((kv_glb_ptr)(src + (long long)row * stride_n + 4 * sl), (kv_lds_ptr)(img + p * 256), 16, 0, 0);
    }
}

template <int DT>
__global__ __launch_bounds__(512, 2) void attn_bwd_kv4_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(DT == 2, "256-byte rows (D = 64)");
    constexpr int D = 32 * DT, NW = 8, NG = 4 * DT, PPT = 32 * D / 256;      // LDS-DMA pieces per 32-row tile (8)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hf = lane >> 5;
    const int N = a.N, nkt = a.nkt, NP = nkt * 32, nbh = a.B * a.H;
    const int rows = kv_a4_rows(N), BUF = rows * D, npieces = rows * D / 256;
    float* ld_base = smem + 3 * BUF + (NP - rows) * D;                // [2][NP][2]: lse * log2(e) (+inf on pad rows), rowsum(dO*O)
    // done[qt]: how many (wave, head) pairs have finished query tile qt -- monotonic, so nkt * (t + 1) once every computing
    // wave is through tile qt of head t.  The loader polls it (it has nothing else to do); the computing waves never wait.
    unsigned* done = reinterpret_cast<unsigned*>(ld_base + 2 * (2 * NP));
    if (tid < 8) done[tid] = 0u;
    const int jt = wave;                                              // this wave's key tile
    const bool loader = jt >= nkt;                                    // host-checked: exactly the waves nkt .. 7; the FIRST of them loads
    const float sc2 = a.scale * LOG2E;
    const int nq_last = (N - (nkt - 1) * 32 + 7) >> 3;               // quads of k-steps of the last QUERY tile that hold a valid row
    int koff[NG];                                                      // row-walking fragment reads (lane = row l31 of a tile), see attn_fwd4_kernel
#pragma unroll
    for (int g = 0; g < NG; ++g) koff[g] = l31 * D + 4 * ((8 * hf + g) ^ (l31 & 15));
    int voff[4];                                                       // column-walking reads (lane = column l31 + 32*dt of row 8q + 4hf + e)
#pragma unroll
    for (int e = 0; e < 4; ++e) voff[e] = (4 * hf) * D + 4 * ((l31 >> 2) ^ (4 * hf + e)) + (l31 & 3);

    // (lse * log2 e, rowsum(dO * O)) of query rows [n_lo, n_hi) of head bhx -> ld[NP][2]; one wave, 4 lanes per row
    auto table_rows = [&](int bhx, float* __restrict__ ld, int n_lo, int n_hi) {
        const int bx = bhx / a.H, hx = bhx - bx * a.H;
        const float* ob = a.o + bx * a.osb + hx * a.osh;
        const float* dob = a.d_o + bx * a.osb + hx * a.osh;
        const int sub = lane & 3, r0 = lane >> 2;
        for (int n0 = n_lo; n0 < n_hi; n0 += 16) {
            const int n = n0 + r0;
            float dl = 0.0f;
            if (n < N) {
                const float* orow = ob + (long long)n * a.osn + 16 * sub;
                const float* drow = dob + (long long)n * a.osn + 16 * sub;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 o4 = *reinterpret_cast<const f32x4*>(orow + 4 * c);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(drow + 4 * c);
                    dl += o4[0] * d4[0] + o4[1] * d4[1] + o4[2] * d4[2] + o4[3] * d4[3];
                }
            }
            dl += __shfl_xor(dl, 1);
            dl += __shfl_xor(dl, 2);
            if (sub == 0 && n < n_hi) {
                float2 v = {INFINITY, 0.0f};                          // pad rows: p = exp2(s - inf) = 0
                if (n < N) v = {a.lse_in[(long long)bhx * N + n] * LOG2E, dl};
                *reinterpret_cast<float2*>(ld + 2 * n) = v;
            }
        }
    };
    float kf[16 * DT], vf[16 * DT];
    auto load_kv_rows = [&](int bhx) {                                // this lane's K and V rows (B-operand fragments) of head bhx
        const int bx = bhx / a.H, hx = bhx - bx * a.H;
        const int key = jt * 32 + l31;
        const bool ok = key < N;
        bf16x8_t unused0[1], unused1[1];
        row_frags<DT, false>(a.k + bx * a.ksb + hx * a.ksh + (long long)(ok ? key : 0) * a.ksn, ok, hf, kf, unused0);
        row_frags<DT, false>(a.v + bx * a.vsb + hx * a.vsh + (long long)(ok ? key : 0) * a.vsn, ok, hf, vf, unused1);
    };

    int t = 0;
    int bh = blockIdx.x;
    KV_CLK_BEGIN()
    if (bh < nbh) {                                                    // prologue: the first head's images (all waves), table, K / V rows
        const int bi = bh / a.H, hi = bh - bi * a.H;
        kv_fill_glds<DT>(smem, a.q + bi * a.qsb + hi * a.qsh, a.qsn, rows, N, wave, NW, lane);
        kv_fill_glds<DT>(smem + BUF, a.d_o + bi * a.osb + hi * a.osh, a.osn, rows, N, wave, NW, lane);
        if (jt == nkt) table_rows(bh, ld_base, 0, NP);
        if (!loader) load_kv_rows(bh);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 16 * DT; ++e) {
            asm volatile("" : "+v"(kf[e]));
            asm volatile("" : "+v"(vf[e]));
        }
    }
    for (; bh < nbh; bh += gridDim.x, ++t) {
        const int bi = bh / a.H, hi = bh - bi * a.H;
        const int qb_i = (2 * t) % 3;                                  // buffer of Q_t; dO_t sits in (2t + 1) % 3, (2t + 2) % 3 is free
        float* Q_s = smem + qb_i * BUF;
        float* dO_s = smem + ((2 * t + 1) % 3) * BUF;
        const float* ld_s = ld_base + (t & 1) * (2 * NP);
        const int bhn = bh + gridDim.x;
        const bool more = bhn < nbh;
        // ---- head boundary: what the loader fetched during head t - 1 has arrived (the loader waits for its own LDS-DMA; the
        //      computing waves waited for their K / V rows before their last stores, below, and leave those stores draining)
        if (loader) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        KV_CLK_PHASE(0)
        if (loader) {
            // one seventh of the next head's Q image and table per query-tile interval, then -- once the computing waves have
            // all left this tile's Q rows -- the dO rows of the next head into them
            const int bn = more ? bhn / a.H : 0, hn = more ? bhn - bn * a.H : 0;
            float* Qn = smem + ((2 * t + 2) % 3) * BUF;
            float* ldn = ld_base + ((t + 1) & 1) * (2 * NP);
            const int qchunk = (npieces + nkt - 1) / nkt;
            for (int qt = 0; qt < nkt; ++qt) {
                if (more && jt == nkt) {
                    const int p0 = qt * qchunk, p1 = (p0 + qchunk < npieces) ? p0 + qchunk : npieces;
                    kv_fill_glds_range<DT>(Qn, a.q + bn * a.qsb + hn * a.qsh, a.qsn, N, p0, p1, lane);
                    table_rows(bhn, ldn, qt * 32, qt * 32 + 32);
                }
                if (more && jt == nkt) {                               // every computing wave is done with the Q rows of tile qt?
                    const unsigned want = (unsigned)nkt * (unsigned)(t + 1);
                    while (*reinterpret_cast<volatile unsigned*>(done + qt) < want) __builtin_amdgcn_s_sleep(16);
                }
                if (more && jt == nkt) {
                    const int p0 = qt * PPT, p1 = (p0 + PPT < npieces) ? p0 + PPT : npieces;
                    kv_fill_glds_range<DT>(Q_s, a.d_o + bn * a.osb + hn * a.osh, a.osn, N, p0, p1, lane);
                }
            }
            continue;                                                  // wave-uniform: a loader has no key tile
        }

        const int key = jt * 32 + l31;
        const bool key_ok = key < N;
        f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dkacc[dt][r] = 0.0f;
                dvacc[dt][r] = 0.0f;
            }
        // dS[q][key] rows of this head and key tile: a wave-UNIFORM base (scalar registers) + one per-lane offset; written
        // as 16 per-lane 64-bit row pointers the compiler keeps 32 vector registers for them and spills inside the tile loop
        float* dsb = a.ds + ((long long)bh * NP) * NP + jt * 32;
        const int ds_lane = 4 * hf * NP + l31;
        for (int qt = 0; qt < nkt; ++qt) {
            const float* qrp = Q_s + qt * 32 * D;
            const float* drp = dO_s + qt * 32 * D;
            // ---- S[q][key] = Q.K^T and dP[q][key] = dO.V^T, two independent accumulation chains interleaved
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.0f;
                pacc[r] = 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f32x4 qa = *reinterpret_cast<const f32x4*>(qrp + koff[g]);
                const f32x4 da = *reinterpret_cast<const f32x4*>(drp + koff[g]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[e], kf[4 * g + e], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(da[e], vf[4 * g + e], pacc, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (qt == nkt - 1 && more) load_kv_rows(bhn);             // the fragment registers are dead: the next head's rows land under the rest of this tile
            KV_CLK_PHASE(2)
            const int nq = (qt == nkt - 1) ? nq_last : 4;
            // second products' operands: dO^T / Q^T columns of query rows qt*32 + 8q + 2p + {0, 1} + 4hf -- half quads (two
            // k-steps) read one half quad ahead into a second register set (a full quad ahead spills)
            float da2[2][2][DT], qa2[2][2][DT];
            auto read_half = [&](int hq, int b) {          // hq = 2*q + p
                const int q = hq >> 1, e0 = 2 * (hq & 1);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const int off = voff[e0 + e] + (8 * q + e0 + e) * D + 32 * (dt ^ (q & 1));
                        da2[b][e][dt] = drp[off];
                        qa2[b][e][dt] = qrp[off];
                    }
            };
            read_half(0, 0);
            // ---- p = exp(s*scale - lse), ds = p * scale * (dp - delta)        (utils.py:278-287)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = qt * 32 + kv_acc_row(r, hf);
                const float2 ld = *reinterpret_cast<const float2*>(ld_s + 2 * qrow);
                float p = __builtin_amdgcn_exp2f(sacc[r] * sc2 - ld.x);      // v_exp_f32: exp2(-inf) = 0 on pad rows
                if (!key_ok || (a.causal && key > qrow) || qrow >= N) p = 0.0f;      // (rows past the image hold another buffer's bits)
                sacc[r] = p;
                pacc[r] = p == 0.0f ? 0.0f : p * a.scale * (pacc[r] - ld.y);
            }
            KV_CLK_PHASE(3)
            {       // dS[q][key] for the dQ kernel: row = q (register), 32 consecutive keys per lane half
                float* dsp = dsb + (long long)(qt * 32) * NP;                                     // uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) (dsp + ((r & 3) + 8 * (r >> 2)) * NP)[ds_lane] = pacc[r];      // row kv_acc_row(r, hf)
            }
            KV_CLK_PHASE(4)
            // ---- dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]: k-step r <-> query row
            //      qt*32 + kv_acc_row(r, hf); the ragged last query tile contributes only its valid rows (p = ds = 0 elsewhere)
#pragma unroll
            for (int hq = 0; hq < 8; ++hq) {
                if (hq + 1 < 8) read_half(hq + 1, (hq + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                if ((hq >> 1) < nq) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(da2[hq & 1][e][dt], sacc[2 * hq + e], dvacc[dt], 0, 0, 0);
                            dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa2[hq & 1][e][dt], pacc[2 * hq + e], dkacc[dt], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            KV_CLK_PHASE(5)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // my LDS reads of this tile's Q rows have returned:
            if (lane == 0) __hip_atomic_fetch_add(done + qt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // tell the loader
            KV_CLK_PHASE(6)
        }
        if (more) {
            // the next head's K / V rows (requested under the last first products): an empty statement "using" the registers
            // makes hipcc place ITS wait for those loads here, in front of the dK / dV stores -- a counted wait that leaves the
            // dS stores draining; the head boundary then waits for nothing
#pragma unroll
            for (int e = 0; e < 16 * DT; ++e) {
                asm volatile("" : "+v"(kf[e]));
                asm volatile("" : "+v"(vf[e]));
            }
        }
        if (key_ok) {
            store_acc_rows<DT>(a.dk + bi * a.ksb + hi * a.ksh + (long long)key * a.ksn, dkacc, hf, 1.0f);
            store_acc_rows<DT>(a.dv + bi * a.vsb + hi * a.vsh + (long long)key * a.vsn, dvacc, hf, 1.0f);
        }
        KV_CLK_PHASE(7)
    }
    KV_CLK_END()
}

// dQ from the stored dS: dQ^T[d][q] = sum_key K^T[d][key] dS[q][key].  K in LDS (A operand, lanes walk d: one ds_read2_b32
// for the two d tiles of a key); the B operand is 16 consecutive keys of the lane's own dS row per key tile (lane half h
// takes keys 16h + s: float4 loads one key tile ahead).  Fragments are read one quad of k-steps ahead; the ragged last
// key tile contributes only its valid keys (the dS of the others was stored as exact zeros).
template <int DT, int NKT>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq3_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 32 * DT, KS = kv_pad4(D), NTHR = 512, NW = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int N = a.N, nkt = a.nkt, NP = nkt * 32, nbh = a.B * a.H;
    float* K_s = smem;                    // [NP][KS]
    for (int bh = blockIdx.x; bh < nbh; bh += gridDim.x) {       // persistent over (batch, head) pairs
    const int bi = bh / a.H, hi = bh - bi * a.H;
    float* dqb = a.dq + bi * a.qsb + hi * a.qsh;
    __syncthreads();
    fill_rows_f32<DT, NTHR>(K_s, a.k + bi * a.ksb + hi * a.ksh, a.ksn, NP, N, tid);
    __syncthreads();
    const int nlast = N - (nkt - 1) * 32;                         // valid keys of the last tile (1..32)
    const int nq_last = ((nlast < 16 ? nlast : 16) + 3) >> 2;     // quads of k-steps (lane half 0 holds keys 0..15 of a tile)
    for (int qt = wave; qt < nkt; qt += NW) {
        const int qrow = qt * 32 + l31;
        const float* dsrow = a.ds + ((long long)bh * NP + qrow) * NP + 16 * hf;
        f32x16 dqacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.0f;
        f32x4 nxt[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) nxt[e] = *reinterpret_cast<const f32x4*>(dsrow + 4 * e);
        const float* kcp = K_s + (16 * hf) * KS + l31;
        float ka[2][4][DT];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) ka[0][e][dt] = kcp[e * KS + dt * 32];
        for (int j = 0; j < nkt; ++j) {
            float dsv[16];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) dsv[4 * e + c] = nxt[e][c];
            if (j + 1 < nkt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) nxt[e] = *reinterpret_cast<const f32x4*>(dsrow + (j + 1) * 32 + 4 * e);
            }
            const int nq = (j == nkt - 1) ? nq_last : 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nq) {
                    // next quad: keys 4(q+1) .. of this tile, or the first quad of the next tile (always inside the padded image)
                    const int nrow = (q + 1 < 4) ? j * 32 + 4 * (q + 1) : (j + 1) * 32;
                    const bool more = (q + 1 < 4) || (j + 1 < nkt);
                    if (more) {       // wave-uniform; no select on the loaded value (it would force a wait right here)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int dt = 0; dt < DT; ++dt) ka[(q + 1) & 1][e][dt] = kcp[(nrow + e) * KS + dt * 32];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
                            dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[q & 1][e][dt], dsv[4 * q + e], dqacc[dt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (qrow < N) store_acc_rows<DT>(dqb + (long long)qrow * a.qsn, dqacc, hf, 1.0f);
    }
    }   // heads
}

int check_desc(const kanvit_attn_desc* d, const char* who) {
    if (!d) return kv_fail(KANVIT_EINVAL, "%s: null descriptor", who);
    if (d->B < 0 || d->H < 1 || d->N < 1 || d->D < 1) return kv_fail(KANVIT_EINVAL, "%s: bad sizes", who);
    if (d->N > KANVIT_ATTN_MAX_N) return kv_fail(KANVIT_EINVAL, "%s: N=%d exceeds %d (one head must fit a workgroup)", who, d->N, KANVIT_ATTN_MAX_N);
    if (d->D > KANVIT_ATTN_MAX_D || (d->D & 1)) return kv_fail(KANVIT_EINVAL, "%s: D=%d must be even and <= %d", who, d->D, KANVIT_ATTN_MAX_D);
    if ((long long)d->B * d->H > 0x7fffffffLL) return kv_fail(KANVIT_EINVAL, "%s: B*H too large", who);
    const int ks = (d->D <= 32 ? 32 : 64) + 1, np = (d->N + 31) / 32 * 32;
    const size_t lds = sizeof(float) * ((size_t)2 * np * ks + 2 * (size_t)np + (size_t)4 * 32 * ks);
    if (lds > 160 * 1024)
        return kv_fail(KANVIT_EINVAL, "%s: N=%d with D=%d needs %zu bytes of LDS (> 160 KiB): one head must fit a CU", who,
                       d->N, d->D, lds);
    return 0;
}

AttnArgs make_args(const kanvit_attn_desc* d) {
    AttnArgs a{};
    a.B = d->B; a.H = d->H; a.N = d->N; a.D = d->D; a.causal = d->causal; a.scale = d->scale;
    a.nkt = (d->N + 31) / 32;
    a.qsb = d->q_stride_b; a.qsh = d->q_stride_h; a.qsn = d->q_stride_n;
    a.ksb = d->k_stride_b; a.ksh = d->k_stride_h; a.ksn = d->k_stride_n;
    a.vsb = d->v_stride_b; a.vsh = d->v_stride_h; a.vsn = d->v_stride_n;
    a.osb = d->o_stride_b; a.osh = d->o_stride_h; a.osn = d->o_stride_n;
    const long long strides[] = {a.qsb, a.qsh, a.qsn, a.ksb, a.ksh, a.ksn, a.vsb, a.vsh, a.vsn, a.osb, a.osh, a.osn};
    a.vec = (d->D % 4 == 0);
    for (long long sd : strides) a.vec = a.vec && (sd % 4 == 0);
    return a;
}

template <int DT, int NKT, bool BF>
int launch_fwd(const AttnArgs& a, hipStream_t st) {
    constexpr int KS = 32 * DT + 1;
    const size_t lds = sizeof(float) * ((size_t)2 * a.nkt * 32 * KS + (size_t)4 * 32 * KS);
    KV_ALLOW_LDS(160 * 1024, attn_fwd_kernel<DT, NKT, BF>);
    hipLaunchKernelGGL((attn_fwd_kernel<DT, NKT, BF>), dim3((unsigned)(a.B * a.H)), dim3(ATHR), lds, st, a);
    KV_LAUNCH_CHECK("attn_fwd_kernel");
    return 0;
}

template <int DT, int NKT, bool BF>
int launch_fwd2(const AttnArgs& a, hipStream_t st) {
    constexpr int D = 32 * DT;
    const int NP = a.nkt * 32;
    const size_t lds = BF ? sizeof(unsigned short) * ((size_t)NP * (D + 8) + (size_t)D * (NP + 8))
                          : sizeof(float) * (size_t)2 * NP * (D + 1);
    KV_ALLOW_LDS(160 * 1024, attn_fwd2_kernel<DT, NKT, BF>);
    hipLaunchKernelGGL((attn_fwd2_kernel<DT, NKT, BF>), dim3((unsigned)(a.B * a.H)), dim3(BF ? 256 : 512), lds, st, a);
    KV_LAUNCH_CHECK("attn_fwd2_kernel");
    return 0;
}

template <int DT, int NKT>
int launch_fwd3(const AttnArgs& a, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)2 * a.nkt * 32 * kv_pad4(32 * DT);
    KV_ALLOW_LDS(160 * 1024, attn_fwd3_kernel<DT, NKT>);
    hipLaunchKernelGGL((attn_fwd3_kernel<DT, NKT>), dim3((unsigned)kv_persistent_grid(a.B * a.H, lds)), dim3(KV_A3_THREADS), lds, st, a);
    KV_LAUNCH_CHECK("attn_fwd3_kernel");
    return 0;
}

template <int NKT>
int launch_fwd4(const AttnArgs& a, hipStream_t st) {
    const size_t lds = kv_a4_lds(a.N, 64);
    KV_ALLOW_LDS(160 * 1024, (attn_fwd4_kernel<2, NKT>));
    const int nbh = a.B * a.H;
    const int g = kv_config().attn_grid > 0 ? kv_config().attn_grid : KV_N_CU;       // one work-group per CU (three images fill its LDS)
    hipLaunchKernelGGL((attn_fwd4_kernel<2, NKT>), dim3((unsigned)(nbh < g ? nbh : g)), dim3(KV_A3_THREADS), lds, st, a);
    KV_LAUNCH_CHECK("attn_fwd4_kernel");
    return 0;
}

template <int DT, bool BF>
int dispatch_fwd(const AttnArgs& a, hipStream_t st) {
    if constexpr (!BF && DT == 2) {
        // 16-row tiles (round 4, csrc/attention16.hip): D = 64, 64 < N <= 204
        if (!kv_config().attn_v1 && !kv_config().attn_v2 && !kv_config().attn_v3) {
            const int rc = kv_attn16_fwd(a, st);
            if (rc <= 0) return rc;
        }
        // fourth form: D = 64, 64 < N (below, a head is a few tiles and the third form's single fill is cheap), three swizzled
        // images within the 160 KiB, 16-byte aligned rows for the LDS-DMA fills
        if (a.vec && a.D == 64 && a.nkt >= 3 && a.nkt <= 7 && kv_a4_lds(a.N, 64) <= 160 * 1024 && !kv_config().attn_v1 && !kv_config().attn_v2 &&
            !kv_config().attn_v3 && (((uintptr_t)a.out | (uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) % 16 == 0)) {
            if (a.nkt <= 4) return launch_fwd4<4>(a, st);
            return launch_fwd4<7>(a, st);
        }
    }
    if constexpr (BF && DT == 2) {      // bf16 matrix cores on the 16-row-tile structure (round 4): N = 193 .. 204
        if (!kv_config().attn_v1 && !kv_config().attn_v2 && !kv_config().attn_v3) {
            const int rc = kv_attn16_fwd_bf16(a, st);
            if (rc <= 0) return rc;
        }
    }
    if constexpr (!BF) {
        if (a.vec && a.D == 32 * DT && ((uintptr_t)a.out % 16 == 0) && !kv_config().attn_v1 && !kv_config().attn_v2) {
            if (a.nkt <= 1) return launch_fwd3<DT, 1>(a, st);
            if (a.nkt <= 2) return launch_fwd3<DT, 2>(a, st);
            if (a.nkt <= 4) return launch_fwd3<DT, 4>(a, st);
            if (a.nkt <= 7) return launch_fwd3<DT, 7>(a, st);
        }
    }
    if (a.vec && a.D == 32 * DT && ((uintptr_t)a.out % 16 == 0) && !kv_config().attn_v1) {
        if (a.nkt <= 1) return launch_fwd2<DT, 1, BF>(a, st);
        if (a.nkt <= 2) return launch_fwd2<DT, 2, BF>(a, st);
        if (a.nkt <= 4) return launch_fwd2<DT, 4, BF>(a, st);
        if (a.nkt <= 7) return launch_fwd2<DT, 7, BF>(a, st);
    }
    if (a.nkt <= 1) return launch_fwd<DT, 1, BF>(a, st);
    if (a.nkt <= 2) return launch_fwd<DT, 2, BF>(a, st);
    if (a.nkt <= 4) return launch_fwd<DT, 4, BF>(a, st);
    if (a.nkt <= 7) return launch_fwd<DT, 7, BF>(a, st);
    return launch_fwd<DT, 8, BF>(a, st);
}

template <int DT, bool BF>
int launch_bwd2(const AttnArgs& a, hipStream_t st) {
    constexpr int D = 32 * DT;
    const int NP = a.nkt * 32;
    const size_t row = sizeof(unsigned short) * (size_t)NP * (D + 8), tr = sizeof(unsigned short) * (size_t)D * (NP + 8);
    const size_t f32img = sizeof(float) * (size_t)NP * (D + 1);
    const size_t lds_kv = (BF ? 2 * row + 2 * tr : 2 * f32img) + sizeof(float) * 2 * (size_t)NP;
    const size_t lds_q = BF ? 2 * row + tr : 2 * f32img;
    KV_ALLOW_LDS(160 * 1024, (attn_bwd_kv2_kernel<DT, BF, false>));
    KV_ALLOW_LDS(160 * 1024, (attn_bwd_q2_kernel<DT, BF>));
    if constexpr (!BF) {
        if constexpr (DT == 2) {
            // fourth form of the key-stationary kernel (LDS-DMA ring, loader wave): D = 64, 64 < N, a wave without a key tile
            if (a.ds && a.third && a.nkt >= 3 && a.nkt <= 7 && kv_b4_lds(a.N, D) <= 160 * 1024 && !kv_config().attn_v3 &&
                (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.o | (uintptr_t)a.d_o) % 16 == 0)) {
                const size_t lds4 = kv_b4_lds(a.N, D), ldsq = sizeof(float) * (size_t)NP * kv_pad4(D);
                const int nbh = a.B * a.H;
                const int g4 = kv_config().attn_grid > 0 ? kv_config().attn_grid : KV_N_CU;
                if (a.nkt <= 4) {
                    KV_ALLOW_LDS(160 * 1024, (attn_bwd_kv4_kernel<2>));
                    KV_ALLOW_LDS(160 * 1024, (attn_bwd_dq3_kernel<DT, 4>));
                    hipLaunchKernelGGL((attn_bwd_kv4_kernel<2>), dim3((unsigned)(nbh < g4 ? nbh : g4)), dim3(512), lds4, st, a);
                    KV_LAUNCH_CHECK("attn_bwd_kv4_kernel");
                    hipLaunchKernelGGL((attn_bwd_dq3_kernel<DT, 4>), dim3((unsigned)kv_persistent_grid(nbh, ldsq)), dim3(512), ldsq, st, a);
                } else {
                    KV_ALLOW_LDS(160 * 1024, (attn_bwd_kv4_kernel<2>));
                    KV_ALLOW_LDS(160 * 1024, (attn_bwd_dq3_kernel<DT, 8>));
                    hipLaunchKernelGGL((attn_bwd_kv4_kernel<2>), dim3((unsigned)(nbh < g4 ? nbh : g4)), dim3(512), lds4, st, a);
                    KV_LAUNCH_CHECK("attn_bwd_kv4_kernel");
                    hipLaunchKernelGGL((attn_bwd_dq3_kernel<DT, 8>), dim3((unsigned)kv_persistent_grid(nbh, ldsq)), dim3(512), ldsq, st, a);
                }
                KV_LAUNCH_CHECK("attn_bwd_dq3_kernel");
                return 0;
            }
        }
        if (a.ds && a.third) {      // third form: same dS hand-off, pipelined LDS reads, rowsum(dO*O) in the prologue
            const size_t lds3 = sizeof(float) * ((size_t)2 * NP * kv_pad4(D) + 2 * (size_t)NP);
            const size_t ldsq = sizeof(float) * (size_t)NP * kv_pad4(D);
            const int nbh = a.B * a.H;
            if (a.nkt <= 2) {
                KV_ALLOW_LDS(160 * 1024, (attn_bwd_kv3_kernel<DT, 2>));
                KV_ALLOW_LDS(160 * 1024, (attn_bwd_dq3_kernel<DT, 2>));
                hipLaunchKernelGGL((attn_bwd_kv3_kernel<DT, 2>), dim3((unsigned)kv_persistent_grid(nbh, lds3)), dim3(512), lds3, st, a);
                KV_LAUNCH_CHECK("attn_bwd_kv3_kernel");
                hipLaunchKernelGGL((attn_bwd_dq3_kernel<DT, 2>), dim3((unsigned)kv_persistent_grid(nbh, ldsq)), dim3(512), ldsq, st, a);
                KV_LAUNCH_CHECK("attn_bwd_dq3_kernel");
            } else if (a.nkt <= 4) {
                KV_ALLOW_LDS(160 * 1024, (attn_bwd_kv3_kernel<DT, 4>));
                KV_ALLOW_LDS(160 * 1024, (attn_bwd_dq3_kernel<DT, 4>));
                hipLaunchKernelGGL((attn_bwd_kv3_kernel<DT, 4>), dim3((unsigned)kv_persistent_grid(nbh, lds3)), dim3(512), lds3, st, a);
                KV_LAUNCH_CHECK("attn_bwd_kv3_kernel");
                hipLaunchKernelGGL((attn_bwd_dq3_kernel<DT, 4>), dim3((unsigned)kv_persistent_grid(nbh, ldsq)), dim3(512), ldsq, st, a);
                KV_LAUNCH_CHECK("attn_bwd_dq3_kernel");
            } else {
                KV_ALLOW_LDS(160 * 1024, (attn_bwd_kv3_kernel<DT, 8>));
                KV_ALLOW_LDS(160 * 1024, (attn_bwd_dq3_kernel<DT, 8>));
                hipLaunchKernelGGL((attn_bwd_kv3_kernel<DT, 8>), dim3((unsigned)kv_persistent_grid(nbh, lds3)), dim3(512), lds3, st, a);
                KV_LAUNCH_CHECK("attn_bwd_kv3_kernel");
                hipLaunchKernelGGL((attn_bwd_dq3_kernel<DT, 8>), dim3((unsigned)kv_persistent_grid(nbh, ldsq)), dim3(512), ldsq, st, a);
                KV_LAUNCH_CHECK("attn_bwd_dq3_kernel");
            }
            return 0;
        }
        if (a.ds) {      // dS spill: the key-stationary kernel stores dS, dQ is one plain product (5 MFMA products instead of 7)
            KV_ALLOW_LDS(160 * 1024, (attn_bwd_kv2_kernel<DT, false, true>));
            KV_ALLOW_LDS(160 * 1024, (attn_bwd_dq_kernel<DT>));
            hipLaunchKernelGGL((attn_bwd_kv2_kernel<DT, false, true>), dim3((unsigned)(a.B * a.H)), dim3(512), lds_kv, st, a);
            KV_LAUNCH_CHECK("attn_bwd_kv2_kernel");
            hipLaunchKernelGGL((attn_bwd_dq_kernel<DT>), dim3((unsigned)(a.B * a.H)), dim3(512), f32img, st, a);
            KV_LAUNCH_CHECK("attn_bwd_dq_kernel");
            return 0;
        }
    }
    if constexpr (BF) {
        if (a.ds && a.nkt <= 8) {      // bf16 dS hand-off: the key-stationary kernel stores dS as bf16, dQ is one product (no recomputation)
            KV_ALLOW_LDS(160 * 1024, (attn_bwd_kv2_kernel<DT, true, true>));
            hipLaunchKernelGGL((attn_bwd_kv2_kernel<DT, true, true>), dim3((unsigned)(a.B * a.H)), dim3(512), lds_kv, st, a);
            KV_LAUNCH_CHECK("attn_bwd_kv2_kernel");
            if (a.nkt <= 4) hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<DT, 4>), dim3((unsigned)(a.B * a.H)), dim3(512), tr, st, a);
            else hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<DT, 8>), dim3((unsigned)(a.B * a.H)), dim3(512), tr, st, a);
            KV_LAUNCH_CHECK("attn_bwd_dq_bf16_kernel");
            return 0;
        }
    }
    hipLaunchKernelGGL((attn_bwd_kv2_kernel<DT, BF, false>), dim3((unsigned)(a.B * a.H)), dim3(512), lds_kv, st, a);
    KV_LAUNCH_CHECK("attn_bwd_kv2_kernel");
    hipLaunchKernelGGL((attn_bwd_q2_kernel<DT, BF>), dim3((unsigned)(a.B * a.H)), dim3(512), lds_q, st, a);
    KV_LAUNCH_CHECK("attn_bwd_q2_kernel");
    return 0;
}

template <int DT, bool BF>
int launch_bwd(const AttnArgs& a, hipStream_t st) {
    if (a.vec && a.D == 32 * DT && !kv_config().attn_v1 &&
        (((uintptr_t)a.dq | (uintptr_t)a.dk | (uintptr_t)a.dv) % 16 == 0)) {
        const int NPc = a.nkt * 32;
        const size_t need = BF ? sizeof(unsigned short) * ((size_t)2 * NPc * (32 * DT + 8) + (size_t)2 * 32 * DT * (NPc + 8)) + 8 * (size_t)NPc
                               : sizeof(float) * ((size_t)2 * NPc * (32 * DT + 1) + 2 * (size_t)NPc);
        if (need <= 160 * 1024) return launch_bwd2<DT, BF>(a, st);
    }
    constexpr int KS = 32 * DT + 1;
    const int NP = a.nkt * 32;
    const size_t lds_kv = sizeof(float) * ((size_t)2 * NP * KS + 2 * (size_t)NP + (size_t)4 * 32 * KS);
    const size_t lds_q = sizeof(float) * ((size_t)2 * NP * KS + (size_t)4 * 32 * KS);
    KV_ALLOW_LDS(160 * 1024, (attn_bwd_kv_kernel<DT, BF>));
    KV_ALLOW_LDS(160 * 1024, (attn_bwd_q_kernel<DT, BF>));
    hipLaunchKernelGGL((attn_bwd_kv_kernel<DT, BF>), dim3((unsigned)(a.B * a.H)), dim3(ATHR), lds_kv, st, a);
    KV_LAUNCH_CHECK("attn_bwd_kv_kernel");
    hipLaunchKernelGGL((attn_bwd_q_kernel<DT, BF>), dim3((unsigned)(a.B * a.H)), dim3(ATHR), lds_q, st, a);
    KV_LAUNCH_CHECK("attn_bwd_q_kernel");
    return 0;
}

// =============================================================================================
// Small heads (N <= 32, D <= 32; exact fp32): ONE WAVE per (batch, head), 4 heads per work-group, no barriers.
// train.py's own default geometry is this case (32x32 images, 4x4 patches -> N = 17; d = 64, 8 heads -> D = 8): 1024 heads of a
// 17x17x8 problem, for which the general kernels (one 512-thread work-group per head, 32-wide padded tiles through LDS) spend
// 10 + 40 us of launch and fill latency per block.  Here a head is one 32x32 MFMA tile:
//   S^T[key][q] = K Q^T           A = the lane's own K row, B = the lane's own Q row (half h takes d = h*ceil(D/2) + s)
//   softmax over keys = over the 16 accumulator registers and the partner lane (l ^ 32), in registers
//   O^T[d][q]  = V^T P^T          the contraction index (key) is the accumulator REGISTER index: P feeds the matrix pipe as is
// Backward (same wave, same orientation): S and dP^T = V dO^T recomputed, dS in registers, dQ^T = K^T dS^T straight from
// registers; dV^T = dO^T P and dK^T = Q^T dS contract over q (the lane index), so P^T and dS^T take one trip through a
// wave-private LDS tile.  Every output row is the storing lane's own row.
// =============================================================================================
__device__ __forceinline__ void attn_small_store_row(float* __restrict__ rowp, const f32x16& acc, int hf, int D, float mul) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = kv_acc_row(r, hf);
        if (d < D) rowp[d] = acc[r] * mul;
    }
}

__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const AttnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= a.B * a.H) return;
    const int bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, D = a.D, DHc = (D + 1) / 2;
    const bool row_ok = l31 < N;
    const float* qrow = a.q + bi * a.qsb + hi * a.qsh + (long long)(row_ok ? l31 : 0) * a.qsn;
    const float* krow = a.k + bi * a.ksb + hi * a.ksh + (long long)(row_ok ? l31 : 0) * a.ksn;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    float vt[16];                                  // V[key(r, h)][d = this lane]: the A operand of the second product
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = kv_acc_row(r, hf);
        vt[r] = (key < N && l31 < D) ? vb[(long long)key * a.vsn + l31] : 0.0f;
    }
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
    for (int s = 0; s < DHc; ++s) {
        const int d = hf * DHc + s;
        const bool ok = row_ok && d < D;
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? krow[d] : 0.0f, ok ? qrow[d] : 0.0f, sacc, 0, 0, 0);
    }
    const float sc2 = a.scale * LOG2E;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = kv_acc_row(r, hf);
        if (key >= N || (a.causal && key > l31)) sacc[r] = -INFINITY;
        mx = fmaxf(mx, sacc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = exp2f((sacc[r] - mx) * sc2);       // masked entries: exp2(-inf) = 0
        sacc[r] = p;
        sum += p;
    }
    sum += __shfl_xor(sum, 32);
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[r], sacc[r], oacc, 0, 0, 0);
    if (row_ok) {
        attn_small_store_row(a.out + bi * a.osb + hi * a.osh + (long long)l31 * a.osn, oacc, hf, D, 1.0f / sum);
        if (hf == 0 && a.lse) a.lse[(long long)bh * N + l31] = mx * a.scale + logf(sum);
    }
}

__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const AttnArgs a) {
    __shared__ float tiles[4][2][32 * 33];         // per wave: P^T[key][q] and dS^T[key][q]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= a.B * a.H) return;
    const int bi = bh / a.H, hi = bh - bi * a.H;
    const int N = a.N, D = a.D, DHc = (D + 1) / 2;
    const bool row_ok = l31 < N;
    const long long r0 = row_ok ? l31 : 0;
    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    const float* ob = a.o + bi * a.osb + hi * a.osh;
    const float* dob = a.d_o + bi * a.osb + hi * a.osh;
    // gathers for the three register/LDS-operand products: lane = feature d, 16 rows each
    float kt[16], dot[16], qt[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = kv_acc_row(r, hf);                                   // dQ^T: contraction over keys in register order
        kt[r] = (key < N && l31 < D) ? kb[(long long)key * a.ksn + l31] : 0.0f;
        const int qq = 2 * r + hf;                                           // dV^T / dK^T: contraction over q, k-step r, half h
        const bool ok = qq < N && l31 < D;
        dot[r] = ok ? dob[(long long)qq * a.osn + l31] : 0.0f;
        qt[r] = ok ? qb[(long long)qq * a.qsn + l31] : 0.0f;
    }
    f32x16 sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sacc[r] = 0.0f;
        pacc[r] = 0.0f;
    }
    float dl = 0.0f;
    for (int s = 0; s < DHc; ++s) {
        const int d = hf * DHc + s;
        const bool ok = row_ok && d < D;
        const float kv = ok ? kb[r0 * a.ksn + d] : 0.0f, qv = ok ? qb[r0 * a.qsn + d] : 0.0f;
        const float vv = ok ? vb[r0 * a.vsn + d] : 0.0f, dv = ok ? dob[r0 * a.osn + d] : 0.0f;
        const float ov = ok ? ob[r0 * a.osn + d] : 0.0f;
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qv, sacc, 0, 0, 0);     // S^T[key][q]
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, dv, pacc, 0, 0, 0);     // dP^T[key][q]
        dl += dv * ov;                                                           // rowsum(dO * O) of this lane's query (utils.py:286)
    }
    dl += __shfl_xor(dl, 32);
    const float sc2 = a.scale * LOG2E;
    const float lse2 = row_ok ? a.lse_in[(long long)bh * N + l31] * LOG2E : INFINITY;      // pad queries: p = 0
    float* pt = tiles[wave][0];
    float* dst = tiles[wave][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = kv_acc_row(r, hf);
        float p = exp2f(sacc[r] * sc2 - lse2);
        if (key >= N || (a.causal && key > l31)) p = 0.0f;
        const float ds = p * a.scale * (pacc[r] - dl);                           // utils.py:278-287
        sacc[r] = p;
        pacc[r] = ds;
        pt[key * 33 + l31] = p;
        dst[key * 33 + l31] = ds;
    }
    f32x16 dqacc, dkacc, dvacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dqacc[r] = 0.0f;
        dkacc[r] = 0.0f;
        dvacc[r] = 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[r], pacc[r], dqacc, 0, 0, 0);    // dQ^T[d][q]
    __builtin_amdgcn_wave_barrier();               // the tile is wave-private: LDS executes this wave's writes before its reads
#pragma unroll
    for (int r = 0; r < 16; ++r) {                 // B[k = q = 2r + h][col = key = lane]
        dvacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dot[r], pt[l31 * 33 + 2 * r + hf], dvacc, 0, 0, 0);          // dV^T[d][key]
        dkacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qt[r], dst[l31 * 33 + 2 * r + hf], dkacc, 0, 0, 0);          // dK^T[d][key]
    }
    if (row_ok) {
        attn_small_store_row(a.dq + bi * a.qsb + hi * a.qsh + (long long)l31 * a.qsn, dqacc, hf, D, 1.0f);
        attn_small_store_row(a.dk + bi * a.ksb + hi * a.ksh + (long long)l31 * a.ksn, dkacc, hf, D, 1.0f);
        attn_small_store_row(a.dv + bi * a.vsb + hi * a.vsh + (long long)l31 * a.vsn, dvacc, hf, D, 1.0f);
    }
}

bool attn_small_ok(const kanvit_attn_desc* d) { return d->N <= 32 && d->D <= 32 && !kv_config().attn_v1 && !kv_config().attn_v2; }

}  // namespace

extern "C" {

#ifdef KANVIT_CLOCK_PROBE
__attribute__((visibility("default"))) int kanvit_debug_clock(unsigned long long* out) {      // diagnostic build only: {shader cycles, 100 MHz ticks} of the stamped work-group
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kv_clk), 10 * sizeof(unsigned long long)) == hipSuccess ? 0 : -5;
}
#endif

int kanvit_attn_fwd(const kanvit_attn_desc* d, const float* q, const float* k, const float* v, float* o, float* lse,
                    void* stream) {
    if (int rc = check_desc(d, "kanvit_attn_fwd")) return rc;
    if (!q || !k || !v || !o) return kv_fail(KANVIT_EINVAL, "kanvit_attn_fwd: null q/k/v/o");
    if (d->B == 0) return 0;
    AttnArgs a = make_args(d);
    a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse;
    a.vec = a.vec && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (attn_small_ok(d)) {           // one wave per head (exact fp32; the bf16 flag allows, never requires, bf16 products)
        hipLaunchKernelGGL(attn_small_fwd_kernel, dim3((unsigned)((d->B * d->H + 3) / 4)), dim3(256), 0, st, a);
        KV_LAUNCH_CHECK("attn_small_fwd_kernel");
        return 0;
    }
    if ((d->flags & KANVIT_FLAG_BF16_MFMA) && d->D % 16 == 0 && !kv_config().no_bf16)
        return d->D <= 32 ? dispatch_fwd<1, true>(a, st) : dispatch_fwd<2, true>(a, st);
    return d->D <= 32 ? dispatch_fwd<1, false>(a, st) : dispatch_fwd<2, false>(a, st);
}

// rowsum(dO*O) [B*H*N] (rounded up to 16 bytes), then -- exact fp32 path with D in {32, 64} only -- dS [B*H][NP][NP]
static size_t attn_delta_bytes(const kanvit_attn_desc* d) { return (sizeof(float) * (size_t)d->B * d->H * d->N + 15) / 16 * 16; }
static bool attn_bf16_mode(const kanvit_attn_desc* d) { return (d->flags & KANVIT_FLAG_BF16_MFMA) && d->D % 16 == 0 && !kv_config().no_bf16; }
// dS hand-off from the key-stationary kernel to the dQ kernel: fp32 values on the exact path, bf16 values in bf16 mode (there the
// dQ product consumes them as a bf16 operand anyway; needs one query tile per wave of the 512-thread kernels: N <= 256)
static bool attn_ds_spill(const kanvit_attn_desc* d) {
    if (!(d->D == 32 || d->D == 64) || d->causal || kv_config().attn_no_ds || kv_config().attn_v1) return false;
    if (kv_attn16_bwd_ok(d)) return false;       // the 16-row-tile backward forms dQ in the same kernel: nothing crosses HBM
    return attn_bf16_mode(d) ? d->N <= 256 : true;
}
size_t kanvit_attn_bwd_workspace(const kanvit_attn_desc* d) {
    if (!d || d->B < 0 || d->H < 1 || d->N < 1) return 0;
    size_t n = attn_delta_bytes(d);
    if (attn_ds_spill(d)) {
        const size_t np = (size_t)(d->N + 31) / 32 * 32;
        n += (attn_bf16_mode(d) ? sizeof(unsigned short) : sizeof(float)) * (size_t)d->B * d->H * np * np;
    }
    return n;
}

/* dq/dk/dv are fully written (no accumulation into the outputs). */
int kanvit_attn_bwd(const kanvit_attn_desc* d, const float* q, const float* k, const float* v, const float* o,
                    const float* lse, const float* d_o, float* dq, float* dk, float* dv, void* workspace,
                    size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(d, "kanvit_attn_bwd")) return rc;
    if (!q || !k || !v || !o || !lse || !d_o || !dq || !dk || !dv)
        return kv_fail(KANVIT_EINVAL, "kanvit_attn_bwd: null argument");
    if (d->B == 0) return 0;
    if (!workspace || workspace_bytes < kanvit_attn_bwd_workspace(d))
        return kv_fail(KANVIT_ENOMEM, "kanvit_attn_bwd: workspace %zu bytes < required %zu", workspace_bytes,
                       kanvit_attn_bwd_workspace(d));
    float* delta_ws = (float*)workspace;
    AttnArgs a = make_args(d);
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse_in = lse; a.d_o = d_o;
    a.dq = dq; a.dk = dk; a.dv = dv; a.delta = delta_ws; a.delta_in = delta_ws;
    a.ds = attn_ds_spill(d) ? (float*)((char*)workspace + attn_delta_bytes(d)) : nullptr;
    a.vec = a.vec && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)d_o) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (attn_small_ok(d)) {
        hipLaunchKernelGGL(attn_small_bwd_kernel, dim3((unsigned)((d->B * d->H + 3) / 4)), dim3(256), 0, st, a);
        KV_LAUNCH_CHECK("attn_small_bwd_kernel");
        return 0;
    }
    if (kv_attn16_bwd_ok(d) && a.vec) {      // one kernel, five products (csrc/attention16.hip); a refusal (alignment) falls through to the older forms
        const int rc = kv_attn16_bwd(a, st);
        if (rc <= 0) return rc;
    }
    if (attn_bf16_mode(d) && a.vec && !kv_config().attn_v1 && !kv_config().attn_v2 && !kv_config().attn_v3 && !kv_config().attn_no_ds) {      // the same kernel on the bf16 matrix cores
        const int rc = kv_attn16_bwd_bf16(a, st);
        if (rc <= 0) return rc;
    }
    const long long rows = (long long)d->B * d->H * d->N;
    // the third-form fp32 kernels form rowsum(dO*O) themselves; every other path reads it from the workspace
    const size_t lds3 = sizeof(float) * ((size_t)2 * a.nkt * 32 * kv_pad4(d->D) + 2 * (size_t)a.nkt * 32);
    const bool third = a.ds && !attn_bf16_mode(d) && !kv_config().attn_v2 && !kv_config().attn_v1 && a.vec && (d->D == 32 || d->D == 64) &&
                       (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)o) % 16 == 0) && lds3 <= 160 * 1024;
    a.third = third ? 1 : 0;
    if (!third) {
        hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, a);
        KV_LAUNCH_CHECK("attn_delta_kernel");
    }
    if (attn_bf16_mode(d)) return d->D <= 32 ? launch_bwd<1, true>(a, st) : launch_bwd<2, true>(a, st);
    return d->D <= 32 ? launch_bwd<1, false>(a, st) : launch_bwd<2, false>(a, st);
}

}  // extern "C"
