// General attention core: independent query / key lengths and a boolean attend-mask (exact fp32).
//
// FlashAttentionFunction (utils.py:134-295) tiles over independent q / k lengths (utils.py:150-160, cross-attention through
// FlashAttention(context=...), attention.py:59-109) and takes a mask broadcastable to [b, h, q_len, k_len] (a [b, k_len]
// key-padding mask is viewed as [b, 1, 1, k_len], utils.py:156-157).  `causal` (key j > query i is dead, utils.py:178-190) is taken
// for k_len <= q_len only: with k_len > q_len the reference shifts the diagonal the wrong way (utils.py:169 SUBTRACTS qk_len_diff:
// query i sees keys j <= i - (k_len - q_len)), the first k_len - q_len queries have no visible key and its answer for them depends
// on the bucket sizes -- refused here (KANVIT_EINVAL) instead of imitated.  The ViT path never uses either (attention.hip's kernels: one length, no
// mask); these kernels cover the rest of the function's domain with the same tile orientation as attention.hip's first form:
//   forward       work-group = eight 32-query tiles of one (batch, head), one per wave; the keys are swept in LDS chunks of 128 rows
//                 (K, V images [128][KS]); S^T = K.Q^T with the key index in the accumulator register index -> softmax over the
//                 chunk in registers, running (max, sum) across chunks (utils.py:199-221) -> O^T = V^T.P^T from registers
//   backward      rowsum(dO*O) (attn_x_delta_kernel), then a key-stationary kernel (a wave owns a key tile: dK^T, dV^T in
//                 accumulators; Q, dO swept in LDS chunks) and a query-stationary one (dQ^T in accumulators; K, V swept in
//                 chunks): no atomics, bitwise reproducible
// ANY sequence length (the swept operand never has to fit the LDS): ops.attention also routes self-attention heads that do not fit
// the ViT kernels of attention.hip (N > 224 at D = 64, e.g. ViT-B/16 at 384 x 384: N = 577) here.
// A position is dead when key >= k_len, or causal and key > query, or the mask says so.  A query whose keys are ALL
// dead gets o = 0, lse = -FLT_MAX and zero gradients (the reference's clamp(min=EPSILON) row sum gives the same o = 0 and
// lse = log(1e-10) - FLT_MAX for a fully MASKED row).
// Limits (host-checked): D even and <= 64.
#include "kanvit_common.h"

#include <float.h>
#include <initializer_list>

namespace {

constexpr int XTHR = 512;        // 8 waves: two per SIMD (one LDS chunk of the swept operand serves eight 32-row tiles)
constexpr int XW = XTHR / 64;
constexpr float LOG2E_X = 1.4426950408889634f;

struct AttnXArgs {
    const float* q;
    const float* k;
    const float* v;
    const float* o;
    const float* lse_in;
    const float* d_o;
    const float* delta_in;
    const unsigned char* mask;      // nonzero = attend; element (b, h, i, j) at mask[b*msb + h*msh + i*msq + j*msk] (0 strides broadcast)
    float* out;
    float* lse;
    float* dq;
    float* dk;
    float* dv;
    float* delta;
    long long msb, msh, msq, msk;
    long long qsb, qsh, qsn, ksb, ksh, ksn, vsb, vsh, vsn, osb, osh, osn;
    int B, H, Nq, Nk, D, causal, nqt, nkt;
    int vec;          // rows are 16-byte aligned pieces (D % 4 == 0, strides % 4 == 0, aligned bases): tile fills use 16-byte loads
    float scale;
};

__device__ __forceinline__ bool x_dead(const AttnXArgs& a, const unsigned char* mrow_base, int qrow, int key) {
    if (key >= a.Nk || (a.causal && key > qrow)) return true;
    if (mrow_base) {
        const int qi = qrow < a.Nq ? qrow : a.Nq - 1;
        return mrow_base[(long long)qi * a.msq + (long long)key * a.msk] == 0;
    }
    return false;
}

// Dead positions of one 32x32 accumulator tile as a bit per accumulator register.  The mask bytes are read HERE, before the
// accumulators are touched: with the loads inside the select on the accumulator value, hipcc 7.2 reused the register holding the
// element for the loaded byte and every masked launch returned wrong numbers (an all-true mask changed the result).
// KEY_IN_REG: the register index runs over keys and the lane is the query (forward, dQ); else the register index runs over queries
// and the lane is the key (dK, dV).
template <bool KEY_IN_REG>
__device__ __forceinline__ unsigned x_dead_bits(const AttnXArgs& a, const unsigned char* mrow_base, int lane_index, int tile0, int hf) {
    unsigned bits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int reg_index = tile0 + kv_acc_row(r, hf);
        const bool d = KEY_IN_REG ? x_dead(a, mrow_base, lane_index, reg_index) : x_dead(a, mrow_base, reg_index, lane_index);
        bits |= (d ? 1u : 0u) << r;
    }
    return bits;
}

// dst[rows][KS] <- src rows row0.. (row stride stride_n), zero beyond n_valid rows and D columns
template <int DT>
__device__ __forceinline__ void x_load_tile(float* __restrict__ dst, const float* __restrict__ src, long long stride_n, int row0,
                                            int rows, int n_valid, int D, int tid, int nthr, bool vec) {
    constexpr int W = 32 * DT, KS = W + 1;
    if (vec) {
        constexpr int W4 = W / 4;
        for (int idx = tid; idx < rows * W4; idx += nthr) {
            const int r = idx / W4, c = (idx - r * W4) * 4;
            const int n = row0 + r;
            f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
            if (n < n_valid && c < D) t = *reinterpret_cast<const f32x4*>(src + (long long)n * stride_n + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[r * KS + c + e] = t[e];
        }
        return;
    }
    for (int idx = tid; idx < rows * W; idx += nthr) {
        const int r = idx / W, c = idx - r * W;
        const int n = row0 + r;
        dst[r * KS + c] = (n < n_valid && c < D) ? src[(long long)n * stride_n + c] : 0.0f;
    }
}

template <int DT>
__device__ __forceinline__ void x_store_tile(float* __restrict__ dstg, long long stride_n, int row0, int n_valid, int D,
                                             const float* __restrict__ T, int lane) {
    constexpr int KS = 32 * DT + 1;
    for (int idx = lane; idx < 32 * D; idx += 64) {
        const int r = idx / D, c = idx - r * D;
        const int n = row0 + r;
        if (n < n_valid) dstg[(long long)n * stride_n + c] = T[r * KS + c];
    }
}

constexpr int XCH = 4;            // tiles of 32 rows per LDS chunk of the swept operand (128 rows)

// forward: grid (B*H, ceil(q tiles / 8)); a wave owns a 32-query tile and sweeps the keys in chunks of XCH tiles with the running
// (max, sum) rescale of utils.py:199-221 -- in this orientation (O^T[d][query]: the query is the lane) the rescale of the partial
// output is one multiplication of the lane's accumulators by a lane-private factor
template <int DT>
__global__ __launch_bounds__(XTHR) void attn_x_fwd_kernel(const AttnXArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 32 * DT + 1, CH = XCH * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int D = a.D, nkt = a.nkt;
    float* K_s = smem;
    float* V_s = K_s + CH * KS;
    float* Q_w = V_s + CH * KS + wave * 32 * KS;

    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    float* ob = a.out + bi * a.osb + hi * a.osh;
    const unsigned char* mb = a.mask ? a.mask + bi * a.msb + hi * a.msh : nullptr;

    const float sc2 = a.scale * LOG2E_X;
    const int qt = blockIdx.y * XW + wave;       // tiles past nqt run on zero rows and store nothing
    const int qrow = qt * 32 + l31;
    x_load_tile<DT>(Q_w, qb, a.qsn, qt * 32, 32, (qt < a.nqt) ? a.Nq : 0, D, lane, 64, a.vec);
    __syncthreads();
    float qf[16 * DT];
#pragma unroll
    for (int s = 0; s < 16 * DT; ++s) qf[s] = Q_w[l31 * KS + 2 * s + hf];

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.0f;
    float mrun = -INFINITY, lrun = 0.0f;         // running max (raw scores) and sum of this lane's query

    for (int kc = 0; kc < nkt; kc += XCH) {
        __syncthreads();                         // every wave is done with the previous chunk
        x_load_tile<DT>(K_s, kb, a.ksn, kc * 32, CH, a.Nk, D, tid, XTHR, a.vec);
        x_load_tile<DT>(V_s, vb, a.vsn, kc * 32, CH, a.Nk, D, tid, XTHR, a.vec);
        __syncthreads();
        f32x16 sacc[XCH];
        unsigned dbits[XCH];
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[j][r] = 0.0f;
            dbits[j] = (kc + j < nkt) ? x_dead_bits<true>(a, mb, qrow, (kc + j) * 32, hf) : 0xffffu;
            if (kc + j < nkt) {
                const float* kp = K_s + (j * 32 + l31) * KS + hf;
#pragma unroll
                for (int s = 0; s < 16 * DT; ++s) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s], qf[s], sacc[j], 0, 0, 0);
            }
        }
        float mx = mrun;
#pragma unroll
        for (int j = 0; j < XCH; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sv = ((dbits[j] >> r) & 1u) ? -INFINITY : sacc[j][r];
                sacc[j][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mxs = (mx == -INFINITY) ? 0.0f : mx * sc2;
        const float corr = exp2f(mrun * sc2 - mxs);          // 0 while nothing was live before (mrun = -inf)
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < XCH; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2f(sacc[j][r] * sc2 - mxs);
                sacc[j][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 32);
        lrun = lrun * corr + sum;
        mrun = mx;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= corr;
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            if (kc + j < nkt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* vp = V_s + (j * 32 + kv_acc_row(r, hf)) * KS + l31;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[dt * 32], sacc[j][r], oacc[dt], 0, 0, 0);
                }
            }
        }
    }
    const float inv = lrun > 0.0f ? 1.0f / lrun : 0.0f;      // every key dead: o = 0
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) Q_w[l31 * KS + dt * 32 + kv_acc_row(r, hf)] = oacc[dt][r] * inv;
    __syncthreads();
    if (qt < a.nqt) {
        x_store_tile<DT>(ob, a.osn, qt * 32, a.Nq, D, Q_w, lane);
        if (hf == 0 && qrow < a.Nq && a.lse) a.lse[(long long)bh * a.Nq + qrow] = lrun > 0.0f ? mrun * a.scale + logf(lrun) : -FLT_MAX;
    }
}

__global__ __launch_bounds__(256) void attn_x_delta_kernel(const AttnXArgs a) {
    const int sub = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long long rows = (long long)a.B * a.H * a.Nq;
    float s = 0.0f;
    if (row < rows) {
        const int n = (int)(row % a.Nq);
        const long long bh = row / a.Nq;
        const int hi = (int)(bh % a.H);
        const long long bi = bh / a.H;
        const float* op = a.o + bi * a.osb + hi * a.osh + (long long)n * a.osn;
        const float* dp = a.d_o + bi * a.osb + hi * a.osh + (long long)n * a.osn;
        for (int c = sub; c < a.D; c += 16) s += op[c] * dp[c];
    }
    s += __shfl_xor(s, 8);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 1);
    if (row < rows && sub == 0) a.delta[row] = s;
}

// dK, dV: key-stationary (utils.py:262-291).  grid (B*H, ceil(key tiles / 8)); a wave owns a 32-key tile (K, V fragments in registers,
// dK^T / dV^T in accumulators) and sweeps the queries in LDS chunks of XCH tiles.
template <int DT>
__global__ __launch_bounds__(XTHR) void attn_x_bwd_kv_kernel(const AttnXArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 32 * DT + 1, CH = XCH * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int D = a.D;
    float* Q_s = smem;
    float* dO_s = Q_s + CH * KS;
    float* lse_s = dO_s + CH * KS;
    float* dl_s = lse_s + CH;
    float* T_w = dl_s + CH + wave * 32 * KS;

    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    const float* dob = a.d_o + bi * a.osb + hi * a.osh;
    float* dkb = a.dk + bi * a.ksb + hi * a.ksh;
    float* dvb = a.dv + bi * a.vsb + hi * a.vsh;
    const unsigned char* mb = a.mask ? a.mask + bi * a.msb + hi * a.msh : nullptr;
    const float sc2 = a.scale * LOG2E_X;

    const int jt = blockIdx.y * XW + wave;
    const int key = jt * 32 + l31;
    float kf[16 * DT], vf[16 * DT];
    x_load_tile<DT>(T_w, kb, a.ksn, jt * 32, 32, (jt < a.nkt) ? a.Nk : 0, D, lane, 64, a.vec);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16 * DT; ++s) kf[s] = T_w[l31 * KS + 2 * s + hf];
    __syncthreads();
    x_load_tile<DT>(T_w, vb, a.vsn, jt * 32, 32, (jt < a.nkt) ? a.Nk : 0, D, lane, 64, a.vec);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16 * DT; ++s) vf[s] = T_w[l31 * KS + 2 * s + hf];

    f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dkacc[dt][r] = 0.0f;
            dvacc[dt][r] = 0.0f;
        }
    for (int qc = 0; qc < a.nqt; qc += XCH) {
        __syncthreads();
        x_load_tile<DT>(Q_s, qb, a.qsn, qc * 32, CH, a.Nq, D, tid, XTHR, a.vec);
        x_load_tile<DT>(dO_s, dob, a.osn, qc * 32, CH, a.Nq, D, tid, XTHR, a.vec);
        for (int n = tid; n < CH; n += XTHR) {
            const int qn = qc * 32 + n;
            lse_s[n] = (qn < a.Nq) ? a.lse_in[(long long)bh * a.Nq + qn] * LOG2E_X : INFINITY;      // exp2(-inf) = 0 on pad rows
            dl_s[n] = (qn < a.Nq) ? a.delta_in[(long long)bh * a.Nq + qn] : 0.0f;
        }
        __syncthreads();
        for (int qq = 0; qq < XCH && qc + qq < a.nqt; ++qq) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.0f;
                pacc[r] = 0.0f;
            }
            const float* qp = Q_s + (qq * 32 + l31) * KS + hf;
            const float* dp = dO_s + (qq * 32 + l31) * KS + hf;
#pragma unroll
            for (int s = 0; s < 16 * DT; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[2 * s], kf[s], sacc, 0, 0, 0);      // S[q][key]
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[2 * s], vf[s], pacc, 0, 0, 0);      // dP[q][key]
            }
            const unsigned db = x_dead_bits<false>(a, mb, key, (qc + qq) * 32, hf);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = qq * 32 + kv_acc_row(r, hf), qrow = qc * 32 + ql;
                float p = exp2f(sacc[r] * sc2 - lse_s[ql]);
                if (jt >= a.nkt || qrow >= a.Nq || ((db >> r) & 1u)) p = 0.0f;          // select, never multiply: exp2 may be inf on a dead row
                sacc[r] = p;
                pacc[r] = p == 0.0f ? 0.0f : p * a.scale * (pacc[r] - dl_s[ql]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (qq * 32 + kv_acc_row(r, hf)) * KS + l31;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dO_s[row + dt * 32], sacc[r], dvacc[dt], 0, 0, 0);
                    dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Q_s[row + dt * 32], pacc[r], dkacc[dt], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) T_w[l31 * KS + dt * 32 + kv_acc_row(r, hf)] = dkacc[dt][r];
    __syncthreads();
    if (jt < a.nkt) x_store_tile<DT>(dkb, a.ksn, jt * 32, a.Nk, D, T_w, lane);
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) T_w[l31 * KS + dt * 32 + kv_acc_row(r, hf)] = dvacc[dt][r];
    __syncthreads();
    if (jt < a.nkt) x_store_tile<DT>(dvb, a.vsn, jt * 32, a.Nk, D, T_w, lane);
}

// dQ: query-stationary mirror of the forward kernel (grid (B*H, ceil(q tiles / 8)); keys swept in LDS chunks of XCH tiles)
template <int DT>
__global__ __launch_bounds__(XTHR) void attn_x_bwd_q_kernel(const AttnXArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 32 * DT + 1, CH = XCH * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.x, bi = bh / a.H, hi = bh - bi * a.H;
    const int D = a.D, nkt = a.nkt;
    float* K_s = smem;
    float* V_s = K_s + CH * KS;
    float* T_w = V_s + CH * KS + wave * 32 * KS;

    const float* qb = a.q + bi * a.qsb + hi * a.qsh;
    const float* kb = a.k + bi * a.ksb + hi * a.ksh;
    const float* vb = a.v + bi * a.vsb + hi * a.vsh;
    const float* dob = a.d_o + bi * a.osb + hi * a.osh;
    float* dqb = a.dq + bi * a.qsb + hi * a.qsh;
    const unsigned char* mb = a.mask ? a.mask + bi * a.msb + hi * a.msh : nullptr;
    const float sc2 = a.scale * LOG2E_X;

    const int qt = blockIdx.y * XW + wave;
    const int qrow = qt * 32 + l31;
    const bool q_ok = (qt < a.nqt) && (qrow < a.Nq);
    float qf[16 * DT], dof[16 * DT];
    x_load_tile<DT>(T_w, qb, a.qsn, qt * 32, 32, (qt < a.nqt) ? a.Nq : 0, D, lane, 64, a.vec);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16 * DT; ++s) qf[s] = T_w[l31 * KS + 2 * s + hf];
    __syncthreads();
    x_load_tile<DT>(T_w, dob, a.osn, qt * 32, 32, (qt < a.nqt) ? a.Nq : 0, D, lane, 64, a.vec);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16 * DT; ++s) dof[s] = T_w[l31 * KS + 2 * s + hf];
    const float lse2 = q_ok ? a.lse_in[(long long)bh * a.Nq + qrow] * LOG2E_X : INFINITY;
    const float dl = q_ok ? a.delta_in[(long long)bh * a.Nq + qrow] : 0.0f;

    f32x16 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.0f;
    for (int kc = 0; kc < nkt; kc += XCH) {
        __syncthreads();
        x_load_tile<DT>(K_s, kb, a.ksn, kc * 32, CH, a.Nk, D, tid, XTHR, a.vec);
        x_load_tile<DT>(V_s, vb, a.vsn, kc * 32, CH, a.Nk, D, tid, XTHR, a.vec);
        __syncthreads();
        for (int j = 0; j < XCH && kc + j < nkt; ++j) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.0f;
                pacc[r] = 0.0f;
            }
            const float* kp = K_s + (j * 32 + l31) * KS + hf;
            const float* vp = V_s + (j * 32 + l31) * KS + hf;
#pragma unroll
            for (int s = 0; s < 16 * DT; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s], qf[s], sacc, 0, 0, 0);      // S^T[key][q]
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[2 * s], dof[s], pacc, 0, 0, 0);     // dP^T[key][q]
            }
            const unsigned db = x_dead_bits<true>(a, mb, qrow, (kc + j) * 32, hf);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2f(sacc[r] * sc2 - lse2);
                const bool dead = !q_ok || ((db >> r) & 1u);
                pacc[r] = dead ? 0.0f : p * a.scale * (pacc[r] - dl);                             // dS^T
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* kr = K_s + (j * 32 + kv_acc_row(r, hf)) * KS + l31;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[dt * 32], pacc[r], dqacc[dt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) T_w[l31 * KS + dt * 32 + kv_acc_row(r, hf)] = dqacc[dt][r];
    __syncthreads();
    if (qt < a.nqt) x_store_tile<DT>(dqb, a.qsn, qt * 32, a.Nq, D, T_w, lane);
}

int x_check(const kanvit_attn_desc* d, const kanvit_attn_ext* e, const char* who) {
    if (!d || !e) return kv_fail(KANVIT_EINVAL, "%s: null descriptor", who);
    if (d->B < 0 || d->H < 1 || d->N < 1 || e->Nk < 1 || d->D < 1) return kv_fail(KANVIT_EINVAL, "%s: bad sizes", who);
    if (d->D > KANVIT_ATTN_MAX_D || (d->D & 1)) return kv_fail(KANVIT_EINVAL, "%s: D=%d must be even and <= %d", who, d->D, KANVIT_ATTN_MAX_D);
    if ((long long)d->B * d->H > 0x7fffffffLL) return kv_fail(KANVIT_EINVAL, "%s: B*H too large", who);
    if (!(d->scale > 0.0f)) return kv_fail(KANVIT_EINVAL, "%s: scale=%g must be positive (the running maximum is taken over the raw scores)", who, (double)d->scale);
    if (d->causal && e->Nk > d->N)
        return kv_fail(KANVIT_EINVAL, "%s: causal with k_len=%d > q_len=%d is ill-defined in the reference (utils.py:169,183: the first k_len - q_len queries see no key)", who, e->Nk, d->N);
    if (d->flags & KANVIT_FLAG_BF16_MFMA) return kv_fail(KANVIT_EINVAL, "%s: the general attention kernels are exact fp32 (no KANVIT_FLAG_BF16_MFMA)", who);
    if ((long long)(d->N + 127) / 128 > 65535 || (long long)(e->Nk + 127) / 128 > 65535) return kv_fail(KANVIT_EINVAL, "%s: sequence too long for one launch", who);
    return 0;
}

AttnXArgs x_args(const kanvit_attn_desc* d, const kanvit_attn_ext* e) {
    AttnXArgs a{};
    a.B = d->B; a.H = d->H; a.Nq = d->N; a.Nk = e->Nk; a.D = d->D; a.causal = d->causal; a.scale = d->scale;
    a.nqt = (d->N + 31) / 32;
    a.nkt = (e->Nk + 31) / 32;
    a.qsb = d->q_stride_b; a.qsh = d->q_stride_h; a.qsn = d->q_stride_n;
    a.ksb = d->k_stride_b; a.ksh = d->k_stride_h; a.ksn = d->k_stride_n;
    a.vsb = d->v_stride_b; a.vsh = d->v_stride_h; a.vsn = d->v_stride_n;
    a.osb = d->o_stride_b; a.osh = d->o_stride_h; a.osn = d->o_stride_n;
    a.vec = (d->D % 4 == 0);
    for (long long sd : {a.qsb, a.qsh, a.qsn, a.ksb, a.ksh, a.ksn, a.vsb, a.vsh, a.vsn, a.osb, a.osh, a.osn}) a.vec = a.vec && (sd % 4 == 0);
    a.mask = (const unsigned char*)e->mask;
    a.msb = e->mask_stride_b; a.msh = e->mask_stride_h; a.msq = e->mask_stride_q; a.msk = e->mask_stride_k;
    return a;
}

template <int DT>
int x_launch_fwd(const AttnXArgs& a, hipStream_t st) {
    constexpr int KS = 32 * DT + 1, CH = XCH * 32;
    const size_t lds = sizeof(float) * ((size_t)2 * CH * KS + (size_t)XW * 32 * KS);
    KV_ALLOW_LDS(160 * 1024, (attn_x_fwd_kernel<DT>));
    hipLaunchKernelGGL((attn_x_fwd_kernel<DT>), dim3((unsigned)(a.B * a.H), (unsigned)((a.nqt + XW - 1) / XW)), dim3(XTHR), lds, st, a);
    KV_LAUNCH_CHECK("attn_x_fwd_kernel");
    return 0;
}

template <int DT>
int x_launch_bwd(const AttnXArgs& a, hipStream_t st) {
    constexpr int KS = 32 * DT + 1, CH = XCH * 32;
    const size_t lds_kv = sizeof(float) * ((size_t)2 * CH * KS + 2 * (size_t)CH + (size_t)XW * 32 * KS);
    const size_t lds_q = sizeof(float) * ((size_t)2 * CH * KS + (size_t)XW * 32 * KS);
    KV_ALLOW_LDS(160 * 1024, (attn_x_bwd_kv_kernel<DT>));
    KV_ALLOW_LDS(160 * 1024, (attn_x_bwd_q_kernel<DT>));
    hipLaunchKernelGGL((attn_x_bwd_kv_kernel<DT>), dim3((unsigned)(a.B * a.H), (unsigned)((a.nkt + XW - 1) / XW)), dim3(XTHR), lds_kv, st, a);
    KV_LAUNCH_CHECK("attn_x_bwd_kv_kernel");
    hipLaunchKernelGGL((attn_x_bwd_q_kernel<DT>), dim3((unsigned)(a.B * a.H), (unsigned)((a.nqt + XW - 1) / XW)), dim3(XTHR), lds_q, st, a);
    KV_LAUNCH_CHECK("attn_x_bwd_q_kernel");
    return 0;
}

}  // namespace

extern "C" {

int kanvit_attn_x_fwd(const kanvit_attn_desc* d, const kanvit_attn_ext* e, const float* q, const float* k, const float* v, float* o,
                      float* lse, void* stream) {
    if (int rc = x_check(d, e, "kanvit_attn_x_fwd")) return rc;
    if (!q || !k || !v || !o) return kv_fail(KANVIT_EINVAL, "kanvit_attn_x_fwd: null q/k/v/o");
    if (d->B == 0) return 0;
    AttnXArgs a = x_args(d, e);
    a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse;
    a.vec = a.vec && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    return d->D <= 32 ? x_launch_fwd<1>(a, st) : x_launch_fwd<2>(a, st);
}

size_t kanvit_attn_x_bwd_workspace(const kanvit_attn_desc* d, const kanvit_attn_ext* e) {
    if (!d || !e || d->B < 0 || d->H < 1 || d->N < 1) return 0;
    return (sizeof(float) * (size_t)d->B * d->H * d->N + 15) / 16 * 16;         // rowsum(dO*O), "D" of utils.py:286
}

int kanvit_attn_x_bwd(const kanvit_attn_desc* d, const kanvit_attn_ext* e, const float* q, const float* k, const float* v, const float* o,
                      const float* lse, const float* d_o, float* dq, float* dk, float* dv, void* workspace, size_t workspace_bytes,
                      void* stream) {
    if (int rc = x_check(d, e, "kanvit_attn_x_bwd")) return rc;
    if (!q || !k || !v || !o || !lse || !d_o || !dq || !dk || !dv) return kv_fail(KANVIT_EINVAL, "kanvit_attn_x_bwd: null argument");
    if (d->B == 0) return 0;
    const size_t need = kanvit_attn_x_bwd_workspace(d, e);
    if (!workspace || workspace_bytes < need) return kv_fail(KANVIT_ENOMEM, "kanvit_attn_x_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
    AttnXArgs a = x_args(d, e);
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse_in = lse; a.d_o = d_o;
    a.dq = dq; a.dk = dk; a.dv = dv; a.delta = (float*)workspace; a.delta_in = (const float*)workspace;
    a.vec = a.vec && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)d_o) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)d->B * d->H * d->N;
    hipLaunchKernelGGL(attn_x_delta_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, a);
    KV_LAUNCH_CHECK("attn_x_delta_kernel");
    return d->D <= 32 ? x_launch_bwd<1>(a, st) : x_launch_bwd<2>(a, st);
}

}  // extern "C"
