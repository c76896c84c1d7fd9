// fp32 -> three-term bf16 split images for the "bf16x3" feed-forward mode (kanvit/dense.py):
//     v = hi + lo + O(2^-17 |v|),  hi = bf16(v),  lo = bf16(v - hi)
// A product a.b is then formed as a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on the bf16 matrix cores with fp32 accumulation (the
// a_lo.b_lo term, ~2^-16 relative, is dropped): one bf16 GEMM over a K axis that is three times as long.  This kernel writes
// the K-concatenated operand image in one pass, optionally fusing what sits between two GEMMs of the reference's
// Linear -> ReLU -> Linear block (model.py:25-29): + bias, ReLU, or the ReLU mask of the backward pass.
//   pattern 0 ("A"): out[m] = [ hi | hi | lo ]        pattern 1 ("B"): out[m] = [ hi | lo | hi ]
// so that  A-image . B-image^T  =  hi.hi + hi.lo + lo.hi.
#include "../../include/kanvit.h"
#include "kanvit_common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short f2bf(float v) {      // round to nearest even
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// one thread = 8 consecutive columns of one row
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                     const unsigned short* __restrict__ mask, long long mask_ld,
                                                     unsigned short* __restrict__ out, long long M, int K, int relu, int pattern) {
    const int k8 = K / 8;
    const long long total = M * k8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long m = idx / k8;
        const int c = (int)(idx - m * k8) * 8;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + m * K + c);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(x + m * K + c + 4);
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias[c + e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
        }
        if (mask) {             // keep where the saved activation (its hi part) is positive: ReLU backward
            const u32x4_t mk = *reinterpret_cast<const u32x4_t*>(mask + m * mask_ld + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned short h = (unsigned short)((mk[e >> 1] >> (16 * (e & 1))) & 0xffffu);
                if (!(bf2f(h) > 0.0f)) v[e] = 0.0f;
            }
        }
        unsigned short hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hi[e] = f2bf(v[e]);
            lo[e] = f2bf(v[e] - bf2f(hi[e]));
        }
        const u32x4_t H = {(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16),
                           (unsigned)hi[4] | ((unsigned)hi[5] << 16), (unsigned)hi[6] | ((unsigned)hi[7] << 16)};
        const u32x4_t L = {(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16),
                           (unsigned)lo[4] | ((unsigned)lo[5] << 16), (unsigned)lo[6] | ((unsigned)lo[7] << 16)};
        unsigned short* o = out + m * 3 * K + c;
        *reinterpret_cast<u32x4_t*>(o) = H;
        *reinterpret_cast<u32x4_t*>(o + K) = pattern ? L : H;
        *reinterpret_cast<u32x4_t*>(o + 2 * K) = pattern ? H : L;
    }
}

}  // namespace

extern "C" {

int kanvit_split3_bf16(int64_t M, int K, const float* x, const float* bias, int relu, const void* mask_hi, int64_t mask_ld,
                       void* out, int pattern, void* stream) {
    if (M < 0 || K < 8 || (K & 7)) return kv_fail(KANVIT_EINVAL, "kanvit_split3_bf16: K=%d must be a positive multiple of 8", K);
    if (M == 0) return 0;
    if (!x || !out) return kv_fail(KANVIT_EINVAL, "kanvit_split3_bf16: null x/out");
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)(mask_hi ? mask_hi : out)) & 15 || (mask_hi && (mask_ld & 7)))
        return kv_fail(KANVIT_EINVAL, "kanvit_split3_bf16: pointers must be 16-byte aligned (mask_ld a multiple of 8)");
    const long long total = (long long)M * (K / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias,
                       (const unsigned short*)mask_hi, (long long)mask_ld, (unsigned short*)out, (long long)M, K, relu, pattern ? 1 : 0);
    KV_LAUNCH_CHECK("split3_kernel");
    return 0;
}

}  // extern "C"
