// Microbenchmark ladder for the fp32 register-form KAN kernels: what keeps v_mfma_f32_32x32x2_f32 from issuing back to back?
// Each variant adds one ingredient of kan_fwd_reg_kernel<CHEBY, NT=2, NSH=3, ICH=4> (20 k-steps x 6 MFMAs per chunk):
//   V0  MFMAs only (register operands)                              -> the ceiling of the structure
//   V1  + W fragments read from LDS one k-step ahead (3 x ds_read2_b32 per step)
//   V2  + one __syncthreads() per chunk
//   V3  + W chunk staging: 30 KB global -> registers at chunk start, ds_write_b128 after the MFMAs, barrier
//   V4  + the basis VALU work of a chunk (4 x tanh + Chebyshev recurrences) and the x loads
// Run with 1 and 2 work-groups per CU (dynamic LDS size decides), all 256 CUs, one round of work-groups (no tail).
// Prints TFLOP/s, the MFMA duty it implies at the in-kernel clock (s_memtime / s_memrealtime), and that clock.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe && tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WROW = 192, KC = 40, VH = 20, NTT = 6;

template <int V>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ w, const float* __restrict__ x, float* __restrict__ out, int nch,
                                                unsigned long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
    float* W_s = smem;                     // [2][KC][WROW]
    constexpr int WSZ = KC * WROW;
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0 && blockIdx.x == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int i = tid; i < 2 * WSZ; i += 256) W_s[i] = w[i % (KC * WROW)];
    __syncthreads();
    f32x16 acc[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int wc = (tid & 15) * 4, wr0 = tid >> 4;      // staging map: 16 rows per pass, 4 columns
    f32x4 wreg[3][3];
    const float* xrow = x + (size_t)(blockIdx.x * 128 + (tid >> 6) * 32 + l31) * 64 + hf * 4;
    f32x4 xv = {0.1f, 0.2f, 0.3f, 0.4f};
    for (int c = 0; c < nch; ++c) {
        if constexpr (V >= 3) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int row = wr0 + q * 16;
                    if (row < KC) wreg[p][q] = *reinterpret_cast<const f32x4*>(w + ((size_t)(c & 7) * KC + row) * WROW + p * 64 + wc);
                }
        }
        float phi[VH];
        if constexpr (V >= 4) {
            const f32x4 xn = *reinterpret_cast<const f32x4*>(xrow + (c & 7) * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float tt = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * xv[j]) + 1.0f);
                float p0 = 1.0f, p1 = tt;
                phi[j * 5] = 1.0f;
                phi[j * 5 + 1] = tt;
#pragma unroll
                for (int g = 2; g < 5; ++g) {
                    const float p2 = 2.0f * tt * p1 - p0;
                    phi[j * 5 + g] = p2;
                    p0 = p1;
                    p1 = p2;
                }
            }
            xv = xn;
        } else {
#pragma unroll
            for (int s = 0; s < VH; ++s) phi[s] = 0.01f * (s + 1) + lane * 1e-4f;
        }
        const float* wp = W_s + (c & 1) * WSZ + hf * WROW + l31;
        float wa[2][NTT];
        if constexpr (V >= 1) {
#pragma unroll
            for (int t = 0; t < NTT; ++t) wa[0][t] = wp[t * 32];
        } else {
#pragma unroll
            for (int t = 0; t < NTT; ++t) wa[0][t] = wa[1][t] = 0.5f + t + lane * 1e-3f;
        }
#pragma unroll
        for (int s2 = 0; s2 < VH; ++s2) {
            if constexpr (V >= 1) {
                if (s2 + 1 < VH) {
#pragma unroll
                    for (int t = 0; t < NTT; ++t) wa[(s2 + 1) & 1][t] = wp[(2 * (s2 + 1)) * WROW + t * 32];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NTT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s2 & 1][t], phi[s2], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (V >= 3) {
            float* dst = W_s + ((c + 1) & 1) * WSZ + wc;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int row = wr0 + q * 16;
                    if (row < KC) *reinterpret_cast<f32x4*>(dst + row * WROW + p * 64) = wreg[p][q];
                }
        }
        if constexpr (V >= 2) __syncthreads();
    }
    if constexpr (V >= 5) {      // the kernel's epilogue: 24 float4 row-segment stores per lane into y[M][2304]
        const size_t row = (size_t)(blockIdx.x / 12) * 128 + (tid >> 6) * 32 + l31;
        const int gs = blockIdx.x % 12;
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            float* yp = out + row * 2304 + (size_t)((t / 2) * 12 + gs) * 64 + (t % 2) * 32 + 4 * hf;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                *reinterpret_cast<f32x4*>(yp + 8 * q) = v;
            }
        }
    }
    float r = 0.0f;
#pragma unroll
    for (int t = 0; t < NTT; ++t) r += acc[t][0] + acc[t][7];
    if (r == 123.456f) out[tid] = r;
    if (tid == 0 && blockIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int V>
void run(const float* w, const float* x, float* out, unsigned long long* clk, int per_cu, int nch, int grid_override = 0) {
    const size_t lds = per_cu == 1 ? 120 * 1024 : 2 * KC * WROW * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int grid = grid_override ? grid_override : 256 * per_cu;
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    probe<V><<<grid, 256, lds>>>(w, x, out, nch, clk);
    hipDeviceSynchronize();
    hipEventRecord(s);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) probe<V><<<grid, 256, lds>>>(w, x, out, nch, clk);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    ms /= reps;
    unsigned long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0) ;          // memrealtime ticks at 100 MHz
    const double flops = (double)grid * 4 * nch * VH * NTT * 4096.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    const double duty = (double)nch * VH * NTT * 64.0 * per_cu / (double)h[0];      // MFMA cycles of one SIMD / kernel cycles (block 0's lifetime)
    printf("V%d  %d wg/CU grid %5d nch %4d : %8.3f ms  %7.1f TFLOP/s  clock %.2f GHz  mfma duty %.3f  (err %s)\n", V, per_cu, grid, nch, ms, tf, ghz, duty,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    float *w, *x, *out;
    unsigned long long* clk;
    hipMalloc(&w, 8 * KC * WROW * sizeof(float));
    hipMalloc(&x, (size_t)2364 * 128 * 64 * sizeof(float));
    hipMalloc(&out, (size_t)25216 * 2304 * sizeof(float));
    hipMalloc(&clk, 64);
    float* hw = (float*)malloc(8 * KC * WROW * sizeof(float));
    for (int i = 0; i < 8 * KC * WROW; ++i) hw[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
    hipMemcpy(w, hw, 8 * KC * WROW * sizeof(float), hipMemcpyHostToDevice);
    float* hx = (float*)malloc((size_t)512 * 128 * 64 * sizeof(float));
    for (size_t i = 0; i < (size_t)512 * 128 * 64; ++i) hx[i] = (float)((i * 40503u) & 0xffff) / 65536.0f - 0.5f;
    hipMemcpy(x, hx, (size_t)512 * 128 * 64 * sizeof(float), hipMemcpyHostToDevice);
    // the launch geometry of the real q|k|v forward (ViT-B, B = 128): 197 row tiles x 12 head sets = 2364 work-groups of 8 chunks
    for (int rep = 0; rep < 2; ++rep) {
        run<2>(w, x, out, clk, 2, 8, 2364);
        run<3>(w, x, out, clk, 2, 8, 2364);
        run<4>(w, x, out, clk, 2, 8, 2364);
        run<5>(w, x, out, clk, 2, 8, 2364);
        run<5>(w, x, out, clk, 2, 8, 512);
        run<5>(w, x, out, clk, 2, 64, 512);
    }
    for (int nch : {512}) {
        for (int per_cu : {1, 2}) {
            run<0>(w, x, out, clk, per_cu, nch);
            run<1>(w, x, out, clk, per_cu, nch);
            run<2>(w, x, out, clk, per_cu, nch);
            run<3>(w, x, out, clk, per_cu, nch);
            run<4>(w, x, out, clk, per_cu, nch);
        }
    }
    return 0;
}
