// Intra-wave probe: one wave per SIMD (256-thread blocks) issues NV independent v_fma between
// consecutive MFMAs.  If MFMA execution is asynchronous w.r.t. the wave's VALU stream, time stays
// flat until NV*4 cycles exceeds the MFMA's pass time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float vfma(float x) { const float c1 = 1.0001f, c2 = 0.5f; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2)); return x; }
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, int n_mfma) {
    f32x16 acc0 = {0}, acc1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8}, bb = {8, 7, 6, 5, 4, 3, 2, 1};
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-4f + j;
    for (int i = 0; i < n_mfma; i += 2) {
        if (KIND == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j & 7] = vfma(x[j & 7]);
        if (KIND == 0) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ab, acc1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j & 7] = vfma(x[j & 7]);
    }
    float r = acc0[0] + acc1[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) r += x[j];
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int KIND, int NV, int WAVES>
float run(float* d, int nm) {
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    probe<KIND, NV, WAVES><<<256, 64 * WAVES>>>(d, nm);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 5; ++i) probe<KIND, NV, WAVES><<<256, 64 * WAVES>>>(d, nm);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    return ms / 5;
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    printf("f32 32x32x2 (64 cyc), 1 wave/SIMD, NV fma per MFMA: NV=0 %.3f  4 %.3f  8 %.3f  12 %.3f  16 %.3f  24 %.3f  32 %.3f ms\n",
           run<0, 0, 4>(d, 20000), run<0, 4, 4>(d, 20000), run<0, 8, 4>(d, 20000), run<0, 12, 4>(d, 20000),
           run<0, 16, 4>(d, 20000), run<0, 24, 4>(d, 20000), run<0, 32, 4>(d, 20000));
    printf("f32 32x32x2, 2 waves/SIMD (each 10000 MFMA):            NV=0 %.3f  4 %.3f  8 %.3f  12 %.3f  16 %.3f  24 %.3f  32 %.3f ms\n",
           run<0, 0, 8>(d, 10000), run<0, 4, 8>(d, 10000), run<0, 8, 8>(d, 10000), run<0, 12, 8>(d, 10000),
           run<0, 16, 8>(d, 10000), run<0, 24, 8>(d, 10000), run<0, 32, 8>(d, 10000));
    printf("bf16 32x32x16 (32 cyc), 1 wave/SIMD:                   NV=0 %.3f  2 %.3f  4 %.3f  6 %.3f  8 %.3f  12 %.3f  16 %.3f ms\n",
           run<1, 0, 4>(d, 40000), run<1, 2, 4>(d, 40000), run<1, 4, 4>(d, 40000), run<1, 6, 4>(d, 40000),
           run<1, 8, 4>(d, 40000), run<1, 12, 4>(d, 40000), run<1, 16, 4>(d, 40000));
    return 0;
}
