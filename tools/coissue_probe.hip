// Microbenchmark: do VALU instructions of one wave overlap with MFMAs of the partner wave on the
// same SIMD?  Block = 8 waves (2 per SIMD): waves 0-3 issue MFMAs, waves 4-7 issue VALU FMAs.
// mode bit0: run the MFMA waves, bit1: run the VALU waves.  Prints ms for each combination.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float vfma(float x) { const float c1 = 1.0001f, c2 = 0.5f; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2)); return x; }
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int n_mfma, int n_valu, int mode) {
    int wave = threadIdx.x >> 6;
    if (mode & 8) wave = 7 - wave;                       // swap roles: VALU waves are the older ones
    if ((mode & 4) && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(3);
    if ((mode & 16) && __builtin_amdgcn_readfirstlane(wave) < 4) __builtin_amdgcn_s_setprio(3);
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            if (KIND == 0) {
                f32x16 acc0 = {0}, acc1 = {0};
                float a = threadIdx.x * 1e-3f, b = 1.0f;
                for (int i = 0; i < n_mfma; i += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
                }
                r = acc0[0] + acc1[3];
            } else {
                f32x16 acc0 = {0}, acc1 = {0};
                bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
                for (int i = 0; i < n_mfma; i += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
                }
                r = acc0[0] + acc1[3];
            }
        }
    } else if (mode & 2) {
        float x0 = threadIdx.x * 1e-4f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
        for (int i = 0; i < n_valu; i += 4) {
            x0 = vfma(x0);
            x1 = vfma(x1);
            x2 = vfma(x2);
            x3 = vfma(x3);
        }
        r = x0 + x1 + x2 + x3;
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int KIND>
float run(float* d, int nm, int nv, int mode) {
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    probe<KIND><<<256, 512>>>(d, nm, nv, mode);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 5; ++i) probe<KIND><<<256, 512>>>(d, nm, nv, mode);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    return ms / 5;
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    const int nm32 = 20000;            // f32 32x32x2: 64 cycles each
    for (int nv : {80000, 160000}) {
        printf("f32 mfma x%d | valu x%d:  mfma %.3f  valu %.3f  both %.3f | valu-prio %.3f | mfma-prio %.3f | swapped %.3f | swapped+valu-prio %.3f ms\n", nm32, nv,
               run<0>(d, nm32, nv, 1), run<0>(d, nm32, nv, 2), run<0>(d, nm32, nv, 3), run<0>(d, nm32, nv, 3 | 4),
               run<0>(d, nm32, nv, 3 | 16), run<0>(d, nm32, nv, 3 | 8), run<0>(d, nm32, nv, 3 | 8 | 4));
    }
    const int nmb = 40000;             // bf16 32x32x16: 32 cycles each
    for (int nv : {80000, 160000}) {
        printf("bf16 mfma x%d | valu x%d:  mfma %.3f  valu %.3f  both %.3f | valu-prio %.3f | mfma-prio %.3f | swapped %.3f | swapped+valu-prio %.3f ms\n", nmb, nv,
               run<1>(d, nmb, nv, 1), run<1>(d, nmb, nv, 2), run<1>(d, nmb, nv, 3), run<1>(d, nmb, nv, 3 | 4),
               run<1>(d, nmb, nv, 3 | 16), run<1>(d, nmb, nv, 3 | 8), run<1>(d, nmb, nv, 3 | 8 | 4));
    }
    return 0;
}
