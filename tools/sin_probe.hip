// Accuracy of hardware v_sin_f32 / v_cos_f32 (argument in revolutions) against the Cody-Waite + polynomial kv_sincos of
// csrc/kan_basis.h, both measured against double precision, over the argument range the SineKAN layers see.
//   hipcc -O3 --offload-arch=gfx950 tools/sin_probe.hip -o tools/sin_probe && tools/sin_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../kan-vit_amd/csrc/kan_basis.h"

__device__ __forceinline__ float hw_sin(float x) {       // sin(x), x in radians
    const float r = x * 0.15915494309189535f;             // revolutions
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r));
}
__device__ __forceinline__ float hw_sin2(float x) {      // two-term product for the revolutions (compensated 1/2pi)
    const float C = 0.15915494309189535f, CL = -6.1232339957e-10f; // placeholder low part
    const float r = x * C;
    const float lo = __builtin_fmaf(x, C, -r);
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r) + lo + x * CL);
}
__global__ void k(const float* x, float* a, float* b, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] = kv_sin(x[i]);
    b[i] = hw_sin(x[i]);
    c[i] = hw_sin2(x[i]);
}
int main() {
    const int n = 1 << 22;
    for (float range : {4.0f, 16.0f, 64.0f, 256.0f}) {
        std::vector<float> x(n), a(n), b(n), c(n);
        for (int i = 0; i < n; ++i) x[i] = range * (2.0f * (float)rand() / RAND_MAX - 1.0f);
        float *dx, *da, *db, *dc;
        hipMalloc(&dx, 4 * n); hipMalloc(&da, 4 * n); hipMalloc(&db, 4 * n); hipMalloc(&dc, 4 * n);
        hipMemcpy(dx, x.data(), 4 * n, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, dc, n);
        hipMemcpy(a.data(), da, 4 * n, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), db, 4 * n, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, 4 * n, hipMemcpyDeviceToHost);
        double ea = 0, eb = 0, ec = 0, ra = 0, rb = 0;
        for (int i = 0; i < n; ++i) {
            const double r = sin((double)x[i]);
            ea = fmax(ea, fabs(a[i] - r)); eb = fmax(eb, fabs(b[i] - r)); ec = fmax(ec, fabs(c[i] - r));
            ra += (a[i] - r) * (a[i] - r); rb += (b[i] - r) * (b[i] - r);
        }
        printf("|x| < %5.0f: max abs error  kv_sin %.3e   v_sin_f32(fract(x/2pi)) %.3e   compensated %.3e   rms %.3e / %.3e\n", range, ea, eb, ec,
               sqrt(ra / n), sqrt(rb / n));
        hipFree(dx); hipFree(da); hipFree(db); hipFree(dc);
    }
    return 0;
}
