// Device-side basis functions of the five KAN families (+ the identity "linear"
// family) and their derivatives.  Values are produced straight into an LDS
// tile (dst[j*stride]); nothing here touches HBM except the small per-group
// parameter tables (knots / centres / freq / phase), which are L1/L2 resident.
//
// Reference arithmetic being restated (paths relative to the reference repo):
//   CHEBY   models/cheby.py:37-43     T_d(tanh x) = cos(d*acos(tanh x)); evaluated by the
//                                     three-term recurrence ("Transcendental cost")
//   BSPLINE models/effkan.py:99-132   Cox-de Boor recursion, half-open order-0 indicator
//   RBF     models/fastkan.py:29-30   exp(-((u-c)/h)^2);  base path silu(x) fastkan.py:74
//   SINE    models/sinekan.py:86      sin(x*freq + phase)
//   FOURIER models/nfkan.py:44-45     cos(k x), sin(k x), k = 1..G (rotation recurrence)
#pragma once
#include <hip/hip_runtime.h>

#define KV_LINEAR 0
#define KV_CHEBY 1
#define KV_BSPLINE 2
#define KV_RBF 3
#define KV_SINE 4
#define KV_FOURIER 5
#define KV_SINE_DF 6      // internal (not an ABI family): SINE's frequency-derivative operand x * cos(x f + p), weight-gradient kernels only (KANVIT_FLAG_SINE_DFREQ)

#define KV_MAX_KNOTS 40   // generic B-spline path: grid_size + 2*order + 1 <= 40

struct BasisArgs {
    int G;             // family count parameter (see kanvit.h)
    int GP;            // generated columns per input feature
    int order;         // BSPLINE
    int nk;            // BSPLINE knots per feature
    int has_base;      // BSPLINE / RBF
    float inv_h;       // RBF
    const float* bp;   // this group's parameter table
    int uniform;       // BSPLINE: KANVIT_FLAG_UNIFORM_KNOTS (closed-form cubic)
};

// tanh(x) = 1 - 2/(exp(2x)+1): v_exp_f32 + v_rcp_f32, absolute error ~2e-7 over the whole range
// (saturates cleanly: exp -> inf gives 1, exp -> 0 gives -1).  ocml's tanhf is ~10x the instructions.
__device__ __forceinline__ float kv_tanh(float x) {
    // v_rcp_f32 (1 ulp) instead of the correctly rounded quotient: __frcp_rn expands to the ten-instruction IEEE division
    // sequence, and the matrix pipe waits for every VALU instruction of the fp32 kernels
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}

// FastKAN's own grid (models/fastkan.py:22-27: G = 8 centres linspace(min, max), denominator = their spacing h): with
// t = (u - c0)/h the basis is exp(-(t - j)^2) and phi_{j+1} = phi_j * exp(2t - 1) * exp(-2j).  Two anchors (j = 2 and 5) by
// exp, every other value at most two steps away: 3 v_exp + 1 v_rcp + 12 v_mul for the eight values instead of eight times
// (sub, mul, mul, mul, exp) -- the RBF kernels are bound by this VALU work, not by the matrix pipe.
// Error: two steps x ~2 ulp (the ratio exp(2t - 1) is formed with a compensated argument, kv_exp_comp) ~ 5e-7 RELATIVE wherever the
// anchor is a normal number.  v_exp_f32 has no denormal range, so an anchor that underflows (|t - 2| or |t - 5| > ~9.35) takes its
// neighbours to exactly 0 with it: the largest value lost that way is p[0] = exp(-t^2) just past t = -7.35, 3.5e-24 (ABSOLUTE error
// <= 4e-24; the relative error is 100 % in that tail, which no training step can see).  t is clamped so exp(2t - 1) stays
// finite (everything is 0 out there); the compare form of the clamp lets a NaN input stay a NaN.
// exp(x) with the rounding of x*log2(e) compensated (the fast __expf loses |x| * 6e-8 relative there: 8e-7 at x = 13, and the
// recurrence below multiplies by this value up to twice); v_exp_f32's own ~1 ulp remains
__device__ __forceinline__ float kv_exp_comp(float x) {
    const float L2E = 1.44269502162933349609375f, L2E_LO = 1.925963033500e-8f;
    const float y = x * L2E;
    const float lo = __builtin_fmaf(x, L2E, -y) + x * L2E_LO;
    const float e = __builtin_amdgcn_exp2f(y);
    return __builtin_fmaf(e, lo * 0.69314718f, e);
}
__device__ __forceinline__ void kv_rbf8(float t, float (&p)[8]) {
    const float tc = t < -20.0f ? -20.0f : (t > 30.0f ? 30.0f : t);
    const float a2 = tc - 2.0f, a5 = tc - 5.0f;
    const float e2 = __expf(-a2 * a2), e5 = __expf(-a5 * a5);
    const float R = kv_exp_comp(2.0f * tc - 1.0f);
    const float Ri = __builtin_amdgcn_rcpf(R);
    p[2] = e2;
    p[3] = e2 * (R * 1.8315639e-02f);        // exp(-4)
    p[1] = e2 * (Ri * 7.3890561f);           // exp(2)
    p[0] = p[1] * Ri;
    p[5] = e5;
    p[6] = e5 * (R * 4.5399930e-05f);        // exp(-10)
    p[7] = p[6] * (R * 6.1442124e-06f);      // exp(-12)
    p[4] = e5 * (Ri * 2.9809580e+03f);       // exp(8)
}
// p[j] without a runtime-indexed register array (which would live in scratch): folds to one register when j is a constant
__device__ __forceinline__ float kv_sel8(const float (&p)[8], int j) {
    const float a = (j & 1) ? p[1] : p[0], b = (j & 1) ? p[3] : p[2], c = (j & 1) ? p[5] : p[4], d = (j & 1) ? p[7] : p[6];
    const float ab = (j & 2) ? b : a, cd = (j & 2) ? d : c;
    return (j & 4) ? cd : ab;
}

// sin and cos together, |x| up to ~5e4 with <= 9e-8 absolute error (libm's float sin: 7e-8): k = rint(x * 2/pi), a
// three-term Cody-Waite reduction r = x - k*pi/2 (exact products through fma), then the Cephes minimax polynomials on
// [-pi/4, pi/4] and a quadrant swap.  ~20 VALU ops for both values; ocml's sinf + cosf (Payne-Hanek path included) are
// several times that, and the Sine/Fourier layers evaluate one per (row, feature, grid point).
__device__ __forceinline__ void kv_sincos(float x, float& s, float& c) {
    const float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(k, -1.5707963705062866f, x);
    r = fmaf(k, 4.371138828673793e-08f, r);
    r = fmaf(k, 1.7763568394002505e-15f, r);
    const int q = (int)k;
    const float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    cp = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    const bool swap = (q & 1) != 0;
    const float ss = swap ? cp : sp, cc = swap ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}
// One value only (SineKAN evaluates one sine -- or, in the input gradient, one cosine -- per (row, feature, grid point), and
// these layers are bound by exactly this VALU work): reduce by PI instead of PI/2, k = rint(x/pi), r = x - k*pi in
// [-pi/2, pi/2] (three-term Cody-Waite, exact products through fma), sin(x) = (-1)^k sin(r) with ONE odd polynomial
// r + r^3 (S1 + S2 r^2 + S3 r^4 + S4 r^6) (weighted minimax fit, approximation error 4.7e-9; 1.2e-7 absolute evaluated in
// fp32 for |x| up to 256, against 9e-8 for kv_sincos) and the sign as an XOR on the sign bit: 14 VALU instructions where
// kv_sincos needs 24 for a value whose quadrant swap drags both polynomials along.  cos(x) = sin(x + pi/2) is the same with
// the half-integer multiplier k - 1/2, k = rint(x/pi + 1/2): (k - 1/2) * PI_A is still an exact fp32 product.
__device__ __forceinline__ float kv_sin_poly(float r, float k) {
    const float r2 = r * r;
    float p = fmaf(r2, 2.6001134756370448e-06f, -0.00019806642376352102f);
    p = fmaf(p, r2, 0.00833301804959774f);
    p = fmaf(p, r2, -0.16666656732559204f);
    const float s = fmaf(p * r2, r, r);
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s) ^ ((unsigned)(int)k << 31));      // odd k: negate
}
__device__ __forceinline__ float kv_sin(float x) {
    const float k = rintf(x * 0.3183098861837907f);
    float r = fmaf(k, -3.140625f, x);
    r = fmaf(k, -9.67502593994140625e-4f, r);
    r = fmaf(k, -1.509957990978376432e-7f, r);
    return kv_sin_poly(r, k);
}
__device__ __forceinline__ float kv_cos(float x) {
    const float k = rintf(fmaf(x, 0.3183098861837907f, 0.5f));
    const float kk = k - 0.5f;
    float r = fmaf(kk, -3.140625f, x);
    r = fmaf(kk, -9.67502593994140625e-4f, r);
    r = fmaf(kk, -1.509957990978376432e-7f, r);
    return kv_sin_poly(r, k);
}

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (11 instructions): silu is evaluated once per (row, feature) in the
// B-spline and FastKAN kernels, which are bound by exactly this VALU work; 2 ulp of a value of order 1 against the 1e-4 parity bound.
__device__ __forceinline__ float kv_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float kv_dsilu(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}

// ---------------------------------------------------------------------------------------------
// B-spline helpers.  Both run the Cox-de Boor recursion up to order-1, then finish the last
// level producing values and (optionally) derivatives in one pass:
//   dB_{j,p}/dx = p*(B_{j,p-1}/(t_{j+p}-t_j) - B_{j+1,p-1}/(t_{j+p+1}-t_{j+1}))
// _fixed: compile-time knot count / order, fully unrolled, everything in registers (the
//         grid_size=5, spline_order=3 configuration of every reference call site);
// _rt:    runtime sizes, local arrays (slow path for other ctor arguments).
// ---------------------------------------------------------------------------------------------
template <int NK, int ORD, bool DER>
__device__ __forceinline__ void kv_bspline_fixed(const float* __restrict__ kn_g, float xv, float* __restrict__ val,
                                                 float* __restrict__ der) {
    float kn[NK];
    float bb[NK - 1];
#pragma unroll
    for (int j = 0; j < NK; ++j) kn[j] = kn_g[j];
#pragma unroll
    for (int j = 0; j < NK - 1; ++j) bb[j] = (xv >= kn[j] && xv < kn[j + 1]) ? 1.0f : 0.0f;   // effkan.py:115
#pragma unroll
    for (int k = 1; k < ORD; ++k) {
#pragma unroll
        for (int j = 0; j < NK - 1 - k; ++j) {
            const float l = __fdividef(xv - kn[j], kn[j + k] - kn[j]);
            const float r = __fdividef(kn[j + k + 1] - xv, kn[j + k + 1] - kn[j + 1]);
            bb[j] = l * bb[j] + r * bb[j + 1];                                                  // effkan.py:117-125
        }
    }
#pragma unroll
    for (int j = 0; j < NK - 1 - ORD; ++j) {
        const float il = __fdividef(1.0f, kn[j + ORD] - kn[j]);
        const float ir = __fdividef(1.0f, kn[j + ORD + 1] - kn[j + 1]);
        val[j] = (xv - kn[j]) * il * bb[j] + (kn[j + ORD + 1] - xv) * ir * bb[j + 1];
        if (DER) der[j] = (float)ORD * (bb[j] * il - bb[j + 1] * ir);
    }
}

__device__ __noinline__ void kv_bspline_rt(const float* __restrict__ kn, int nk, int ord, float xv,
                                           float* __restrict__ val, float* __restrict__ der) {
    float bb[KV_MAX_KNOTS];
    for (int j = 0; j < nk - 1; ++j) bb[j] = (xv >= kn[j] && xv < kn[j + 1]) ? 1.0f : 0.0f;
    for (int k = 1; k < ord; ++k)
        for (int j = 0; j < nk - 1 - k; ++j) {
            const float l = __fdividef(xv - kn[j], kn[j + k] - kn[j]);
            const float r = __fdividef(kn[j + k + 1] - xv, kn[j + k + 1] - kn[j + 1]);
            bb[j] = l * bb[j] + r * bb[j + 1];
        }
    if (ord == 0) {
        for (int j = 0; j < nk - 1; ++j) {
            val[j] = bb[j];
            if (der) der[j] = 0.0f;
        }
        return;
    }
    for (int j = 0; j < nk - 1 - ord; ++j) {
        const float il = __fdividef(1.0f, kn[j + ord] - kn[j]);
        const float ir = __fdividef(1.0f, kn[j + ord + 1] - kn[j + 1]);
        val[j] = (xv - kn[j]) * il * bb[j] + (kn[j + ord + 1] - xv) * ir * bb[j + 1];
        if (der) der[j] = (float)ord * (bb[j] * il - bb[j + 1] * ir);
    }
}

// Uniform cubic B-splines (knots g0 + j*h): for x in knot interval j0 the four non-zero bases are j0-3 .. j0 with
//   (1-u)^3/6, (3u^3-6u^2+4)/6, (-3u^3+3u^2+3u+1)/6, u^3/6,  u = (x-g0)/h - j0   (same values as Cox-de Boor, e.g. x = 0 on
// the reference grid gives [.., 1/48, 23/48, 23/48, 1/48, ..]); outside [g0, g0 + (nk-1)h) everything is zero, which is the
// half-open order-0 indicator of models/effkan.py:115.
__device__ __forceinline__ bool kv_bspline_uniform(const float* __restrict__ kn, int nk, float xv, int& j0, float (&bv)[4],
                                                   float (&dv)[4], bool want_der) {
    const float g0 = kn[0], h = (kn[nk - 1] - kn[0]) / (float)(nk - 1);
    const float ih = __frcp_rn(h);
    const float t = (xv - g0) * ih;
    const float fl = floorf(t);
    j0 = (int)fl;
    const bool in = (t >= 0.0f) && j0 < nk - 1;
    // the values are ALWAYS formed, and finite (u = 0 outside the knot range, for NaN and for infinities): kv_bsel4 multiplies
    // them by 0/1 masks, and a callers' `in ? ... : 0` select would cost the same instruction
    const float u = in ? t - fl : 0.0f, u2 = u * u, u3 = u2 * u, om = 1.0f - u;
    const float s6 = 1.0f / 6.0f;
    bv[0] = om * om * om * s6;
    bv[1] = (3.0f * u3 - 6.0f * u2 + 4.0f) * s6;
    bv[2] = (-3.0f * u3 + 3.0f * u2 + 3.0f * u + 1.0f) * s6;
    bv[3] = u3 * s6;
    if (want_der) {
        dv[0] = -0.5f * om * om * ih;
        dv[1] = (1.5f * u2 - 2.0f * u) * ih;
        dv[2] = (-1.5f * u2 + u + 0.5f) * ih;
        dv[3] = 0.5f * u2 * ih;
    }
    return in;
}

// Value of basis jg for a point in knot interval j0 (the four non-zero values v[e] belong to bases j0-3+e), branch-free:
// lane masks (j0 == k) as 0/1 floats -- jg is a constant after unrolling, so each mask is formed once and shared between
// the bases -- times the four values: 11 v_cndmask + 4 v_fma per basis, exact (one factor is 1, the others 0; the values
// are finite).  A chain of `j0 == c ? v : ...` is what one would write; the compiler turns that into a switch over j0 with
// divergent branches around every basis value.  Points outside the knot range carry j0 = KV_BSPLINE_OUT, which matches
// nothing (every basis 0, models/effkan.py:115).
constexpr int KV_BSPLINE_OUT = -64;
__device__ __forceinline__ float kv_bsel4(int j0, int jg, const float (&v)[4]) {
    const float m0 = (j0 == jg + 3) ? 1.0f : 0.0f, m1 = (j0 == jg + 2) ? 1.0f : 0.0f;
    const float m2 = (j0 == jg + 1) ? 1.0f : 0.0f, m3 = (j0 == jg) ? 1.0f : 0.0f;
    return __builtin_fmaf(m0, v[0], __builtin_fmaf(m1, v[1], __builtin_fmaf(m2, v[2], m3 * v[3])));
}

// ---------------------------------------------------------------------------------------------
// forward: write GP values for feature i (value xv; RBF spline path uses uv) to dst[j*stride]
// ---------------------------------------------------------------------------------------------
template <int FAM>
__device__ __forceinline__ void basis_fwd(const BasisArgs& b, float xv, float uv, int i, float* __restrict__ dst,
                                          int stride) {
    if constexpr (FAM == KV_LINEAR) {
        dst[0] = xv;
    } else if constexpr (FAM == KV_CHEBY) {
        const float t = kv_tanh(xv);
        float p0 = 1.0f, p1 = t;
        dst[0] = 1.0f;
        if (b.G > 1) dst[stride] = t;
        for (int g = 2; g < b.G; ++g) {
            const float p2 = 2.0f * t * p1 - p0;
            dst[g * stride] = p2;
            p0 = p1;
            p1 = p2;
        }
    } else if constexpr (FAM == KV_BSPLINE) {
        const float* kn = b.bp + (long long)i * b.nk;
        if (b.uniform) {                           // order 3, uniform knots (host-checked): 4 non-zero bases
            int j0;
            float bv[4], dv[4];
            const bool in = kv_bspline_uniform(b.bp, b.nk, xv, j0, bv, dv, false);
            for (int j = 0; j < b.G; ++j) dst[j * stride] = 0.0f;
            if (in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = j0 - 3 + e;
                    if (idx >= 0 && idx < b.G) dst[idx * stride] = bv[e];
                }
            }
        } else if (b.nk == 12 && b.order == 3) {   // grid_size 5, spline_order 3: every call site of the reference
            float val[8];
            kv_bspline_fixed<12, 3, false>(kn, xv, val, nullptr);
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j * stride] = val[j];
        } else {
            float val[KV_MAX_KNOTS];
            kv_bspline_rt(kn, b.nk, b.order, xv, val, nullptr);
            for (int j = 0; j < b.G; ++j) dst[j * stride] = val[j];
        }
        if (b.has_base) dst[b.G * stride] = kv_silu(xv);
    } else if constexpr (FAM == KV_RBF) {
        for (int g = 0; g < b.G; ++g) {
            const float d = (uv - b.bp[g]) * b.inv_h;
            dst[g * stride] = __expf(-d * d);
        }
        if (b.has_base) dst[b.G * stride] = kv_silu(xv);
    } else if constexpr (FAM == KV_SINE) {
        const float* fr = b.bp;
        const float* ph = b.bp + b.G + (long long)i * b.G;
        for (int g = 0; g < b.G; ++g) dst[g * stride] = kv_sin(__fadd_rn(__fmul_rn(xv, fr[g]), ph[g]));
    } else if constexpr (FAM == KV_FOURIER) {
        float s1, c1;
        kv_sincos(xv, s1, c1);
        float ck = c1, sk = s1;
        for (int k = 0; k < b.G; ++k) {
            dst[k * stride] = ck;
            dst[(b.G + k) * stride] = sk;
            const float cn = ck * c1 - sk * s1;
            sk = sk * c1 + ck * s1;
            ck = cn;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward: dA[j*stride] = d loss / d phi_j for this (row, feature).  Returns d loss / d x in dx
// and (RBF only) d loss / d u in du.  SINE: dfreq[g] (LDS, one slot per wave, owned by the
// calling wave) accumulates d loss / d freq_g through a wave reduction -- deterministic order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float kv_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int FAM>
__device__ __forceinline__ void basis_bwd(const BasisArgs& b, float xv, float uv, int i, bool valid,
                                          const float* __restrict__ dA, int stride, float& dx, float& du,
                                          float* __restrict__ dfreq_wave) {
    dx = 0.0f;
    du = 0.0f;
    if constexpr (FAM == KV_LINEAR) {
        dx = dA[0];
    } else if constexpr (FAM == KV_CHEBY) {
        const float t = kv_tanh(xv);
        float u0 = 1.0f, u1 = 2.0f * t;   // U_0, U_1 (second kind): dT_g/dt = g*U_{g-1}
        float acc = (b.G > 1) ? dA[stride] : 0.0f;
        for (int g = 2; g < b.G; ++g) {
            acc += dA[g * stride] * ((float)g * u1);
            const float u2 = 2.0f * t * u1 - u0;
            u0 = u1;
            u1 = u2;
        }
        dx = acc * (1.0f - t * t);
    } else if constexpr (FAM == KV_BSPLINE) {
        const float* kn = b.bp + (long long)i * b.nk;
        float acc = 0.0f;
        if (b.uniform) {
            int j0;
            float bv[4], dv[4];
            if (kv_bspline_uniform(b.bp, b.nk, xv, j0, bv, dv, true)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = j0 - 3 + e;
                    if (idx >= 0 && idx < b.G) acc += dA[idx * stride] * dv[e];
                }
            }
        } else if (b.nk == 12 && b.order == 3) {
            float val[8], der[8];
            kv_bspline_fixed<12, 3, true>(kn, xv, val, der);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += dA[j * stride] * der[j];
        } else {
            float val[KV_MAX_KNOTS], der[KV_MAX_KNOTS];
            kv_bspline_rt(kn, b.nk, b.order, xv, val, der);
            for (int j = 0; j < b.G; ++j) acc += dA[j * stride] * der[j];
        }
        if (b.has_base) acc += dA[b.G * stride] * kv_dsilu(xv);
        dx = acc;
    } else if constexpr (FAM == KV_RBF) {
        float acc = 0.0f;
        for (int g = 0; g < b.G; ++g) {
            const float d = (uv - b.bp[g]) * b.inv_h;
            acc += dA[g * stride] * (__expf(-d * d) * (-2.0f * d * b.inv_h));
        }
        du = acc;
        if (b.has_base) dx = dA[b.G * stride] * kv_dsilu(xv);
    } else if constexpr (FAM == KV_SINE) {
        const float* fr = b.bp;
        const float* ph = b.bp + b.G + (long long)i * b.G;
        float acc = 0.0f;
        for (int g = 0; g < b.G; ++g) {
            const float f = fr[g];
            const float c = kv_cos(__fadd_rn(__fmul_rn(xv, f), ph[g]));
            const float da = valid ? dA[g * stride] : 0.0f;
            acc += da * c * f;
            const float part = kv_wave_sum(da * c * xv);     // all 64 lanes take part (uniform trip count)
            if ((threadIdx.x & 63) == 0) dfreq_wave[g] += part;
        }
        dx = acc;
    } else if constexpr (FAM == KV_FOURIER) {
        float s1, c1;
        kv_sincos(xv, s1, c1);
        float ck = c1, sk = s1, acc = 0.0f;
        for (int k = 0; k < b.G; ++k) {
            const float kf = (float)(k + 1);
            acc += kf * (dA[(b.G + k) * stride] * ck - dA[k * stride] * sk);   // d cos = -k sin, d sin = k cos
            const float cn = ck * c1 - sk * s1;
            sk = sk * c1 + ck * s1;
            ck = cn;
        }
        dx = acc;
    }
}


// ---------------------------------------------------------------------------------------------
// BasisGen with the per-lane constants hoisted: prepare() once per (lane, feature), then init()/next() per token touch no
// memory.  The streaming weight-gradient kernel evaluates the basis inside a software-pipelined token loop whose body is
// conditional, so the compiler cannot speculate the parameter loads (knots, centres, frequencies, phases) out of it; left
// in the loop they put an L1 round trip on the critical path of every step.
// ---------------------------------------------------------------------------------------------
template <int FAM, int GP, int J0C = -1>      // J0C >= 0: the window start is a compile-time constant (B-spline / RBF windows)
struct BasisGenP {
    static constexpr bool SINE_ = (FAM == KV_SINE || FAM == KV_SINE_DF);
    static constexpr int NC0 = SINE_ ? GP : 1;         // RBF: only the first centre (uniform grid, kv_rbf8)
    static constexpr int NC1 = SINE_ ? GP : 1;
    float c0[NC0], c1[NC1];
    float g0, ih, inv_h;
    int nkm1, G;
    float x, u, t, p0, p1;
    float bv[4];
    float pr[(FAM == KV_RBF) ? 8 : 1];      // RBF (uniform grid, host-checked): the eight values of this token (kv_rbf8)
    int j0;
    bool in;

    // `GP` here is the number of basis functions this generator produces: all of them, or a window [j0, j0 + GP) of a
    // wide basis (SINE / FOURIER with G = 28, contracted in several passes); next(j) takes the index INSIDE the window.
    int j0w;
    float s1, c1r, sk, ck;
    bool sin_half;
    __device__ __forceinline__ void prepare(const BasisArgs& b, int feat, int j0 = 0) {
        G = b.G;
        inv_h = b.inv_h;
        j0w = j0;
        if constexpr (SINE_) {
#pragma unroll
            for (int j = 0; j < GP; ++j) {
                c0[j] = b.bp[j0 + j];
                c1[j] = b.bp[b.G + (long long)feat * b.G + j0 + j];
            }
        } else if constexpr (FAM == KV_FOURIER) {       // window lies inside the cos block (j0 < G) or the sin block
            sin_half = j0 >= b.G;
        } else if constexpr (FAM == KV_RBF) {               // uniform 8-centre grid and j0 == 0 (host-checked: the register kernels' only RBF layout)
            c0[0] = b.bp[0];
        } else if constexpr (FAM == KV_BSPLINE) {       // uniform knots (host-checked)
            g0 = b.bp[0];
            nkm1 = b.nk - 1;
            ih = __frcp_rn((b.bp[nkm1] - g0) / (float)nkm1);
        }
    }
    __device__ __forceinline__ void init(float xv, float uv) {
        x = xv;
        u = uv;
        if constexpr (FAM == KV_RBF) kv_rbf8((uv - c0[0]) * inv_h, pr);
        if constexpr (FAM == KV_CHEBY) {
            t = kv_tanh(xv);
            p0 = 1.0f;
            p1 = t;
        } else if constexpr (FAM == KV_FOURIER) {       // cos / sin of k0*x directly (as the reference forms k*x), then rotate by x
            kv_sincos(xv, s1, c1r);
            const float k0 = (float)((sin_half ? j0w - G : j0w) + 1);
            kv_sincos(__fmul_rn(k0, xv), sk, ck);
        } else if constexpr (FAM == KV_BSPLINE) {       // same arithmetic as kv_bspline_uniform
            const float tt = (xv - g0) * ih;
            const float fl = floorf(tt);
            j0 = (int)fl;
            in = (tt >= 0.0f) && (j0 < nkm1);
            if (!in) j0 = KV_BSPLINE_OUT;
            const float uu = in ? tt - fl : 0.0f, u2 = uu * uu, u3 = u2 * uu, om = 1.0f - uu;      // finite also for NaN / infinite x (kv_bsel4 multiplies by masks)
            const float s6 = 1.0f / 6.0f;
            bv[0] = om * om * om * s6;
            bv[1] = (3.0f * u3 - 6.0f * u2 + 4.0f) * s6;
            bv[2] = (-3.0f * u3 + 3.0f * u2 + 3.0f * uu + 1.0f) * s6;
            bv[3] = u3 * s6;
        }
    }
    __device__ __forceinline__ float next(int j) {
        if constexpr (FAM == KV_LINEAR) {
            return x;
        } else if constexpr (FAM == KV_CHEBY) {
            if (j == 0) return 1.0f;
            if (j == 1) return t;
            const float p2 = 2.0f * t * p1 - p0;
            p0 = p1;
            p1 = p2;
            return p2;
        } else if constexpr (FAM == KV_BSPLINE) {
            const int jg = (J0C >= 0 ? J0C : j0w) + j;
            if (J0C >= 0 && jg > 8) return 0.0f;                  // idle slot of the last window (9 = 5 + 4)
            if (jg >= (J0C >= 0 ? 8 : G)) return kv_silu(x);      // compile-time windows: G = 8 (GP = 9 with the silu column; host-checked)
            return kv_bsel4(j0, jg, bv);
        } else if constexpr (FAM == KV_RBF) {
            const int jg = (J0C >= 0 ? J0C : j0w) + j;
            if (J0C >= 0 && jg > 8) return 0.0f;
            if (jg >= (J0C >= 0 ? 8 : G)) return kv_silu(x);
            return kv_sel8(pr, jg);            // compile-time jg: the unused Gaussians of a window are never computed (dead code)
        } else if constexpr (FAM == KV_FOURIER) {
            const float v = sin_half ? sk : ck;
            const float cn = ck * c1r - sk * s1;
            sk = sk * c1r + ck * s1;
            ck = cn;
            return v;
        } else if constexpr (FAM == KV_SINE_DF) {   // d sin(x f + p) / d f = x cos(x f + p): same argument arithmetic as BasisDGen<KV_SINE>
            return x * kv_cos(__fadd_rn(__fmul_rn(x, c0[j < NC0 ? j : 0]), c1[j < NC1 ? j : 0]));
        } else {   // SINE
            return kv_sin(__fadd_rn(__fmul_rn(x, c0[j < NC0 ? j : 0]), c1[j < NC1 ? j : 0]));
        }
    }
};

// SINE input gradient with the frequency-gradient partial sums kept in registers (G <= KV_SINE_REG_G): the caller
// wave-reduces dfq once per tile.  Same arithmetic as basis_bwd<KV_SINE>.
constexpr int KV_SINE_REG_G = 32;
__device__ __forceinline__ void basis_bwd_sine_reg(const BasisArgs& b, float xv, int i, const float* __restrict__ dA, int stride,
                                                   float& dx, float (&dfq)[KV_SINE_REG_G]) {
    const float* fr = b.bp;
    const float* ph = b.bp + b.G + (long long)i * b.G;
    float acc = 0.0f;
#pragma unroll
    for (int g = 0; g < KV_SINE_REG_G; ++g) {
        if (g < b.G) {
            const float f = fr[g];
            const float c = kv_cos(__fadd_rn(__fmul_rn(xv, f), ph[g]));
            const float da = dA[g * stride];
            acc += da * c * f;
            dfq[g] += da * c * xv;
        }
    }
    dx = acc;
}

// ---------------------------------------------------------------------------------------------
// Generator form of the same bases: init() once per (row, feature), then next(j) for j = 0 .. GP-1 in order.  Used by
// the register-operand forward kernel, where every lane feeds its MFMA A operand directly (no LDS basis tile), so the
// values must come out one at a time without runtime-indexed register arrays.
// ---------------------------------------------------------------------------------------------
// GC >= 0: the number of grid functions G is a compile-time constant (register kernels with a compile-time basis size: RBF / BSPLINE
// GP - 1 with the base column host-checked, FOURIER GP / 2, SINE GP).  With a runtime G every `j >= G` below is a wave-uniform
// BRANCH around its own copy of the silu code -- nine copies and nine branches per feature in the unrolled kernels (the FastKAN
// bf16 forward ran 34 VALU instructions per MFMA, profiles/r03_sq_pmc_fast_vits_bf16.md).
constexpr int kv_gc(int fam, int gp) {
    return gp <= 0 ? -1 : (fam == KV_RBF || fam == KV_BSPLINE) ? gp - 1 : (fam == KV_FOURIER ? gp / 2 : (fam == KV_SINE ? gp : -1));
}
template <int FAM, int GC = -1>
struct BasisGen {
    __device__ __forceinline__ int gcount() const { return GC >= 0 ? GC : G; }
    float x, u, t, p0, p1, c1, s1, ck, sk;
    float pr[(FAM == KV_RBF) ? 8 : 1];      // RBF (uniform grid, host-checked): kv_rbf8
    float bv[4];
    int j0, G, i;
    bool in;
    const float* bp;
    float inv_h;
    int has_base;

    __device__ __forceinline__ void init(const BasisArgs& b, float xv, float uv, int feat) {
        x = xv;
        u = uv;
        G = b.G;
        i = feat;
        bp = b.bp;
        inv_h = b.inv_h;
        has_base = b.has_base;
        if constexpr (FAM == KV_RBF) kv_rbf8((uv - b.bp[0]) * b.inv_h, pr);
        if constexpr (FAM == KV_CHEBY) {
            t = kv_tanh(xv);
            p0 = 1.0f;
            p1 = t;
        } else if constexpr (FAM == KV_BSPLINE) {
            float dv[4];
            in = kv_bspline_uniform(b.bp, b.nk, xv, j0, bv, dv, false);     // uniform knots only (host-checked)
            if (!in) j0 = KV_BSPLINE_OUT;
        } else if constexpr (FAM == KV_FOURIER) {
            kv_sincos(xv, s1, c1);
            ck = c1;
            sk = s1;
        }
    }
    __device__ __forceinline__ float next(int j) {
        if constexpr (FAM == KV_LINEAR) {
            return x;
        } else if constexpr (FAM == KV_CHEBY) {
            if (j == 0) return 1.0f;
            if (j == 1) return t;
            const float p2 = 2.0f * t * p1 - p0;
            p0 = p1;
            p1 = p2;
            return p2;
        } else if constexpr (FAM == KV_BSPLINE) {
            if (j >= gcount()) return kv_silu(x);
            return kv_bsel4(j0, j, bv);
        } else if constexpr (FAM == KV_RBF) {
            if (j >= gcount()) return kv_silu(x);
            return kv_sel8(pr, j);
        } else if constexpr (FAM == KV_SINE) {
            return kv_sin(__fadd_rn(__fmul_rn(x, bp[j]), bp[gcount() + (long long)i * gcount() + j]));
        } else {   // FOURIER: cos(k x) for j < G, then sin(k x)
            if (j == 0) return c1;
            if (j == gcount()) {
                ck = c1;
                sk = s1;
                return s1;
            }
            const float cn = ck * c1 - sk * s1;
            sk = sk * c1 + ck * s1;
            ck = cn;
            return j < gcount() ? ck : sk;
        }
    }
};


// Derivative generator: next(j) = d phi_j / d(input) for j = 0 .. GP-1 in order (RBF: d/du for j < G, d silu/dx for j = G).
template <int FAM, int GC = -1>
struct BasisDGen {
    __device__ __forceinline__ int gcount() const { return GC >= 0 ? GC : G; }
    float x, u, t, sech2, u0, u1, c1, s1, ck, sk, inv_h;
    float pr[(FAM == KV_RBF) ? 8 : 1];
    float lastc;   // SINE: cos(x f_j + p_ij) of the last next() (the caller needs it for d loss / d freq)
    float dv[4];
    int j0, G, i;
    bool in;
    const float* bp;

    __device__ __forceinline__ void init(const BasisArgs& b, float xv, float uv, int feat) {
        i = feat;
        x = xv;
        u = uv;
        G = b.G;
        bp = b.bp;
        inv_h = b.inv_h;
        if constexpr (FAM == KV_RBF) {
            t = (uv - b.bp[0]) * b.inv_h;
            kv_rbf8(t, pr);
        }
        if constexpr (FAM == KV_CHEBY) {
            t = kv_tanh(xv);
            sech2 = 1.0f - t * t;
            u0 = 1.0f;
            u1 = 2.0f * t;
        } else if constexpr (FAM == KV_BSPLINE) {
            float bv[4];
            in = kv_bspline_uniform(b.bp, b.nk, xv, j0, bv, dv, true);
            if (!in) j0 = KV_BSPLINE_OUT;
        } else if constexpr (FAM == KV_FOURIER) {
            kv_sincos(xv, s1, c1);
            ck = c1;
            sk = s1;
        }
    }
    __device__ __forceinline__ float next(int j) {
        if constexpr (FAM == KV_LINEAR) {
            return 1.0f;
        } else if constexpr (FAM == KV_CHEBY) {     // dT_j/dx = j * U_{j-1}(t) * (1 - t^2)
            if (j == 0) return 0.0f;
            if (j == 1) return sech2;
            const float r = (float)j * u1 * sech2;
            const float u2 = 2.0f * t * u1 - u0;
            u0 = u1;
            u1 = u2;
            return r;
        } else if constexpr (FAM == KV_BSPLINE) {
            if (j >= gcount()) return kv_dsilu(x);
            return kv_bsel4(j0, j, dv);
        } else if constexpr (FAM == KV_RBF) {
            if (j >= gcount()) return kv_dsilu(x);
            return kv_sel8(pr, j) * (-2.0f * (t - (float)j) * inv_h);
        } else if constexpr (FAM == KV_SINE) {      // d sin(x f + p)/dx = f cos(x f + p)
            const float f = bp[j];
            lastc = kv_cos(__fadd_rn(__fmul_rn(x, f), bp[gcount() + (long long)i * gcount() + j]));
            return lastc * f;
        } else if constexpr (FAM == KV_FOURIER) {   // d cos(kx) = -k sin(kx) for j < G, d sin(kx) = k cos(kx) after
            if (j == gcount()) {
                ck = c1;
                sk = s1;
            } else if (j != 0) {
                const float cn = ck * c1 - sk * s1;
                sk = sk * c1 + ck * s1;
                ck = cn;
            }
            const float kf = (float)((j < gcount() ? j : j - gcount()) + 1);
            return j < gcount() ? -kf * sk : kf * ck;
        } else {
            return 0.0f;                            // SINE is not handled by the register kernels (dfreq reduction)
        }
    }
};
