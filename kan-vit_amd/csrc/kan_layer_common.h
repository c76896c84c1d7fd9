// Internal header of the fused KAN layer kernels (csrc/kan_*.hip): launch arguments, tile constants, the plan structs the
// host entry points (kan_layer.hip) share with the kernel translation units, and the functions that cross between them.
// One translation unit per kernel generation keeps a rebuild at the size of the kernel that changed:
//     kan_layer.hip               C ABI entry points, validation, run-time switches
//     kan_tile.hip                general LDS-tile producer/consumer kernels (every shape; fp32 and bf16 contraction)
//     kan_fwd_reg.hip             register-form forward (fp32 exact; also the fused patch embedding)
//     kan_fwd_reg_bf16.hip        register-form and W-stationary forward on the bf16 matrix cores
//     kan_bwd_input_reg.hip       register-form input gradient (fp32 exact)
//     kan_bwd_input_reg_bf16.hip  register-form input gradient on the bf16 matrix cores
//     kan_bwd_weight_reg.hip      streaming register-form weight gradient + the ordered slab reduction
#pragma once
#include "kan_basis.h"

#include <type_traits>
#include "kanvit_common.h"

#include <stdlib.h>

constexpr int BM = 128;          // rows per block in fwd / bwd_input (4 consumer waves x 32 rows)
constexpr int NTHR = 512;        // 4 consumer (MFMA) waves + 4 producer (load / basis) waves
constexpr int NPROD = 256;       // producer threads
constexpr int AS = BM + 1;       // row stride of the K-major LDS tiles
constexpr int BIN_NC = 32;       // dY columns staged per step in bwd_input
constexpr int BW_ROWS = 32;      // rows staged per step in bwd_weight
constexpr int BW_AS = BW_ROWS + 1;
constexpr int BW_NT = 2;         // 64 output columns per bwd_weight block
constexpr int BW_TPW = 5;        // max 32x32 MFMA tiles per consumer wave in bwd_weight
constexpr int BW_KC_MAX = 288;   // (BW_TPW*4 tiles / BW_NT) * 32 = 320 >= 288
constexpr int N_CU = 256;

struct LayerArgs {
    const float* x;
    const float* u;
    const float* w;
    const float* bp;
    const float* bias;
    float* y;
    const float* dy;
    float* dx;
    float* du;
    float* dparam;
    float* slab;
    const unsigned short* wb;   // bf16 fragment-major repack of w (KANVIT_FLAG_BF16_MFMA)
    const unsigned short* wb2;  // bf16 repack for the input-gradient kernel: [g][chunk][O/8][KCT][8 n]
    long long M, ldx, ldu, ldy, bp_stride, rows_per_split;
    int I, O, groups, xmod, G, GP, order, nk, has_base, K, IC, msplit, nchunks_n;
    float rbf_inv_h;
    int flags;
    // patch gather (kanvit_patch_embed_*): x is an NCHW image batch, row m = (sample m / P, patch m % P), feature i = (c, iy, ix)
    // of the patch (model.py:111-126); y rows are shifted behind `pg_pre` prepended rows per sample (the class token), `pos`
    // ([P + pg_pre][O]) is added and the class-token row cls + pos[0] is written by the lanes that own a sample's first patch
    int pg, pg_C, pg_H, pg_W, pg_n, pg_pre;
    const float* cls;
    const float* pos;
    // KANVIT_FLAG_FUSED_LN (RBF): the spline-path input u = LayerNorm(x slice) * gamma + beta (models/fastkan.py:68) is formed
    // in the kernels; bparams of a group = [centres(G) | gamma(I) | beta(I)]; stats[M][xmod][2] = (mean, rstd) per row and x
    // slice, written by the forward kernel and read by the two backward kernels
    int vcols;            // bf16 input gradient of ONE wide layer (O = 64*v): the v column chunks run as "groups" sharing x, basis and chain rule
    int ln;
    float ln_eps;
    float* stats;
    // launch tail (kv_tail_tiles): row tiles >= tail_y0 are cut into smaller work-groups that sit at the END of the grid and
    // fill the wave slots the last whole work-groups leave idle (forward: one projection each; input gradient: one feature chunk each)
    int tail_y0;
};

// Row tiles [T1, T) of a launch of T row tiles x P work-groups per tile (two resident work-groups per CU) that should run as
// sub-divided work-groups.  Measured on ViT-B (197 tiles x 12 heads = 9.23 work-groups per CU): a CU runs its work-groups in
// pairs and a last single one at twice the speed, so the launch takes as long as 10 per CU although 60 of the 256 CUs get a
// tenth; cutting the tiles beyond 9 per CU into thirds / halves spreads that remainder over all CUs.
inline int kv_tail_first_tile(long long tiles, int per_tile) {
    const int cfg = kv_config().tail;
    if (cfg == 0 || tiles < 2) return (int)tiles;
    if (cfg > 0) return (int)(tiles - (cfg < tiles ? cfg : tiles - 1));
    const long long wgs = tiles * per_tile;
    const long long per_cu = wgs / N_CU;                      // whole work-groups every CU gets
    if (per_cu < 4) return (int)tiles;                       // short launches: the pieces' fixed costs outweigh the tail
    // the remainder beyond per_cu per CU, plus a quarter of a work-group per CU: the dispatcher hands work-groups to whichever CU
    // frees a slot, and a reserve of small pieces at the end evens out what that leaves uneven (measured: the launch time is flat
    // between 8 and 16 tail tiles of 12 and rises below 5; the pieces cost ~20 % more than the whole they replace)
    const long long rest = wgs - per_cu * N_CU + N_CU / 4;
    long long t1 = (wgs - rest) / per_tile;
    if (t1 < tiles / 2) t1 = tiles / 2;
    return (int)t1;
}

__device__ __forceinline__ BasisArgs make_basis(const LayerArgs& a, int g) {
    BasisArgs b;
    b.G = a.G;
    b.GP = a.GP;
    b.order = a.order;
    b.nk = a.nk;
    b.has_base = a.has_base;
    b.inv_h = a.rbf_inv_h;
    b.bp = a.bp ? a.bp + (long long)g * a.bp_stride : nullptr;
    b.uniform = (a.flags & KANVIT_FLAG_UNIFORM_KNOTS) && a.order == 3;
    return b;
}

// LayerNorm statistics of one row of I features held by a lane pair: this lane sees ICH consecutive features of every
// chunk of 2*ICH (xh points at its first one), its partner lane (l ^ 32) the others.  Two passes (mean, then centred
// sum of squares: the accuracy of torch's Welford kernel), biased variance, rstd = rsqrt(var + eps) as nn.LayerNorm.
template <int ICH>
__device__ __forceinline__ void kv_ln_row_stats(const float* __restrict__ xh, int nch, int I, float eps, float& mean, float& rstd) {
    float s = 0.0f;
    for (int c = 0; c < nch; ++c)
#pragma unroll
        for (int e = 0; e < ICH; ++e) s += xh[c * 2 * ICH + e];
    s += __shfl_xor(s, 32);
    mean = s / (float)I;
    float q = 0.0f;
    for (int c = 0; c < nch; ++c)
#pragma unroll
        for (int e = 0; e < ICH; ++e) {
            const float d = xh[c * 2 * ICH + e] - mean;
            q = fmaf(d, d, q);
        }
    q += __shfl_xor(q, 32);
    rstd = rsqrtf(q / (float)I + eps);
}

// Patch rows that exist only as an NCHW image batch (kanvit_patch_embed_*; model.py:111-126): row m = (image m / P, patch
// m % P).  The weight-gradient kernels visit the rows of a slab in order, a few per step, so the walker keeps the position
// of ONE row -- element offset of the patch origin (channel 0) from the image base, element offset of the row's dY row
// from the dY base (dY has pg_pre extra rows per image in front of the patch tokens) -- and advances it row by row: adds and
// compares on wave-uniform values (the scalar unit), no division after init().  Feature i = (c, iy, ix) of a patch adds
// the lane constant kv_patch_feature_offset().
struct PatchWalk {
    int px, py, xoff, dyoff;
    int n, pw, line_wrap, img_wrap, ldy, pre_ldy;
    __device__ __forceinline__ void init(const LayerArgs& a, int m) {
        n = a.pg_n;
        const int P = n * n, ph = a.pg_H / n;
        pw = a.pg_W / n;
        const int smp = m / P, pidx = m - smp * P;
        py = pidx / n;
        px = pidx - py * n;
        xoff = (smp * a.pg_C * a.pg_H + py * ph) * a.pg_W + px * pw;
        ldy = (int)a.ldy;
        pre_ldy = a.pg_pre * ldy;
        dyoff = (m + (smp + 1) * a.pg_pre) * ldy;
        line_wrap = ph * a.pg_W - n * pw;                       // last patch of a patch row -> first patch of the next one
        img_wrap = (a.pg_C * a.pg_H - n * ph) * a.pg_W;         // last patch of an image -> first patch of the next image
    }
    __device__ __forceinline__ void step() {      // branch-free: compares and selects on wave-uniform values (s_cmp / s_cselect)
        ++px;
        const bool w1 = px == n;                  // past the last patch of a patch row
        px = w1 ? 0 : px;
        py += w1 ? 1 : 0;
        const bool w2 = py == n;                  // past the last patch row of an image
        py = w2 ? 0 : py;
        xoff += pw + (w1 ? line_wrap : 0) + (w2 ? img_wrap : 0);
        dyoff += ldy + (w2 ? pre_ldy : 0);
    }
};

__device__ __forceinline__ int kv_patch_feature_offset(const LayerArgs& a, int f) {
    const int ph = a.pg_H / a.pg_n, pw = a.pg_W / a.pg_n;
    const int c = f / (ph * pw), r = f - c * (ph * pw), iy = r / pw, ix = r - iy * pw;
    return (c * a.pg_H + iy) * a.pg_W + ix;
}

__device__ __forceinline__ int kv_pow2_ge(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned kv_pack_bf16(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};        // hipcc -O3: one v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}


// ---------------------------------------------------------------------------------------------
// host side: what the translation units share
// ---------------------------------------------------------------------------------------------
inline int gp_of(const kanvit_layer_desc* d) {
    switch (d->family) {
        case KANVIT_LINEAR: return 1;
        case KANVIT_CHEBY: return d->G;
        case KANVIT_BSPLINE: return d->G + (d->has_base ? 1 : 0);
        case KANVIT_RBF: return d->G + (d->has_base ? 1 : 0);
        case KANVIT_SINE: return d->G;
        case KANVIT_FOURIER: return 2 * d->G;
        default: return -1;
    }
}

// RBF in the register kernels: only FastKAN's own uniform 8-centre grid (kv_rbf8: two exp anchors + recurrence); the caller
// vouches with KANVIT_FLAG_UNIFORM_KNOTS, anything else takes the LDS-tile kernels (direct exp per centre)
inline bool kv_rbf_reg_ok(int flags, int G) { return (flags & KANVIT_FLAG_UNIFORM_KNOTS) && G == 8; }

// families that get shared-basis (NSH = 3 / SHARED) kernel instantiations ...
template <int FAM>
constexpr bool kv_shared_basis() { return FAM == KV_LINEAR || FAM == KV_CHEBY || FAM == KV_FOURIER || FAM == KV_BSPLINE; }
// ... and whether a given launch may use them: parameter-free families always, BSPLINE when the caller vouches that the
// groups sharing x also share the knot table (KANVIT_FLAG_SHARED_BPARAMS)
inline bool kv_share_ok(int family, int flags) {
    return family == KANVIT_LINEAR || family == KANVIT_CHEBY || family == KANVIT_FOURIER ||
           (family == KANVIT_BSPLINE && (flags & KANVIT_FLAG_SHARED_BPARAMS));
}


#define KV_FAMILY_SWITCH(fam, CALL)                                   \
    switch (fam) {                                                    \
        case KANVIT_LINEAR: return CALL(KV_LINEAR);                   \
        case KANVIT_CHEBY: return CALL(KV_CHEBY);                     \
        case KANVIT_BSPLINE: return CALL(KV_BSPLINE);                 \
        case KANVIT_RBF: return CALL(KV_RBF);                         \
        case KANVIT_SINE: return CALL(KV_SINE);                       \
        case KANVIT_FOURIER: return CALL(KV_FOURIER);                 \
        default: return kv_fail(KANVIT_EINVAL, "unknown family %d", fam); \
    }

// ---- plans: pure host functions of the descriptor (the workspace queries and the launches must agree) ----
struct FwdRegBf16Plan {
    bool ok;
    int gp, nt, nsh, ich, vs, nch;
    size_t lds, ws_bytes;
};

struct FwdBf16Plan {
    bool ok;
    int ic, nt, nsh, kc, kcp, nch;
    size_t lds, ws_bytes;
};

struct BwdRegBf16Plan {
    bool ok;
    int gp, kt, fph, nci;
    int vcols;            // > 0: one wide layer (groups = 1, O = 64*vcols) contracted 64 columns at a time into the same accumulators
    size_t lds, ws_bytes;
};

struct BwPlan {
    int ic, nfchunks, nchunks_n, msplit, nsh;
    long long rows_per_split;
};

struct BwRegPlan {
    bool ok;
    int gp, nt, nfb, nos, tiles_per_bg, nbg, shared, slabs, njc;
    int t16;              // 1: the 16-row-tile kernel (nt = 16-column tiles per wave, nfb = 16-feature blocks)
    int bf;               // 1: the planned kernel contracts on the bf16 matrix cores (KANVIT_FLAG_BF16_MFMA allows it; the exact kernels ignore it)
    int dma;              // 1: the LDS-DMA form (kan_bwd_weight_dma.hip): a work-group = four row ranges of one wave unit, slabs counts WORK-GROUP slabs
    long long rows_per_slab;
    size_t ws_bytes;
};


// ---- kan_tile.hip: the general LDS-tile kernels -------------------------------------------------------------------
int kv_tile_fwd(int family, LayerArgs& a, hipStream_t st);
FwdBf16Plan plan_fwd_bf16(const kanvit_layer_desc* d);
int kv_tile_fwd_bf16(int family, LayerArgs& a, const FwdBf16Plan& p, void* ws, hipStream_t st);
int kv_tile_bwd_input(int family, LayerArgs& a, hipStream_t st);        // a.wb2 != nullptr: bf16 contraction (workspace = repacked W)
size_t kv_tile_bwd_input_ws(const kanvit_layer_desc* d);
BwPlan plan_bwd_weight(const kanvit_layer_desc* d);
int kv_tile_bwd_weight(int family, const LayerArgs& a, const BwPlan& p, bool bf, hipStream_t st);
// ---- register-form kernels: "try" functions return 1 when the shape is not covered (the caller falls back), 0 on success, < 0 on error
int kv_try_fwd_reg(int family, const LayerArgs& a, hipStream_t st);
FwdRegBf16Plan plan_fwd_reg_bf16(const kanvit_layer_desc* d);
int kv_fwd_reg_bf16(int family, LayerArgs& a, const FwdRegBf16Plan& p, void* ws, hipStream_t st);
int kv_try_bwd_input_reg(int family, const LayerArgs& a, hipStream_t st);
BwdRegBf16Plan plan_bwd_input_reg_bf16(const kanvit_layer_desc* d);
bool bwd_input_bf16_ok(const kanvit_layer_desc* d);
int kv_bwd_input_reg_bf16(int family, LayerArgs& a, const BwdRegBf16Plan& p, hipStream_t st);
BwRegPlan plan_bwd_weight_reg(const kanvit_layer_desc* d);
int kv_bwd_weight_reg(int family, LayerArgs& a, const BwRegPlan& p, bool bf, hipStream_t st);
int kv_slab_reduce(const float* slab, float* dw, long long total, int slabs, hipStream_t st);
bool kv_bwd_weight_reg_pg_ok(const kanvit_layer_desc* d, const BwRegPlan& p);      // the plan's kernel exists in the patch-gather form
int kv_bwd_weight_dma(int family, LayerArgs& a, const BwRegPlan& p, bool bf, hipStream_t st);      // kan_bwd_weight_dma.hip
bool kv_bwd_weight_dma_aligned(const LayerArgs& a);
