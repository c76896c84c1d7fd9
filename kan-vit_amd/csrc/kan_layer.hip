// Fused KAN layer kernels for gfx950 (MI355X): the C ABI entry points (include/kanvit.h), descriptor validation, kernel
// selection and the run-time switches.  The kernels themselves live in one translation unit per generation
// (kan_layer_common.h lists them); every family is
//     Y[M x O] = Phi(X)[M x K] . W[K x O],   K = I*GP,  Phi generated on the fly from X[M x I]
// so the three operations are GEMMs whose generated operand never exists in HBM:
//     fwd         Y    = Phi(X) . W
//     bwd_input   dPhi = dY . W^T, then dX = sum_j dPhi_j * phi_j'(X)
//     bwd_weight  dW   = Phi(X)^T . dY          (split over row ranges -> slabs -> ordered reduce)
// Selection order per call: tiny per-head layers (kan_tiny.hip) -> register-form kernels (the shapes the reference
// instantiates on its 224x224 path) -> the general LDS-tile kernels (every other shape).
#include "kan_layer_common.h"

extern "C" int kanvit_layer_ln_fusable(const kanvit_layer_desc* d);

namespace {

int validate(const kanvit_layer_desc* d, const char* who) {
    if (!d) return kv_fail(KANVIT_EINVAL, "%s: null descriptor", who);
    const int gp = gp_of(d);
    if (gp < 1) return kv_fail(KANVIT_EINVAL, "%s: unknown family %d or G=%d", who, d->family, d->G);
    if (gp > 80) return kv_fail(KANVIT_EINVAL, "%s: %d generated columns per feature exceeds the supported 80", who, gp);
    if (d->groups < 1 || d->x_group_mod < 1 || d->groups % d->x_group_mod != 0)
        return kv_fail(KANVIT_EINVAL, "%s: groups=%d must be a positive multiple of x_group_mod=%d", who, d->groups,
                       d->x_group_mod);
    if (d->groups > 65535) return kv_fail(KANVIT_EINVAL, "%s: groups=%d exceeds 65535", who, d->groups);
    if ((d->M + BM - 1) / BM > 65535) return kv_fail(KANVIT_EINVAL, "%s: M=%lld exceeds %d rows per launch", who,
                                                     (long long)d->M, 65535 * BM);
    if (d->I < 1 || d->O < 1 || d->M < 0) return kv_fail(KANVIT_EINVAL, "%s: bad sizes M=%lld I=%d O=%d", who,
                                                          (long long)d->M, d->I, d->O);
    if ((long long)d->I * gp > 0x7fffffffLL / 4) return kv_fail(KANVIT_EINVAL, "%s: K too large", who);
    if (d->ldx < (int64_t)d->x_group_mod * d->I) return kv_fail(KANVIT_EINVAL, "%s: ldx=%lld < x_group_mod*I", who,
                                                                 (long long)d->ldx);
    if (d->ldy < (int64_t)d->groups * d->O) return kv_fail(KANVIT_EINVAL, "%s: ldy=%lld < groups*O", who,
                                                            (long long)d->ldy);
    if (d->family == KANVIT_BSPLINE) {
        const int nk = d->G + d->spline_order + 1;
        if (d->spline_order < 0 || d->G < 1 || nk > KV_MAX_KNOTS)
            return kv_fail(KANVIT_EINVAL, "%s: bspline G=%d order=%d unsupported (knots %d > %d)", who, d->G,
                           d->spline_order, nk, KV_MAX_KNOTS);
        if (d->bparam_stride < (int64_t)d->I * nk) return kv_fail(KANVIT_EINVAL, "%s: bparam_stride too small", who);
    }
    if (d->family == KANVIT_RBF && d->bparam_stride < d->G) return kv_fail(KANVIT_EINVAL, "%s: bparam_stride too small", who);
    if ((d->flags & KANVIT_FLAG_SINE_DFREQ) && d->family != KANVIT_SINE)
        return kv_fail(KANVIT_EINVAL, "%s: KANVIT_FLAG_SINE_DFREQ is a SINE flag", who);
    if (d->flags & KANVIT_FLAG_FUSED_LN) {
        if (d->family != KANVIT_RBF) return kv_fail(KANVIT_EINVAL, "%s: KANVIT_FLAG_FUSED_LN is an RBF (FastKAN) flag", who);
        if (d->bparam_stride < (int64_t)d->G + 2 * (int64_t)d->I)
            return kv_fail(KANVIT_EINVAL, "%s: KANVIT_FLAG_FUSED_LN needs bparams = [centres(G) | gamma(I) | beta(I)] per group", who);
        if (!kanvit_layer_ln_fusable(d)) return kv_fail(KANVIT_EINVAL, "%s: shape not covered by the register kernels that fuse the LayerNorm", who);
    }
    if (d->family == KANVIT_SINE && d->bparam_stride < (int64_t)d->G * (1 + d->I))
        return kv_fail(KANVIT_EINVAL, "%s: bparam_stride too small", who);
    return 0;
}

LayerArgs base_args(const kanvit_layer_desc* d) {
    LayerArgs a{};
    a.M = d->M;
    a.ldx = d->ldx;
    a.ldu = d->ldu;
    a.ldy = d->ldy;
    a.bp_stride = d->bparam_stride;
    a.I = d->I;
    a.O = d->O;
    a.groups = d->groups;
    a.xmod = d->x_group_mod;
    a.G = d->G;
    a.GP = gp_of(d);
    a.order = d->spline_order;
    a.nk = d->G + d->spline_order + 1;
    a.has_base = d->has_base;
    a.K = d->I * a.GP;
    a.rbf_inv_h = d->rbf_inv_h;
    a.flags = d->flags;
    a.ln = (d->family == KANVIT_RBF && (d->flags & KANVIT_FLAG_FUSED_LN)) ? 1 : 0;
    a.ln_eps = d->ln_eps;
    a.tail_y0 = 0x7fffffff;             // no sub-divided tail unless a launcher sets one (kv_tail_first_tile)
    return a;
}

int needs_bparams(int family) { return family == KANVIT_BSPLINE || family == KANVIT_RBF || family == KANVIT_SINE; }

// forward / input gradient: register-form kernel when the shape allows it, else the LDS-tile kernel
int dispatch_fwd(int family, LayerArgs& a, hipStream_t st) {
    const int rc = kv_try_fwd_reg(family, a, st);
    if (rc <= 0) return rc;
    if (a.ln) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: KANVIT_FLAG_FUSED_LN needs the register kernel (alignment / shape)");
    return kv_tile_fwd(family, a, st);
}

int dispatch_bwd_input(int family, LayerArgs& a, hipStream_t st) {
    if ((long long)BM * a.ldx >= (1LL << 30) || (long long)BM * a.ldy >= (1LL << 30) || (long long)BM * a.ldu >= (1LL << 30) ||
        (long long)a.K * a.O >= (1LL << 30))
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: row strides / weight slab too large for 32-bit tile offsets");
    if (!a.wb2) {                                       // wb2 is set by the entry point when the bf16 path applies
        const int rc = kv_try_bwd_input_reg(family, a, st);
        if (rc <= 0) return rc;
        if (a.ln) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: KANVIT_FLAG_FUSED_LN needs the register kernel (alignment / shape)");
    }
    return kv_tile_bwd_input(family, a, st);
}

}  // namespace

static KvTinyArgs tiny_args(const kanvit_layer_desc* d) {
    KvTinyArgs t{};
    t.M = d->M; t.ldx = d->ldx; t.ldy = d->ldy; t.bp_stride = d->bparam_stride;
    t.family = d->family; t.I = d->I; t.O = d->O; t.groups = d->groups; t.xmod = d->x_group_mod; t.G = d->G;
    t.GP = gp_of(d); t.K = d->I * t.GP; t.order = d->spline_order; t.nk = d->G + d->spline_order + 1;
    t.has_base = d->has_base; t.flags = d->flags;
    return t;
}

thread_local char g_kanvit_err[512] = "";

// ---- run-time switches: read once, reported, never consulted through getenv on the launch path ----
static KvConfig g_kv_config;
static int g_kv_config_state = 0;      // 0 = not loaded
static void kv_config_load() {
    KvConfig c{};
    auto flag = [](const char* n) { const char* v = getenv(n); return (v && *v && !(v[0] == '0' && !v[1])) ? 1 : 0; };
    auto num = [](const char* n) { const char* v = getenv(n); return v ? atoi(v) : 0; };
    c.no_reg = flag("KANVIT_NO_REG");
    c.no_reg_bw = flag("KANVIT_NO_REG_BW");
    c.bw_no_t16 = flag("KANVIT_BW_NO_T16");
    c.bw_no_dma = flag("KANVIT_BW_NO_DMA");
    c.bi_no_res = flag("KANVIT_BI_NO_RES");
    c.bw_dma_force = flag("KANVIT_BW_DMA_FORCE");
    c.ws_no_strip = flag("KANVIT_WS_NO_STRIP");
    c.no_fast = flag("KANVIT_NO_FAST");
    c.no_pipe = flag("KANVIT_NO_PIPE");
    c.no_ws = flag("KANVIT_NO_WS");
    c.no_bf16 = flag("KANVIT_NO_BF16");
    c.no_fused_ln = flag("KANVIT_NO_FUSED_LN");
    c.no_tiny = flag("KANVIT_NO_TINY");
    c.attn_v1 = flag("KANVIT_ATTN_V1");
    c.attn_v2 = flag("KANVIT_ATTN_V2");
    c.attn_v3 = flag("KANVIT_ATTN_V3");
    c.attn_v4 = flag("KANVIT_ATTN_V4");
    c.attn_no_ds = flag("KANVIT_ATTN_NO_DS");
    c.attn_grid = num("KANVIT_ATTN_GRID");
    c.ff_grid = num("KANVIT_FF_GRID");
    c.bf16_nsh = num("KANVIT_BF16_NSH");
    c.bf16_ic = num("KANVIT_BF16_IC");
    c.bs_bw_bf16 = num("KANVIT_BSPLINE_BW_BF16");
    c.tail = getenv("KANVIT_TAIL") ? atoi(getenv("KANVIT_TAIL")) : -1;
    snprintf(c.text, sizeof(c.text),
             "no_reg=%d no_reg_bw=%d bw_no_t16=%d no_fast=%d no_pipe=%d no_ws=%d no_bf16=%d no_fused_ln=%d no_tiny=%d attn_v1=%d attn_v2=%d attn_v3=%d attn_v4=%d attn_no_ds=%d attn_grid=%d bf16_nsh=%d bf16_ic=%d ff_grid=%d bs_bw_bf16=%d tail=%d bw_no_dma=%d bi_no_res=%d bw_dma_force=%d ws_no_strip=%d",
             c.no_reg, c.no_reg_bw, c.bw_no_t16, c.no_fast, c.no_pipe, c.no_ws, c.no_bf16, c.no_fused_ln, c.no_tiny, c.attn_v1, c.attn_v2, c.attn_v3, c.attn_v4, c.attn_no_ds, c.attn_grid, c.bf16_nsh, c.bf16_ic, c.ff_grid, c.bs_bw_bf16, c.tail, c.bw_no_dma, c.bi_no_res, c.bw_dma_force, c.ws_no_strip);
    g_kv_config = c;
    __atomic_store_n(&g_kv_config_state, 1, __ATOMIC_RELEASE);
}
const KvConfig& kv_config() {
    if (!__atomic_load_n(&g_kv_config_state, __ATOMIC_ACQUIRE)) kv_config_load();     // idempotent: a race loads the same values twice
    return g_kv_config;
}

extern "C" {

const char* kanvit_last_error(void) { return g_kanvit_err; }
const char* kanvit_config(void) { return kv_config().text; }
int kanvit_config_reload(void) {
    kv_config_load();
    return 0;
}
int kanvit_abi_version(void) { return KANVIT_ABI_VERSION; }
int kanvit_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return kv_fail(KANVIT_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

size_t kanvit_layer_fwd_workspace(const kanvit_layer_desc* d) {
    if (!d || !(d->flags & KANVIT_FLAG_BF16_MFMA) || gp_of(d) < 1 || d->groups < 1 || d->x_group_mod < 1 || d->I < 1 || d->O < 1)
        return 0;
    const FwdRegBf16Plan pr = plan_fwd_reg_bf16(d);
    const FwdBf16Plan p = plan_fwd_bf16(d);          // fallback when the register kernel's alignment checks fail at launch
    const size_t a1 = pr.ok ? pr.ws_bytes : 0, a2 = p.ok ? p.ws_bytes : 0;
    return a1 > a2 ? a1 : a2;
}

int kanvit_layer_fwd(const kanvit_layer_desc* d, const float* x, const float* u, const float* w, const float* bparams,
                     const float* bias, float* y, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = validate(d, "kanvit_layer_fwd")) return rc;
    if (d->flags & KANVIT_FLAG_SINE_DFREQ) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: KANVIT_FLAG_SINE_DFREQ is a kanvit_layer_bwd_weight flag");
    if (d->M == 0) return 0;
    if (!x || !w || !y) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: null x/w/y");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: family %d needs bparams", d->family);
    if (d->family == KANVIT_RBF && u && d->ldu < (int64_t)d->groups * d->I)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: ldu < groups*I");
    if (d->M == 0) return 0;
    if (kv_tiny_ok(d)) {              // tiny per-head layers (I, O <= 16): vector-pipe kernels, csrc/kan_tiny.hip
        KvTinyArgs t = tiny_args(d);
        t.x = x; t.w = w; t.bp = bparams; t.bias = bias; t.y = y;
        return kv_tiny_fwd(t, (hipStream_t)stream);
    }
    LayerArgs a = base_args(d);
    a.x = x;
    a.u = u;
    a.w = w;
    a.bp = bparams;
    a.bias = bias;
    a.y = y;
    if (a.ln) {                       // the u slot carries the statistics buffer [M][x_group_mod][2] (written here)
        if (!u || ((uintptr_t)u & 7)) return kv_fail(KANVIT_EINVAL, "kanvit_layer_fwd: KANVIT_FLAG_FUSED_LN needs the (8-byte aligned) statistics buffer in the u argument");
        a.stats = const_cast<float*>(u);
        a.u = nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    if ((d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16) {
        const FwdRegBf16Plan pr = plan_fwd_reg_bf16(d);
        if (pr.ok && !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)(u ? u : x) | (uintptr_t)(bias ? bias : x) | (uintptr_t)(bparams ? bparams : x)) & 15)) {
            if (!workspace || workspace_bytes < pr.ws_bytes || ((uintptr_t)workspace & 15))
                return kv_fail(KANVIT_ENOMEM, "kanvit_layer_fwd: workspace %zu bytes < required %zu (or not 16-byte aligned)",
                               workspace_bytes, pr.ws_bytes);
            return kv_fwd_reg_bf16(d->family, a, pr, workspace, st);
        }
        const FwdBf16Plan p = plan_fwd_bf16(d);
        if (p.ok && !a.ln) {
            if (!workspace || workspace_bytes < p.ws_bytes || ((uintptr_t)workspace & 15))
                return kv_fail(KANVIT_ENOMEM, "kanvit_layer_fwd: workspace %zu bytes < required %zu (or not 16-byte aligned)",
                               workspace_bytes, p.ws_bytes);
            return kv_tile_fwd_bf16(d->family, a, p, workspace, st);
        }
    }
    return dispatch_fwd(d->family, a, st);
}

/* 1 when the three register kernels that can form the FastKAN LayerNorm in-kernel cover this layer (pure host function) */
int kanvit_layer_ln_fusable(const kanvit_layer_desc* d) {
    if (!d || d->family != KANVIT_RBF || !d->has_base || !kv_rbf_reg_ok(d->flags, d->G) || d->groups < 1 || d->x_group_mod < 1) return 0;
    if (d->I % 32 || d->O % 32 || d->M < 256 || (d->ldx & 3) || (d->ldy & 3)) return 0;
    if (d->O > 64 && d->O % 128) return 0;                 // forward column tiling: 32, 64 or multiples of 128
    {                                                      // kanvit_layer_ln_bwd's lane-group layout
        const int ns = d->groups / d->x_group_mod;
        if (d->groups % d->x_group_mod || (ns != 1 && ns != 3) || d->I > (ns == 3 ? 512 : 1024)) return 0;
    }
    if (kv_config().no_reg || kv_config().no_reg_bw || kv_config().no_fused_ln) return 0;
    kanvit_layer_desc e = *d;
    e.flags |= KANVIT_FLAG_FUSED_LN;
    if (!plan_bwd_weight_reg(&e).ok) return 0;
    if ((d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16 && !plan_fwd_reg_bf16(&e).ok) return 0;
    return 1;
}

// ---- fused patch embedding (model.py:111-126 patchify, :144-152 class token + position embedding) ----
static int patch_validate(const kanvit_layer_desc* d, const kanvit_patch_desc* p, const char* who) {
    if (!p) return kv_fail(KANVIT_EINVAL, "%s: null patch descriptor", who);
    if (p->C < 1 || p->H < 1 || p->W < 1 || p->n_patches < 1 || p->H % p->n_patches || p->W % p->n_patches)
        return kv_fail(KANVIT_EINVAL, "%s: image %dx%dx%d is not divisible into %d x %d patches", who, p->C, p->H, p->W, p->n_patches,
                       p->n_patches);
    if (p->prepend_rows != 0 && p->prepend_rows != 1) return kv_fail(KANVIT_EINVAL, "%s: prepend_rows must be 0 or 1", who);
    const long long P = (long long)p->n_patches * p->n_patches, I = (long long)p->C * (p->H / p->n_patches) * (p->W / p->n_patches);
    if (d->groups != 1 || d->x_group_mod != 1) return kv_fail(KANVIT_EINVAL, "%s: one layer per launch (groups = 1)", who);
    if (d->I != I) return kv_fail(KANVIT_EINVAL, "%s: I=%d but a patch has %lld pixels", who, d->I, I);
    if (d->M % P) return kv_fail(KANVIT_EINVAL, "%s: M=%lld is not a whole number of images (%lld patches each)", who, (long long)d->M, P);
    if ((long long)p->C * p->H * p->W >= (1LL << 30)) return kv_fail(KANVIT_EINVAL, "%s: image too large for 32-bit offsets", who);
    if (d->family == KANVIT_RBF) return kv_fail(KANVIT_EINVAL, "%s: FastKAN needs its LayerNorm'ed input u (use kanvit_layer_fwd)", who);
    return 0;
}

static void patch_args(LayerArgs& a, const kanvit_patch_desc* p, const float* cls, const float* pos) {
    a.pg = 1;
    a.pg_C = p->C;
    a.pg_H = p->H;
    a.pg_W = p->W;
    a.pg_n = p->n_patches;
    a.pg_pre = p->prepend_rows;
    a.cls = cls;
    a.pos = pos;
    a.ldx = a.I;                         // unused by the gather; keeps the alignment checks of the dispatcher meaningful
}

int kanvit_patch_embed_fwd_ws(const kanvit_layer_desc* d, const kanvit_patch_desc* p, const float* images, const float* w,
                              const float* bparams, const float* bias, const float* cls, const float* pos, float* y,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = validate(d, "kanvit_patch_embed_fwd")) return rc;
    if (int rc = patch_validate(d, p, "kanvit_patch_embed_fwd")) return rc;
    if (d->M == 0) return 0;
    if (!images || !w || !y) return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: null images/w/y");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: family %d needs bparams", d->family);
    if (p->prepend_rows && !cls) return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: prepend_rows = 1 needs the class token");
    if (((uintptr_t)images | (uintptr_t)(cls ? cls : w) | (uintptr_t)(pos ? pos : w)) & 15)
        return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: images / cls / pos must be 16-byte aligned");
    LayerArgs a = base_args(d);
    a.x = images;
    a.w = w;
    a.bp = bparams;
    a.bias = bias;
    a.y = y;
    patch_args(a, p, cls, pos);
    hipStream_t st = (hipStream_t)stream;
    int rc = 1;
    if ((d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16) {
        // bf16 matrix cores: the register-form forward in its patch form (one wide layer: four column tiles per work-group; a
        // lane's feature chunk must be consecutive pixels of one image line)
        const FwdRegBf16Plan pr = plan_fwd_reg_bf16(d);
        const int pw = p->W / p->n_patches;
        if (pr.ok && pr.nt == 4 && pr.nsh == 1 && d->family != KANVIT_RBF && d->family != KANVIT_LINEAR && pw % (2 * pr.ich) == 0 &&
            (pr.ich < 4 || (p->W & 3) == 0) && !(((uintptr_t)y | (uintptr_t)(bias ? bias : w) | (uintptr_t)(bparams ? bparams : w)) & 15)) {
            if (!workspace || workspace_bytes < pr.ws_bytes || ((uintptr_t)workspace & 15))
                return kv_fail(KANVIT_ENOMEM, "kanvit_patch_embed_fwd: workspace %zu bytes < required %zu (or not 16-byte aligned)",
                               workspace_bytes, pr.ws_bytes);
            rc = kv_fwd_reg_bf16(d->family, a, pr, workspace, st);
        }
    } else {
        rc = (kv_config().no_reg || d->family == KANVIT_RBF) ? 1 : kv_try_fwd_reg(d->family, a, st);
    }
    if (rc == 1)
        return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: shape not covered by the fused kernel (O %% 32, patch width %% chunk, "
                                      "basis size); use patchify + kanvit_layer_fwd");
    return rc;
}

int kanvit_patch_embed_fwd(const kanvit_layer_desc* d, const kanvit_patch_desc* p, const float* images, const float* w,
                           const float* bparams, const float* bias, const float* cls, const float* pos, float* y, void* stream) {
    if (d && (d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16)
        return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_fwd: KANVIT_FLAG_BF16_MFMA needs a workspace (kanvit_patch_embed_fwd_ws)");
    return kanvit_patch_embed_fwd_ws(d, p, images, w, bparams, bias, cls, pos, y, nullptr, 0, stream);
}

// Weight gradient of the patch-embedding layer with the same gather: x rows from the NCHW images, dY rows from the token-sequence
// gradient [B][P + prepend_rows][ldy] (the class-token rows are stepped over) -- no transient patch matrix, no dY copy.
int kanvit_patch_embed_bwd_weight_ok(const kanvit_layer_desc* d, const kanvit_patch_desc* p) {
    if (!d || !p || gp_of(d) < 1 || d->groups != 1 || d->x_group_mod != 1 || d->I < 1 || d->O < 1 || d->M < 1) return 0;
    if (p->C < 1 || p->H < 1 || p->W < 1 || p->n_patches < 1 || p->H % p->n_patches || p->W % p->n_patches) return 0;
    if (p->prepend_rows != 0 && p->prepend_rows != 1) return 0;
    const long long P = (long long)p->n_patches * p->n_patches;
    if (d->I != (long long)p->C * (p->H / p->n_patches) * (p->W / p->n_patches) || d->M % P) return 0;
    const long long B = d->M / P;
    // 32-bit element offsets from the image base / the dY base
    if (B * p->C * p->H * p->W >= (1LL << 31) || (d->M + B * p->prepend_rows + 1) * d->ldy >= (1LL << 31)) return 0;
    if (d->family == KANVIT_RBF || kv_tiny_ok(d)) return 0;
    return kv_bwd_weight_reg_pg_ok(d, plan_bwd_weight_reg(d)) ? 1 : 0;
}

size_t kanvit_patch_embed_bwd_weight_workspace(const kanvit_layer_desc* d, const kanvit_patch_desc* p) {
    if (!kanvit_patch_embed_bwd_weight_ok(d, p)) return 0;
    return plan_bwd_weight_reg(d).ws_bytes;
}

int kanvit_patch_embed_bwd_weight(const kanvit_layer_desc* d, const kanvit_patch_desc* p, const float* images, const float* bparams,
                                  const float* dy, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = validate(d, "kanvit_patch_embed_bwd_weight")) return rc;
    if (int rc = patch_validate(d, p, "kanvit_patch_embed_bwd_weight")) return rc;
    if (!dw) return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_bwd_weight: null dw");
    if (d->M == 0) {
        KV_HIP_CHECK(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->I * gp_of(d) * d->O, (hipStream_t)stream));
        return 0;
    }
    if (!images || !dy) return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_bwd_weight: null images/dy");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_bwd_weight: family %d needs bparams", d->family);
    if ((d->flags & KANVIT_FLAG_SINE_DFREQ) && !kanvit_layer_sine_dfreq_ok(d))     // as kanvit_layer_bwd_weight: never hand plain dW back as Q
        return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_bwd_weight: KANVIT_FLAG_SINE_DFREQ is not available for this layer (kanvit_layer_sine_dfreq_ok)");
    if (!kanvit_patch_embed_bwd_weight_ok(d, p))
        return kv_fail(KANVIT_EINVAL, "kanvit_patch_embed_bwd_weight: layer / geometry not covered by the gathering weight-gradient kernels "
                                      "(kanvit_patch_embed_bwd_weight_ok); use patchify + kanvit_layer_bwd_weight");
    const BwRegPlan pr = plan_bwd_weight_reg(d);
    if (pr.ws_bytes > 0 && (!workspace || workspace_bytes < pr.ws_bytes))
        return kv_fail(KANVIT_ENOMEM, "kanvit_patch_embed_bwd_weight: workspace %zu bytes < required %zu", workspace_bytes, pr.ws_bytes);
    LayerArgs a = base_args(d);
    a.x = images;
    a.bp = bparams;
    a.dy = dy;
    patch_args(a, p, nullptr, nullptr);
    hipStream_t st = (hipStream_t)stream;
    const bool bf = pr.bf != 0;
    a.rows_per_split = pr.rows_per_slab;
    a.msplit = pr.slabs;
    a.slab = (pr.slabs > 1) ? (float*)workspace : dw;
    if (int rc = kv_bwd_weight_reg(d->family, a, pr, bf, st)) return rc;
    if (pr.slabs > 1) return kv_slab_reduce((const float*)workspace, dw, (long long)a.K * d->O, pr.slabs, st);
    return 0;
}

int kanvit_layer_sine_dfreq_ok(const kanvit_layer_desc* d) {
    if (!d || d->family != KANVIT_SINE || gp_of(d) < 1 || d->groups < 1 || d->x_group_mod < 1 || d->I < 1 || d->O < 1 || d->M < 1) return 0;
    if (kv_tiny_ok(d)) return 0;
    return plan_bwd_weight_reg(d).ok ? 1 : 0;
}

int64_t kanvit_layer_dparam_tiles(const kanvit_layer_desc* d) {
    if (!d || d->family != KANVIT_SINE) return 0;
    return (d->M + BM - 1) / BM;
}

size_t kanvit_layer_bwd_input_workspace(const kanvit_layer_desc* d) {
    if (!d || gp_of(d) < 1 || d->groups < 1 || d->x_group_mod < 1 || d->I < 1 || d->O < 1 || !bwd_input_bf16_ok(d)) return 0;
    size_t reg_ws = 0;
    {
        const BwdRegBf16Plan pr = plan_bwd_input_reg_bf16(d);
        if (pr.ok) reg_ws = pr.ws_bytes;
    }
    const size_t lds_ws = kv_tile_bwd_input_ws(d);
    return reg_ws > lds_ws ? reg_ws : lds_ws;
}

int kanvit_layer_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,
                           const float* bparams, const float* dy, float* dx, float* du, float* dparam, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (int rc = validate(d, "kanvit_layer_bwd_input")) return rc;
    if (d->M == 0) return 0;
    if (!x || !w || !dy || !dx) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: null x/w/dy/dx");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: family %d needs bparams", d->family);
    if (d->family == KANVIT_SINE && !dparam) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: SINE needs dparam");
    if (d->flags & KANVIT_FLAG_SINE_DFREQ) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: KANVIT_FLAG_SINE_DFREQ is a kanvit_layer_bwd_weight flag");
    if (d->family == KANVIT_RBF && (u || du) && d->ldu < (int64_t)d->groups * d->I)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: ldu < groups*I");
    if (d->family == KANVIT_SINE && (d->groups / d->x_group_mod) * 4 * d->G > 4096)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: SINE G too large");
    if (d->M == 0) return 0;
    if (kv_tiny_ok(d)) {
        KvTinyArgs t = tiny_args(d);
        t.x = x; t.w = w; t.bp = bparams; t.dy = dy; t.dx = dx;
        return kv_tiny_bwd_input(t, (hipStream_t)stream);
    }
    LayerArgs a = base_args(d);
    a.x = x;
    a.u = u;
    a.w = w;
    a.bp = bparams;
    a.dy = dy;
    a.dx = dx;
    a.du = du;
    a.dparam = dparam;
    a.wb2 = nullptr;
    if (a.ln) {
        if (!u) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_input: KANVIT_FLAG_FUSED_LN needs the statistics buffer in the u argument");
        a.stats = const_cast<float*>(u);
        a.u = nullptr;
    }
    if (bwd_input_bf16_ok(d) && (((uintptr_t)dy & 15) == 0)) {
        const size_t need = kanvit_layer_bwd_input_workspace(d);
        if (need) {
            if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15))
                return kv_fail(KANVIT_ENOMEM, "kanvit_layer_bwd_input: workspace %zu bytes < required %zu (or not 16-byte aligned)",
                               workspace_bytes, need);
            a.wb2 = (const unsigned short*)workspace;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (a.wb2) {
        const BwdRegBf16Plan pr = plan_bwd_input_reg_bf16(d);
        if (pr.ok && !(((uintptr_t)x | (uintptr_t)dx | (uintptr_t)dy | (uintptr_t)(u ? u : x) | (uintptr_t)(du ? du : dx)) & 15)) {
            return kv_bwd_input_reg_bf16(d->family, a, pr, st);
        }
        if (a.ln || d->O > 64 || d->family == KANVIT_SINE) a.wb2 = nullptr;     // no LayerNorm fusion / no wide layers in the LDS-tile bf16 kernel: the exact register kernel runs instead
    }
    return dispatch_bwd_input(d->family, a, st);
}

size_t kanvit_layer_bwd_weight_workspace(const kanvit_layer_desc* d) {
    if (!d || gp_of(d) < 1 || d->groups < 1 || d->I < 1 || d->O < 1) return 0;
    if (kv_tiny_ok(d)) {
        const int s = kv_tiny_slabs(d);
        return s > 1 ? sizeof(float) * (size_t)s * d->groups * ((size_t)d->I * gp_of(d)) * d->O : 0;
    }
    const BwRegPlan pr = plan_bwd_weight_reg(d);
    if (pr.ok) return pr.ws_bytes;
    const BwPlan p = plan_bwd_weight(d);
    if (p.msplit <= 1) return 0;
    return sizeof(float) * (size_t)p.msplit * d->groups * ((size_t)d->I * gp_of(d)) * d->O;
}

int kanvit_layer_bwd_weight(const kanvit_layer_desc* d, const float* x, const float* u, const float* bparams,
                            const float* dy, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = validate(d, "kanvit_layer_bwd_weight")) return rc;
    if (d->M == 0) {   // no rows: the gradient is exactly zero
        if (!dw) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: null dw");
        KV_HIP_CHECK(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->groups * d->I * gp_of(d) * d->O, (hipStream_t)stream));
        return 0;
    }
    if (!x || !dy || !dw) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: null x/dy/dw");
    if (needs_bparams(d->family) && !bparams) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: family %d needs bparams", d->family);
    if (d->family == KANVIT_RBF && u && d->ldu < (int64_t)d->groups * d->I)
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: ldu < groups*I");
    const size_t need = kanvit_layer_bwd_weight_workspace(d);
    if (need > 0 && (!workspace || workspace_bytes < need))
        return kv_fail(KANVIT_ENOMEM, "kanvit_layer_bwd_weight: workspace %zu bytes < required %zu", workspace_bytes, need);
    if ((d->flags & KANVIT_FLAG_SINE_DFREQ) && !kanvit_layer_sine_dfreq_ok(d))
        return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: KANVIT_FLAG_SINE_DFREQ needs the register weight-gradient kernel (shape)");
    if (kv_tiny_ok(d)) {
        hipStream_t st = (hipStream_t)stream;
        KvTinyArgs t = tiny_args(d);
        t.x = x; t.bp = bparams; t.dy = dy;
        t.slabs = kv_tiny_slabs(d);
        t.rows_per_slab = ((d->M + t.slabs - 1) / t.slabs + 63) / 64 * 64;
        t.slabs = (int)((d->M + t.rows_per_slab - 1) / t.rows_per_slab);
        t.slab = t.slabs > 1 ? (float*)workspace : dw;
        if (int rc = kv_tiny_bwd_weight(t, st)) return rc;
        if (t.slabs > 1) return kv_slab_reduce((const float*)workspace, dw, (long long)d->groups * t.K * d->O, t.slabs, st);
        return 0;
    }
    const BwPlan p = plan_bwd_weight(d);
    LayerArgs a = base_args(d);
    a.x = x;
    a.u = u;
    a.bp = bparams;
    a.dy = dy;
    if (a.ln) {
        if (!u) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: KANVIT_FLAG_FUSED_LN needs the statistics buffer in the u argument");
        a.stats = const_cast<float*>(u);
        a.u = nullptr;
    }
    {
        const BwRegPlan pr = plan_bwd_weight_reg(d);
        if (!pr.ok && a.ln) return kv_fail(KANVIT_EINVAL, "kanvit_layer_bwd_weight: KANVIT_FLAG_FUSED_LN needs the register kernel (shape)");
        if (pr.ok) {
            hipStream_t st = (hipStream_t)stream;
            const bool bf = pr.bf != 0;
            a.rows_per_split = pr.rows_per_slab;
            a.msplit = pr.slabs;
            a.slab = (pr.slabs > 1) ? (float*)workspace : dw;
            if (int rc = kv_bwd_weight_reg(d->family, a, pr, bf, st)) return rc;
            if (pr.slabs > 1) return kv_slab_reduce((const float*)workspace, dw, (long long)d->groups * a.K * d->O, pr.slabs, st);
            return 0;
        }
    }
    a.IC = p.ic;
    a.msplit = p.msplit;
    a.nchunks_n = p.nchunks_n;
    a.rows_per_split = p.rows_per_split;
    a.slab = (p.msplit > 1) ? (float*)workspace : dw;
    hipStream_t st = (hipStream_t)stream;
    const bool bf = (d->flags & KANVIT_FLAG_BF16_MFMA) && !kv_config().no_bf16;
    const int rc = kv_tile_bwd_weight(d->family, a, p, bf, st);
    if (rc) return rc;
    if (p.msplit > 1) return kv_slab_reduce((const float*)workspace, dw, (long long)d->groups * a.K * d->O, p.msplit, st);
    return 0;
}

// ---- per-family named entry points ---------------------------------------------------------------
#define KV_DEFINE_FAMILY(name, FAMID)                                                                               \
    int kanvit_##name##_fwd(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,            \
                            const float* bp, const float* bias, float* y, void* ws, size_t wsb, void* s) {         \
        if (!d || d->family != FAMID) return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_fwd: descriptor family mismatch"); \
        return kanvit_layer_fwd(d, x, u, w, bp, bias, y, ws, wsb, s);                                               \
    }                                                                                                               \
    int kanvit_##name##_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,      \
                                  const float* bp, const float* dy, float* dx, float* du, float* dp, void* ws,     \
                                  size_t wsb, void* s) {                                                            \
        if (!d || d->family != FAMID) return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_bwd_input: descriptor family mismatch"); \
        return kanvit_layer_bwd_input(d, x, u, w, bp, dy, dx, du, dp, ws, wsb, s);                                  \
    }                                                                                                               \
    int kanvit_##name##_bwd_weight(const kanvit_layer_desc* d, const float* x, const float* u, const float* bp,    \
                                   const float* dy, float* dw, void* ws, size_t wsb, void* s) {                    \
        if (!d || d->family != FAMID) return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_bwd_weight: descriptor family mismatch"); \
        return kanvit_layer_bwd_weight(d, x, u, bp, dy, dw, ws, wsb, s);                                            \
    }                                                                                                               \
    int kanvit_##name##_qkv_fwd(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,        \
                                const float* bp, const float* bias, float* y, void* ws, size_t wsb, void* s) {     \
        if (!d || d->family != FAMID || d->groups != 3 * d->x_group_mod)                                            \
            return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_qkv_fwd: need family match and groups == 3*x_group_mod"); \
        return kanvit_layer_fwd(d, x, u, w, bp, bias, y, ws, wsb, s);                                               \
    }                                                                                                               \
    int kanvit_##name##_qkv_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,  \
                                      const float* bp, const float* dy, float* dx, float* du, float* dp, void* ws, \
                                      size_t wsb, void* s) {                                                        \
        if (!d || d->family != FAMID || d->groups != 3 * d->x_group_mod)                                            \
            return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_qkv_bwd_input: need family match and groups == 3*x_group_mod"); \
        return kanvit_layer_bwd_input(d, x, u, w, bp, dy, dx, du, dp, ws, wsb, s);                                  \
    }                                                                                                               \
    int kanvit_##name##_qkv_bwd_weight(const kanvit_layer_desc* d, const float* x, const float* u, const float* bp, \
                                       const float* dy, float* dw, void* ws, size_t wsb, void* s) {                \
        if (!d || d->family != FAMID || d->groups != 3 * d->x_group_mod)                                            \
            return kv_fail(KANVIT_EINVAL, "kanvit_" #name "_qkv_bwd_weight: need family match and groups == 3*x_group_mod"); \
        return kanvit_layer_bwd_weight(d, x, u, bp, dy, dw, ws, wsb, s);                                            \
    }
KV_DEFINE_FAMILY(linear, KANVIT_LINEAR)
KV_DEFINE_FAMILY(cheby, KANVIT_CHEBY)
KV_DEFINE_FAMILY(bspline, KANVIT_BSPLINE)
KV_DEFINE_FAMILY(rbf, KANVIT_RBF)
KV_DEFINE_FAMILY(sine, KANVIT_SINE)
KV_DEFINE_FAMILY(fourier, KANVIT_FOURIER)

}  // extern "C"

