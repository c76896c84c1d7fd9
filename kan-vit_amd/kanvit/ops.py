"""torch.autograd wrappers around the kanvit C ABI (include/kanvit.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and the autograd tape.
All arithmetic of the hot path (basis evaluation, contractions, attention) runs in the HIP
kernels of libkanvit.so.  There is no fallback: CPU tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BSPLINE, CHEBY, FOURIER, LINEAR, RBF, SINE, AttnDesc, KanvitError, LayerDesc, check  # noqa: F401


@dataclass(frozen=True)
class LayerCfg:
    """Static description of one (grouped) fused KAN layer launch; mirrors kanvit_layer_desc."""
    family: int
    I: int
    O: int
    G: int
    groups: int = 1
    x_group_mod: int = 1
    spline_order: int = 0
    has_base: int = 0
    rbf_inv_h: float = 0.0
    flags: int = 0          # _lib.FLAG_BF16_MFMA: contract on the bf16 matrix cores (set under bf16 autocast)
    ln_eps: float = 0.0     # _lib.FLAG_FUSED_LN (FastKAN): epsilon of the LayerNorm formed inside the kernels

    @property
    def GP(self) -> int:
        return {LINEAR: 1, CHEBY: self.G, BSPLINE: self.G + self.has_base, RBF: self.G + self.has_base,
                SINE: self.G, FOURIER: 2 * self.G}[self.family]

    @property
    def K(self) -> int:
        return self.I * self.GP


class KernelTimer:
    """Optional in-stream timing of the C-ABI launches (bench.py's roofline leg): a pair of events
    recorded on the launch stream brackets each call, so the measured interval is that launch's
    kernels only.  Disabled (None) by default; costs nothing then."""

    def __init__(self):
        self.records = {}          # tag -> list of (start, end, flops, bytes)

    def add(self, tag, start, end, flops, nbytes):
        self.records.setdefault(tag, []).append((start, end, flops, nbytes))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _, _ in recs]
            out[tag] = {"launches": len(recs), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                        "flops": recs[0][2], "bytes": recs[0][3]}
        return out


timer: Optional[KernelTimer] = None


class _timed:
    def __init__(self, tag, flops=0, nbytes=0):
        self.tag, self.flops, self.nbytes = tag, flops, nbytes

    def __enter__(self):
        if timer is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if timer is not None:
            self.e.record()
            timer.add(self.tag, self.s, self.e, self.flops, self.nbytes)
        return False


def _layer_cost(cfg: "LayerCfg", M: int, kind: str):
    """Algorithmic flops / HBM bytes of one launch (SURVEY.md section 8d formulas, fp32)."""
    flops = 2 * M * cfg.K * cfg.O * cfg.groups
    xb, yb, wb = M * cfg.x_group_mod * cfg.I, M * cfg.groups * cfg.O, cfg.groups * cfg.K * cfg.O
    nbytes = 4 * {"fwd": xb + yb + wb, "bwd_input": 2 * xb + yb + wb, "bwd_weight": xb + yb + wb}[kind]
    return flops, nbytes


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu_f32(name: str, t: Optional[torch.Tensor]):
    if t is None:
        return
    if not t.is_cuda:
        raise KanvitError(f"{name} is on {t.device}: the kanvit ops run only on an AMD GPU (no CPU fallback)")
    if t.dtype != torch.float32:
        raise KanvitError(f"{name} has dtype {t.dtype}; the kanvit kernels compute in float32")


def _desc(cfg: LayerCfg, M: int, ldx: int, ldu: int, ldy: int, bp_stride: int) -> LayerDesc:
    return LayerDesc(cfg.family, cfg.groups, cfg.x_group_mod, cfg.I, cfg.O, cfg.G, cfg.spline_order, cfg.has_base,
                     cfg.rbf_inv_h, cfg.flags, M, ldx, ldu, ldy, bp_stride, cfg.ln_eps, 0)


def _workspace(nbytes: int, device):
    """Caller-owned scratch for a C-ABI call (16-byte aligned by the torch allocator)."""
    return torch.empty(max((nbytes + 3) // 4, 4), device=device, dtype=torch.float32)


_fwd_f32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")
# Under torch.autocast (bench.py --amp bf16: the stock FF GEMMs run on the bf16 matrix cores) the kanvit ops
# keep computing in fp32: inputs are cast to fp32 at the boundary and autocast is off inside.


class _KanLayerFn(torch.autograd.Function):
    """y[M, groups*O] = fused_kan(x[M, x_group_mod*I], u, w[groups, K, O], bparams, bias)."""

    @staticmethod
    @_fwd_f32
    def forward(ctx, x, u, w, bparams, bias, cfg: LayerCfg):
        for n, t in (("x", x), ("u", u), ("w", w), ("bparams", bparams), ("bias", bias)):
            _require_gpu_f32(n, t)
        x = x.contiguous()
        w = w.contiguous()
        u = None if u is None else u.contiguous()
        bparams = None if bparams is None else bparams.contiguous()
        bias = None if bias is None else bias.contiguous()
        M, ldx = x.shape
        if ldx != cfg.x_group_mod * cfg.I:
            raise KanvitError(f"x has {ldx} columns, expected x_group_mod*I = {cfg.x_group_mod * cfg.I}")
        if tuple(w.shape) != (cfg.groups, cfg.K, cfg.O):
            raise KanvitError(f"packed weight shape {tuple(w.shape)} != {(cfg.groups, cfg.K, cfg.O)}")
        if u is not None and tuple(u.shape) != (M, cfg.groups * cfg.I):
            raise KanvitError(f"u shape {tuple(u.shape)} != {(M, cfg.groups * cfg.I)}")
        y = torch.empty(M, cfg.groups * cfg.O, device=x.device, dtype=torch.float32)
        d = _desc(cfg, M, ldx, cfg.groups * cfg.I, cfg.groups * cfg.O, 0 if bparams is None else bparams.shape[1])
        tag = ("qkv" if cfg.groups > 1 else "layer") + "_fwd" + ("_bf16" if cfg.flags & _lib.FLAG_BF16_MFMA else "")
        with torch.cuda.device(x.device):
            nbytes = int(_lib.lib().kanvit_layer_fwd_workspace(C.byref(d)))
            ws = _workspace(nbytes, x.device) if nbytes else None
            with _timed(tag, *_layer_cost(cfg, M, "fwd")):
                check(_lib.lib().kanvit_layer_fwd(C.byref(d), _ptr(x), _ptr(u), _ptr(w), _ptr(bparams), _ptr(bias),
                                                  _ptr(y), _ptr(ws), C.c_size_t(nbytes), _stream()), "kanvit_layer_fwd")
        ctx.cfg = cfg
        ctx.has_u = u is not None
        ctx.has_bp = bparams is not None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, u, w, bparams)
        return y

    @staticmethod
    @_bwd
    def backward(ctx, dy):
        x, u, w, bparams = ctx.saved_tensors
        return _kan_backward(ctx.cfg, x, u if ctx.has_u else None, w, bparams, dy.float().contiguous(), ctx.needs_input_grad[:5],
                             ctx.has_u, ctx.has_bias) + (None,)


def _kan_backward(cfg: "LayerCfg", x, u, w, bparams, dy, needs, has_u, has_bias, patch=None):
    """Gradients of one fused KAN launch w.r.t. (x, u, w, bparams, bias): the C-ABI input-gradient and weight-gradient calls.

    patch = (PatchDesc, M): x is the NCHW image batch and dy the token-sequence gradient [B, P + pre, O]; the weight-gradient
    kernels gather both (kanvit_patch_embed_bwd_weight: no patch matrix, no dY copy).  Only weight-side gradients then."""
    if patch is not None:
        pd, M = patch
        ldx = cfg.I
        assert not (needs[0] or needs[1]), "the patch form has no input gradient"
    else:
        pd = None
        M, ldx = x.shape
    need_x, need_u, need_w, need_bp, need_b = needs
    d = _desc(cfg, M, ldx, cfg.groups * cfg.I, cfg.groups * cfg.O, 0 if bparams is None else bparams.shape[1])
    L = _lib.lib()

    def weight_pass(desc, out, tag):
        if pd is not None:
            nbytes = int(L.kanvit_patch_embed_bwd_weight_workspace(C.byref(desc), C.byref(pd)))
            ws = torch.empty(max(nbytes // 4, 1), device=x.device, dtype=torch.float32)
            with _timed(tag, *_layer_cost(cfg, M, "bwd_weight")):
                check(L.kanvit_patch_embed_bwd_weight(C.byref(desc), C.byref(pd), _ptr(x), _ptr(bparams), _ptr(dy), _ptr(out),
                                                      _ptr(ws), C.c_size_t(nbytes), _stream()), "kanvit_patch_embed_bwd_weight")
            return
        nbytes = int(L.kanvit_layer_bwd_weight_workspace(C.byref(desc)))
        ws = torch.empty(max(nbytes // 4, 1), device=x.device, dtype=torch.float32)
        with _timed(tag, *_layer_cost(cfg, M, "bwd_weight")):
            check(L.kanvit_layer_bwd_weight(C.byref(desc), _ptr(x), _ptr(u), _ptr(bparams), _ptr(dy), _ptr(out),
                                            _ptr(ws), C.c_size_t(nbytes), _stream()), "kanvit_layer_bwd_weight")

    dx = du = dw = dbp = db = None
    sine_freq = cfg.family == SINE and need_bp
    # SineKAN's trainable freq (models/sinekan.py:60): d loss / d freq_g = sum_{m,i} x cos(x f_g + p_ig) dPhi[m,i,g] falls out of the
    # input-gradient kernel.  When nobody needs the input gradient itself (the patch embedding: x is the image), the same sum is
    # sum_{i,o} w[(i,g),o] Q[(i,g),o] with Q = a weight-gradient pass over the operand x cos(.) (KANVIT_FLAG_SINE_DFREQ) -- the faster
    # of the two contractions at G = 28 (8.6 instead of 13.9 ms at ViT-B).
    freq_via_w = False
    if sine_freq and not need_x and (pd is not None or not _lib.py_switches()["no_dfreq_w"]):
        with torch.cuda.device(x.device):
            freq_via_w = bool(L.kanvit_layer_sine_dfreq_ok(C.byref(d)))
    if pd is not None and sine_freq and not freq_via_w:
        raise KanvitError("patch form: SineKAN's frequency gradient needs the register weight-gradient kernel")
    with torch.cuda.device(x.device):
        if need_x or (need_u and has_u) or (sine_freq and not freq_via_w):
            dx = torch.empty_like(x)
            du_buf = torch.empty(M, cfg.groups * cfg.I, device=x.device, dtype=torch.float32) if cfg.family == RBF else None
            dpart = None
            if cfg.family == SINE:
                tiles = int(L.kanvit_layer_dparam_tiles(C.byref(d)))
                dpart = torch.empty(tiles, cfg.groups, cfg.G, device=x.device, dtype=torch.float32)
            nb_in = int(L.kanvit_layer_bwd_input_workspace(C.byref(d)))
            ws_in = _workspace(nb_in, x.device) if nb_in else None
            with _timed(("qkv" if cfg.groups > 1 else "layer") + "_bwd_input" + ("_bf16" if nb_in else ""),
                        *_layer_cost(cfg, M, "bwd_input")):
                check(L.kanvit_layer_bwd_input(C.byref(d), _ptr(x), _ptr(u), _ptr(w), _ptr(bparams), _ptr(dy),
                                               _ptr(dx), _ptr(du_buf), _ptr(dpart), _ptr(ws_in), C.c_size_t(nb_in),
                                               _stream()), "kanvit_layer_bwd_input")
            if cfg.family == RBF:
                if has_u:
                    du = du_buf
                else:   # u aliased x (no LayerNorm): fold the spline-path gradient back into dx
                    nshare = cfg.groups // cfg.x_group_mod
                    dx = dx + du_buf.view(M, nshare, cfg.x_group_mod * cfg.I).sum(1)
            if sine_freq:
                dbp = torch.zeros_like(bparams)
                dbp[:, :cfg.G] = dpart.sum(0)
        if need_w:
            dw = torch.empty_like(w)
            weight_pass(d, dw, ("qkv" if cfg.groups > 1 else "layer") + "_bwd_weight" + ("_bf16" if cfg.flags & 1 else ""))
        if freq_via_w:
            dq = LayerDesc(*[getattr(d, f) for f, _ in LayerDesc._fields_])
            dq.flags = cfg.flags | _lib.FLAG_SINE_DFREQ
            q = torch.empty_like(w)
            weight_pass(dq, q, ("qkv" if cfg.groups > 1 else "layer") + "_bwd_freq" + ("_bf16" if cfg.flags & 1 else ""))
            dbp = torch.zeros_like(bparams)
            dbp[:, :cfg.G] = q.mul_(w).view(cfg.groups, cfg.I, cfg.G, cfg.O).sum((1, 3))
        if need_b and has_bias:
            if pd is not None:
                db = dy[:, pd.prepend_rows:, :].sum((0, 1)).view(1, cfg.O)
            else:
                db = dy.view(M, cfg.groups, cfg.O).sum(0)
    return (dx if need_x else None), (du if need_u else None), dw, dbp, db


def ln_fusable(cfg: "LayerCfg", M: int) -> bool:
    """True when the register kernels can form FastKAN's LayerNorm themselves for this launch (kanvit_layer_ln_fusable)."""
    if cfg.family != RBF:
        return False
    f = cfg.flags | (_lib.FLAG_BF16_MFMA if _autocast_flags() else 0)
    d = _desc(LayerCfg(**{**cfg.__dict__, "flags": f}), M, cfg.x_group_mod * cfg.I, cfg.groups * cfg.I, cfg.groups * cfg.O,
              cfg.G + 2 * cfg.I)
    return bool(_lib.lib().kanvit_layer_ln_fusable(C.byref(d)))


class _KanLayerLnFn(torch.autograd.Function):
    """FastKAN launch with the LayerNorm of the spline path formed IN the kernels (KANVIT_FLAG_FUSED_LN):
    y = spline(rbf(LN_g(x_slice))) + base(silu(x_slice)) for every group g, bparams[g] = [centres | gamma_g | beta_g].

    The forward kernel computes each row slice's (mean, rstd), normalises on the fly and writes the statistics [M, x_group_mod, 2];
    the [M, groups*I] normalised tensor u is never materialised (one write + three reads of it per step saved).  Backward: both
    gradient kernels rebuild u from x and the saved statistics; the input-gradient kernel still returns d loss / d u, from which
    the LayerNorm backward (d gamma, d beta, d x) is the standard closed form over the saved statistics (models/fastkan.py:68's
    nn.LayerNorm, biased variance)."""

    @staticmethod
    @_fwd_f32
    def forward(ctx, x, w, bparams, bias, cfg: LayerCfg):
        for n, t in (("x", x), ("w", w), ("bparams", bparams), ("bias", bias)):
            _require_gpu_f32(n, t)
        x, w, bparams = x.contiguous(), w.contiguous(), bparams.contiguous()
        bias = None if bias is None else bias.contiguous()
        M, ldx = x.shape
        if ldx != cfg.x_group_mod * cfg.I:
            raise KanvitError(f"x has {ldx} columns, expected x_group_mod*I = {cfg.x_group_mod * cfg.I}")
        if tuple(w.shape) != (cfg.groups, cfg.K, cfg.O):
            raise KanvitError(f"packed weight shape {tuple(w.shape)} != {(cfg.groups, cfg.K, cfg.O)}")
        if tuple(bparams.shape) != (cfg.groups, cfg.G + 2 * cfg.I):
            raise KanvitError(f"bparams shape {tuple(bparams.shape)} != {(cfg.groups, cfg.G + 2 * cfg.I)} (centres | gamma | beta)")
        y = torch.empty(M, cfg.groups * cfg.O, device=x.device, dtype=torch.float32)
        stats = torch.empty(M, cfg.x_group_mod, 2, device=x.device, dtype=torch.float32)
        d = _desc(cfg, M, ldx, cfg.groups * cfg.I, cfg.groups * cfg.O, bparams.shape[1])
        tag = ("qkv" if cfg.groups > 1 else "layer") + "_fwd" + ("_bf16" if cfg.flags & _lib.FLAG_BF16_MFMA else "")
        with torch.cuda.device(x.device):
            nbytes = int(_lib.lib().kanvit_layer_fwd_workspace(C.byref(d)))
            ws = _workspace(nbytes, x.device) if nbytes else None
            with _timed(tag, *_layer_cost(cfg, M, "fwd")):
                check(_lib.lib().kanvit_layer_fwd(C.byref(d), _ptr(x), _ptr(stats), _ptr(w), _ptr(bparams), _ptr(bias),
                                                  _ptr(y), _ptr(ws), C.c_size_t(nbytes), _stream()), "kanvit_layer_fwd")
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, stats, w, bparams)
        return y

    @staticmethod
    @_bwd
    def backward(ctx, dy):
        x, stats, w, bparams = ctx.saved_tensors
        cfg = ctx.cfg
        need_x, need_w, need_bp, need_b = ctx.needs_input_grad[:4]
        need_ln = need_x or need_bp
        dx, du, dw, _, db = _kan_backward(cfg, x, stats, w, bparams, dy.float().contiguous(),
                                          (need_ln, need_ln, need_w, False, need_b), True, ctx.has_bias)
        dbp = None
        if need_ln:
            # LayerNorm backward from du and the saved statistics: dx += ..., d gamma, d beta in one pass (kanvit_layer_ln_bwd)
            M, I, G = x.shape[0], cfg.I, cfg.G
            d = _desc(cfg, M, x.shape[1], cfg.groups * I, cfg.groups * cfg.O, bparams.shape[1])
            L = _lib.lib()
            dgb = torch.empty(2, cfg.groups, I, device=x.device, dtype=torch.float32)
            with torch.cuda.device(x.device):
                nbytes = int(L.kanvit_layer_ln_bwd_workspace(C.byref(d)))
                ws = _workspace(nbytes, x.device)
                with _timed("ln_bwd", 0, 4 * M * (cfg.groups * I + (3 if need_x else 1) * cfg.x_group_mod * I)):
                    check(L.kanvit_layer_ln_bwd(C.byref(d), _ptr(x), _ptr(stats), _ptr(bparams), _ptr(du), _ptr(dx if need_x else None),
                                                _ptr(dgb[0]), _ptr(dgb[1]), _ptr(ws), C.c_size_t(nbytes), _stream()), "kanvit_layer_ln_bwd")
            if need_bp:
                dbp = torch.cat([torch.zeros(cfg.groups, G, device=x.device, dtype=torch.float32), dgb[0], dgb[1]], dim=1)
        return (dx if need_x else None), dw, dbp, db, None


def kan_layer_ln(x: torch.Tensor, w: torch.Tensor, cfg: LayerCfg, bparams: torch.Tensor, bias: Optional[torch.Tensor],
                 eps: float) -> torch.Tensor:
    """kan_layer() for FastKAN with the LayerNorm fused (call only when ln_fusable(cfg, M)); bparams = [centres | gamma | beta]."""
    from dataclasses import replace
    f = cfg.flags | _lib.FLAG_FUSED_LN | (_lib.FLAG_BF16_MFMA if _autocast_flags() else 0)
    return _KanLayerLnFn.apply(x, w, bparams, bias, replace(cfg, flags=f, ln_eps=float(eps)))


def patchify(images: torch.Tensor, n_patches: int) -> torch.Tensor:
    """(B, C, H, W) -> (B, n^2, C*ph*pw): patches row-major, each flattened in (C, ph, pw) order (model.py:111-126)."""
    b, c, h, w = images.shape
    ph, pw = h // n_patches, w // n_patches
    return images.reshape(b, c, n_patches, ph, n_patches, pw).permute(0, 2, 4, 1, 3, 5).reshape(b, n_patches * n_patches, c * ph * pw)


class _PatchEmbedFn(torch.autograd.Function):
    """images[B, C, H, W] -> tokens[B, P + 1, O] = [cls + pos[0]; layer(patch(b, p)) + pos[1 + p]] in ONE launch (SURVEY.md
    section 8(f)2): the kernel gathers its rows from the NCHW images (no [B, P, I] staging tensor), adds the position
    embedding in its epilogue and writes the class-token rows; with KANVIT_FLAG_BF16_MFMA in cfg.flags (bf16 autocast) the
    bf16 register-form forward does the same.  Backward: the layer's weight gradient (and SineKAN's frequency pass) gathers
    its x rows from the saved images and its dY rows from the token-sequence gradient (kanvit_patch_embed_bwd_weight) -- no
    transient patch matrix, no copy of dY without the class-token rows; layers the gathering kernels do not cover
    (kanvit_patch_embed_bwd_weight_ok) rebuild a transient patch matrix.  d cls = sum_b dy[b, 0]."""

    @staticmethod
    @_fwd_f32
    def forward(ctx, images, w, bparams, bias, cls, pos, cfg: LayerCfg, n_patches: int):
        for n, t in (("images", images), ("w", w), ("bparams", bparams), ("bias", bias), ("cls", cls), ("pos", pos)):
            _require_gpu_f32(n, t)
        images = images.contiguous()
        w = w.contiguous()
        B, Cc, H, W = images.shape
        P = n_patches * n_patches
        y = torch.empty(B, P + 1, cfg.O, device=images.device, dtype=torch.float32)
        d = _desc(cfg, B * P, cfg.I, cfg.I, cfg.O, 0 if bparams is None else bparams.shape[1])
        pd = _lib.PatchDesc(Cc, H, W, n_patches, 1, 0)
        bp = None if bparams is None else bparams.contiguous()
        bs = None if bias is None else bias.contiguous()
        cl, po = cls.contiguous(), pos.contiguous()
        bf = bool(cfg.flags & _lib.FLAG_BF16_MFMA)
        with torch.cuda.device(images.device):
            nbytes = int(_lib.lib().kanvit_layer_fwd_workspace(C.byref(d))) if bf else 0
            ws = _workspace(nbytes, images.device) if nbytes else None
            with _timed("layer_fwd" + ("_bf16" if bf else ""), *_layer_cost(cfg, B * P, "fwd")):
                check(_lib.lib().kanvit_patch_embed_fwd_ws(C.byref(d), C.byref(pd), _ptr(images), _ptr(w), _ptr(bp), _ptr(bs), _ptr(cl),
                                                            _ptr(po), _ptr(y), _ptr(ws), C.c_size_t(nbytes), _stream()),
                      "kanvit_patch_embed_fwd")
        ctx.cfg, ctx.n_patches = cfg, n_patches
        ctx.has_bp, ctx.has_bias = bparams is not None, bias is not None
        ctx.save_for_backward(images, w, bp)
        return y

    @staticmethod
    @_bwd
    def backward(ctx, dy):
        images, w, bparams = ctx.saved_tensors
        cfg, P = ctx.cfg, ctx.n_patches ** 2
        dy = dy.float().contiguous()
        B, Cc, H, W = images.shape
        dcls = dy[:, 0, :].sum(0) if ctx.needs_input_grad[4] else None
        needs = (False, False, ctx.needs_input_grad[1], ctx.needs_input_grad[2] and ctx.has_bp, ctx.needs_input_grad[3] and ctx.has_bias)
        pd = _lib.PatchDesc(Cc, H, W, ctx.n_patches, 1, 0)
        d = _desc(cfg, B * P, cfg.I, cfg.I, cfg.O, 0 if bparams is None else bparams.shape[1])
        with torch.cuda.device(images.device):
            gather = bool(_lib.lib().kanvit_patch_embed_bwd_weight_ok(C.byref(d), C.byref(pd))) and not _lib.py_switches()["no_patch_bw"]
        if cfg.family == SINE and _lib.py_switches()["no_dfreq_w"]:
            gather = False          # d loss / d freq through the input-gradient kernel needs the patch matrix
        if gather:
            _, _, dw, dbp, db = _kan_backward(cfg, images, None, w, bparams, dy, needs, False, ctx.has_bias, patch=(pd, B * P))
        else:
            x = patchify(images, ctx.n_patches).reshape(-1, cfg.I)                 # transient
            dyt = dy[:, 1:, :].reshape(-1, cfg.O)                                  # transient (contiguous patch-token rows)
            _, _, dw, dbp, db = _kan_backward(cfg, x, None, w, bparams, dyt, needs, False, ctx.has_bias)
        return None, dw, dbp, db, dcls, None, None, None


def patch_embed(images, w, cfg: LayerCfg, bparams, bias, cls, pos, n_patches: int) -> torch.Tensor:
    """Fused patch embedding (see _PatchEmbedFn); w is the packed weight [1, K, O], cls [O], pos [P + 1, O].  Raises
    KanvitError for shapes the fused kernel does not cover (the caller falls back to patchify + kan_layer).  Under bf16
    autocast the contraction runs on the bf16 matrix cores (KANVIT_FLAG_BF16_MFMA), as kan_layer() does."""
    if _autocast_flags() and not (cfg.flags & _lib.FLAG_BF16_MFMA):
        from dataclasses import replace
        cfg = replace(cfg, flags=cfg.flags | _lib.FLAG_BF16_MFMA)
    return _PatchEmbedFn.apply(images, w, bparams, bias, cls, pos, cfg, n_patches)


def kan_layer(x: torch.Tensor, w: torch.Tensor, cfg: LayerCfg, u: Optional[torch.Tensor] = None,
              bparams: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused basis + contraction for `cfg.groups` layers sharing the rows of x (see include/kanvit.h).

    Under ``torch.autocast('cuda', dtype=torch.bfloat16)`` the contraction is allowed onto the bf16 matrix cores
    (KANVIT_FLAG_BF16_MFMA); without autocast it is always the exact fp32 path."""
    if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 and not (cfg.flags & 1):
        from dataclasses import replace
        cfg = replace(cfg, flags=cfg.flags | _lib.FLAG_BF16_MFMA)
    return _KanLayerFn.apply(x, u, w, bparams, bias, cfg)


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
_attn_flags = 0      # set by attention()/attention_packed() from the ambient autocast state (the Functions run with autocast off)


def _attn_desc(q, k, v, o, causal: bool, scale: float, flags: int = 0) -> AttnDesc:
    B, H, N, D = q.shape
    for n, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        if t.dim() != 4 or tuple(t.shape) != (B, H, N, D):
            # kanvit_attn_desc carries ONE sequence length: a shorter k/v would be read out of bounds, a longer one silently
            # truncated.  The reference's FlashAttentionFunction accepts q_len != k_len (utils.py:150-160); this path does not.
            raise KanvitError(f"attention: {n} has shape {tuple(t.shape)}, expected {(B, H, N, D)} "
                              "(self-attention only: q, k, v and o must agree in batch, heads, length and head size)")
        if t.stride(3) != 1:
            raise KanvitError(f"{n}: innermost dimension must be contiguous")
    return AttnDesc(B, H, N, D, int(bool(causal)), float(scale), int(flags), 0,
                    q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                    v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2))


def _attn_fits_one_workgroup(N: int, D: int) -> bool:
    """The ViT kernels of csrc/attention.hip hold one head in a work-group's LDS (check_desc's bound: N <= 224 at D = 64, 256 at
    D <= 32); longer sequences (e.g. 384 x 384 images at patch 16: N = 577) take the chunked general kernels (csrc/attention_x.hip)."""
    ks, npad = (32 if D <= 32 else 64) + 1, (N + 31) // 32 * 32
    return N <= 256 and 4 * (2 * npad * ks + 2 * npad + 4 * 32 * ks) <= 160 * 1024


def _attn_fwd(q, k, v, o, causal, scale, flags=0):
    if q.dim() == 4 and tuple(k.shape) == tuple(q.shape) and not _attn_fits_one_workgroup(q.shape[2], q.shape[3]):
        return _attn_x_fwd(q, k, v, o, None, causal, scale)
    B, H, N, _ = q.shape
    lse = torch.empty(B, H, N, device=q.device, dtype=torch.float32)
    d = _attn_desc(q, k, v, o, causal, scale, flags)
    flops, nbytes = 4 * B * H * N * N * q.shape[3], 4 * 4 * B * H * N * q.shape[3]
    with torch.cuda.device(q.device), _timed("attn_fwd" + ("_bf16" if flags & 1 else ""), flops, nbytes):
        check(_lib.lib().kanvit_attn_fwd(C.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _stream()),
              "kanvit_attn_fwd")
    return lse


def _attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal, scale, flags=0):
    if q.dim() == 4 and tuple(k.shape) == tuple(q.shape) and not _attn_fits_one_workgroup(q.shape[2], q.shape[3]):
        return _attn_x_bwd(q, k, v, o, lse, do, dq, dk, dv, None, causal, scale)
    d = _attn_desc(q, k, v, o, causal, scale, flags)
    if (do.stride() != o.stride()) or dq.stride() != q.stride() or dk.stride() != k.stride() or dv.stride() != v.stride():
        raise KanvitError("attention backward: gradient layouts must match their forward tensors")
    L = _lib.lib()
    nbytes = int(L.kanvit_attn_bwd_workspace(C.byref(d)))
    ws = torch.empty(max(nbytes // 4, 1), device=q.device, dtype=torch.float32)
    B, H, N, D = q.shape
    # algorithmic work of a recompute backward: 5 products (S, dP, dV, dK, dQ) = 2.5x the forward; the two-kernel bf16 path
    # executes 7 (S and dP twice), the fp32 dS-spill path exactly 5
    with torch.cuda.device(q.device), _timed("attn_bwd" + ("_bf16" if flags & 1 else ""), 10 * B * H * N * N * D, 4 * 9 * B * H * N * D):
        check(L.kanvit_attn_bwd(C.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _ptr(do), _ptr(dq), _ptr(dk),
                                _ptr(dv), _ptr(ws), C.c_size_t(nbytes), _stream()), "kanvit_attn_bwd")


def _attn_x_desc(q, k, v, o, mask, causal: bool, scale: float):
    """Descriptor pair of the general attention kernels (kanvit_attn_x_*): q, o [B, H, Nq, D]; k, v [B, H, Nk, D]; mask None or a
    torch.bool tensor already expanded (views, no copy) to [B, H, Nq, Nk]."""
    for n, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        _require_gpu_f32(n, t)
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    if tuple(k.shape) != (B, H, Nk, D) or tuple(v.shape) != (B, H, Nk, D) or tuple(o.shape) != (B, H, Nq, D):
        raise KanvitError(f"attention: q {tuple(q.shape)}, k {tuple(k.shape)}, v {tuple(v.shape)}, o {tuple(o.shape)} do not agree")
    for n, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        if t.stride(3) != 1:
            raise KanvitError(f"{n}: innermost dimension must be contiguous")
    d = AttnDesc(B, H, Nq, D, int(bool(causal)), float(scale), 0, 0,
                 q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                 v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2))
    if mask is None:
        e = _lib.AttnExt(Nk, 0, None, 0, 0, 0, 0)
    else:
        if mask.dtype != torch.bool or not mask.is_cuda or tuple(mask.shape) != (B, H, Nq, Nk):
            raise KanvitError(f"attention mask: need a torch.bool GPU tensor expanded to {(B, H, Nq, Nk)}, got {mask.dtype} {tuple(mask.shape)}")
        e = _lib.AttnExt(Nk, 0, mask.data_ptr(), mask.stride(0), mask.stride(1), mask.stride(2), mask.stride(3))
    return d, e


def _attn_x_fwd(q, k, v, o, mask, causal, scale):
    B, H, Nq, D = q.shape
    lse = torch.empty(B, H, Nq, device=q.device, dtype=torch.float32)
    d, e = _attn_x_desc(q, k, v, o, mask, causal, scale)
    Nk = k.shape[2]
    with torch.cuda.device(q.device), _timed("attn_x_fwd", 4 * B * H * Nq * Nk * D, 4 * 2 * B * H * (Nq + Nk) * D):
        check(_lib.lib().kanvit_attn_x_fwd(C.byref(d), C.byref(e), _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _stream()), "kanvit_attn_x_fwd")
    return lse


def _attn_x_bwd(q, k, v, o, lse, do, dq, dk, dv, mask, causal, scale):
    d, e = _attn_x_desc(q, k, v, o, mask, causal, scale)
    if (do.stride() != o.stride()) or dq.stride() != q.stride() or dk.stride() != k.stride() or dv.stride() != v.stride():
        raise KanvitError("attention backward: gradient layouts must match their forward tensors")
    L = _lib.lib()
    nbytes = int(L.kanvit_attn_x_bwd_workspace(C.byref(d), C.byref(e)))
    ws = torch.empty(max(nbytes // 4, 1), device=q.device, dtype=torch.float32)
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    with torch.cuda.device(q.device), _timed("attn_x_bwd", 14 * B * H * Nq * Nk * D, 4 * 4 * B * H * (Nq + Nk) * D):
        check(L.kanvit_attn_x_bwd(C.byref(d), C.byref(e), _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _ptr(do), _ptr(dq), _ptr(dk),
                                  _ptr(dv), _ptr(ws), C.c_size_t(nbytes), _stream()), "kanvit_attn_x_bwd")


class _AttnPackedFn(torch.autograd.Function):
    """qkv[B, N, 3, H, D] (the layout the grouped q|k|v KAN launch writes) -> o[B, N, H*D]."""

    @staticmethod
    @_fwd_f32
    def forward(ctx, qkv, causal: bool, scale: float, flags: int = 0):
        _require_gpu_f32("qkv", qkv)
        qkv = qkv.contiguous()
        B, N, three, H, D = qkv.shape
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        o = torch.empty(B, N, H, D, device=qkv.device, dtype=torch.float32)
        lse = _attn_fwd(q, k, v, o.permute(0, 2, 1, 3), causal, scale, flags)
        ctx.save_for_backward(qkv, o, lse)
        ctx.causal, ctx.scale, ctx.flags = causal, scale, flags
        return o.view(B, N, H * D)

    @staticmethod
    @_bwd
    def backward(ctx, do):
        do = do.float()
        qkv, o, lse = ctx.saved_tensors
        B, N, _, H, D = qkv.shape
        do = do.contiguous().view(B, N, H, D)
        dqkv = torch.empty_like(qkv)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        _attn_bwd(q, k, v, o.permute(0, 2, 1, 3), lse, do.permute(0, 2, 1, 3), dq, dk, dv, ctx.causal, ctx.scale, ctx.flags)
        return dqkv, None, None, None


class _AttnFn(torch.autograd.Function):
    """Separate q, k, v of shape (B, H, N, D) with arbitrary outer strides -> o (B, H, N, D)."""

    @staticmethod
    @_fwd_f32
    def forward(ctx, q, k, v, causal: bool, scale: float, flags: int = 0):
        for n, t in (("q", q), ("k", k), ("v", v)):
            _require_gpu_f32(n, t)
        q, k, v = (t if t.stride(3) == 1 else t.contiguous() for t in (q, k, v))
        o = torch.empty(q.shape, device=q.device, dtype=torch.float32)
        lse = _attn_fwd(q, k, v, o, causal, scale, flags)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.causal, ctx.scale, ctx.flags = causal, scale, flags
        return o

    @staticmethod
    @_bwd
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        do = do.float().contiguous()
        dq, dk, dv = (torch.empty_strided(t.shape, t.stride(), device=t.device, dtype=t.dtype) for t in (q, k, v))
        _attn_bwd(q, k, v, o, lse, do, dq, dk, dv, ctx.causal, ctx.scale, ctx.flags)
        return dq, dk, dv, None, None, None


def _autocast_flags() -> int:
    """bf16 autocast allows the products onto the bf16 matrix cores; otherwise the exact fp32 kernels run."""
    on = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
    return _lib.FLAG_BF16_MFMA if on else 0


def attention_packed(qkv: torch.Tensor, causal: bool = False, scale: Optional[float] = None) -> torch.Tensor:
    return _AttnPackedFn.apply(qkv, causal, qkv.shape[-1] ** -0.5 if scale is None else scale, _autocast_flags())


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = False,
              scale: Optional[float] = None) -> torch.Tensor:
    return _AttnFn.apply(q, k, v, causal, q.shape[-1] ** -0.5 if scale is None else scale, _autocast_flags())


# ------------------------------------------------------------------------------------------------
# residual add + LayerNorm (TransformerBlock assembly, reference model.py:31-37)
# ------------------------------------------------------------------------------------------------
class _AddLayerNormFn(torch.autograd.Function):
    """(x, delta | None, gamma, beta) -> (s = x + delta, LayerNorm(s)) in one pass; backward in one pass + a tiny reduce.

    The residual stream (x, s, their gradients) is fp32.  Under torch.autocast(bfloat16) the tensors that come from / go to the
    feed-forward GEMMs may be bf16 and are read / written as they are (kanvit_addln_*_ex) instead of through cast kernels:
    `delta` (the previous block's feed-forward output), `y` when `y_bf16` (the feed-forward's input), the gradient arriving
    on y, and the gradient returned for a bf16 delta (a second, rounded copy of dx written by the same pass)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, delta, gamma, beta, eps, y_bf16):
        if not x.is_cuda:
            raise KanvitError(f"x is on {x.device}: the kanvit ops run only on an AMD GPU (no CPU fallback)")
        L = _lib.lib()
        D = x.shape[-1]
        x2 = x.float().contiguous().view(-1, D)
        d2 = None
        if delta is not None:
            d2 = (delta if delta.dtype in (torch.float32, torch.bfloat16) else delta.float()).contiguous().view(-1, D)
        M = x2.shape[0]
        y = torch.empty(M, D, device=x.device, dtype=torch.bfloat16 if y_bf16 else torch.float32)
        s = torch.empty_like(x2) if d2 is not None else x2
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        g, b = gamma.float().contiguous(), beta.float().contiguous()
        with torch.cuda.device(x.device):          # launch on x's device and ITS current stream, whatever the ambient device is
            check(L.kanvit_addln_fwd_ex(M, D, float(eps), _ptr(x2), _ptr(d2), int(d2 is not None and d2.dtype == torch.bfloat16), _ptr(g), _ptr(b),
                                        _ptr(s) if d2 is not None else None, _ptr(y), int(bool(y_bf16)), _ptr(mean), _ptr(rstd), _stream()),
                  "kanvit_addln_fwd")
        ctx.save_for_backward(s, g, mean, rstd)
        ctx.delta_dtype = None if d2 is None else d2.dtype
        ctx.shape = x.shape
        return s.view(x.shape), y.view(x.shape)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gs, gy):
        s, g, mean, rstd = ctx.saved_tensors
        L = _lib.lib()
        M, D = s.shape
        if gy is None:
            gy2 = torch.zeros_like(s)
        else:
            gy2 = gy.contiguous().view(M, D)
            if gy2.dtype != torch.bfloat16:
                gy2 = gy2.float()
        gs2 = None if gs is None else gs.contiguous().view(M, D).float()
        dx = torch.empty_like(s)
        dxb = torch.empty(M, D, device=s.device, dtype=torch.bfloat16) if ctx.delta_dtype == torch.bfloat16 else None
        dg = torch.empty(D, device=s.device, dtype=torch.float32)
        db = torch.empty(D, device=s.device, dtype=torch.float32)
        nbytes = int(L.kanvit_addln_bwd_workspace(M, D))
        ws = _workspace(nbytes, s.device)
        with torch.cuda.device(s.device):
            check(L.kanvit_addln_bwd_ex(M, D, _ptr(s), _ptr(g), _ptr(mean), _ptr(rstd), _ptr(gy2), int(gy2.dtype == torch.bfloat16), _ptr(gs2),
                                        _ptr(dx), _ptr(dxb), _ptr(dg), _ptr(db), _ptr(ws), C.c_size_t(nbytes), _stream()), "kanvit_addln_bwd")
        dx = dx.view(ctx.shape)
        dd = None if ctx.delta_dtype is None else (dxb.view(ctx.shape) if dxb is not None else dx)
        return dx, dd, dg, db, None, None


def add_layernorm(x: torch.Tensor, delta: Optional[torch.Tensor], norm: torch.nn.LayerNorm, y_bf16: bool = False):
    """(x + delta, norm(x + delta)) -- delta None gives (x, norm(x)).  Shapes the kernel does not cover (last dim not a
    multiple of 4 or > 1024, non-affine norms) take the stock ops; like every kanvit op it needs CUDA tensors.
    y_bf16: write norm(...) as bfloat16 (the caller feeds it to bf16 GEMMs under autocast: no cast pass in between)."""
    D = x.shape[-1]
    if norm.weight is None or norm.bias is None or len(norm.normalized_shape) != 1 or D % 4 or D > 1024 or D < 4:
        s = x if delta is None else x + delta
        return s, norm(s)
    return _AddLayerNormFn.apply(x, delta, norm.weight, norm.bias, norm.eps, bool(y_bf16))
