"""Dense (nn.Linear) layers of the feed-forward block with a token-split weight gradient.

The reference's feed-forward is two stock nn.Linear layers (model.py:25-29); their forward and input-gradient GEMMs
run at 110-130 TFLOP/s fp32 in hipBLASLt on MI355X, but the weight gradient  dW[N,K] = dYᵀ[N,M] · X[M,K]  contracts
over the M = B*N_tokens axis into a small output (768 x 3072 = 36 macro-tiles for 256 CUs) and the library does not
split the contraction: 53 TFLOP/s, 2.2 ms per call, 53 ms of a 150 ms ViT-B step (tools/probe_ff_gemm.py).  Splitting
the token axis into S slabs -- one batched GEMM into [S, N, K] partials plus an ordered sum over S, the same
slab-then-reduce scheme the KAN weight-gradient kernel uses -- fills the chip: 0.83 ms, 143 TFLOP/s
(tools/probe_wgrad.py).  Deterministic (fixed summation order), and slightly MORE accurate than the single long
contraction."""
import torch
import torch.nn.functional as F

from . import _lib

_TILE = 256
_HALF = (torch.bfloat16, torch.float16)
_CUS = 256


def wgrad_slabs(M: int, N: int, K: int) -> int:
    """Number of token slabs: the smallest divisor of M (<= 32) that yields >= one macro-tile per CU while keeping at
    least 128 tokens per slab; 1 when the output alone already fills the chip or M is too small to split."""
    tiles = -(-N // _TILE) * -(-K // _TILE)
    if tiles >= _CUS:
        return 1
    best = 1
    for s in range(2, 33):
        if M % s or M // s < 128:
            continue
        best = s
        if s * tiles >= _CUS:
            break
    return best


class _DenseFn(torch.autograd.Function):
    """y = act(x @ W^T + b) for 2-D x, act = identity or ReLU.

    Under torch.autocast the operands are cast ONCE here (and the casts saved for backward) instead of once per use; the
    ReLU variant runs bias + ReLU in the GEMM epilogue (torch._addmm_activation: bit-identical to linear + relu_, one
    pass over [M, 4d] less) and applies the ReLU mask to the incoming gradient itself."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        ac = x.is_cuda and torch.is_autocast_enabled("cuda")
        dt = torch.get_autocast_dtype("cuda") if ac else x.dtype
        xc, wc = x.to(dt), weight.to(dt)
        bc = None if bias is None else bias.to(dt)
        with torch.autocast("cuda", enabled=False):
            if relu and bc is not None and x.is_cuda:
                y = torch._addmm_activation(bc, xc, wc.t(), use_gelu=False)
            else:
                y = F.linear(xc, wc, bc)
                if relu:
                    y = torch.relu_(y)
        ctx.save_for_backward(xc, wc, y if relu else None)
        ctx.has_bias = bias is not None
        ctx.relu = relu
        ctx.in_dtypes = (x.dtype, weight.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc, y = ctx.saved_tensors
        xdt, wdt = ctx.in_dtypes
        dx = dw = db = None
        with torch.autocast("cuda", enabled=False):
            dy = dy.contiguous().to(xc.dtype)
            if ctx.relu:
                if (dy.is_cuda and dy.dtype == y.dtype and dy.dtype in (torch.float32, torch.bfloat16) and y.is_contiguous() and dy.shape[1] % 4 == 0
                        and ctx.has_bias and ctx.needs_input_grad[2] and dy.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0
                        and not _lib.py_switches()["no_ff_epi"]):
                    dy, db = _relu_bwd_bias(dy, y)      # ReLU mask + bias gradient in one pass (csrc/ff_epilogue.hip)
                else:
                    dy = torch.ops.aten.threshold_backward(dy, y, 0)
            if ctx.needs_input_grad[0]:
                dx = (dy @ wc).to(xdt)
            if ctx.needs_input_grad[1]:
                M, N = dy.shape
                K = xc.shape[1]
                S = wgrad_slabs(M, N, K)
                if S > 1:
                    part = torch.bmm(dy.view(S, M // S, N).transpose(1, 2), xc.view(S, M // S, K))
                    dw = part.sum(0, dtype=torch.float32) if part.dtype in _HALF else part.sum(0)
                else:
                    dw = dy.t() @ xc
                dw = dw.to(wdt)
            if db is not None:
                db = db.to(wdt)
            elif ctx.has_bias and ctx.needs_input_grad[2]:
                db = (dy.sum(0, dtype=torch.float32) if dy.dtype in _HALF else dy.sum(0)).to(wdt)
        return dx, dw, db, None


def _relu_bwd_bias(dy: torch.Tensor, y: torch.Tensor):
    """(dy masked by y > 0, column sums of the result): kanvit_relu_bwd_bias -- what threshold_backward + sum(0) compute in two
    passes over [M, 4d].  The masked gradient goes to a fresh tensor (the incoming dy may be shared with other consumers)."""
    import ctypes as C
    from . import ops
    M, N = dy.shape
    L = _lib.lib()
    db = torch.empty(N, device=dy.device, dtype=torch.float32)
    out = torch.empty_like(dy)
    fn = L.kanvit_relu_bwd_bias_bf16 if dy.dtype == torch.bfloat16 else L.kanvit_relu_bwd_bias      # autocast: bf16 tensors, fp32 column sums
    with torch.cuda.device(dy.device):
        nbytes = int(L.kanvit_relu_bwd_bias_workspace(M, N))
        ws = ops._workspace(nbytes, dy.device)
        _lib.check(fn(M, N, ops._ptr(dy), ops._ptr(y), ops._ptr(out), ops._ptr(db), ops._ptr(ws), C.c_size_t(nbytes), ops._stream()),
                   "kanvit_relu_bwd_bias")
    return out, db


def dense(x: torch.Tensor, linear: torch.nn.Linear, relu: bool = False) -> torch.Tensor:
    """linear(x) (then ReLU if asked) for a 2-D x, with the token-split weight gradient in backward."""
    assert x.dim() == 2
    return _DenseFn.apply(x, linear.weight, linear.bias, relu)


# ------------------------------------------------------------------------------------------------
# opt-in "bf16x3" feed-forward: fp32-accurate GEMMs on the bf16 matrix cores
# ------------------------------------------------------------------------------------------------
# The fp32 matrix pipe of MI355X peaks at 157 TFLOP/s and the stock fp32 GEMMs already run at 125-143 of it; they are 65 %
# of the fp32 train step.  Splitting every fp32 operand into two bf16 terms (v = hi + lo) and forming a.b as
# hi.hi + hi.lo + lo.hi -- ONE bf16 GEMM over a three times longer K axis, fp32 accumulate -- reproduces the fp32 product
# to ~5e-6 relative (the fp32 GEMM itself is at 1.4e-6, tools/probe_split_gemm.py) at 2.5x the speed.  Whether the step
# may spend part of its 1e-4 parity budget here is a policy decision, so this is OFF by default:
#     kanvit.dense.FF_MODE = "bf16x3"      (bench.py --ff bf16x3, or KANVIT_FF=bf16x3)
FF_MODE = "fp32"


def _split3(x: torch.Tensor, pattern: int, bias=None, relu=False, mask=None) -> torch.Tensor:
    import ctypes as C
    L = _lib.lib()
    M, K = x.shape
    out = torch.empty(M, 3 * K, device=x.device, dtype=torch.bfloat16)
    x = x.contiguous()
    with torch.cuda.device(x.device):              # x's device and its current stream, whatever the ambient device is
        _lib.check(L.kanvit_split3_bf16(M, K, C.c_void_p(x.data_ptr()), None if bias is None else C.c_void_p(bias.data_ptr()),
                                        int(relu), None if mask is None else C.c_void_p(mask.data_ptr()),
                                        0 if mask is None else mask.stride(0), C.c_void_p(out.data_ptr()), pattern,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "kanvit_split3_bf16")
    return out


def _wgrad3(G3: torch.Tensor, X3: torch.Tensor, N: int, K: int) -> torch.Tensor:
    """dW[N, K] = G^T X from the pattern-0 images G3 = [g_hi|g_hi|g_lo] ([M, 3N]) and X3 = [x_hi|x_hi|x_lo] ([M, 3K]):
    g_hi^T [x_hi | x_lo] + g_lo^T x_hi, token axis split into slabs like the fp32 path."""
    M = G3.shape[0]
    S = wgrad_slabs(M, N, K)
    g, x = G3.view(S, M // S, 3 * N), X3.view(S, M // S, 3 * K)
    t1 = torch.bmm(g[:, :, :N].transpose(1, 2), x[:, :, K:], out_dtype=torch.float32).sum(0)        # [N, 2K]
    t2 = torch.bmm(g[:, :, 2 * N:].transpose(1, 2), x[:, :, :K], out_dtype=torch.float32).sum(0)    # [N, K]
    return t1[:, :K] + t1[:, K:] + t2


class _FFSplitFn(torch.autograd.Function):
    """Linear -> ReLU -> Linear (reference model.py:25-29) with every GEMM as a three-term bf16 split product."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w1, b1, w2, b2):
        A3 = _split3(x, 0)
        y1 = torch.mm(A3, _split3(w1, 1).t(), out_dtype=torch.float32)
        H3 = _split3(y1, 0, bias=b1, relu=True)
        y2 = torch.mm(H3, _split3(w2, 1).t(), out_dtype=torch.float32)
        y2.add_(b2)
        ctx.save_for_backward(A3, H3, w1, w2)
        return y2

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy2):
        A3, H3, w1, w2 = ctx.saved_tensors
        N1, K = w1.shape
        N2 = w2.shape[0]
        dy2 = dy2.contiguous().float()
        D3 = _split3(dy2, 0)
        dH = torch.mm(D3, _split3(w2.t().contiguous(), 1).t(), out_dtype=torch.float32)          # [M, N1]
        DP3 = _split3(dH, 0, mask=H3)                                                                # ReLU mask from H's hi part
        dx = torch.mm(DP3, _split3(w1.t().contiguous(), 1).t(), out_dtype=torch.float32)           # [M, K]
        dw2 = _wgrad3(D3, H3, N2, N1)
        dw1 = _wgrad3(DP3, A3, N1, K)
        db2 = dy2.sum(0)
        db1 = DP3[:, :N1].sum(0, dtype=torch.float32) + DP3[:, 2 * N1:].sum(0, dtype=torch.float32)
        return dx, dw1, db1, dw2, db2


class _FFSmallFn(torch.autograd.Function):
    """Linear -> ReLU -> Linear of the small geometries (d = 64, 4d = 256: the reference's own defaults, model.py:49 and
    train.py:18-20) as ONE forward launch and ONE backward launch + an ordered reduce (csrc/ff_small.hip, SURVEY.md section
    8(f)1) instead of 2 + ~14 stock launches of ~5 us each.  fp32 products and sums on the matrix cores; the backward
    recomputes the hidden activations instead of saving [M, 4d]."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w1, b1, w2, b2):
        from . import ops
        x, w1, b1, w2, b2 = (t.contiguous() for t in (x, w1, b1, w2, b2))
        M, D = x.shape
        F_ = w1.shape[0]
        y = torch.empty(M, D, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device), ops._timed("ff_small_fwd", 4 * M * D * F_, 4 * (2 * M * D + 2 * D * F_)):
            _lib.check(_lib.lib().kanvit_ff_small_fwd(M, D, F_, ops._ptr(x), ops._ptr(w1), ops._ptr(b1), ops._ptr(w2), ops._ptr(b2),
                                                      ops._ptr(y), ops._stream()), "kanvit_ff_small_fwd")
        ctx.save_for_backward(x, w1, b1, w2)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        from . import ops
        import ctypes as C
        x, w1, b1, w2 = ctx.saved_tensors
        M, D = x.shape
        F_ = w1.shape[0]
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        dw1, db1, dw2, db2 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty(D, device=x.device, dtype=torch.float32)
        L = _lib.lib()
        with torch.cuda.device(x.device):
            nbytes = int(L.kanvit_ff_small_bwd_workspace(M, D, F_))
            ws = ops._workspace(nbytes, x.device)
            with ops._timed("ff_small_bwd", 8 * M * D * F_, 4 * (3 * M * D + 4 * D * F_)):
                _lib.check(L.kanvit_ff_small_bwd(M, D, F_, ops._ptr(x), ops._ptr(w1), ops._ptr(b1), ops._ptr(w2), ops._ptr(dy), ops._ptr(dx),
                                                 ops._ptr(dw1), ops._ptr(db1), ops._ptr(dw2), ops._ptr(db2), ops._ptr(ws), C.c_size_t(nbytes),
                                                 ops._stream()), "kanvit_ff_small_bwd")
        return dx, dw1, db1, dw2, db2


class _LnFFSmallFn(torch.autograd.Function):
    """(x, delta | None, LN2, FF) -> (s = x + delta, FF(LayerNorm(s))): the second half of a small-geometry TransformerBlock
    (model.py:36) in ONE forward launch; backward ONE launch + the ordered reduce, returning d s (the gradient of x and of
    delta), d gamma, d beta and the four feed-forward parameter gradients (csrc/ff_small.hip, LN variants)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, delta, gamma, beta, w1, b1, w2, b2, eps):
        from . import ops
        D = x.shape[-1]
        x2 = x.contiguous().view(-1, D)
        d2 = None if delta is None else delta.contiguous().view(-1, D)
        gamma, beta, w1, b1, w2, b2 = (t.contiguous() for t in (gamma, beta, w1, b1, w2, b2))
        M, F_ = x2.shape[0], w1.shape[0]
        s = torch.empty_like(x2)
        y = torch.empty_like(x2)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device), ops._timed("ff_small_fwd", 4 * M * D * F_, 4 * (4 * M * D + 2 * D * F_)):
            _lib.check(_lib.lib().kanvit_lnff_small_fwd(M, D, F_, float(eps), ops._ptr(x2), ops._ptr(d2), ops._ptr(gamma), ops._ptr(beta),
                                                        ops._ptr(w1), ops._ptr(b1), ops._ptr(w2), ops._ptr(b2), ops._ptr(s), ops._ptr(mean),
                                                        ops._ptr(rstd), ops._ptr(y), ops._stream()), "kanvit_lnff_small_fwd")
        ctx.save_for_backward(s, mean, rstd, gamma, beta, w1, b1, w2)
        ctx.has_delta = delta is not None
        ctx.shape = x.shape
        return s.view(x.shape), y.view(x.shape)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gs, gy):
        from . import ops
        import ctypes as C
        s, mean, rstd, gamma, beta, w1, b1, w2 = ctx.saved_tensors
        M, D = s.shape
        F_ = w1.shape[0]
        gy2 = gy.contiguous().view(M, D).float() if gy is not None else torch.zeros_like(s)
        gs2 = None if gs is None else gs.contiguous().view(M, D).float()
        ds = torch.empty_like(s)
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        dw1, db1, dw2, db2 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty(D, device=s.device, dtype=torch.float32)
        L = _lib.lib()
        with torch.cuda.device(s.device):
            nbytes = int(L.kanvit_ff_small_bwd_workspace(M, D, F_))
            ws = ops._workspace(nbytes, s.device)
            with ops._timed("ff_small_bwd", 8 * M * D * F_, 4 * (5 * M * D + 4 * D * F_)):
                _lib.check(L.kanvit_lnff_small_bwd(M, D, F_, ops._ptr(s), ops._ptr(mean), ops._ptr(rstd), ops._ptr(gamma), ops._ptr(beta),
                                                   ops._ptr(w1), ops._ptr(b1), ops._ptr(w2), ops._ptr(gy2), ops._ptr(gs2), ops._ptr(ds),
                                                   ops._ptr(dg), ops._ptr(db), ops._ptr(dw1), ops._ptr(db1), ops._ptr(dw2), ops._ptr(db2),
                                                   ops._ptr(ws), C.c_size_t(nbytes), ops._stream()), "kanvit_lnff_small_bwd")
        ds = ds.view(ctx.shape)
        return ds, (ds if ctx.has_delta else None), dg, db, dw1, db1, dw2, db2, None


def ln_feed_forward(x: torch.Tensor, delta, norm: torch.nn.LayerNorm, lin1: torch.nn.Linear, lin2: torch.nn.Linear):
    """(s, f) with s = x + delta and f = lin2(relu(lin1(norm(s)))) -- model.py:36 without its final residual add.  One fused
    launch for the small geometries; otherwise add_layernorm + feed_forward (the same arithmetic as two ops)."""
    from . import ops
    d = x.shape[-1]
    if (norm.elementwise_affine and norm.bias is not None and tuple(norm.normalized_shape) == (d,) and (delta is None or delta.shape == x.shape)
            and not _lib.py_switches()["no_lnff"] and _ff_small_ok(x.reshape(-1, d), lin1, lin2)):
        return _LnFFSmallFn.apply(x, delta, norm.weight, norm.bias, lin1.weight, lin1.bias, lin2.weight, lin2.bias, norm.eps)
    s, h = ops.add_layernorm(x, delta, norm)
    return s, feed_forward(h.reshape(-1, d), lin1, lin2).view(x.shape)


def ff_mode() -> str:
    """"fp32" or "bf16x3": KANVIT_FF (read once, _lib.py_switches) overrides the module default FF_MODE."""
    return _lib.py_switches()["ff"] or FF_MODE


def ff_small_supported(d: int, f: int) -> bool:
    """Host-only query: does the fused small feed-forward (csrc/ff_small.hip) exist for Linear(d, f) -> ReLU -> Linear(f, d)?"""
    return (not _lib.py_switches()["no_ff_small"]) and bool(_lib.lib().kanvit_ff_small_supported(d, f))


def _ff_small_ok(x, lin1, lin2) -> bool:
    if _lib.py_switches()["no_ff_small"] or not x.is_cuda or x.dtype != torch.float32 or lin1.bias is None or lin2.bias is None:
        return False
    if ff_mode() == "bf16x3":
        return False                       # the opt-in split-product mode was asked for explicitly: it takes the feed-forward
    if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") != torch.float32:
        return False                       # bf16 autocast keeps the stock bf16 GEMMs
    if lin1.weight.dtype != torch.float32 or lin2.out_features != lin1.in_features or lin2.in_features != lin1.out_features:
        return False
    L = _lib.lib()
    return bool(L.kanvit_ff_small_supported(lin1.in_features, lin1.out_features)) and 0 < x.shape[0] <= int(L.kanvit_ff_small_max_rows())


def feed_forward(x: torch.Tensor, lin1: torch.nn.Linear, lin2: torch.nn.Linear) -> torch.Tensor:
    """lin2(relu(lin1(x))) for a 2-D x: the stock-GEMM path (`dense`) or, when FF_MODE == "bf16x3" and the shapes allow
    it (CUDA, fp32 parameters, widths multiples of 8, biases present, M divisible into slabs), the split-bf16 path."""
    if _ff_small_ok(x, lin1, lin2):
        return _FFSmallFn.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
    mode = ff_mode()
    if (mode == "bf16x3" and x.is_cuda and lin1.bias is not None and lin2.bias is not None and
            lin1.in_features % 8 == 0 and lin1.out_features % 8 == 0 and lin2.out_features % 8 == 0 and
            not (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") != torch.float32)):
        return _FFSplitFn.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
    return dense(dense(x, lin1, relu=True), lin2)
