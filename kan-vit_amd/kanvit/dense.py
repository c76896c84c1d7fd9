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

_TILE = 256
_HALF = (torch.bfloat16, torch.float16)
_CUS = 256


def wgrad_slabs(M: int, N: int, K: int) -> int:
    """Number of token slabs: the smallest divisor of M (<= 32) that yields >= one macro-tile per CU while keeping at
    least 512 tokens per slab; 1 when the output alone already fills the chip or M is too small to split."""
    tiles = -(-N // _TILE) * -(-K // _TILE)
    if tiles >= _CUS:
        return 1
    best = 1
    for s in range(2, 33):
        if M % s or M // s < 512:
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
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = (dy.sum(0, dtype=torch.float32) if dy.dtype in _HALF else dy.sum(0)).to(wdt)
        return dx, dw, db, None


def dense(x: torch.Tensor, linear: torch.nn.Linear, relu: bool = False) -> torch.Tensor:
    """linear(x) (then ReLU if asked) for a 2-D x, with the token-split weight gradient in backward."""
    assert x.dim() == 2
    return _DenseFn.apply(x, linear.weight, linear.bias, relu)
