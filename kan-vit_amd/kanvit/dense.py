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
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        dy = dy.contiguous()
        if ctx.needs_input_grad[0]:
            dx = dy @ weight.to(dy.dtype)
        if ctx.needs_input_grad[1]:
            M, N = dy.shape
            K = x.shape[1]
            xs = x.to(dy.dtype)
            S = wgrad_slabs(M, N, K)
            if S > 1:
                part = torch.bmm(dy.view(S, M // S, N).transpose(1, 2), xs.view(S, M // S, K))
                dw = part.sum(0, dtype=torch.float32) if part.dtype in _HALF else part.sum(0)
            else:
                dw = dy.t() @ xs
            dw = dw.to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = (dy.sum(0, dtype=torch.float32) if dy.dtype in _HALF else dy.sum(0)).to(weight.dtype)
        return dx, dw, db


def dense(x: torch.Tensor, linear: torch.nn.Linear) -> torch.Tensor:
    """linear(x) for a 2-D x, with the token-split weight gradient in backward."""
    assert x.dim() == 2
    return _DenseFn.apply(x, linear.weight, linear.bias)
