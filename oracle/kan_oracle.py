"""CPU oracle for the ViKANformer hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, in plain functional PyTorch on the CPU, the arithmetic of
the reference path named by BASELINE.json's north_star (the five KAN layers,
the per-head multi-head attention, the tiled "flash" attention function, the
VisionTransformer assembly and one Adam train step).  It exists only so that
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can CHECK (or
time, as a CPU baseline) the HIP kernels.  Nothing under kan-vit_amd/ imports
it; the product path fails loudly when the HIP library is missing.

Pinning: every function here is checked against tensors produced by importing
the real reference in the build container (tests/golden/make_golden.py ->
tests/golden/*.npz; tests/test_oracle_golden.py), so parity is PINNED, not
merely restated.

Conventions
-----------
* All functions are dtype generic: pass float64 tensors for a high-precision
  truth, float32 tensors for reference-like rounding.
* Parameters are passed as plain tensors using the reference's state_dict
  names/shapes, so a reference ``state_dict()`` can be
  fed directly.
* Gradients come from torch.autograd over these functions -- the same
  mechanism the reference itself relies on (it has no hand-written backward
  except utils.py:229-295, restated in ``flash_attention_backward``).

All ``file:line`` citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# Operand rounding (checker for the bf16 matrix-core mode of the HIP kernels)
# --------------------------------------------------------------------------
# The bf16 mode of the kernels (BASELINE configs[2], [4]) rounds the two operands of every contraction -- the generated
# basis values Phi(x) and the packed coefficients -- to bf16 and accumulates in fp32.  Inside `operand_rounding(fn)` the
# layer functions below apply `fn` to exactly those operands -- in the forward products AND in the products autograd forms
# for the backward (_RoundedMM, _RoundedAttention) -- so a float64 evaluation becomes a TIGHT oracle for the bf16 kernels
# (agreement ~1e-4..1e-3 instead of the ~1e-2, or far worse on cancelling sums, that unrounded arithmetic allows).  The reference has
# no such mode (train.py has no AMP; SURVEY.md section 7 "bf16 parity"); with no rounding installed -- the default --
# these functions are the plain restatement of the reference.
_ROUND = None
# bench.py's cpu_baseline leg sets this: ChebyKAN / SineKAN then run the reference's own (slower) op sequences, so that the
# timed CPU port costs what the reference costs (tools/calibrate_cpu_baseline.py: within 10 % for every type).  Values are
# identical either way (tests/test_oracle_golden.py runs both).
REFERENCE_OP_SEQUENCE = False


class operand_rounding:
    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _ROUND
        self.prev, _ROUND = _ROUND, self.fn
        return self

    def __exit__(self, *exc):
        global _ROUND
        _ROUND = self.prev
        return False


def bf16_round(t: Tensor) -> Tensor:
    """Round to the nearest bf16-representable value (ties to even), keeping the dtype."""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _RoundedMM(torch.autograd.Function):
    """a @ b with both operands rounded, and -- as the bf16 kernels do -- the BACKWARD products formed from rounded operands
    too: dA = r(dOut) r(B)^T (input-gradient kernel: dY and W^T rounded), dB = r(A)^T r(dOut) (weight-gradient kernel: Phi and
    dY rounded).  Everything else (basis derivatives, bias sums, LayerNorm) stays unrounded, as in the kernels."""

    @staticmethod
    def forward(ctx, a, b, fn):
        ctx.fn = fn
        ctx.save_for_backward(a, b)
        return fn(a) @ fn(b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        r = ctx.fn
        return r(g) @ r(b).transpose(-1, -2), r(a).transpose(-1, -2) @ r(g), None


def _mm(a: Tensor, b: Tensor) -> Tensor:
    """The contraction of a layer: plain a @ b, or its operand-rounded form inside `operand_rounding`."""
    return a @ b if _ROUND is None else _RoundedMM.apply(a, b, _ROUND)


class _RoundedAttention(torch.autograd.Function):
    """softmax(q k^T scale) v with the rounding points of the bf16 attention kernels (csrc/attention.hip):
    forward   S = r(q) r(k)^T;  p = exp(scale S - max) (fp32, unrounded);  o = r(p) r(v) / sum(p)
              (first-form kernels, head sizes other than 32 / 64, round the NORMALISED probabilities: `norm_first`)
    backward  P = exp(scale S - lse), dP = r(do) r(v)^T, delta = rowsum(do * o) (unrounded), dS = P scale (dP - delta);
              dv = r(P)^T r(do);  dk = r(dS)^T r(q);  dq = r(dS) r(k)."""

    @staticmethod
    def forward(ctx, q, k, v, fn, norm_first):
        scale = q.shape[-1] ** -0.5
        s = (fn(q) @ fn(k).transpose(-1, -2)) * scale
        mx = s.amax(dim=-1, keepdim=True)
        p = torch.exp(s - mx)
        l = p.sum(dim=-1, keepdim=True)
        o = fn(p / l) @ fn(v) if norm_first else (fn(p) @ fn(v)) / l
        ctx.fn, ctx.scale = fn, scale
        ctx.save_for_backward(q, k, v, o, (l.log() + mx))
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        r, scale = ctx.fn, ctx.scale
        p = torch.exp((r(q) @ r(k).transpose(-1, -2)) * scale - lse)
        dp = r(do) @ r(v).transpose(-1, -2)
        delta = (do * o).sum(dim=-1, keepdim=True)
        ds = p * scale * (dp - delta)
        return r(ds) @ r(k), r(ds).transpose(-1, -2) @ r(q), r(p).transpose(-1, -2) @ r(do), None, None


def _attention_core(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """softmax(q k^T / sqrt(dh)) v on (..., N, dh) tensors: plain, or with the bf16 kernels' rounding points."""
    if _ROUND is None:
        return torch.softmax(q @ k.transpose(-1, -2) / (q.shape[-1] ** 0.5), dim=-1) @ v
    return _RoundedAttention.apply(q, k, v, _ROUND, q.shape[-1] not in (32, 64))


# --------------------------------------------------------------------------
# a1  ChebyKANLayer.forward          models/cheby.py:36-48
# --------------------------------------------------------------------------
def cheby_forward(x: Tensor, cheby_coeffs: Tensor, faithful: bool = True) -> Tensor:
    """y[m,o] = sum_i sum_d T_d(tanh x[m,i]) * C[i,o,d]; output is ALWAYS 2-D
    (prod(leading), O) like the reference (models/cheby.py:38,47 -- SURVEY D3).

    faithful=True evaluates T_d as cos(d*acos(t)) (models/cheby.py:41-43);
    faithful=False uses the three-term recurrence the HIP kernel uses.
    """
    in_dim, out_dim, deg1 = cheby_coeffs.shape
    t = torch.tanh(x).reshape(-1, in_dim)
    if faithful and REFERENCE_OP_SEQUENCE and _ROUND is None:
        # the reference's own op sequence (models/cheby.py:37-46), for the CPU-baseline timing: acos and cos run on the
        # EXPANDED (M, I, D+1) tensor and the contraction is an einsum over (i, d) -- same values as below, ~1.5x the time
        th = t.reshape(-1, in_dim, 1).expand(-1, -1, deg1).acos()
        th = th * torch.arange(deg1, dtype=t.dtype)
        return torch.einsum("bid,iod->bo", th.cos(), cheby_coeffs)
    if faithful:
        d = torch.arange(deg1, dtype=t.dtype)
        basis = torch.cos(torch.acos(t).unsqueeze(-1) * d)          # (M, I, D+1)
    else:
        cols = [torch.ones_like(t)]
        if deg1 > 1:
            cols.append(t)
        for _ in range(2, deg1):
            cols.append(2.0 * t * cols[-1] - cols[-2])
        basis = torch.stack(cols, dim=-1)
    # contraction over (i, d)  (models/cheby.py:44-46)
    y = _mm(basis.reshape(t.shape[0], in_dim * deg1), cheby_coeffs.permute(0, 2, 1).reshape(in_dim * deg1, out_dim))
    return y


# --------------------------------------------------------------------------
# a2  KANLinear.forward / b_splines   models/effkan.py:99-132,166-187
# --------------------------------------------------------------------------
def bspline_bases(x2d: Tensor, grid: Tensor, spline_order: int) -> Tensor:
    """Cox-de Boor recursion on the per-feature knot vector ``grid[I, nk]``.

    Order-0 bases are the half-open indicators [g_j, g_{j+1})
    (models/effkan.py:115); each level k blends neighbours with the two
    linear ramps of models/effkan.py:117-125.  Returns (M, I, nk-1-order).
    """
    xe = x2d.unsqueeze(-1)
    b = ((xe >= grid[:, :-1]) & (xe < grid[:, 1:])).to(x2d.dtype)
    for k in range(1, spline_order + 1):
        left = (xe - grid[:, : -(k + 1)]) / (grid[:, k:-1] - grid[:, : -(k + 1)])
        right = (grid[:, k + 1:] - xe) / (grid[:, k + 1:] - grid[:, 1:-k])
        b = left * b[:, :, :-1] + right * b[:, :, 1:]
    return b


def kanlinear_forward(x: Tensor, base_weight: Tensor, spline_weight: Tensor,
                      spline_scaler: Optional[Tensor], grid: Tensor, spline_order: int = 3) -> Tensor:
    """y = silu(x) Wb^T + B(x).view(M, I*nb) (Ws * scaler[...,None]).view(O,-1)^T
    (models/effkan.py:174-187, scaled_spline_weight :166-172). Keeps leading dims."""
    out_f, in_f = base_weight.shape
    lead = x.shape[:-1]
    x2 = x.reshape(-1, in_f)
    w = spline_weight if spline_scaler is None else spline_weight * spline_scaler.unsqueeze(-1)
    bases = bspline_bases(x2, grid, spline_order)
    y = _mm(F.silu(x2), base_weight.t()) + _mm(bases.reshape(x2.shape[0], -1), w.reshape(out_f, -1).t())
    return y.reshape(*lead, out_f)


def make_uniform_knots(in_features: int, grid_size: int = 5, spline_order: int = 3,
                       grid_range: Sequence[float] = (-1.0, 1.0), dtype=torch.float32) -> Tensor:
    """Knot buffer built at construction (models/effkan.py:44-53)."""
    h = (grid_range[1] - grid_range[0]) / grid_size
    k = torch.arange(-spline_order, grid_size + spline_order + 1, dtype=torch.float32) * h + grid_range[0]
    return k.to(dtype).expand(in_features, -1).contiguous()


# --------------------------------------------------------------------------
# a3  FastKANLayer.forward            models/fastkan.py:66-76, 29-30
# --------------------------------------------------------------------------
def fastkan_forward(x: Tensor, ln_weight: Tensor, ln_bias: Tensor, rbf_grid: Tensor,
                    spline_weight: Tensor, base_weight: Optional[Tensor], base_bias: Optional[Tensor],
                    denominator: Optional[float] = None, use_layernorm: bool = True) -> Tensor:
    """y = Wsp vec(exp(-((LN(x)_i - g_k)/h)^2)) + Wb silu(x) + b.
    Spline path sees LayerNorm(x) (eps 1e-5), base path sees raw x
    (models/fastkan.py:68,74)."""
    in_f = x.shape[-1]
    ng = rbf_grid.numel()
    if denominator is None:
        denominator = float((rbf_grid[-1] - rbf_grid[0]).item()) / (ng - 1)     # models/fastkan.py:26-27
    u = F.layer_norm(x, (in_f,), ln_weight, ln_bias, 1e-5) if use_layernorm else x
    phi = torch.exp(-(((u.unsqueeze(-1) - rbf_grid) / denominator) ** 2))        # (..., I, ng)
    y = _mm(phi.reshape(*x.shape[:-1], in_f * ng), spline_weight.t())
    if base_weight is not None:
        y = y + _mm(F.silu(x), base_weight.t()) + base_bias
    return y


# --------------------------------------------------------------------------
# a4  NaiveFourierKANLayer.forward    models/nfkan.py:36-52
# --------------------------------------------------------------------------
def fourier_forward(x: Tensor, fouriercoeffs: Tensor, bias: Optional[Tensor]) -> Tensor:
    """y[m,o] = sum_i sum_{k=1..G} cos(k x_mi) F[0,o,i,k-1] + sin(k x_mi) F[1,o,i,k-1] (+ bias).
    The reference materialises an (M,O,I,G) product (models/nfkan.py:47-48);
    the contraction below is the same sum without that tensor."""
    _, out_f, in_f, g = fouriercoeffs.shape
    lead = x.shape[:-1]
    x2 = x.reshape(-1, in_f)
    k = torch.arange(1, g + 1, dtype=x.dtype)
    ang = x2.unsqueeze(-1) * k                                   # (M, I, G)
    y = _mm(torch.cos(ang).reshape(x2.shape[0], -1), fouriercoeffs[0].reshape(out_f, -1).t())
    y = y + _mm(torch.sin(ang).reshape(x2.shape[0], -1), fouriercoeffs[1].reshape(out_f, -1).t())
    if bias is not None:
        y = y + bias
    return y.reshape(*lead, out_f)


# --------------------------------------------------------------------------
# a5  SineKANLayer.forward            models/sinekan.py:81-91, ctor :27-79
# --------------------------------------------------------------------------
SINE_A, SINE_K, SINE_C = 0.9724108095811765, 0.9884401790754128, 0.999449553483052   # models/sinekan.py:47


def sine_phase(input_dim: int, grid_size: int) -> Tensor:
    """Fixed phase buffer (1,1,I,G): (g/(G+1) + pi*i/(I-1)) * prod_{n=1}^{G-1}(A n^-K + C)
    (models/sinekan.py:59-75, forward_step :7-23), evaluated in float32 like the reference."""
    gp = torch.arange(1, grid_size + 1).reshape(1, 1, 1, grid_size) / (grid_size + 1)
    ip = torch.linspace(0, math.pi, input_dim).reshape(1, 1, input_dim, 1)
    ph = gp + ip
    for n in range(1, grid_size):
        ph = (SINE_A * n ** (-SINE_K) + SINE_C) * ph
    return ph


def sine_forward(x: Tensor, amplitudes: Tensor, freq: Tensor, phase: Tensor, bias: Optional[Tensor]) -> Tensor:
    """y[m,o] = sum_i sum_g sin(x_mi * freq_g + phase_ig) * A[o,i,g] (+ bias) (models/sinekan.py:81-91)."""
    out_f, in_f, g = amplitudes.shape
    lead = x.shape[:-1]
    x2 = x.reshape(-1, in_f)
    if REFERENCE_OP_SEQUENCE and _ROUND is None:
        # the reference's own op sequence (models/sinekan.py:84-89), for the CPU-baseline timing: a 4-D broadcast and an
        # einsum over (input, grid) against amplitudes[O, I, G] -- same values as below
        s4 = torch.sin(x2.reshape(x2.shape[0], 1, in_f, 1) * freq.reshape(1, 1, 1, g) + phase.reshape(1, 1, in_f, g))
        y = torch.einsum("ijkl,jkl->ij", s4, amplitudes)
        if bias is not None:
            y = y + bias
        return y.reshape(*lead, out_f)
    s = torch.sin(x2.reshape(-1, in_f, 1) * freq.reshape(1, 1, g) + phase.reshape(1, in_f, g))
    y = _mm(s.reshape(x2.shape[0], -1), amplitudes.reshape(out_f, -1).t())
    if bias is not None:
        y = y + bias
    return y.reshape(*lead, out_f)


# --------------------------------------------------------------------------
# generic dispatch on the reference's state_dict layout
# --------------------------------------------------------------------------
def layer_kind(sd: Dict[str, Tensor], prefix: str) -> str:
    if prefix + "cheby_coeffs" in sd:
        return "cheby"
    if prefix + "spline_weight" in sd:
        return "efficientkan"
    if prefix + "spline_linear.weight" in sd:
        return "fast"
    if prefix + "fouriercoeffs" in sd:
        return "fourier"
    if prefix + "amplitudes" in sd:
        return "sine"
    if prefix + "weight" in sd:
        return "linear"
    raise KeyError(f"no known layer under prefix {prefix!r}")


def layer_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, cheby_keep_2d: bool = True) -> Tensor:
    """Apply whichever layer lives under ``prefix`` in a reference-layout state dict."""
    kind = layer_kind(sd, prefix)
    g = lambda n: sd[prefix + n]
    if kind == "cheby":
        y = cheby_forward(x, g("cheby_coeffs"))
        return y if cheby_keep_2d else y.reshape(*x.shape[:-1], -1)
    if kind == "efficientkan":
        return kanlinear_forward(x, g("base_weight"), g("spline_weight"), sd.get(prefix + "spline_scaler"), g("grid"))
    if kind == "fast":
        return fastkan_forward(x, g("layernorm.weight"), g("layernorm.bias"), g("rbf.grid"),
                               g("spline_linear.weight"), sd.get(prefix + "base_linear.weight"),
                               sd.get(prefix + "base_linear.bias"))
    if kind == "fourier":
        return fourier_forward(x, g("fouriercoeffs"), sd.get(prefix + "bias"))
    if kind == "sine":
        return sine_forward(x, g("amplitudes"), g("freq"), g("phase"), sd.get(prefix + "bias"))
    y = _mm(x, g("weight").t())                          # nn.Linear (attention.py:136-142)
    b = sd.get(prefix + "bias")
    return y if b is None else y + b


# --------------------------------------------------------------------------
# a6  MSA.forward                     attention.py:181-202
# --------------------------------------------------------------------------
def msa_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, n_heads: int, faithful_loop: bool = False) -> Tensor:
    """softmax(q k^T / sqrt(dh)) v per sample and head, heads concatenated, no
    out-projection (attention.py:188-202).

    faithful_loop=True walks sample by sample and head by head exactly like the
    reference (the structure whose CPU time bench.py reports); False folds the
    batch into the row dimension (identical results)."""
    bsz, n, d = x.shape
    dh = d // n_heads
    if faithful_loop:
        outs = []
        for b in range(bsz):
            heads = []
            for h in range(n_heads):
                seq = x[b, :, h * dh:(h + 1) * dh]
                q = layer_forward(sd, f"{prefix}q_mappings.{h}.", seq)
                k = layer_forward(sd, f"{prefix}k_mappings.{h}.", seq)
                v = layer_forward(sd, f"{prefix}v_mappings.{h}.", seq)
                heads.append(_attention_core(q, k, v))
            outs.append(torch.cat(heads, dim=1))
        return torch.stack(outs, dim=0)
    heads = []
    for h in range(n_heads):
        seq = x[:, :, h * dh:(h + 1) * dh].reshape(bsz * n, dh)
        q = layer_forward(sd, f"{prefix}q_mappings.{h}.", seq).reshape(bsz, n, dh)
        k = layer_forward(sd, f"{prefix}k_mappings.{h}.", seq).reshape(bsz, n, dh)
        v = layer_forward(sd, f"{prefix}v_mappings.{h}.", seq).reshape(bsz, n, dh)
        heads.append(_attention_core(q, k, v))
    return torch.cat(heads, dim=-1)


# --------------------------------------------------------------------------
# a7  FlashAttentionFunction          utils.py:134-295 ; FlashAttention attention.py:13-109
# --------------------------------------------------------------------------
def attention_dead(q: Tensor, k: Tensor, causal: bool = False, mask: Optional[Tensor] = None) -> Optional[Tensor]:
    """Boolean [.., q_len, k_len] map of the positions FlashAttentionFunction excludes: a mask entry that is False (a (b, n)
    key-padding mask is viewed as (b, 1, 1, n), utils.py:156-157) and, with `causal`, key j > query i (utils.py:178-190:
    triu(q_start_index - k_start_index + 1)).  causal with k_len > q_len is refused: there utils.py:169 subtracts
    qk_len_diff from q_start_index, query i sees keys j <= i - (k_len - q_len), the first k_len - q_len queries see none and
    the reference's answer for them depends on the bucket sizes (uniform weights over causally masked keys of some buckets)."""
    nq, nk = q.shape[-2], k.shape[-2]
    if causal and nk > nq:
        raise ValueError("causal attention with k_len > q_len is ill-defined in the reference (utils.py:169,183)")
    dead = None
    if mask is not None:
        if mask.dim() == 2:
            mask = mask[:, None, None, :]
        dead = ~mask.to(torch.bool).expand(q.shape[:-2] + (nq, nk))
    if causal:
        c = torch.ones(nq, nk, dtype=torch.bool).triu(1)
        dead = c.expand(q.shape[:-2] + (nq, nk)) if dead is None else (dead | c)
    return dead


def attention_reference(q: Tensor, k: Tensor, v: Tensor, causal: bool = False, mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Dense softmax(q k^T d^-1/2) v and the row log-sum-exp; q (..., q_len, D), k / v (..., k_len, D); excluded positions
    (attention_dead) take no weight; a query with no live key gets o = 0 (the reference's clamp(min=EPSILON) row sum,
    utils.py:204-205,223) and lse = -max."""
    scale = q.shape[-1] ** -0.5
    s = (q @ k.transpose(-1, -2)) * scale
    dead = attention_dead(q, k, causal, mask)
    if dead is None:
        return torch.softmax(s, dim=-1) @ v, torch.logsumexp(s, dim=-1)
    s = s.masked_fill(dead, -torch.finfo(s.dtype).max)
    p = torch.softmax(s, dim=-1).masked_fill(dead, 0.0)
    lse = torch.logsumexp(s, dim=-1)
    none_live = dead.all(dim=-1)
    lse = torch.where(none_live, torch.full_like(lse, -torch.finfo(s.dtype).max), lse)
    return p @ v, lse


def flash_attention_tiled(q: Tensor, k: Tensor, v: Tensor, q_bucket: int, k_bucket: int) -> Tuple[Tensor, Tensor]:
    """Row/column tiled online-softmax forward (utils.py:166-225): running row max
    and row sum per q tile, rescale of the partial output on every k tile, final
    division, lse = log(sum) + max."""
    scale = q.shape[-1] ** -0.5
    o = torch.zeros_like(q)
    lse = q.new_zeros(q.shape[:-1])
    for r0 in range(0, q.shape[-2], q_bucket):
        qc = q[..., r0:r0 + q_bucket, :]
        acc = torch.zeros_like(qc)
        run_max = qc.new_full(qc.shape[:-1] + (1,), -torch.finfo(q.dtype).max)
        run_sum = qc.new_zeros(qc.shape[:-1] + (1,))
        for c0 in range(0, k.shape[-2], k_bucket):
            s = (qc @ k[..., c0:c0 + k_bucket, :].transpose(-1, -2)) * scale
            new_max = torch.maximum(s.amax(dim=-1, keepdim=True), run_max)
            p = torch.exp(s - new_max)
            corr = torch.exp(run_max - new_max)
            run_sum = corr * run_sum + p.sum(dim=-1, keepdim=True).clamp(min=1e-10)
            acc = acc * corr + p @ v[..., c0:c0 + k_bucket, :]
            run_max = new_max
        o[..., r0:r0 + q_bucket, :] = acc / run_sum
        lse[..., r0:r0 + q_bucket] = (run_sum.log() + run_max).squeeze(-1)
    return o, lse


def flash_attention_backward(q: Tensor, k: Tensor, v: Tensor, o: Tensor, lse: Tensor, do: Tensor, causal: bool = False,
                             mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Recompute-based backward (utils.py:229-295) without tiling:
    p = exp(s - lse) (0 on excluded positions, utils.py:272-281); dv = p^T do; dp = do v^T; D = rowsum(do*o);
    ds = p*scale*(dp - D); dq = ds k; dk = ds^T q."""
    scale = q.shape[-1] ** -0.5
    s = (q @ k.transpose(-1, -2)) * scale
    dead = attention_dead(q, k, causal, mask)
    if dead is not None:
        s = s.masked_fill(dead, -torch.finfo(s.dtype).max)
    p = torch.exp(s - lse.unsqueeze(-1))
    if dead is not None:
        p = p.masked_fill(dead, 0.0)
    dv = p.transpose(-1, -2) @ do
    dp = do @ v.transpose(-1, -2)
    dsum = (do * o).sum(dim=-1, keepdim=True)
    ds = p * scale * (dp - dsum)
    return ds @ k, ds.transpose(-1, -2) @ q, dv


def flash_module_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, heads: int, dim_head: int = 64) -> Tensor:
    """FlashAttention.forward (attention.py:59-109): bias-free to_q / to_kv / to_out."""
    b, n, _ = x.shape
    q = x @ sd[prefix + "to_q.weight"].t()
    kv = x @ sd[prefix + "to_kv.weight"].t()
    k, v = kv[..., : heads * dim_head], kv[..., heads * dim_head:]
    split = lambda t: t.reshape(b, n, heads, dim_head).permute(0, 2, 1, 3)
    o, _ = attention_reference(split(q), split(k), split(v))
    return o.permute(0, 2, 1, 3).reshape(b, n, heads * dim_head) @ sd[prefix + "to_out.weight"].t()


# --------------------------------------------------------------------------
# a8  VisionTransformer                model.py:14-169
# --------------------------------------------------------------------------
def patchify(images: Tensor, n_patches: int) -> Tensor:
    """Non-overlapping (C, ph, pw)-ordered flatten, patches row-major (model.py:111-126)."""
    b, c, h, w = images.shape
    ph, pw = h // n_patches, w // n_patches
    t = images.reshape(b, c, n_patches, ph, n_patches, pw).permute(0, 2, 4, 1, 3, 5)
    return t.reshape(b, n_patches * n_patches, c * ph * pw)


def positional_embeddings(seq_len: int, d: int) -> Tensor:
    """pe[i,j] = sin(i / 10000^(j/d)) for even j, cos(...) for odd j, computed in
    float64 and stored as float32 (model.py:128-140; note the exponent is j/d)."""
    i = torch.arange(seq_len, dtype=torch.float64).unsqueeze(1)
    j = torch.arange(d, dtype=torch.float64).unsqueeze(0)
    ang = i / torch.pow(torch.tensor(10000.0, dtype=torch.float64), j / d)
    pe = torch.where((torch.arange(d) % 2 == 0).unsqueeze(0), torch.sin(ang), torch.cos(ang))
    return pe.to(torch.float32)


def vit_forward(sd: Dict[str, Tensor], images: Tensor, n_patches: int, n_heads: int, model_type: str,
                faithful_loop: bool = False) -> Tensor:
    """VisionTransformer.forward (model.py:142-169) + TransformerBlock.forward (:31-37).
    n_blocks / d_hidden are read off the state dict.  For model_type 'cheby' the
    patch-embedding output is reshaped back to (B, P, d) -- the adapter SURVEY D3
    describes (the shipped reference crashes there)."""
    d = sd["v_class"].shape[1]
    n_blocks = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    dt = sd["v_class"].dtype
    patches = patchify(images.to(dt), n_patches)
    tok = layer_forward(sd, "linear_mapper.", patches).reshape(patches.shape[0], patches.shape[1], d)
    tok = torch.cat([sd["v_class"].unsqueeze(0).expand(tok.shape[0], -1, -1), tok], dim=1)
    out = tok + positional_embeddings(tok.shape[1], d).to(dt)
    for l in range(n_blocks):
        p = f"blocks.{l}."
        if model_type == "flash-attn":
            out = flash_module_forward(sd, p, out, n_heads)            # model.py:156-159: no residual / FF
            continue
        hN = F.layer_norm(out, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        out = out + msa_forward(sd, p + "attn.", hN, n_heads, faithful_loop)
        hN = F.layer_norm(out, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
        ff = F.linear(F.relu(F.linear(hN, sd[p + "ff.0.weight"], sd[p + "ff.0.bias"])),
                      sd[p + "ff.2.weight"], sd[p + "ff.2.bias"])
        out = out + ff
    cls = F.layer_norm(out[:, 0], (d,), sd["mlp_head.0.weight"], sd["mlp_head.0.bias"], 1e-5)
    return F.linear(cls, sd["mlp_head.1.weight"], sd["mlp_head.1.bias"])


# --------------------------------------------------------------------------
# a9  one train step                   train.py:31-40
# --------------------------------------------------------------------------
BUFFER_SUFFIXES = ("arange", "grid", "phase", "rbf.grid")


def is_buffer_key(key: str) -> bool:
    """state_dict entries that are not trained (buffers, and FastKAN's frozen rbf.grid)."""
    return key.endswith(BUFFER_SUFFIXES)


def train_steps(sd: Dict[str, Tensor], images: Tensor, labels: Tensor, n_patches: int, n_heads: int,
                model_type: str, steps: int = 1, lr: float = 1e-3, faithful_loop: bool = False
                ) -> Tuple[List[float], Dict[str, Tensor]]:
    """CrossEntropy -> backward -> Adam(lr, default betas/eps) in the order of train.py:34-40.
    Returns the loss trajectory and the updated state dict."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not is_buffer_key(k)}
    fixed = {k: v for k, v in sd.items() if is_buffer_key(k)}
    opt = torch.optim.Adam(list(params.values()), lr=lr)
    losses = []
    for _ in range(steps):
        logits = vit_forward({**params, **fixed}, images, n_patches, n_heads, model_type, faithful_loop)
        loss = F.cross_entropy(logits, labels)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, {**{k: v.detach() for k, v in params.items()}, **fixed}


# --------------------------------------------------------------------------
# RNG-independent known-answer parameter fill
# --------------------------------------------------------------------------
def kat_fill(t: Tensor) -> Tensor:
    """0.1 * linspace(-1, 1, numel) computed in float64, cast to float32, reshaped."""
    return (0.1 * torch.linspace(-1, 1, t.numel(), dtype=torch.float64)).to(torch.float32).reshape(t.shape)
