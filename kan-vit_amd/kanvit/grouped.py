"""Launch helpers: one fused kernel per KAN layer, or ONE kernel for every head's q, k and v
mapping of an MSA block (3*H independent layers that share the rows of x).

A layer module takes part by exposing
    kan_cfg()  -> ops.LayerCfg (groups = 1)
    kan_pack() -> (w[K, O], bparams[stride] | None, bias[O] | None)   built with differentiable
                  torch ops from the module's reference-layout parameters, so autograd scatters
                  the packed-weight gradient back to cheby_coeffs / spline_weight / ... itself
    kan_u(x2d) -> u | None  (FastKAN: the LayerNorm'ed input of the spline path)
The packing is O(|W|) plumbing; every flop of the layer itself runs in the HIP kernel.
"""
from __future__ import annotations

from dataclasses import replace
from typing import Sequence

import torch
import torch.nn.functional as F

from . import ops


class _StackParams(torch.autograd.Function):
    """torch.stack over per-head parameters whose backward costs ONE kernel.

    Stock stack hands autograd 3*H strided slices of the packed-weight gradient; AccumulateGrad then clones each one
    (432 five-microsecond copies per ViT-B step).  Here the incoming gradient is made contiguous once and the per-parameter
    gradients are its unbound slices: contiguous, uniquely referenced, so AccumulateGrad adopts them without a copy."""

    @staticmethod
    def forward(ctx, *tensors):
        return torch.stack(tensors)

    @staticmethod
    def backward(ctx, grad):
        return grad.contiguous().unbind(0)


class _StackPermuted(torch.autograd.Function):
    """torch.stack over per-head parameters WITH a layout permutation, one kernel each way: forward stacks permuted VIEWS (torch.cat's
    strided-input path writes the packed layout directly -- stack followed by permute + reshape was a cat launch and a copy launch per
    block), backward makes the incoming gradient contiguous in the parameters' own layout once and hands out its unbound slices."""

    @staticmethod
    def forward(ctx, perm, *tensors):
        ctx.inv = tuple(sorted(range(len(perm)), key=lambda a: perm[a]))
        return torch.stack([t.permute(*perm) for t in tensors])

    @staticmethod
    def backward(ctx, grad):
        inv = (0,) + tuple(1 + a for a in ctx.inv)
        return (None,) + grad.permute(*inv).contiguous().unbind(0)


def stack_params(tensors: Sequence[torch.Tensor], perm: Sequence[int] = None) -> torch.Tensor:
    """torch.stack(tensors) -- of tensors.permute(perm) when perm is given -- whose backward costs one kernel (see above)."""
    tensors = list(tensors)
    if perm is not None:
        if any(t.requires_grad for t in tensors) and torch.is_grad_enabled():
            return _StackPermuted.apply(tuple(perm), *tensors)
        return torch.stack([t.permute(*perm) for t in tensors])
    if any(t.requires_grad for t in tensors) and torch.is_grad_enabled():
        return _StackParams.apply(*tensors)
    return torch.stack(tensors)


def linear_cfg(lin: torch.nn.Linear) -> ops.LayerCfg:
    return ops.LayerCfg(family=ops.LINEAR, I=lin.in_features, O=lin.out_features, G=1)


def linear_pack(lin: torch.nn.Linear):
    return lin.weight.t(), None, lin.bias


def _cfg_pack(m):
    if isinstance(m, torch.nn.Linear):
        return linear_cfg(m), linear_pack(m)
    return m.kan_cfg(), m.kan_pack()


def run_single(layer, x2d: torch.Tensor) -> torch.Tensor:
    """y[M, O] for one layer on a 2-D input."""
    cfg, (w, bp, bias) = _cfg_pack(layer)
    if hasattr(layer, "kan_ln") and layer.kan_ln() is not None and ops.ln_fusable(cfg, x2d.shape[0]):
        ln = layer.kan_ln()           # FastKAN: the LayerNorm of the spline path is formed inside the kernels
        return ops.kan_layer_ln(x2d, w.unsqueeze(0), cfg, torch.cat([bp, ln.weight, ln.bias]).unsqueeze(0),
                                None if bias is None else bias.reshape(1, -1), ln.eps)
    u = layer.kan_u(x2d) if hasattr(layer, "kan_u") else None
    return ops.kan_layer(x2d, w.unsqueeze(0), cfg, u=u,
                         bparams=None if bp is None else bp.unsqueeze(0),
                         bias=None if bias is None else bias.reshape(1, -1))


def run_qkv(q_layers: Sequence, k_layers: Sequence, v_layers: Sequence, x2d: torch.Tensor) -> torch.Tensor:
    """x2d[M, H*dh] -> qkv[M, 3*H*dh]; column block g = proj*H + head holds layer g's output.

    Equivalent to the reference's per-sample, per-head loop (attention.py:188-197) because every
    layer acts row-wise."""
    H = len(q_layers)
    layers = list(q_layers) + list(k_layers) + list(v_layers)
    if isinstance(layers[0], torch.nn.Linear):
        cfg0 = linear_cfg(layers[0])
    elif hasattr(type(layers[0]), "kan_pack_grouped") and "layers" in layers[0].kan_cfg.__code__.co_varnames:
        cfg0 = layers[0].kan_cfg(layers)          # family whose flags depend on all the layers (efficient-KAN knot tables)
    else:
        cfg0 = layers[0].kan_cfg()
    cfg = replace(cfg0, groups=3 * H, x_group_mod=H)
    # Pack ALL 3*H layers with one stack + one layout transform (a handful of launches) instead of a
    # permute-copy per head (3*H launches forward and again backward: 880 tiny kernels per ViT-B step).
    if isinstance(layers[0], torch.nn.Linear):
        w = stack_params([m.weight for m in layers], perm=(1, 0))      # [g, I, O]: nn.Linear keeps [O, I]
        bp = None
        bias = None if layers[0].bias is None else stack_params([m.bias for m in layers])
    else:
        w, bp, bias = type(layers[0]).kan_pack_grouped(layers)
    if hasattr(layers[0], "kan_ln") and layers[0].kan_ln() is not None and ops.ln_fusable(cfg, x2d.shape[0]):
        gamma = stack_params([l.kan_ln().weight for l in layers])
        beta = stack_params([l.kan_ln().bias for l in layers])
        return ops.kan_layer_ln(x2d, w, cfg, torch.cat([bp, gamma, beta], dim=1), bias, layers[0].kan_ln().eps)
    u = None
    if hasattr(layers[0], "kan_u_grouped"):
        u = type(layers[0]).kan_u_grouped(layers, x2d, H)
    return ops.kan_layer(x2d, w, cfg, u=u, bparams=bp, bias=bias)
