"""FastKANLayer (+ RadialBasisFunction, SplineLinear) -- drop-in for models/fastkan.py:6-76
(family RBF).  The Gaussian basis, the silu base path and both contractions run in one kernel;
the LayerNorm in front of the spline path (models/fastkan.py:68) is torch's native op for now
(SURVEY.md section 8f lists its fusion as a next step)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from kanvit import grouped, ops


_RBF_FACTS = {}


def _rbf_grid_is_default(layers) -> bool:
    """True when every layer's centres are c0 + j*h with h = the layer's denominator and num_grids = 8 -- the grid
    models/fastkan.py:22-27 builds -- and all layers agree.  Then the kernels may use the two-anchor recurrence
    (KANVIT_FLAG_UNIFORM_KNOTS).  Checked once per set of grid buffers (one host sync), cached on their version counters."""
    key = tuple((id(m), m.rbf.grid._version, m.rbf.grid.data_ptr(), float(m.rbf.denominator)) for m in layers)
    hit = _RBF_FACTS.get(key)
    if hit is None:
        l0 = layers[0]
        g = l0.rbf.grid.detach().double()
        h = float(l0.rbf.denominator)
        ok = g.numel() == 8 and h > 0
        if ok:
            ideal = g[0] + h * torch.arange(8, device=g.device, dtype=torch.float64)
            ok = float((g - ideal).abs().max()) <= 2e-6 * (float(g.abs().max()) + 1.0)
        ok = ok and all(float(m.rbf.denominator) == h and m.rbf.grid.shape == l0.rbf.grid.shape and bool((m.rbf.grid == l0.rbf.grid).all())
                        for m in layers[1:])
        if len(_RBF_FACTS) > 256:
            _RBF_FACTS.clear()
        hit = _RBF_FACTS[key] = bool(ok)
    return hit


class SplineLinear(nn.Linear):
    """Bias-free linear whose weight starts as trunc_normal(0, init_scale) (models/fastkan.py:6-12)."""

    def __init__(self, in_features: int, out_features: int, init_scale: float = 0.1, **kw) -> None:
        self.init_scale = init_scale
        super().__init__(in_features, out_features, bias=False, **kw)

    def reset_parameters(self) -> None:
        nn.init.trunc_normal_(self.weight, mean=0, std=self.init_scale)


class RadialBasisFunction(nn.Module):
    """Holds the (frozen) centres ``grid`` and the width ``denominator`` (models/fastkan.py:15-30).
    Its forward is only used for stand-alone calls; inside FastKANLayer the kernel evaluates it."""

    def __init__(self, grid_min: float = -2., grid_max: float = 2., num_grids: int = 8, denominator: float = None):
        super().__init__()
        self.grid = torch.nn.Parameter(torch.linspace(grid_min, grid_max, num_grids), requires_grad=False)
        self.denominator = denominator or (grid_max - grid_min) / (num_grids - 1)

    def forward(self, x):
        eye = torch.eye(self.grid.numel(), device=x.device, dtype=x.dtype)
        n = self.grid.numel()
        cfg = ops.LayerCfg(family=ops.RBF, I=1, O=n, G=n, has_base=0, rbf_inv_h=1.0 / self.denominator)
        y = ops.kan_layer(x.reshape(-1, 1), eye.unsqueeze(0), cfg, bparams=self.grid.reshape(1, -1))
        return y.reshape(*x.shape, n)


class FastKANLayer(nn.Module):
    """y = spline_linear(rbf(layernorm(x))) + base_linear(silu(x)) (models/fastkan.py:66-76).

    state_dict keys as the reference: layernorm.{weight,bias}, rbf.grid, spline_linear.weight
    [O, I*num_grids] (column i*num_grids + k), base_linear.{weight,bias}."""

    def __init__(self, input_dim: int, output_dim: int, grid_min: float = -2., grid_max: float = 2.,
                 num_grids: int = 8, use_base_update: bool = True, base_activation=F.silu,
                 spline_weight_init_scale: float = 0.1) -> None:
        super().__init__()
        self.input_dim, self.output_dim, self.num_grids = input_dim, output_dim, num_grids
        self.layernorm = nn.LayerNorm(input_dim)
        self.rbf = RadialBasisFunction(grid_min, grid_max, num_grids)
        self.spline_linear = SplineLinear(input_dim * num_grids, output_dim, spline_weight_init_scale)
        self.use_base_update = use_base_update
        if use_base_update:
            if base_activation is not F.silu:
                raise NotImplementedError("the fused kernel implements the reference's default base_activation (silu)")
            self.base_activation = base_activation
            self.base_linear = nn.Linear(input_dim, output_dim)

    def kan_cfg(self, layers=None):
        from kanvit import _lib
        uni = _rbf_grid_is_default(layers if layers is not None else [self])
        return ops.LayerCfg(family=ops.RBF, I=self.input_dim, O=self.output_dim, G=self.num_grids,
                            has_base=int(self.use_base_update), rbf_inv_h=1.0 / float(self.rbf.denominator),
                            flags=_lib.FLAG_UNIFORM_KNOTS if uni else 0)

    def kan_pack(self):
        i, g, o = self.input_dim, self.num_grids, self.output_dim
        w = self.spline_linear.weight.view(o, i, g).permute(1, 2, 0)            # [I, G, O]
        bias = None
        if self.use_base_update:
            w = torch.cat([w, self.base_linear.weight.t().unsqueeze(1)], dim=1)   # [I, G+1, O]: base column last
            bias = self.base_linear.bias
        return w.reshape(-1, o), self.rbf.grid.detach(), bias

    @staticmethod
    def kan_pack_grouped(layers):
        l0 = layers[0]
        i, gr, o = l0.input_dim, l0.num_grids, l0.output_dim
        g = len(layers)
        w = grouped.stack_params([m.spline_linear.weight for m in layers]).view(g, o, i, gr).permute(0, 2, 3, 1)   # [g, I, G, O]
        bias = None
        if l0.use_base_update:
            bw = grouped.stack_params([m.base_linear.weight for m in layers]).permute(0, 2, 1).unsqueeze(2)       # [g, I, 1, O]
            w = torch.cat([w, bw], dim=2)
            bias = grouped.stack_params([m.base_linear.bias for m in layers])
        return w.reshape(g, -1, o), torch.stack([m.rbf.grid.detach() for m in layers]), bias

    def kan_u(self, x2d):
        return self.layernorm(x2d) if getattr(self, "_use_ln", True) else None

    def kan_ln(self):
        """The LayerNorm to fuse into the kernels, or None (time_benchmark skips it, models/fastkan.py:67-70)."""
        return self.layernorm if getattr(self, "_use_ln", True) else None

    @staticmethod
    def kan_u_grouped(layers, x2d, n_heads):
        """LayerNorm of every (projection, head) slice: normalise each head slice once, then apply
        the 3*H affine pairs -> u[M, 3*H*dh], column block g = proj*H + head."""
        M = x2d.shape[0]
        dh = layers[0].input_dim
        xhat = F.layer_norm(x2d.view(M, n_heads, dh), (dh,), None, None, layers[0].layernorm.eps)
        gamma = grouped.stack_params([l.layernorm.weight for l in layers]).view(1, 3, n_heads, dh)
        beta = grouped.stack_params([l.layernorm.bias for l in layers]).view(1, 3, n_heads, dh)
        return torch.addcmul(beta, xhat.unsqueeze(1), gamma).reshape(M, 3 * n_heads * dh)

    def forward(self, x, time_benchmark=False):
        self._use_ln = not time_benchmark        # time_benchmark skips the LayerNorm (models/fastkan.py:67-70)
        try:
            y = grouped.run_single(self, x.reshape(-1, self.input_dim))
        finally:
            self._use_ln = True
        return y.reshape(*x.shape[:-1], self.output_dim)
