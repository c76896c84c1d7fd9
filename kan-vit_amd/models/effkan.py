"""KANLinear (efficient-KAN B-spline layer) -- drop-in for models/effkan.py:8-264 (family BSPLINE).

Forward / backward run in the fused kernel: Cox-de Boor bases on the per-feature knot buffer
``grid`` (half-open order-0 intervals, models/effkan.py:115), the silu base path and both
contractions in one pass.  Construction-time host logic (knot vector, the least-squares
initialisation of ``spline_weight``) is plain torch, as in the reference."""
import math

import torch
import torch.nn.functional as F

from kanvit import _lib, grouped, ops


_GRID_FACTS = {}      # (ids, versions, data_ptrs of the knot buffers) -> (uniform, all_equal)


def _grid_facts(layers):
    """Are the knot buffers uniform (g0 + j*h, the only layout the reference ever builds, models/effkan.py:44-53) and
    identical across `layers`?  Checked once per set of buffers (needs a host sync) and cached on their version
    counters, so an in-place change of any grid is noticed."""
    key = tuple((id(m), m.grid._version, m.grid.data_ptr()) for m in layers)
    hit = _GRID_FACTS.get(key)
    if hit is None:
        g0 = layers[0].grid
        row = g0[0].double()
        step = (row[-1] - row[0]) / (row.numel() - 1)              # the kernel derives h the same way
        ideal = row[0] + step * torch.arange(row.numel(), device=row.device, dtype=torch.float64)
        tol = 2e-6 * (float(row.abs().max()) + 1.0)                # a few float32 ulps of the knot range
        uniform = bool(float(step) > 0 and float((row - ideal).abs().max()) <= tol and bool((g0 == g0[0:1]).all()))
        equal = all(m.grid.shape == g0.shape and bool((m.grid == g0).all()) for m in layers[1:])
        if len(_GRID_FACTS) > 256:
            _GRID_FACTS.clear()
        hit = _GRID_FACTS[key] = (uniform and layers[0].spline_order == 3, equal)
    return hit


class KANLinear(torch.nn.Module):
    def __init__(self, in_features, out_features, grid_size=5, spline_order=3, scale_noise=0.1, scale_base=1.0,
                 scale_spline=1.0, enable_standalone_scale_spline=True, base_activation=torch.nn.SiLU, grid_eps=0.02,
                 grid_range=[-1, 1]):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.grid_size = grid_size
        self.spline_order = spline_order
        if base_activation is not torch.nn.SiLU:
            raise NotImplementedError("the fused kernel implements the reference's default base_activation (SiLU)")

        # uniform knots, spline_order extra on each side (models/effkan.py:44-53)
        h = (grid_range[1] - grid_range[0]) / grid_size
        knots = torch.arange(-spline_order, grid_size + spline_order + 1) * h + grid_range[0]
        self.register_buffer("grid", knots.expand(in_features, -1).contiguous())

        n_basis = grid_size + spline_order
        self.base_weight = torch.nn.Parameter(torch.Tensor(out_features, in_features))
        self.spline_weight = torch.nn.Parameter(torch.Tensor(out_features, in_features, n_basis))
        if enable_standalone_scale_spline:
            self.spline_scaler = torch.nn.Parameter(torch.Tensor(out_features, in_features))

        self.scale_noise = scale_noise
        self.scale_base = scale_base
        self.scale_spline = scale_spline
        self.enable_standalone_scale_spline = enable_standalone_scale_spline
        self.base_activation = base_activation()
        self.grid_eps = grid_eps
        self.reset_parameters()

    # ---- construction-time host logic (models/effkan.py:74-97,134-164) ----
    def reset_parameters(self):
        torch.nn.init.kaiming_uniform_(self.base_weight, a=math.sqrt(5) * self.scale_base)
        with torch.no_grad():
            noise = (torch.rand(self.grid_size + 1, self.in_features, self.out_features) - 0.5) \
                * self.scale_noise / self.grid_size
            scale = 1.0 if self.enable_standalone_scale_spline else self.scale_spline
            pts = self.grid.T[self.spline_order: -self.spline_order]
            self.spline_weight.data.copy_(scale * self.curve2coeff(pts, noise))
            if self.enable_standalone_scale_spline:
                torch.nn.init.kaiming_uniform_(self.spline_scaler, a=math.sqrt(5) * self.scale_spline)

    def b_splines(self, x: torch.Tensor):
        """(batch, in) -> (batch, in, grid_size + spline_order) bases.  Host-side torch version used
        by the initialiser; on a GPU tensor it calls the kernel through an identity contraction."""
        assert x.dim() == 2 and x.size(1) == self.in_features
        if x.is_cuda:
            nb = self.grid_size + self.spline_order
            cfg = ops.LayerCfg(family=ops.BSPLINE, I=1, O=nb, G=nb, groups=self.in_features,
                               x_group_mod=self.in_features, spline_order=self.spline_order, has_base=0)
            eye = torch.eye(nb, device=x.device).expand(self.in_features, nb, nb).contiguous()
            return ops.kan_layer(x, eye, cfg, bparams=self.grid).view(x.size(0), self.in_features, nb)
        g = self.grid
        xe = x.unsqueeze(-1)
        b = ((xe >= g[:, :-1]) & (xe < g[:, 1:])).to(x.dtype)
        for k in range(1, self.spline_order + 1):
            left = (xe - g[:, : -(k + 1)]) / (g[:, k:-1] - g[:, : -(k + 1)])
            right = (g[:, k + 1:] - xe) / (g[:, k + 1:] - g[:, 1:-k])
            b = left * b[:, :, :-1] + right * b[:, :, 1:]
        return b.contiguous()

    def curve2coeff(self, x: torch.Tensor, y: torch.Tensor):
        """Least-squares spline coefficients interpolating y (batch, in, out) at x (batch, in)."""
        assert x.dim() == 2 and x.size(1) == self.in_features
        assert y.size() == (x.size(0), self.in_features, self.out_features)
        a = self.b_splines(x).transpose(0, 1)
        sol = torch.linalg.lstsq(a, y.transpose(0, 1)).solution
        return sol.permute(2, 0, 1).contiguous()

    @property
    def scaled_spline_weight(self):
        if self.enable_standalone_scale_spline:
            return self.spline_weight * self.spline_scaler.unsqueeze(-1)
        return self.spline_weight

    # ---- fused-kernel protocol ----
    def kan_cfg(self, layers=None):
        uniform, equal = _grid_facts(layers if layers is not None else [self])
        flags = (_lib.FLAG_UNIFORM_KNOTS if uniform and equal else 0) | (_lib.FLAG_SHARED_BPARAMS if equal and layers is not None else 0)
        return ops.LayerCfg(family=ops.BSPLINE, I=self.in_features, O=self.out_features,
                            G=self.grid_size + self.spline_order, spline_order=self.spline_order, has_base=1, flags=flags)

    def kan_pack(self):
        # [O, I, nb] (scaled) and base [O, I] -> [I, nb+1, O] -> [I*(nb+1), O]; base column last
        w = torch.cat([self.scaled_spline_weight.permute(1, 2, 0), self.base_weight.t().unsqueeze(1)], dim=1)
        return w.reshape(-1, self.out_features), self.grid.reshape(-1), None

    @staticmethod
    def kan_pack_grouped(layers):
        l0 = layers[0]
        sw = grouped.stack_params([m.spline_weight for m in layers])                  # [g, O, I, nb]
        if l0.enable_standalone_scale_spline:
            sw = sw * grouped.stack_params([m.spline_scaler for m in layers]).unsqueeze(-1)
        bw = grouped.stack_params([m.base_weight for m in layers])                    # [g, O, I]
        w = torch.cat([sw.permute(0, 2, 3, 1), bw.permute(0, 2, 1).unsqueeze(2)], dim=2)   # [g, I, nb+1, O]
        g, i, nb1, o = w.shape
        return w.reshape(g, i * nb1, o), torch.stack([m.grid.reshape(-1) for m in layers]), None

    def forward(self, x: torch.Tensor):
        assert x.size(-1) == self.in_features
        y = grouped.run_single(self, x.reshape(-1, self.in_features))
        return y.reshape(*x.shape[:-1], self.out_features)

    @torch.no_grad()
    def update_grid(self, x: torch.Tensor, margin=0.01):
        raise NotImplementedError("update_grid (models/effkan.py:189-242) has no caller in the reference "
                                  "repository and is outside the accelerated path (SURVEY.md section 2)")

    def regularization_loss(self, regularize_activation=1.0, regularize_entropy=1.0):
        """L1 / entropy surrogate on the spline weights (models/effkan.py:244-264); parameter-only math."""
        l1 = self.spline_weight.abs().mean(-1)
        total = l1.sum()
        p = l1 / total
        return regularize_activation * total - regularize_entropy * torch.sum(p * p.log())
