"""ChebyKANLayer -- drop-in for the reference's models/cheby.py:10-48, computed by the fused
gfx950 kernel (kan-vit_amd/csrc/kan_layer.hip, family CHEBY)."""
import torch
import torch.nn as nn

from kanvit import grouped, ops


class ChebyKANLayer(nn.Module):
    """y[b, o] = sum_i sum_{d=0..degree} T_d(tanh x[b, i]) * cheby_coeffs[i, o, d].

    Same constructor, parameter / buffer names and shapes as the reference (models/cheby.py:11-34):
    ``cheby_coeffs[I, O, degree+1]`` ~ N(0, 1/(I*(degree+1))), integer buffer ``arange``.
    Like the reference (models/cheby.py:38,47) the output is ALWAYS 2-D ``(prod(leading), O)``;
    callers that need the leading dims back reshape (model.py does)."""

    def __init__(self, input_dim, output_dim, degree):
        super().__init__()
        self.inputdim = input_dim
        self.outdim = output_dim
        self.degree = degree
        coeffs = torch.empty(input_dim, output_dim, degree + 1)
        nn.init.normal_(coeffs, mean=0.0, std=1.0 / (input_dim * (degree + 1)))
        self.cheby_coeffs = nn.Parameter(coeffs)
        self.register_buffer("arange", torch.arange(0, degree + 1, 1))

    # ---- fused-kernel protocol (kanvit/grouped.py) ----
    def kan_cfg(self):
        return ops.LayerCfg(family=ops.CHEBY, I=self.inputdim, O=self.outdim, G=self.degree + 1)

    def kan_pack(self):
        # [I, O, D+1] -> [I, D+1, O] -> [I*(D+1), O]: row k = i*(D+1) + d
        w = self.cheby_coeffs.permute(0, 2, 1).reshape(self.inputdim * (self.degree + 1), self.outdim)
        return w, None, None

    @staticmethod
    def kan_pack_grouped(layers):
        c = grouped.stack_params([m.cheby_coeffs for m in layers], perm=(0, 2, 1))    # [g, I, D+1, O], written in the packed layout
        g, i, d1, o = c.shape
        return c.view(g, i * d1, o), None, None

    def forward(self, x):
        return grouped.run_single(self, x.reshape(-1, self.inputdim))
