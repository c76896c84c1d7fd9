"""NaiveFourierKANLayer -- drop-in for the reference's models/nfkan.py:5-52 (family FOURIER).

The reference materialises an (M, O, I, G) product (models/nfkan.py:47-48, 12.9 GB per image at
ViT-B patch-embedding shapes); the fused kernel generates cos(kx) / sin(kx) tiles in LDS and
contracts them on the matrix cores, so nothing of that size ever exists."""
import numpy as np
import torch

from kanvit import grouped, ops


class NaiveFourierKANLayer(torch.nn.Module):
    """y[m, o] = sum_i sum_{k=1..G} cos(k x_mi) F[0,o,i,k-1] + sin(k x_mi) F[1,o,i,k-1] (+ bias).

    Signature as the reference (models/nfkan.py:6, including the ``smootorch_initialization``
    spelling).  ``grid_size=`` is accepted as an alias of ``gridsize`` because the reference's own
    model.py:74 calls the layer that way (and crashes -- SURVEY.md D4)."""

    def __init__(self, inputdim, outdim, gridsize=None, addbias=True, smootorch_initialization=False, grid_size=None):
        super().__init__()
        if gridsize is None:
            gridsize = grid_size
        if gridsize is None:
            raise TypeError("NaiveFourierKANLayer needs gridsize")
        self.gridsize = gridsize
        self.addbias = addbias
        self.inputdim = inputdim
        self.outdim = outdim
        norm = (torch.arange(gridsize) + 1) ** 2 if smootorch_initialization else np.sqrt(gridsize)
        self.fouriercoeffs = torch.nn.Parameter(torch.randn(2, outdim, inputdim, gridsize) / (np.sqrt(inputdim) * norm))
        if addbias:
            self.bias = torch.nn.Parameter(torch.zeros(1, outdim))

    def kan_cfg(self):
        return ops.LayerCfg(family=ops.FOURIER, I=self.inputdim, O=self.outdim, G=self.gridsize)

    def kan_pack(self):
        # [2, O, I, G] -> [I, 2, G, O] -> [I*2G, O]: row k = i*2G + c*G + (freq-1)
        w = self.fouriercoeffs.permute(2, 0, 3, 1).reshape(self.inputdim * 2 * self.gridsize, self.outdim)
        return w, None, (self.bias.reshape(-1) if self.addbias else None)

    def forward(self, x):
        y = grouped.run_single(self, x.reshape(-1, self.inputdim))
        return y.reshape(*x.shape[:-1], self.outdim)
