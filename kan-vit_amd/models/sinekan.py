"""SineKANLayer -- drop-in for the reference's models/sinekan.py:7-109 (family SINE)."""
import math

import torch

from kanvit import grouped, ops


def forward_step(i_n, grid_size, A, K, C):
    """One step of the phase recurrence: (A * grid_size^-K + C) * i_n (models/sinekan.py:7-23)."""
    return (A * grid_size ** (-K) + C) * i_n


class SineKANLayer(torch.nn.Module):
    """y[m, o] = sum_i sum_g sin(x_mi * freq_g + phase_ig) * amplitudes[o, i, g] (+ bias).

    Parameters / buffers exactly as the reference: ``amplitudes[O, I, G]``, trainable
    ``freq[1,1,1,G]``, ``bias[1, O]``, fixed buffer ``phase[1,1,I,G]`` built by the recurrence of
    models/sinekan.py:59-75.  The gradient w.r.t. ``freq`` comes from the input-gradient kernel
    (per-row-tile partial sums, reduced in a fixed order)."""

    def __init__(self, input_dim, output_dim, device='cpu', grid_size=5, is_first=False, add_bias=True, norm_freq=True):
        super().__init__()
        self.grid_size = grid_size
        self.device = device
        self.is_first = is_first
        self.add_bias = add_bias
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.A, self.K, self.C = 0.9724108095811765, 0.9884401790754128, 0.999449553483052

        self.grid_norm_factor = (torch.arange(grid_size) + 1).reshape(1, 1, grid_size)
        base = torch.empty(output_dim, input_dim, 1)
        base = base.normal_(0, .4) if is_first else base.uniform_(-1, 1)
        self.amplitudes = torch.nn.Parameter(base / output_dim / self.grid_norm_factor)

        grid_phase = torch.arange(1, grid_size + 1).reshape(1, 1, 1, grid_size) / (grid_size + 1)
        self.input_phase = torch.linspace(0, math.pi, input_dim).reshape(1, 1, input_dim, 1).to(device)
        phase = grid_phase.to(device) + self.input_phase

        freq = torch.arange(1, grid_size + 1).float().reshape(1, 1, 1, grid_size)
        if norm_freq:
            freq = freq / (grid_size + 1) ** (1 - is_first)
        self.freq = torch.nn.Parameter(freq)

        for n in range(1, grid_size):
            phase = forward_step(phase, n, self.A, self.K, self.C)
        self.register_buffer('phase', phase)

        if add_bias:
            self.bias = torch.nn.Parameter(torch.ones(1, output_dim) / output_dim)

    def kan_cfg(self):
        return ops.LayerCfg(family=ops.SINE, I=self.input_dim, O=self.output_dim, G=self.grid_size)

    def kan_pack(self):
        g = self.grid_size
        w = self.amplitudes.permute(1, 2, 0).reshape(self.input_dim * g, self.output_dim)   # row k = i*G + g
        bp = torch.cat([self.freq.reshape(g), self.phase.reshape(self.input_dim * g)])
        return w, bp, (self.bias.reshape(-1) if self.add_bias else None)

    @staticmethod
    def kan_pack_grouped(layers):
        l0 = layers[0]
        g, gs = len(layers), l0.grid_size
        a = grouped.stack_params([m.amplitudes for m in layers])                      # [g, O, I, G]
        w = a.permute(0, 2, 3, 1).reshape(g, l0.input_dim * gs, l0.output_dim)
        bp = torch.cat([grouped.stack_params([m.freq.reshape(gs) for m in layers]),
                        grouped.stack_params([m.phase.reshape(-1) for m in layers])], dim=1)
        bias = grouped.stack_params([m.bias.reshape(-1) for m in layers]) if l0.add_bias else None
        return w, bp, bias

    def forward(self, x):
        y = grouped.run_single(self, x.reshape(-1, self.input_dim))
        return y.reshape(*x.shape[:-1], self.output_dim)

    def forward_step(self, i_n, grid_size, A, K, C):
        return forward_step(i_n, grid_size, A, K, C)
