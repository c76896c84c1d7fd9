"""Diagnostic: phase stamps of one wave of the 16-row-tile attention backward (csrc/attention16.hip) inside a real MSA loop.
Needs a diagnostic build: tools/build_variant.sh clk16 -DKANVIT_CLOCK_PROBE -DB16_CLK_WAVE=<w>; KANVIT_LIB=<that .so> python tools/clock_probe16.py"""
import ctypes
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'kan-vit_amd'))
import torch
from kanvit import _lib
from kanvit import ops
from attention import MSA
torch.manual_seed(0)
m = MSA(768, 12, type='cheby').cuda()
x = torch.randn(128, 197, 768, device='cuda', requires_grad=True)
h = _lib.lib()
h.kanvit_debug_clock16.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
out = (ctypes.c_ulonglong * 16)()
for it in range(10):
    y = m(x)
    y.square().sum().backward()
torch.cuda.synchronize()
h.kanvit_debug_clock16(out)
cyc, ticks = out[0], out[1]
print(f"stamped wave lived {cyc} shader cycles = {ticks / 100:.1f} us -> in-kernel clock {cyc / (ticks * 10):.3f} GHz; 78 steps")
names = ["delta (waves 0-4) / units: partial hand-off", "wait for the slice", "reads + S / dP MFMAs", "wait delta + exp / dS", "dV / dK MFMAs", "dS tile writes (+ dK / dV store)", "issue the fill / units: wait for wave 5", "loop", "units: sum + store dQ, wait for the dS tiles", "units: MFMAs", "wait for the dS slot", "wait for the fill (vmcnt)", "wait for the slice slot", "units: DSR signal + next head's K rows"]
print("    phases (cycles per step): " + ", ".join(f"{n} {out[2 + i] / 78:.0f}" for i, n in enumerate(names)))
