"""Diagnostic: the shader clock the chip holds DURING attn_fwd3_kernel inside a real MSA forward/backward loop.
Needs the diagnostic build of the library (-DKANVIT_CLOCK_PROBE, see tools/README.md) at tools/_diag/libkanvit_clk.so."""
import ctypes
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'kan-vit_amd'))
import torch
from kanvit import _lib
_lib.LIB_PATH = os.environ.get('KANVIT_LIB') or os.path.join(HERE, '_diag', 'libkanvit_clk.so')
fwd_only = 'fwd' in sys.argv[1:]        # stamps of the forward kernel (the backward kernels overwrite them otherwise)
from kanvit import ops
from attention import MSA
torch.manual_seed(0)
m = MSA(768, 12, type='cheby').cuda()
x = torch.randn(128, 197, 768, device='cuda', requires_grad=True)
h = _lib.lib()
h.kanvit_debug_clock.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
out = (ctypes.c_ulonglong * 10)()
for it in range(12):
    y = m(x)
    if not fwd_only:
        y.square().sum().backward()
    if it >= 8:
        torch.cuda.synchronize()
        h.kanvit_debug_clock(out)
        cyc, ticks = out[0], out[1]
        print(f"iter {it}: stamped work-group lived {cyc} shader cycles = {ticks / 100:.1f} us -> in-kernel clock {cyc / (ticks * 10):.3f} GHz")
        names = (["wait+barrier (S)", "issue V fill", "S MFMAs", "softmax", "wait+barrier (PV)", "issue K fill + Q loads", "PV MFMAs", "store"] if fwd_only else
                 ["head boundary (wait + barrier)", "-", "S and dP MFMAs", "exp / dS", "dS stores", "dV and dK MFMAs", "tile barrier", "dK / dV store"])
        print("    wave 0 phases (cycles): " + ", ".join(f"{n} {out[2 + i]}" for i, n in enumerate(names)))
