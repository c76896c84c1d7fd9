"""Time the launches of one MSA block (q|k|v forward, attention, their backwards) with HIP events on the launch stream.
    python tools/time_op.py [amp] [type] [reps]       (environment switches, e.g. KANVIT_NO_PIPE=1, select fallback kernels)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kan-vit_amd'))
import torch
from kanvit import _lib, ops
from attention import MSA
amp = 'amp' in sys.argv[1:]
types = [a for a in sys.argv[1:] if a in ('cheby', 'vanilla', 'fast', 'efficientkan', 'sine')] or ['cheby']
for t in types:
    torch.manual_seed(0)
    m = MSA(768, 12, type=t).cuda()
    x = torch.randn(128, 197, 768, device='cuda', requires_grad=True)
    for it in range(11):
        if it == 3:
            torch.cuda.synchronize()
            ops.timer = ops.KernelTimer()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            y = m(x)
        y.float().square().sum().backward()
    torch.cuda.synchronize()
    for k, v in ops.timer.summary().items():
        print(f"{t:12s} {k:24s} {v['avg_ms']*1e3:8.1f} us   {v['flops']/v['avg_ms']/1e9:7.1f} TF/s  {v['bytes']/v['avg_ms']/1e6:7.0f} GB/s   [{_lib.active_config()}]")
    ops.timer = None
