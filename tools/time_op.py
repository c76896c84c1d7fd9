"""Time the launches of one MSA block (q|k|v forward, attention, their backwards) with HIP events on the launch stream.
    python tools/time_op.py [amp] [type] [b=<batch>[,<batch>...]]      (environment switches, e.g. KANVIT_NO_PIPE=1, select fallback kernels)
b=110,128,138 probes the launch tail: 12 heads x ceil(197 b / 128) row tiles on 512 resident work-groups = 3.98 / 4.62 / 4.99 rounds."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kan-vit_amd'))
import torch
from kanvit import _lib, ops
from attention import MSA
amp = 'amp' in sys.argv[1:]
types = [a for a in sys.argv[1:] if a in ('cheby', 'vanilla', 'fast', 'efficientkan', 'sine')] or ['cheby']
batches = [int(v) for a in sys.argv[1:] if a.startswith('b=') for v in a[2:].split(',')] or [128]
for t, bsz in [(t, b) for t in types for b in batches]:
    torch.manual_seed(0)
    m = MSA(768, 12, type=t).cuda()
    x = torch.randn(bsz, 197, 768, device='cuda', requires_grad=True)
    for it in range(11):
        if it == 3:
            torch.cuda.synchronize()
            ops.timer = ops.KernelTimer()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            y = m(x)
        y.float().square().sum().backward()
    torch.cuda.synchronize()
    for k, v in ops.timer.summary().items():
        print(f"{t:12s} b={bsz:<4d} {k:24s} {v['avg_ms']*1e3:8.1f} us   {v['flops']/v['avg_ms']/1e9:7.1f} TF/s  {v['bytes']/v['avg_ms']/1e6:7.0f} GB/s")
    ops.timer = None
print(_lib.active_config())
