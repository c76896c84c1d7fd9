"""Time the q|k|v forward (and optionally backward) launches of one MSA block with HIP events, per KANVIT_DBG setting."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kan-vit_amd'))
import torch
from kanvit import ops
from attention import MSA
amp = len(sys.argv) > 1 and sys.argv[1] == 'amp'
torch.manual_seed(0)
m = MSA(768, 12, type='cheby').cuda()
x = torch.randn(128, 197, 768, device='cuda', requires_grad=True)
for it in range(11):
    if it == 3:
        torch.cuda.synchronize(); ops.timer = ops.KernelTimer()
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        y = m(x)
    y.float().square().sum().backward()
torch.cuda.synchronize()
for k, v in ops.timer.summary().items():
    print(f"{os.environ.get('KANVIT_DBG','0'):>3s} {k:24s} {v['avg_ms']*1e3:8.1f} us")
