"""Run one kanvit op in a loop (for rocprofv3 --pmc passes on a single kernel).
  python tools/bench_op.py fwd|bwd [amp] [iters]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kan-vit_amd'))
import torch
from attention import MSA
what = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
amp = len(sys.argv) > 2 and sys.argv[2] == 'amp'
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
m = MSA(768, 12, type='cheby').cuda()
x = torch.randn(128, 197, 768, device='cuda', requires_grad=True)
for _ in range(iters):
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        y = m(x)
    if what == 'bwd':
        y.float().square().sum().backward()
torch.cuda.synchronize()
print('done')
