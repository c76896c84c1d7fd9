"""Which python-level ops launch the D2D copy kernels in one train step? (torch profiler, GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from model import VisionTransformer

torch.manual_seed(0)
m = VisionTransformer((3, 224, 224), 14, 2, 768, 12, 100, type="cheby").cuda()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
x = torch.randn(64, 3, 224, 224, device="cuda"); y = torch.randint(0, 100, (64,), device="cuda")
def step():
    loss = torch.nn.functional.cross_entropy(m(x), y); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.device_time_total > 0 and ("copy" in e.key.lower() or "contiguous" in e.key.lower() or "clone" in e.key.lower() or "cat" in e.key.lower() or "stack" in e.key.lower()):
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
for r in sorted(rows, reverse=True)[:18]:
    print(f"{r[0]/1e3:9.3f} ms  n={r[1]:4d}  {r[2]:28s} {r[3]}")
