"""Attribute the small kernels of one training step to aten ops (torch.profiler), to find launch-bound leftovers."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kan-vit_amd'))
import torch
from model import VisionTransformer
amp = len(sys.argv) > 1 and sys.argv[1] == 'amp'
torch.manual_seed(0)
m = VisionTransformer((3, 224, 224), n_patches=14, n_blocks=12, d_hidden=768, n_heads=12, out_d=1000, type='cheby').cuda()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
x = torch.randn(128, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (128,), device='cuda')
def step():
    opt.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        loss = torch.nn.functional.cross_entropy(m(x), y)
    loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.count >= 12 and (e.key.startswith("aten::") or "Backward" in e.key or "Fn" in e.key)]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:70]:
    print(f"{e.key[:44]:44s} n={e.count:5d} cuda_total={e.device_time_total/1e3:8.3f} ms  self={e.self_device_time_total/1e3:8.3f} ms  shapes={str(e.input_shapes)[:90]}")
