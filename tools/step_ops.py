"""Which torch ops launch the small kernels of one training step?  (torch.profiler, eager step of a bench.py workload)
    python tools/step_ops.py [workload] [amp]
Prints every aten / autograd-Function op that owns device time, with its launch count per step and the kernels under it."""
import os
import sys
from collections import defaultdict
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'kan-vit_amd'))
sys.path.insert(0, R)
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from model import VisionTransformer
name = next((a for a in sys.argv[1:] if a in bench.WORKLOADS), 'mnist-cheby-tiny')
amp = 'amp' in sys.argv[1:]
wl = bench.WORKLOADS[name]
torch.manual_seed(0)
m = VisionTransformer(wl["chw"], wl["n_patches"], wl["n_blocks"], wl["d"], wl["heads"], wl["out_d"], type=wl["type"]).cuda()
opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True, capturable=True)
x = torch.randn(wl["batch"], *wl["chw"], device='cuda')
y = torch.randint(0, wl["out_d"], (wl["batch"],), device='cuda')
crit = torch.nn.CrossEntropyLoss()


def step():
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        loss = crit(m(x), y)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
kernels = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print(f"{name}: {len(kernels)} device launches in one step, {sum(e.device_time for e in kernels) / 1e3:.3f} ms of device time")
rows = defaultdict(lambda: [0, 0.0, defaultdict(int)])
cpu_ops = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
for e in cpu_ops:
    ks = [k for k in e.kernels] if hasattr(e, "kernels") else []
    if not ks:
        continue
    r = rows[e.name]
    r[0] += len(ks)
    r[1] += sum(k.duration for k in ks)
    for k in ks:
        r[2][k.name[:60]] += 1
for k, (n, us, names) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:46]:46s} launches {n:4d}  {us:8.1f} us   " + "; ".join(f"{a} x{b}" for a, b in sorted(names.items(), key=lambda t: -t[1])[:3]))

print("\n-- aten::copy_ / aten::cat / aten::sum / aten::fill_ launches with shapes and the innermost python frame --")
for e in cpu_ops:
    if e.name in ("aten::copy_", "aten::cat", "aten::sum", "aten::fill_", "aten::add", "aten::mul") and getattr(e, "kernels", None):
        frames = [f for f in (e.stack or []) if "kan-vit_amd" in f or "bench.py" in f]
        print(f"{e.name:12s} {str(e.input_shapes)[:70]:70s} {frames[0][-90:] if frames else ''}")
