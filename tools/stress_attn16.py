"""Stress of the 16-row-tile attention kernels (csrc/attention16.hip): many launches over head counts that give the persistent
work-groups 1 .. 40 heads each, every N of the one-kernel backward's domain, fp32 and bf16 mode -- every launch must finish (the
backward's hand-offs are polled LDS counters: a protocol error would hang, so run this under `timeout`) and repeat bitwise.
    timeout -k 10 600 python tools/stress_attn16.py [iterations]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kan-vit_amd'))
import torch
from kanvit import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)
cases = [(1, 1), (1, 7), (3, 5), (2, 12), (16, 12), (128, 12), (256, 6), (77, 3), (512, 20)]
ns = [193, 197, 200, 204]
first = {}
t0 = time.time()
for it in range(iters):
    b, h = cases[it % len(cases)]
    n = ns[(it // len(cases)) % len(ns)]
    amp = (it // (len(cases) * len(ns))) % 2 == 1
    g = torch.Generator(device='cuda').manual_seed(1000 * n + 10 * b + h)
    q, k, v, do = (torch.randn(b, h, n, 64, device='cuda', generator=g) for _ in range(4))
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        o = ops.attention(q, k, v)
    o.backward(do)
    res = [t.detach().clone() for t in (o, q.grad, k.grad, v.grad)]
    assert all(torch.isfinite(t).all() for t in res), (b, h, n, amp)
    key = (b, h, n, amp)
    if key in first:
        assert all(torch.equal(a, c) for a, c in zip(first[key], res)), ("not bitwise reproducible", key)
    elif b * h <= 64:
        first[key] = res
    if it % 20 == 0:
        torch.cuda.synchronize()
        print(f"iter {it}: B={b} H={h} N={n} amp={amp} ok ({time.time() - t0:.1f} s)", flush=True)
torch.cuda.synchronize()
print(f"STRESS OK: {iters} launches of each direction, {len(first)} configurations compared bitwise across repeats")
