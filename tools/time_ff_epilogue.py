import sys, os, time
sys.path.insert(0, 'kan-vit_amd')
import torch
from kanvit import dense
M, N = 25216, 3072
dy = torch.randn(M, N, device='cuda'); y = torch.relu(torch.randn(M, N, device='cuda'))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
def stock():
    d = torch.ops.aten.threshold_backward(dy, y, 0); return d, d.sum(0)
print("stock threshold_backward + sum(0): %.1f us" % t(stock))
print("fused kanvit_relu_bwd_bias:        %.1f us" % t(lambda: dense._relu_bwd_bias(dy, y)))
print("threshold only: %.1f us, sum only: %.1f us" % (t(lambda: torch.ops.aten.threshold_backward(dy, y, 0)), t(lambda: dy.sum(0))))
