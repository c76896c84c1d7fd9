"""Probe: does a 3-term bf16 split GEMM (fp32 accumulate, fp32 out) through the stock library beat the fp32 GEMM, and how
accurate is it?  Shapes: the ViT-B feed-forward at B=128 (M=25088, 768 <-> 3072)."""
import time, torch
torch.manual_seed(0)
dev = "cuda"
M, K, N = 25088, 768, 3072
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.03

def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

ref64 = (a.double() @ w.double().t())
y32 = a @ w.t()
print("fp32 gemm ms", t(lambda: a @ w.t()), "err", float((y32.double() - ref64).abs().max() / ref64.abs().max()))

def split(x):
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return hi, lo

try:
    ah, al = split(a); wh, wl = split(w)
    y = torch.mm(ah, wh.t(), out_dtype=torch.float32)
    print("out_dtype works", y.dtype)
    def x3():
        ah, al = split(a)
        A = torch.cat([ah, ah, al], 1)
        B = torch.cat([wh, wl, wh], 1)
        return torch.mm(A, B.t(), out_dtype=torch.float32)
    y3 = x3()
    print("bf16x3 concat ms", t(x3), "err", float((y3.double() - ref64).abs().max() / ref64.abs().max()))
    A = torch.cat([ah, ah, al], 1); B = torch.cat([wh, wl, wh], 1)
    print("  gemm only ms", t(lambda: torch.mm(A, B.t(), out_dtype=torch.float32)))
    print("  split only ms", t(lambda: torch.cat([*split(a)[:1], *split(a)], 1)))
    yb = torch.mm(ah, wh.t(), out_dtype=torch.float32)
    print("bf16x1 ms", t(lambda: torch.mm(ah, wh.t(), out_dtype=torch.float32)), "err", float((yb.double() - ref64).abs().max() / ref64.abs().max()))
except Exception as e:
    print("out_dtype path failed:", repr(e))
