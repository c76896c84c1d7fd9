"""Probe: fused bias+ReLU GEMM epilogue (torch._addmm_activation) and fused Adam on this ROCm build."""
import time, torch
dev = "cuda"
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
M, K, N = 25216, 768, 3072
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
ref = torch.relu(torch.nn.functional.linear(x, w, b))
try:
    y = torch._addmm_activation(b, x, w.t(), use_gelu=False)
    print("addmm_activation max diff", float((y - ref).abs().max()))
    print("linear+relu_ ms", t(lambda: torch.relu_(torch.nn.functional.linear(x, w, b))), " addmm_activation ms", t(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)))
    xb, wb, bb = x.bfloat16(), w.bfloat16(), b.bfloat16()
    print("bf16: linear+relu_ ms", t(lambda: torch.relu_(torch.nn.functional.linear(xb, wb, bb))), " addmm_activation ms", t(lambda: torch._addmm_activation(bb, xb, wb.t(), use_gelu=False)))
except Exception as e:
    print("addmm_activation failed:", repr(e)[:300])
ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in [64 * 64 * 5] * 432 + [768 * 3072] * 24 + [3840 * 768]]
for p in ps: p.grad = torch.randn_like(p)
for kw in ({}, {"fused": True}):
    try:
        opt = torch.optim.Adam(ps, lr=1e-3, **kw)
        print("Adam", kw, "ms/step", t(opt.step))
    except Exception as e:
        print("Adam", kw, "failed:", repr(e)[:200])
