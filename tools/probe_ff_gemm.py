"""Probe: the six GEMMs of the ViT-B feed-forward (fwd, dgrad, wgrad for 768->3072 and 3072->768) at M=25088 in fp32,
in every operand layout torch can express, to find which library kernel is the slow one."""
import time, torch
torch.manual_seed(0)
dev = "cuda"
M = 25088

def t(f, n=8):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for (K, N) in [(768, 3072), (3072, 768)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; dy = torch.randn(M, N, device=dev)
    wt = w.t().contiguous(); xt = x.t().contiguous(); dyt = dy.t().contiguous()
    gf = 2 * M * K * N / 1e9
    print(f"--- Linear {K}->{N}  ({gf:.0f} GF each)")
    for name, f in [
        ("fwd  x @ w.t()            ", lambda: x @ w.t()),
        ("fwd  x @ wt (contig)      ", lambda: x @ wt),
        ("fwd  F.linear(x,w,b=None) ", lambda: torch.nn.functional.linear(x, w)),
        ("dgrad dy @ w              ", lambda: dy @ w),
        ("dgrad dy @ wt.t()         ", lambda: dy @ wt.t()),
        ("wgrad dy.t() @ x          ", lambda: dy.t() @ x),
        ("wgrad (x.t() @ dy).t()    ", lambda: (x.t() @ dy)),
        ("wgrad dyt @ x             ", lambda: dyt @ x),
        ("wgrad dyt @ xt.t()        ", lambda: dyt @ xt.t()),
    ]:
        ms = t(f)
        print(f"{name} {ms:7.3f} ms  {gf / ms:7.1f} TF/s")
