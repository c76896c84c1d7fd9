"""Probe: can the fp32 forward / input-gradient GEMMs of the feed-forward go faster than the default library pick?"""
import time, torch
dev = "cuda"
M = 25216
def t(f, n=8):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for lib in ("default", "hipblaslt", "hipblas"):
    try:
        if lib != "default": torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "n/a", repr(e)[:80]); continue
    for (K, N) in [(768, 3072), (3072, 768)]:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev)
        gf = 2 * M * K * N / 1e9
        rows = [("linear", lambda: torch.nn.functional.linear(x, w, b)), ("dgrad dy@w", lambda: dy @ w)]
        for S in (2, 4, 8):
            rows.append((f"fwd bmm S={S}", lambda S=S: torch.matmul(x.view(S, M // S, K), w.t())))
            rows.append((f"dgrad bmm S={S}", lambda S=S: torch.matmul(dy.view(S, M // S, N), w)))
        rows.append(("fwd (w @ x^T)^T", lambda: (w @ x.t())))
        for name, f in rows:
            ms = t(f)
            print(f"{lib:9s} {K}->{N} {name:18s} {ms:6.3f} ms {gf/ms:6.1f} TF/s")
