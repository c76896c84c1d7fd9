"""Probe: split-K (over tokens) weight-gradient GEMM through bmm, fp32 and bf16x3."""
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
    x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev)
    ref = dy.double().t() @ x.double()
    gf = 2 * M * K * N / 1e9
    print(f"--- wgrad [{N}x{K}]")
    for S in (2, 4, 7, 8, 14, 16, 28, 32):
        f = lambda: torch.bmm(dy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)).sum(0)
        ms = t(f); err = float((f().double() - ref).abs().max() / ref.abs().max())
        print(f"S={S:3d} {ms:7.3f} ms {gf/ms:7.1f} TF/s err {err:.2e}")
        f2 = lambda: torch.bmm(x.view(S, M // S, K).transpose(1, 2), dy.view(S, M // S, N)).sum(0)
        ms = t(f2)
        print(f"   T  {ms:7.3f} ms {gf/ms:7.1f} TF/s")
    def split(a):
        hi = a.bfloat16(); lo = (a - hi.float()).bfloat16(); return hi, lo
    def x3():
        xh, xl = split(x); dh, dl = split(dy)
        A = torch.cat([dh, dh, dl], 0)      # [3M, N]
        B = torch.cat([xh, xl, xh], 0)      # [3M, K]
        return torch.mm(A.t(), B, out_dtype=torch.float32)
    ms = t(x3); err = float((x3().double() - ref).abs().max() / ref.abs().max())
    print(f"bf16x3 mm {ms:7.3f} ms err {err:.2e}")
    for S in (8, 16):
        def x3s():
            xh, xl = split(x); dh, dl = split(dy)
            A = torch.cat([dh.view(S, M // S, N), dh.view(S, M // S, N), dl.view(S, M // S, N)], 1)
            B = torch.cat([xh.view(S, M // S, K), xl.view(S, M // S, K), xh.view(S, M // S, K)], 1)
            return torch.bmm(A.transpose(1, 2), B, out_dtype=torch.float32).sum(0)
        try:
            ms = t(x3s); err = float((x3s().double() - ref).abs().max() / ref.abs().max())
            print(f"bf16x3 bmm S={S} {ms:7.3f} ms err {err:.2e}")
        except Exception as e:
            print("bmm out_dtype failed", repr(e)[:200])
