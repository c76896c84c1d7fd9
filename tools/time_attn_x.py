"""Event-timed general attention kernels (csrc/attention_x.hip) next to the tuned ViT kernels: python tools/time_attn_x.py
Prints ms and algorithmic TFLOP/s (forward 2 products, backward 5) for self-attention at N = 197 (both kernel families) and at lengths only
the chunked kernels take (N = 577: ViT-B/16 at 384 x 384)."""
import os
import sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'kan-vit_amd'))
import torch
from kanvit import ops


def run(fn, n_it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n_it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n_it


for (B, H, N, D, general) in ((128, 12, 197, 64, False), (128, 12, 197, 64, True), (32, 12, 577, 64, True), (8, 12, 1025, 64, True)):
    q, k, v = (torch.randn(B, H, N, D, device="cuda") for _ in range(3))
    o = torch.empty_like(q)
    do = torch.randn_like(q)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    sc = D ** -0.5
    if general:
        f = lambda: ops._attn_x_fwd(q, k, v, o, None, False, sc)
        lse = f()
        b = lambda: ops._attn_x_bwd(q, k, v, o, lse, do, dq, dk, dv, None, False, sc)
    else:
        f = lambda: ops._attn_fwd(q, k, v, o, False, sc)
        lse = f()
        b = lambda: ops._attn_bwd(q, k, v, o, lse, do, dq, dk, dv, False, sc)
    tf, tb = run(f), run(b)
    fl = 4.0 * B * H * N * N * D
    print(f"{'general' if general else 'ViT    '} B={B} H={H} N={N} D={D}: fwd {tf:.3f} ms ({fl / tf / 1e9:.1f} TF/s)  bwd {tb:.3f} ms ({2.5 * fl / tb / 1e9:.1f} TF/s)")
