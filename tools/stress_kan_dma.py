"""Stress of the two round-4 KAN backward forms whose waits on LDS-DMA fills are explicit s_waitcnt counts (csrc/kan_bwd_weight_dma.hip,
kan_bwd_input_res_bf16_kernel): an accounting error there is silent wrong data, not a hang -- so: the ViT-B q|k|v launch repeated (bitwise
equal to its first run, fp32 and bf16 mode), then random row counts forced through the LDS-DMA weight gradient, each against the register
ring (KANVIT_BW_NO_DMA) and the streaming input gradient (KANVIT_BI_NO_RES) of the same launch.
    timeout -k 10 600 python tools/stress_kan_dma.py [repeats] [random shapes]"""
import os
import random
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kan-vit_amd'))
import torch
from kanvit import _lib, grouped
from attention import MSA

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nrand = int(sys.argv[2]) if len(sys.argv) > 2 else 40


def grads(msa, x, w, amp):
    msa.zero_grad()
    x.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        y = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)
    (y * w).sum().backward()
    return torch.cat([x.grad.flatten()] + [p.grad.flatten() for p in msa.parameters() if p.grad is not None]).clone()


def with_env(env, fn):
    for k, v in env.items():
        os.environ[k] = v
    _lib.reload_config()
    try:
        return fn()
    finally:
        for k in env:
            del os.environ[k]
        _lib.reload_config()


torch.manual_seed(0)
for fam in ("cheby", "efficientkan", "fast"):
    msa = MSA(768, 12, type=fam).cuda()
    x = torch.randn(128 * 197, 768, device='cuda', requires_grad=True)
    w = torch.randn(128 * 197, 2304, device='cuda')
    for amp in (False, True):
        first = grads(msa, x, w, amp)
        assert torch.isfinite(first).all()
        for _ in range(reps):
            assert torch.equal(first, grads(msa, x, w, amp)), ("not bitwise reproducible", fam, amp)
        old = with_env({"KANVIT_BW_NO_DMA": "1", "KANVIT_BI_NO_RES": "1"}, lambda: grads(msa, x, w, amp))
        err = float((first - old).abs().max()) / float(old.abs().max())
        assert err < 1e-5, (fam, amp, err)
        print(f"{fam:12s} amp={amp}: {reps} repeats bitwise, max rel. difference to the round-3 forms {err:.2e}", flush=True)
    del msa, x, w

rng = random.Random(1)
msa = MSA(256, 4, type="cheby").cuda()
for i in range(nrand):
    rows = rng.choice([256, 257, 300, 511, 793, 1024, 1500, 2049, 4097, rng.randrange(256, 6000)])
    amp = bool(i & 1)
    x = torch.randn(rows, 256, device='cuda', requires_grad=True)
    w = torch.randn(rows, 768, device='cuda')
    new = with_env({"KANVIT_BW_DMA_FORCE": "1"}, lambda: [grads(msa, x, w, amp) for _ in range(2)])
    old = with_env({"KANVIT_BW_NO_DMA": "1", "KANVIT_BI_NO_RES": "1"}, lambda: grads(msa, x, w, amp))
    assert torch.equal(new[0], new[1]), ("not bitwise reproducible", rows, amp)
    err = float((new[0] - old).abs().max()) / float(old.abs().max())
    assert err < 1e-5, (rows, amp, err)
print(f"STRESS OK: {nrand} random row counts through the forced LDS-DMA weight gradient")
