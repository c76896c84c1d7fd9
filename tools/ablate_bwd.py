"""Timing-only ablation of the fused backward kernels (KANVIT_DBG mask; results are wrong when set)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
import torch
from attention import MSA
from kanvit import grouped, ops

torch.manual_seed(0)
fam = sys.argv[1] if len(sys.argv) > 1 else "cheby"
msa = MSA(768, 12, type=fam).cuda()
x = torch.randn(128 * 197, 768, device="cuda", requires_grad=True)
dy = torch.randn(128 * 197, 2304, device="cuda")
names = {0: "full", 11: "empty", 27: "empty -x/dx io", 91: "empty -io -dA park", 32: "return at entry"}
for rnd in range(2):
    for mask, nm in names.items():
        os.environ["KANVIT_DBG"] = "0"
        y = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)
        os.environ["KANVIT_DBG"] = str(mask)
        ops.timer = ops.KernelTimer()
        for _ in range(4):
            y.backward(dy, retain_graph=True)
        res = ops.timer.summary()
        ops.timer = None
        print(f"round {rnd} {fam} mask {mask:2d} {nm:16s} bwd_input {res['qkv_bwd_input']['avg_ms']:.3f} ms   bwd_weight {res['qkv_bwd_weight']['avg_ms']:.3f} ms")
