"""Timing-only ablation of the fused forward kernel (KANVIT_DBG mask; outputs are wrong when set)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
import torch
from attention import MSA
from kanvit import grouped

torch.manual_seed(0)
fam = sys.argv[1] if len(sys.argv) > 1 else "cheby"
msa = MSA(768, 12, type=fam).cuda()
x = torch.randn(128 * 197, 768, device="cuda")
names = {0: "full (producer prio)", 16: "full, no prio", 7: "-all producers", 8: "-mfma", 24: "-mfma, no prio", 15: "empty loop"}
with torch.no_grad():
    for rnd in range(2):
        for mask, nm in names.items():
            os.environ["KANVIT_DBG"] = str(mask)
            for _ in range(2):
                grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)
            e.record()
            torch.cuda.synchronize()
            print(f"round {rnd} {fam} mask {mask:2d} {nm:16s} {s.elapsed_time(e) / 5:.3f} ms (includes ~0.1 ms of packing ops)")
