import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
import torch
from attention import MSA
from kanvit import grouped, ops
torch.manual_seed(0)
msa = MSA(768, 12, type="cheby").cuda()
x = torch.randn(128 * 197, 768, device="cuda")
names = {0: "full", 32: "-epilogue", 8: "-mfma loop", 40: "-mfma -epilogue", 2: "-basis", 5: "-loads", 7: "-producers", 47: "nothing"}
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for rnd in range(2):
        for mask, nm in names.items():
            os.environ["KANVIT_DBG"] = str(mask)
            for _ in range(2):
                grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)
            ops.timer = ops.KernelTimer()
            for _ in range(5):
                grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)
            r = ops.timer.summary(); ops.timer = None
            if rnd: print(f"mask {mask:2d} {nm:18s} {r['qkv_fwd_bf16']['avg_ms']:.3f} ms")
