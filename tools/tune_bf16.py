"""Try chunk / sharing variants of the bf16 forward kernel (env knobs read per launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
import torch
from attention import MSA
from kanvit import grouped, ops
torch.manual_seed(0)
fam = sys.argv[1] if len(sys.argv) > 1 else "cheby"
msa = MSA(768, 12, type=fam).cuda()
x = torch.randn(128 * 197, 768, device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for nsh in ("3", "1"):
        for ic in ("16", "8"):
            os.environ["KANVIT_BF16_NSH"], os.environ["KANVIT_BF16_IC"] = nsh, ic
            for _ in range(2):
                grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)
            ops.timer = ops.KernelTimer()
            for _ in range(5):
                grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)
            r = ops.timer.summary(); ops.timer = None
            print(f"{fam} nsh={nsh} ic={ic}: {r['qkv_fwd_bf16']['avg_ms']:.3f} ms  ({313e6 / r['qkv_fwd_bf16']['avg_ms'] / 1e6:.0f} GB/s algorithmic)")
