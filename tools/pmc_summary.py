#!/usr/bin/env python3
"""tools/pmc_kernel.sh output (per-dispatch averages of SQ counters) -> a markdown table with the derived figures DESIGN.md quotes.
    python tools/pmc_summary.py gpurun_out/r02/sq_pmc_fp32.txt > profiles/r02_sq_pmc_fp32.md
Derivations (gfx950, 8 XCDs x 32 CUs x 4 SIMDs):
  kernel cycles      = GRBM_GUI_ACTIVE / 8                      (the counter is summed over the 8 XCDs)
  MFMA pipe busy     = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs)
                       (the counter is SIMD-cycles: it equals SQ_INSTS_MFMA * passes*4, e.g. x64 for v_mfma_f32_32x32x2_f32)
  VALU per MFMA      = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA     (SQ_INSTS_VALU includes the MFMAs)
  LDS conflict rate  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wave occupancy     = 4 * SQ_WAVE_CYCLES / (kernel cycles * 1024)          (resident waves per SIMD, time-averaged; 4-cycle ticks)"""
import re
import sys

cur, data = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"\s+(\w+)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
    if m and cur:
        data[cur][m.group(1)] = float(m.group(2))
        data[cur]["_n"] = int(m.group(3))
    elif line.strip() and not line.startswith(" "):
        cur = re.sub(r"^void ", "", line.strip())
        data[cur] = {}
print("| kernel | dispatches | kernel cycles | MFMA pipe busy | VALU per MFMA | LDS instr per MFMA | LDS bank-conflict rate | waves/SIMD (avg) | VMEM rd / wr instr |")
print("|---|---|---|---|---|---|---|---|---|")
for k, c in data.items():
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    mf = c.get("SQ_INSTS_MFMA", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0) if cyc else 0.0
    valu = (c.get("SQ_INSTS_VALU", 0.0) - mf) / mf if mf else float("nan")
    lds = c.get("SQ_INSTS_LDS", 0.0) / mf if mf else float("nan")
    conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0.0
    occ = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / (cyc * 1024.0) if cyc else 0.0     # the counter ticks once per 4 cycles
    print(f"| `{k}` | {c['_n']} | {cyc:,.0f} | {100 * busy:.1f} % | {valu:.2f} | {lds:.2f} | {100 * conf:.1f} % | {occ:.2f} | "
          f"{c.get('SQ_INSTS_VMEM_RD', 0):,.0f} / {c.get('SQ_INSTS_VMEM_WR', 0):,.0f} |")
