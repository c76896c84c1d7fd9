#!/usr/bin/env python3
"""Register / LDS / occupancy table of the kernels in a hipcc `-Rpass-analysis=kernel-resource-usage` log (stderr).
    hipcc ... -c kan_layer.hip -Rpass-analysis=kernel-resource-usage 2> usage.txt;  python tools/kernel_usage.py usage.txt [regex]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
rows, cur = [], {}
for b in txt.split("remark: "):
    m = re.search(r"Function Name: (\S+)", b)
    if m:
        if cur:
            rows.append(cur)
        cur = {"name": m.group(1)}
    for key, short in (("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("SGPRs", "sgpr"), (r"ScratchSize \[bytes/lane\]", "scratch"),
                       (r"Occupancy \[waves/SIMD\]", "occ"), (r"LDS Size \[bytes/block\]", "lds")):
        m = re.search(key + r": (\d+)", b)
        if m:
            cur[short] = int(m.group(1))
if cur:
    rows.append(cur)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, names):
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\(.*", "", d)
    if pat is None or pat.search(d):
        print(f"{d[:64]:64s} vgpr {r.get('vgpr', 0):4d} agpr {r.get('agpr', 0):4d} sgpr {r.get('sgpr', 0):4d} scratch {r.get('scratch', 0):5d} "
              f"occ {r.get('occ', 0)} lds {r.get('lds', 0)}")
