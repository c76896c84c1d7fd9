#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/round_profiles.sh <tag> 1|2|3) -> the committed profiles/<tag>_* files.
    python tools/assemble_profiles.py r04 "<one line saying which kernels these are>"
bench lines and rocprofv3 kernel-stat tables are copied under the names DESIGN.md cites; the FETCH_SIZE / WRITE_SIZE passes of the
headline workload (fp32 and bf16 autocast) are merged into <tag>_pmc_traffic.json, the file bench.py's `roofline.traffic` reads; the SQ
counter passes go through tools/pmc_summary.py."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
have = set(os.listdir(src))
for f in sorted(have):
    if f.startswith("bench_") and f.endswith(".json"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
    elif f.endswith("_kernel_stats.md") or f.endswith("_pmc_FETCH_SIZE.md") or f.endswith("_pmc_WRITE_SIZE.md"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
    elif f.startswith("sq_pmc_") and f.endswith(".txt"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f[:-4]}_raw.txt"))
        md = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(src, f)], capture_output=True, text=True).stdout
        open(os.path.join(dst, f"{tag}_{f[:-4]}.md"), "w").write(md)
kern = {}
for f in ("vitb16_cheby_traffic.json", "vitb16_cheby_amp_bf16_traffic.json"):
    if f in have:
        kern.update(json.load(open(os.path.join(src, f))))
if kern:
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) through tools/round_profiles.sh {tag} 1 -> tools/profile_box.sh + "
                     f"tools/pmc_to_traffic.py; {note}",
           "workload": "vitb16-224-cheby", "per_gpu_batch": 128,
           "note": "raw counter values per launch in KiB; bench.py reports traffic = (2*fetch_kib + write_kib)*1024 bytes (gfx950: FETCH_SIZE tallies 128-B "
                   "requests at 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM section); an op = every kernel its entry point launches "
                   "(bwd_weight = streaming kernel + slab reduce; bf16 ops include the weight repack)",
           "kernels": kern}
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(sorted(f for f in os.listdir(dst) if f.startswith(tag + "_")))
