#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs -> per-launch HBM traffic of the kanvit ops, keyed by the names
bench.py's kernel timer uses (qkv_fwd, layer_bwd_weight, attn_bwd, ... with a _bf16 suffix under --amp bf16).

  python tools/pmc_to_traffic.py <pmc_FETCH_SIZE_counter_collection.csv> <pmc_WRITE_SIZE_counter_collection.csv> [suffix]

An op = every kernel its entry point launches (e.g. attn_bwd = delta + kv + q kernels; bwd_weight = streaming kernel +
slab reduce; bf16 ops include their weight-repack kernel).  Launches of one kernel name are split by grid size: the
q|k|v launches (12 per step) and the patch-embedding launch (1 per step) share kernel names.  Values are KiB per op
launch, raw counter units (see profiles/r01_pmc_traffic.json "note" for the gfx950 correction bench.py applies)."""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])     # (kernel, grid) -> [dispatches, sum]
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*", "", name)
        if not name.startswith(("kan_", "attn_", "attn16_")):
            continue
        k = (name, r.get("Grid_Size", "?"))
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def classify(agg):
    """(kernel, grid) -> op name; q|k|v launches are the ones dispatched >= 6x as often as the rarest kan_ launch."""
    kan_counts = [n for (k, _), (n, _) in agg.items() if k.startswith("kan_") and "slab" not in k and "pack" not in k]
    lo = min(kan_counts) if kan_counts else 1
    ops = defaultdict(lambda: [0.0, 0])     # op -> [sum, launches of the op's main kernel]
    for (k, grid), (n, v) in agg.items():
        grouped = n >= 6 * lo
        pre = "qkv" if grouped else "layer"
        if k.startswith(("attn_fwd", "attn16_fwd")):
            op, main = "attn_fwd", True
        elif k.startswith(("attn_", "attn16_")):         # (round 4: attn16_bwd_kernel is the whole backward)
            op, main = "attn_bwd", k.startswith(("attn_bwd_kv", "attn16_bwd"))
        elif k.startswith(("kan_fwd", "kan_pack_w_fwd")):
            op, main = pre + "_fwd", k.startswith("kan_fwd")
        elif k.startswith(("kan_bwd_input", "kan_pack_w_bwd")):
            op, main = pre + "_bwd_input", k.startswith("kan_bwd_input")
        elif k.startswith(("kan_bwd_weight", "kan_slab_reduce")):
            op, main = pre + "_bwd_weight", k.startswith("kan_bwd_weight")
        else:
            continue
        ops[op][0] += v
        if main:
            ops[op][1] += n
    return {op: v / n for op, (v, n) in ops.items() if n}


if __name__ == "__main__":
    suffix = sys.argv[3] if len(sys.argv) > 3 else ""
    f = classify(load(sys.argv[1], "FETCH_SIZE"))
    w = classify(load(sys.argv[2], "WRITE_SIZE"))
    out = {op + suffix: {"fetch_kib": round(f.get(op, 0.0)), "write_kib": round(w.get(op, 0.0))} for op in sorted(set(f) | set(w))}
    print(json.dumps(out, indent=1))
