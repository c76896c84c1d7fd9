#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the short tables committed under profiles/.

  kernel stats : python tools/summarize_prof.py stats <kernel_stats.csv> [steps] > profiles/rNN_<tag>_kernel_stats.md
  PMC counters : python tools/summarize_prof.py pmc <counter_collection.csv> <COUNTER> > profiles/rNN_<tag>_pmc_<COUNTER>.md
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(Cijk_\w+?_MT\d+x\d+x\d+)", name)
    if m:
        return "hipBLASLt " + m.group(1)
    name = re.sub(r"\(.*", "", name)                     # drop the argument list
    name = re.sub(r"at::native::", "", name)
    return name[:90]


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = short(r["Name"])
        agg[k][0] += int(r["Calls"])
        agg[k][1] += float(r["TotalDurationNs"])
    total = sum(v[1] for v in agg.values())
    print(f"| kernel | calls | total ms | avg us | % | ms/step ({steps} steps) |\n|---|---|---|---|---|---|")
    for k, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"| `{k}` | {calls} | {ns / 1e6:.3f} | {ns / calls / 1e3:.1f} | {100 * ns / total:.2f} | {ns / 1e6 / steps:.3f} |")
    print(f"\ntotal GPU kernel time {total / 1e6:.2f} ms = {total / 1e6 / steps:.2f} ms/step")


def pmc(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    print(f"| kernel | dispatches | {counter} sum | {counter} per dispatch |\n|---|---|---|---|")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"| `{k}` | {n} | {v:.4g} | {v / n:.4g} |")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    else:
        pmc(sys.argv[2], sys.argv[3])
