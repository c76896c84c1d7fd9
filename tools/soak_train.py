"""Soak: the drop-in train.py on a synthetic stream at its default (C) geometry and at the MNIST-tiny (T) geometry, eager and
--graph, a few hundred steps each, twice: losses must stay finite, go down, and be bitwise identical between the two runs.
    python tools/soak_train.py [steps]"""
import os
import sys
import tempfile
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'kan-vit_amd'))
import torch
import train

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
GEOS = {"C-cheby": ["--model-type", "cheby"],
        "T-cheby": ["--model-type", "cheby", "--image-size", "28", "--in-chans", "1", "--n-patches", "7", "--n-blocks", "4", "--n-heads", "2",
                    "--out-d", "10"],
        "C-vanilla": ["--model-type", "vanilla"], "C-efficientkan": ["--model-type", "efficientkan"]}
ok = True
for name, geo in GEOS.items():
    for graph in (False, True):
        runs = []
        for rep in range(2):
            with tempfile.TemporaryDirectory() as tmp:
                argv = ["--synthetic", "--epochs", "1", "--steps-per-epoch", str(steps), "--log-dir", tmp, "--seed", "3", "--no-step-metrics"] + geo + (["--graph"] if graph else [])
                args = train.parse(argv)
                hist = train.main(args)
                runs.append(torch.tensor([float(l) for l in hist["losses"]]))
        a, b = runs
        fin = bool(torch.isfinite(a).all())
        # efficient-KAN initialises its spline weights with a least-squares solve (models/effkan.py:72-81, as the reference), which is
        # not bitwise reproducible on the CPU (two constructions under one seed differ by ~1e-8): trajectories within 1e-4 there
        same = bool(torch.equal(a, b)) if "efficientkan" not in name else bool(float((a - b).abs().max()) < 1e-4)
        down = float(a[-20:].mean()) < float(a[:20].mean())
        print(f"{name:16s} graph={graph!s:5s} steps={len(a)} first {float(a[:20].mean()):.4f} last {float(a[-20:].mean()):.4f} finite={fin} reproducible={same} decreasing={down}", flush=True)
        ok = ok and fin and same and down
print("SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
