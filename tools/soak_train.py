"""Soak: the drop-in train.py on a synthetic stream at its default (C) geometry and at the MNIST-tiny (T) geometry, eager and
--graph, a few hundred steps each, twice FROM THE SAME INITIAL STATE: losses must stay finite, go down, and be BITWISE identical
between the two repetitions and between eager and --graph -- for every family, efficient-KAN included.

The model is built once per geometry and its state dict handed to every run (`train.main(args, init_state=...)`), so the
comparison sees only the kernels and the optimizer: efficient-KAN's least-squares initialisation (models/effkan.py:72-81, a CPU
`lstsq` that is not bitwise reproducible from one construction to the next) no longer enters it.  A difference here is a
nondeterministic kernel.
    python tools/soak_train.py [steps] [summary.json]"""
import json
import os
import sys
import tempfile
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'kan-vit_amd'))
import torch
import train
from model import VisionTransformer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
T_GEO = ["--image-size", "28", "--in-chans", "1", "--n-patches", "7", "--n-blocks", "4", "--n-heads", "2", "--out-d", "10"]
GEOS = {"C-cheby": ["--model-type", "cheby"],
        "T-cheby": ["--model-type", "cheby"] + T_GEO,
        "C-vanilla": ["--model-type", "vanilla"],
        "C-efficientkan": ["--model-type", "efficientkan"],
        "T-efficientkan": ["--model-type", "efficientkan"] + T_GEO,
        "C-fast": ["--model-type", "fast"],                       # frozen rbf.grid parameter under --graph
        "T-sine,fourier": ["--model-type", "sine,fourier"] + T_GEO}
ok = True
summary = {"steps": steps, "runs": []}
for name, geo in GEOS.items():
    base = ["--synthetic", "--epochs", "1", "--steps-per-epoch", str(steps), "--seed", "3", "--no-step-metrics"] + geo
    a0 = train.parse(base)
    torch.manual_seed(a0.seed)
    init = {k: v.clone() for k, v in VisionTransformer((a0.in_chans, a0.image_size, a0.image_size), n_patches=a0.n_patches,
                                                       n_blocks=a0.n_blocks, d_hidden=a0.d_hidden, n_heads=a0.n_heads, out_d=a0.out_d,
                                                       type=a0.model_type).state_dict().items()}
    traj = {}
    for graph in (False, True):
        runs = []
        for rep in range(2):
            with tempfile.TemporaryDirectory() as tmp:
                args = train.parse(base + ["--log-dir", tmp] + (["--graph"] if graph else []))
                hist = train.main(args, init_state=init)
                runs.append(torch.tensor([float(l) for l in hist["losses"]], dtype=torch.float64))
        a, b = runs
        fin = bool(torch.isfinite(a).all())
        same = bool(torch.equal(a, b))
        down = float(a[-20:].mean()) < float(a[:20].mean())
        traj[graph] = a
        rec = {"config": name, "graph": graph, "steps": len(a), "first20": round(float(a[:20].mean()), 4), "last20": round(float(a[-20:].mean()), 4),
               "finite": fin, "bitwise_reproducible": same, "decreasing": down, "max_abs_diff_between_reps": float((a - b).abs().max())}
        summary["runs"].append(rec)
        print(f"{name:16s} graph={graph!s:5s} steps={len(a)} first {rec['first20']:.4f} last {rec['last20']:.4f} finite={fin} "
              f"bitwise_reproducible={same} decreasing={down}", flush=True)
        ok = ok and fin and same and down
    eg = bool(torch.equal(traj[False], traj[True]))
    summary["runs"].append({"config": name, "eager_equals_graph_bitwise": eg})
    print(f"{name:16s} eager == graph bitwise: {eg}", flush=True)
    ok = ok and eg
summary["ok"] = ok
if len(sys.argv) > 2:
    json.dump(summary, open(sys.argv[2], "w"), indent=1)
print("SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
