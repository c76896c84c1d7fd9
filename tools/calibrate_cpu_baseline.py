#!/usr/bin/env python3
"""Calibrate bench.py's `cpu_baseline` (kind "port": oracle/kan_oracle.py in reference-faithful loop mode) against the REAL
reference imported from /root/reference -- build container only (the reference does not exist on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python tools/calibrate_cpu_baseline.py [--batch 32] [--out profiles/r02_cpu_baseline_calibration.json]

Same protocol as BASELINE.md section 2/3: full train step (forward + CrossEntropy + backward + Adam.step), MNIST-shaped
geometry of model.py:49, seeded synthetic batch, all host threads, 1 warm-up + best of 3.  SURVEY.md section 8(d) asks for
the port's step time to be within +-10 % of the reference's; the ratio per type is written next to the two times."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
import torch  # noqa: E402


def best_of(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--types", default="vanilla,cheby,fast,efficientkan,sine")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_baseline_calibration.json"))
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    sys.path.insert(0, ROOT)
    from oracle import kan_oracle as ko
    ko.REFERENCE_OP_SEQUENCE = True           # as bench.py's cpu_baseline leg runs it
    sys.path.insert(0, "/root/reference")
    import model as ref_model                                  # the reference (CPU)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(args.batch, 1, 28, 28, generator=g)
    y = torch.randint(0, 10, (args.batch,), generator=g)
    rows = {}
    for t in args.types.split(","):
        torch.manual_seed(0)
        m = ref_model.VisionTransformer((1, 28, 28), 7, 4, 64, 2, 10, type=t)
        if t == "cheby":                                       # harness adapter of SURVEY D3 (the shipped model crashes at torch.cat)
            inner = m.linear_mapper.forward
            m.linear_mapper.forward = lambda v, inner=inner: inner(v).reshape(v.shape[0], v.shape[1], -1)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)

        def ref_step():
            loss = torch.nn.functional.cross_entropy(m(x), y)
            opt.zero_grad()
            loss.backward()
            opt.step()

        # the container's cores are shared: interleave the two measurements and keep the best of 6 each
        port_step = lambda: ko.train_steps(sd, x, y, 7, 2, t, steps=1, faithful_loop=True)      # noqa: E731
        ref_step(), port_step()
        t_ref = t_port = float("inf")
        for _ in range(6):
            t0 = time.perf_counter(); ref_step(); t_ref = min(t_ref, time.perf_counter() - t0)      # noqa: E702
            t0 = time.perf_counter(); port_step(); t_port = min(t_port, time.perf_counter() - t0)   # noqa: E702
        rows[t] = {"reference_s": round(t_ref, 4), "port_s": round(t_port, 4), "port_over_reference": round(t_port / t_ref, 3),
                   "reference_images_per_s": round(args.batch / t_ref, 2), "port_images_per_s": round(args.batch / t_port, 2)}
        print(t, rows[t], flush=True)
    out = {"protocol": "train step fwd+CE+bwd+Adam, (1,28,28) np7 L4 d64 H2, batch %d, %d threads, 1 warm-up + best of 6, reference and port interleaved" % (args.batch, torch.get_num_threads()),
           "host": "build container (8 vCPU)", "types": rows}
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
