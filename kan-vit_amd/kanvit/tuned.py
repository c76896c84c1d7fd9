"""Recorded GEMM kernel selections for the stock (library) GEMMs of the feed-forward block.

The feed-forward Linears are hipBLASLt / rocBLAS GEMMs (not KAN kernels).  The libraries' default heuristics
pick kernels that run the ViT-B shapes at 125-133 TFLOP/s fp32; PyTorch's TunableOp, run once on an MI355X
(`PYTORCH_TUNABLEOP_ENABLED=1 python bench.py`), found solutions at 140-148 TFLOP/s (0.93 -> 0.80-0.84 ms per GEMM, 5.5 % of
the fp32 train step) and similar gains for the bf16 shapes.  `tunable_gfx950.csv` holds those selections for the ViT-B and
ViT-S feed-forward shapes; enable_tuned_gemms() makes torch USE them without ever tuning at run time.  The file carries
validators (torch / HIP / hipBLASLt / rocBLAS versions, gfx950): on any other build torch ignores it and the default
heuristics apply.  Same libraries, same exact fp32 arithmetic -- only the tile shape / kernel choice changes."""
import os

import torch

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunable_gfx950.csv")


def enable_tuned_gemms(path: str = RESULTS) -> bool:
    """Returns True when the recorded selections were accepted by this torch build."""
    if not torch.cuda.is_available() or not os.path.exists(path):
        return False
    try:
        import torch.cuda.tunable as tn
        tn.enable(True)
        tn.tuning_enable(False)             # never tune at run time: use what is recorded, default heuristics otherwise
        if hasattr(tn, "write_file_on_exit"):
            tn.write_file_on_exit(False)
        import tempfile
        tn.set_filename(os.path.join(tempfile.gettempdir(), f"kanvit_tunableop_{os.getpid()}.csv"))   # never write in-tree
        return bool(tn.read_file(path))
    except Exception:
        return False
