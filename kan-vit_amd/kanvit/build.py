"""Compile the HIP sources in ../csrc into kanvit/libkanvit.so for gfx950 (MI355X).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the working tree snapshot.

    python kan-vit_amd/kanvit/build.py [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIB = os.path.join(HERE, "libkanvit.so")
# the fused KAN layer kernels are one translation unit per kernel generation (kan_layer_common.h): parallel compiles, and a
# rebuild costs only the kernel that changed
SOURCES = ["kan_tile.hip", "kan_fwd_reg.hip", "kan_fwd_reg_bf16.hip", "kan_bwd_input_reg.hip", "kan_bwd_input_reg_bf16.hip",
           "kan_bwd_weight_reg.hip", "kan_bwd_weight_dma.hip", "kan_layer.hip", "attention.hip", "attention16.hip", "attention_x.hip", "addln.hip", "split3.hip", "ff_small.hip", "ff_epilogue.hip", "kan_tiny.hip"]
HEADERS = ["kan_basis.h", "kanvit_common.h", "kan_layer_common.h", "attention_common.h", os.path.join(INCLUDE, "kanvit.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=on", "-fno-finite-math-only", "-fvisibility=hidden"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """(Re)build libkanvit.so when a source is newer than it. Returns the library path."""
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, s.replace(".hip", ".o"))
        cmd = [_hipcc(), *FLAGS, "-I", INCLUDE, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(LIB + ".tmp", LIB)
    for o in objs:
        os.remove(o)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
