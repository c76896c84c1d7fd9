"""Fixture loading helpers shared by the CPU and GPU tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def bf16_bits_to_f32(a):
    return torch.from_numpy(a.astype(np.int32) << 16).view(torch.float32).reshape(a.shape).clone()


def state_dict_from(blob, prefix=""):
    """Rebuild a torch state dict from '<prefix>sd.<key>' / '<prefix>sdbf16.<key>' entries."""
    sd = {}
    for k, v in blob.items():
        if k.startswith(prefix + "sd."):
            sd[k[len(prefix) + 3:]] = torch.from_numpy(np.array(v))
        elif k.startswith(prefix + "sdbf16."):
            sd[k[len(prefix) + 7:]] = bf16_bits_to_f32(v)
    return sd


def grads_from(blob, prefix=""):
    return {k[len(prefix) + 5:]: torch.from_numpy(np.array(v)) for k, v in blob.items() if k.startswith(prefix + "grad.")}


def T(a):
    return torch.from_numpy(np.array(a))


def max_err(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(max|b|, floor): scale-aware error for gradient tensors.  The
    floor keeps tensors that are mathematically zero (e.g. the key-bias gradient of
    softmax attention, which is shift invariant) from being compared noise-to-noise."""
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def close(a, b, rtol=1e-4, atol=2e-6):
    """max|a-b| <= atol + rtol*max|b| -- for tensors that may be mathematically zero."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) <= atol + rtol * float(b.abs().max())
