#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by importing the REAL
reference (/root/reference, read-only) on the CPU of the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The fixtures are data only (inputs, parameters, outputs, gradients); nothing
of the reference's source travels.  The reference does not exist on the GPU
box, so tests read these files and never import /root/reference.

Adapters -- applied HERE, never to the reference:
  * cheby  : ChebyKANLayer flattens leading dims, so VisionTransformer(type=
             'cheby') crashes at torch.cat; the harness reshapes the patch
             embedding output back to (B, P, d).
  * fourier: model.py passes grid_size= but the ctor wants gridsize=; the
             harness forwards the kwarg under the right name.
Fixtures produced through an adapter carry adapter=1 in their metadata.
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = os.environ.get("KANVIT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch

import model as ref_model                      # noqa: E402  (reference)
from attention import MSA                      # noqa: E402
from models.cheby import ChebyKANLayer         # noqa: E402
from models.effkan import KANLinear            # noqa: E402
from models.fastkan import FastKANLayer        # noqa: E402
from models.nfkan import NaiveFourierKANLayer  # noqa: E402
from models.sinekan import SineKANLayer        # noqa: E402
from utils import FlashAttentionFunction       # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def to_bf16_exact(t):
    """Round to the nearest bf16-representable value, keep float32 storage."""
    return t.to(torch.bfloat16).to(torch.float32)


def bf16_bits(t):
    return npy(t.to(torch.bfloat16).view(torch.int16)).astype(np.uint16)


def kat_fill(t):
    return (0.1 * torch.linspace(-1, 1, t.numel(), dtype=torch.float64)).to(torch.float32).reshape(t.shape)


def make_x(shape, mode, gen):
    if mode == "normal":
        return torch.randn(*shape, generator=gen)
    if mode == "uniform":
        return torch.rand(*shape, generator=gen)
    if mode == "wide":
        return (3.0 * torch.randn(*shape, generator=gen)).clamp(-7.5, 7.5)
    raise ValueError(mode)


LAYER_CTORS = {
    "cheby": lambda i, o, big: ChebyKANLayer(i, o, 4),
    "efficientkan": lambda i, o, big: KANLinear(i, o),
    "fast": lambda i, o, big: FastKANLayer(i, o),
    "fourier": lambda i, o, big: NaiveFourierKANLayer(i, o, 28 if big else 3),
    "sine": lambda i, o, big: SineKANLayer(i, o, grid_size=28 if big else 4),
}
LAYER_SHAPES = [((6, 4), 3, False), ((50, 32), 32, False), ((4, 49, 16), 64, True)]
MODES = ["normal", "uniform", "wide"]


def run_layer(layer, x):
    x = x.clone().requires_grad_(True)
    y = layer(x)
    loss = y.square().sum()
    layer.zero_grad()
    loss.backward()
    return y, x.grad


def gen_layers():
    for fam, ctor in LAYER_CTORS.items():
        blob = {}
        n = 0
        for shape, out_f, big in LAYER_SHAPES:
            for seed, mode in enumerate(MODES):
                torch.manual_seed(seed)
                layer = ctor(shape[-1], out_f, big)
                gen = torch.Generator().manual_seed(100 + seed)
                x = make_x(shape, mode, gen)
                y, gx = run_layer(layer, x)
                p = f"c{n}."
                blob[p + "x"] = npy(x)
                blob[p + "y"] = npy(y)
                blob[p + "grad_x"] = npy(gx)
                for k, v in layer.state_dict().items():
                    blob[p + "sd." + k] = npy(v)
                for k, v in layer.named_parameters():
                    if v.grad is not None:
                        blob[p + "grad." + k] = npy(v.grad)
                n += 1
        # RNG-independent known-answer case
        torch.manual_seed(0)
        layer = {"cheby": lambda: ChebyKANLayer(4, 3, 4), "efficientkan": lambda: KANLinear(4, 3),
                 "fast": lambda: FastKANLayer(4, 3), "fourier": lambda: NaiveFourierKANLayer(4, 3, 3),
                 "sine": lambda: SineKANLayer(4, 3, grid_size=4)}[fam]()
        with torch.no_grad():
            for _, prm in layer.named_parameters():
                if prm.requires_grad:
                    prm.copy_(kat_fill(prm))
        x = torch.linspace(-1.5, 1.5, 24).reshape(6, 4)
        y, gx = run_layer(layer, x)
        blob["kat.x"] = npy(x)
        blob["kat.y"] = npy(y)
        blob["kat.grad_x"] = npy(gx)
        for k, v in layer.state_dict().items():
            blob["kat.sd." + k] = npy(v)
        for k, v in layer.named_parameters():
            if v.grad is not None:
                blob["kat.grad." + k] = npy(v.grad)
        blob["n_cases"] = np.int64(n)
        np.savez_compressed(os.path.join(OUT, f"layer_{fam}.npz"), **blob)
        print("layer", fam, n, "cases")


def gen_msa():
    blob = {}
    for t in ["vanilla", "cheby", "fast", "efficientkan", "sine"]:
        torch.manual_seed(7)
        msa = MSA(64, 2, type=t)
        x = torch.randn(3, 50, 64, generator=torch.Generator().manual_seed(11)).requires_grad_(True)
        y = msa(x)
        (y * torch.linspace(-1, 1, y.numel()).reshape(y.shape)).sum().backward()
        p = t + "."
        blob[p + "x"], blob[p + "y"], blob[p + "grad_x"] = npy(x), npy(y), npy(x.grad)
        for k, v in msa.state_dict().items():
            blob[p + "sd." + k] = npy(v)
        for k, v in msa.named_parameters():
            if v.grad is not None:
                blob[p + "grad." + k] = npy(v.grad)
    np.savez_compressed(os.path.join(OUT, "msa.npz"), **blob)
    print("msa done")


def gen_msa_big():
    """The headline head geometry: MSA(128, 2) -> dh = 64, N = 197 (two 32-feature chunks per head, the register-form
    kernels with 2 heads sharing a launch); parameters rounded to bf16-representable values so they store in 2 bytes."""
    blob = {}
    for t in ["vanilla", "cheby", "fast", "efficientkan", "sine"]:
        torch.manual_seed(17)
        msa = MSA(128, 2, type=t)
        with torch.no_grad():
            for prm in msa.parameters():
                prm.copy_(to_bf16_exact(prm))
        x = torch.randn(2, 197, 128, generator=torch.Generator().manual_seed(23)).requires_grad_(True)
        y = msa(x)
        wgt = torch.sin(torch.arange(y.numel(), dtype=torch.float32) * 0.37).reshape(y.shape)
        (y * wgt).sum().backward()
        p = t + "."
        blob[p + "x"], blob[p + "y"], blob[p + "grad_x"], blob[p + "wgt"] = npy(x), npy(y), npy(x.grad), npy(wgt)
        for k, v in msa.state_dict().items():
            if v.dtype == torch.float32 and not k.endswith(("grid", "phase")):
                blob[p + "sdbf16." + k] = bf16_bits(v)
            else:
                blob[p + "sd." + k] = npy(v)
        for k, v in msa.named_parameters():
            if v.grad is not None:
                blob[p + "grad." + k] = npy(v.grad)
    np.savez_compressed(os.path.join(OUT, "msa197.npz"), **blob)
    print("msa197 done")


def gen_metrics():
    """Output of the reference's reporting helpers (utils.py:13-47, 79-94): the text save_metrics writes (train block with
    flag 0, test block with flag 1) and calculate_metrics on a fixed 300-sample, 100-class input."""
    import tempfile
    import utils as ref_utils
    blob = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)                                   # the reference creates ./logs unconditionally (utils.py:79)
        try:
            fn = os.path.join(td, "m.txt")
            ref_utils.save_metrics(fn, 20, "Train", 1.23456789, 0.5, 0.25, 0.125, 0.0625, 0)
            ref_utils.save_metrics(fn, 20, "Test", 4.60517, 0.01, 0.0123456, 0.99995, 0.5, 1)
            blob["save_metrics_text"] = np.frombuffer(open(fn, "rb").read(), dtype=np.uint8)
        finally:
            os.chdir(cwd)
    g = torch.Generator().manual_seed(9)
    y_true = (torch.arange(300) % 100).numpy()
    proba = torch.softmax(2.0 * torch.randn(300, 100, generator=g), dim=1)
    proba[torch.arange(0, 300, 3), torch.arange(0, 300, 3) % 100] += 1.0          # a third of the samples classified right
    proba = (proba / proba.sum(1, keepdim=True)).numpy()
    y_pred = proba.argmax(1)
    blob["cm.y_true"], blob["cm.y_pred"], blob["cm.proba"] = y_true, y_pred, proba.astype(np.float32)
    blob["cm.out"] = np.array(ref_utils.calculate_metrics(list(y_true), list(y_pred), list(proba.astype(np.float32))), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **blob)
    print("metrics done", blob["cm.out"])


def gen_flash():
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(1, 3, 197, 64, generator=g).requires_grad_(True) for _ in range(3))
    do = torch.randn(1, 3, 197, 64, generator=g)
    blob = {"q": npy(q), "k": npy(k), "v": npy(v), "do": npy(do)}
    for tag, (qb, kb) in {"small": (64, 128), "big": (512, 1024)}.items():
        for t in (q, k, v):
            t.grad = None
        o = FlashAttentionFunction.apply(q, k, v, None, False, qb, kb)
        o.backward(do)
        blob[f"{tag}.o"] = npy(o)
        blob[f"{tag}.dq"], blob[f"{tag}.dk"], blob[f"{tag}.dv"] = npy(q.grad), npy(k.grad), npy(v.grad)
    # causal, single tile
    for t in (q, k, v):
        t.grad = None
    o = FlashAttentionFunction.apply(q, k, v, None, True, 512, 1024)
    o.backward(do)
    blob["causal.o"] = npy(o)
    blob["causal.dq"], blob["causal.dk"], blob["causal.dv"] = npy(q.grad), npy(k.grad), npy(v.grad)
    np.savez_compressed(os.path.join(OUT, "flash.npz"), **blob)
    print("flash done")


def gen_flash_x():
    """FlashAttentionFunction beyond the ViT path: key-padding / full masks and q_len != k_len (utils.py:141-195, 229-295),
    the reference's own outputs and gradients for the general attention kernels (csrc/attention_x.hip).  Not generated: causal
    with k_len > q_len -- utils.py:169,183 shift the diagonal the wrong way (q_start_index = ind*bucket MINUS qk_len_diff: query i
    sees keys j <= i - (k_len - q_len)), so the first k_len - q_len queries have no visible key and the reference returns a
    bucket-size dependent average over causally masked keys for them; this build refuses that combination."""
    g = torch.Generator().manual_seed(11)
    cases = {
        # tag: (b, h, q_len, k_len, d, causal, mask kind, (q_bucket, k_bucket))
        "keypad": (2, 3, 50, 50, 32, False, "bn", (32, 16)),
        "cross": (2, 2, 37, 197, 64, False, None, (512, 1024)),
        "cross_mask4": (2, 2, 70, 33, 64, False, "b1qk", (64, 16)),
        "short_causal": (1, 2, 60, 24, 16, True, None, (512, 1024)),
        "keypad_causal": (2, 2, 45, 45, 32, True, "bn_tail", (512, 1024)),
    }
    blob = {}
    for tag, (b, h, nq, nk, d, causal, mk, (qb, kb)) in cases.items():
        q = torch.randn(b, h, nq, d, generator=g).requires_grad_(True)
        k = torch.randn(b, h, nk, d, generator=g).requires_grad_(True)
        v = torch.randn(b, h, nk, d, generator=g).requires_grad_(True)
        do = torch.randn(b, h, nq, d, generator=g)
        mask = None
        if mk == "bn":
            mask = torch.rand(b, nk, generator=g) > 0.3
            mask[:, 0] = True
        elif mk == "bn_tail":                    # padding at the end of the sequence; key 0 stays visible to every causal row
            mask = torch.ones(b, nk, dtype=torch.bool)
            mask[0, nk - 7:] = False
            mask[1, nk - 20:] = False
        elif mk == "b1qk":
            mask = torch.rand(b, 1, nq, nk, generator=g) > 0.4
            mask[..., 0] = True                  # every query keeps at least one key
        o = FlashAttentionFunction.apply(q, k, v, mask, causal, qb, kb)
        o.backward(do)
        blob[f"{tag}.q"], blob[f"{tag}.k"], blob[f"{tag}.v"], blob[f"{tag}.do"] = npy(q), npy(k), npy(v), npy(do)
        blob[f"{tag}.o"], blob[f"{tag}.dq"], blob[f"{tag}.dk"], blob[f"{tag}.dv"] = npy(o), npy(q.grad), npy(k.grad), npy(v.grad)
        blob[f"{tag}.causal"] = np.array(int(causal))
        if mask is not None:
            blob[f"{tag}.mask"] = mask.numpy()
    # a fully masked batch row: o = 0 through the reference's clamp(min=EPSILON) row sum
    q, k, v = (torch.randn(2, 1, 8, 16, generator=g) for _ in range(3))
    mask = torch.ones(2, 8, dtype=torch.bool)
    mask[1] = False
    o = FlashAttentionFunction.apply(q, k, v, mask, False, 512, 1024)
    blob["allmasked.q"], blob["allmasked.k"], blob["allmasked.v"], blob["allmasked.mask"], blob["allmasked.o"] = npy(q), npy(k), npy(v), mask.numpy(), npy(o)
    np.savez_compressed(os.path.join(OUT, "flash_x.npz"), **blob)
    print("flash_x done")


def build_reference_vit(chw, n_patches, n_blocks, d, heads, out_d, t):
    """Instantiate the reference model; apply the D3/D4 harness adapters when needed."""
    adapter = 0
    if t == "fourier":
        real = ref_model.NaiveFourierKANLayer
        ref_model.NaiveFourierKANLayer = lambda i, o, grid_size=28: real(i, o, gridsize=grid_size)
        try:
            m = ref_model.VisionTransformer(chw, n_patches, n_blocks, d, heads, out_d, type=t)
        finally:
            ref_model.NaiveFourierKANLayer = real
        adapter = 1
    else:
        m = ref_model.VisionTransformer(chw, n_patches, n_blocks, d, heads, out_d, type=t)
    if t == "cheby":
        inner = m.linear_mapper.forward
        m.linear_mapper.forward = lambda x: inner(x).reshape(x.shape[0], x.shape[1], -1)
        adapter = 1
    return m, adapter


GEOMS = {
    # name: (chw, n_patches, n_blocks, d, heads, out_d, input mode)
    "T": ((1, 28, 28), 7, 2, 64, 2, 10, "uniform"),       # MNIST-shaped defaults of model.py:49 (2 blocks)
    "C": ((3, 32, 32), 4, 2, 64, 8, 100, "normal"),       # train.py:18-20 geometry (2 blocks)
}
TYPES = ["vanilla", "flash-attn", "efficientkan", "sine", "fourier", "cheby", "fast"]


def gen_models():
    for gname, (chw, npatch, nblk, d, heads, out_d, mode) in GEOMS.items():
        for t in TYPES:
            torch.manual_seed(0)
            m, adapter = build_reference_vit(chw, npatch, nblk, d, heads, out_d, t)
            # parameters rounded to bf16-representable values so they store in 2 bytes, exactly
            with torch.no_grad():
                for prm in m.parameters():
                    prm.copy_(to_bf16_exact(prm))
            gen = torch.Generator().manual_seed(3)
            x = make_x((4,) + chw, mode, gen)
            labels = torch.arange(4) % out_d
            blob = {"x": npy(x), "labels": npy(labels), "adapter": np.int64(adapter),
                    "cfg": np.array([chw[0], chw[1], chw[2], npatch, nblk, d, heads, out_d], dtype=np.int64)}
            for k, v in m.state_dict().items():
                if v.dtype == torch.float32 and not k.endswith(("grid", "phase")):
                    blob["sdbf16." + k] = bf16_bits(v)
                else:
                    blob["sd." + k] = npy(v)
            m.train()
            logits = m(x)
            loss = torch.nn.functional.cross_entropy(logits, labels)
            m.zero_grad()
            loss.backward()
            blob["logits"], blob["loss"] = npy(logits), npy(loss)
            names = [k for k, v in m.named_parameters() if v.grad is not None]
            blob["grad_names"] = np.array(names)
            blob["grad_norms"] = np.array([float(dict(m.named_parameters())[k].grad.double().norm()) for k in names])
            if gname == "T":
                for k in names:
                    blob["grad." + k] = npy(dict(m.named_parameters())[k].grad)
            else:
                for k in names[:2] + names[-2:] + [n for n in names if "q_mappings.0" in n][:2]:
                    if dict(m.named_parameters())[k].numel() <= 70000:
                        blob["grad." + k] = npy(dict(m.named_parameters())[k].grad)
            # 3 Adam steps from the same state (train.py:31-40 order)
            opt = torch.optim.Adam(m.parameters(), lr=1e-3)
            traj = []
            for _ in range(3):
                lg = m(x)
                ls = torch.nn.functional.cross_entropy(lg, labels)
                opt.zero_grad()
                ls.backward()
                opt.step()
                traj.append(float(ls))
            blob["adam_losses"] = np.array(traj)
            blob["adam_v_class"] = npy(m.v_class)
            first_w = [k for k in names if k.startswith("linear_mapper.")][0]
            blob["adam_w_name"] = np.array(first_w)
            blob["adam_w"] = npy(dict(m.named_parameters())[first_w]).reshape(-1)[:8192]   # leading slice
            np.savez_compressed(os.path.join(OUT, f"model_{gname}_{t}.npz"), **blob)
            print("model", gname, t, "loss", float(loss), "adapter", adapter)


def gen_misc():
    """Structural known answers and the pos-emb / patchify helpers."""
    blob = {}
    kl = KANLinear(1, 1)
    for name, val in {"x0": 0.0, "x1": 1.0, "xm": -2.3, "xp": 2.2, "xh": 0.37}.items():
        blob["bspline." + name] = npy(kl.b_splines(torch.tensor([[val]]))[0, 0])
    blob["bspline.grid"] = npy(kl.grid)
    f = FastKANLayer(4, 3)
    blob["rbf.grid"] = npy(f.rbf.grid)
    blob["rbf.h"] = np.float64(f.rbf.denominator)
    s = SineKANLayer(32, 32, grid_size=4)
    blob["sine4.freq"], blob["sine4.phase"] = npy(s.freq), npy(s.phase)
    s = SineKANLayer(16, 64, grid_size=28)
    blob["sine28.phase"] = npy(s.phase)
    vt = ref_model.VisionTransformer((1, 28, 28), 7, 1, 64, 2, 10, type="vanilla")
    blob["pos_emb_50_64"] = npy(vt.pos_embeddings)
    img = torch.arange(2 * 3 * 8 * 8, dtype=torch.float32).reshape(2, 3, 8, 8)
    vt2 = ref_model.VisionTransformer((3, 8, 8), 2, 1, 16, 2, 10, type="vanilla")
    blob["patchify_in"], blob["patchify_out"] = npy(img), npy(vt2.patchify(img, 2))
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **blob)
    print("misc done")


if __name__ == "__main__":
    only = set(sys.argv[1:])                   # e.g. `make_golden.py msa197 metrics` regenerates just those files
    gens = {"misc": gen_misc, "layers": gen_layers, "msa": gen_msa, "msa197": gen_msa_big, "metrics": gen_metrics,
            "flash": gen_flash, "flash_x": gen_flash_x, "models": gen_models}
    for name, fn in gens.items():
        if not only or name in only:
            fn()
