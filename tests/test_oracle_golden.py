"""Pin the CPU oracle (oracle/kan_oracle.py) against tensors produced by the real
reference (tests/golden/*.npz, made by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import kan_oracle as ko
from tests._util import T, close, grads_from, load_npz, max_err, rel_err, state_dict_from

FAMS = ["cheby", "efficientkan", "fast", "fourier", "sine"]


def _run(sd, x, keep2d=True):
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k)) for k, v in sd.items()}
    x = x.clone().requires_grad_(True)
    y = ko.layer_forward(params, "", x, cheby_keep_2d=keep2d)
    y.square().sum().backward()
    return y.detach(), x.grad, {k: v.grad for k, v in params.items() if v.grad is not None}


@pytest.mark.parametrize("fam", FAMS)
def test_layer_cases_fp32(fam):
    blob = load_npz(f"layer_{fam}.npz")
    for c in range(int(blob["n_cases"])):
        p = f"c{c}."
        sd = state_dict_from(blob, p)
        y, gx, gp = _run(sd, T(blob[p + "x"]))
        assert y.shape == T(blob[p + "y"]).shape
        assert max_err(y, T(blob[p + "y"])) < 2e-5, (fam, c)
        assert rel_err(gx, T(blob[p + "grad_x"])) < 2e-4, (fam, c)
        for k, g in grads_from(blob, p).items():
            assert rel_err(gp[k], g) < 2e-4, (fam, c, k)


@pytest.mark.parametrize("fam", ["cheby", "sine"])
def test_reference_op_sequence_gives_the_same_values(fam, monkeypatch):
    """bench.py's cpu_baseline leg times ChebyKAN / SineKAN in the reference's own op sequence (REFERENCE_OP_SEQUENCE): the
    values must be the ones pinned above."""
    monkeypatch.setattr(ko, "REFERENCE_OP_SEQUENCE", True)
    blob = load_npz(f"layer_{fam}.npz")
    for c in range(int(blob["n_cases"])):
        p = f"c{c}."
        y, gx, gp = _run(state_dict_from(blob, p), T(blob[p + "x"]))
        assert max_err(y, T(blob[p + "y"])) < 2e-5, (fam, c)
        assert rel_err(gx, T(blob[p + "grad_x"])) < 2e-4, (fam, c)
        for k, g in grads_from(blob, p).items():
            assert rel_err(gp[k], g) < 2e-4, (fam, c, k)


@pytest.mark.parametrize("fam", FAMS)
def test_layer_cases_fp64_truth(fam):
    """The float64 oracle is the 'truth' the GPU tests measure against; it must sit
    within fp32 rounding of the reference's fp32 output."""
    blob = load_npz(f"layer_{fam}.npz")
    for c in range(int(blob["n_cases"])):
        p = f"c{c}."
        sd = {k: (v.double() if v.is_floating_point() else v) for k, v in state_dict_from(blob, p).items()}
        y, _, _ = _run(sd, T(blob[p + "x"]).double())
        assert max_err(y, T(blob[p + "y"])) < 2e-5, (fam, c)


KAT = {
    "cheby": ([-0.064203, -0.055008, -0.045813], -0.161604, 0.142479, -0.020922),
    "efficientkan": ([0.099154, 0.001224, -0.062292], 0.359072, 0.509438, 0.004092),
    "fast": ([-0.49536, 0.001911, 0.499181], 0.141934, 2.529268, -0.005969),
    "fourier": ([0.244269, 0.070648, -0.102973], -1.740116, -0.096076, 0.096371),
    "sine": ([-0.31465, -0.009308, 0.296034], -0.165729, -0.101462, -0.015121),
}


@pytest.mark.parametrize("fam", FAMS)
def test_known_answers(fam):
    blob = load_npz(f"layer_{fam}.npz")
    sd = state_dict_from(blob, "kat.")
    for k, v in sd.items():                       # the fill is RNG independent: rebuild it here
        if not ko.is_buffer_key(k):
            assert torch.equal(v, ko.kat_fill(v)), k
    x = torch.linspace(-1.5, 1.5, 24).reshape(6, 4)
    y, gx, _ = _run(sd, x)
    y0, ysum, gxsum, gx00 = KAT[fam]
    assert np.allclose(y[0].numpy(), y0, atol=2e-6)
    assert abs(float(y.sum()) - ysum) < 5e-6
    assert abs(float(gx.sum()) - gxsum) < 5e-6
    assert abs(float(gx[0, 0]) - gx00) < 2e-6
    assert max_err(y, T(blob["kat.y"])) < 1e-6


def test_structural_known_answers():
    m = load_npz("misc.npz")
    grid = ko.make_uniform_knots(1)
    assert max_err(grid, T(m["bspline.grid"])) == 0.0
    for name, val in {"x0": 0.0, "x1": 1.0, "xm": -2.3, "xp": 2.2, "xh": 0.37}.items():
        b = ko.bspline_bases(torch.tensor([[val]]), grid, 3)[0, 0]
        assert max_err(b, T(m["bspline." + name])) < 1e-6
    b0 = ko.bspline_bases(torch.tensor([[0.0]], dtype=torch.float64), grid.double(), 3)[0, 0]
    assert torch.allclose(b0, torch.tensor([0, 0, 1 / 48, 23 / 48, 23 / 48, 1 / 48, 0, 0], dtype=torch.float64), atol=1e-6)
    assert float(ko.bspline_bases(torch.tensor([[2.2]]), grid, 3).abs().sum()) == 0.0
    assert max_err(torch.linspace(-2, 2, 8), T(m["rbf.grid"])) == 0.0
    assert abs(float(m["rbf.h"]) - 4 / 7) < 1e-12
    assert max_err(ko.sine_phase(32, 4), T(m["sine4.phase"])) < 1e-6
    assert max_err(ko.sine_phase(16, 28), T(m["sine28.phase"])) < 2e-5
    assert np.allclose(m["sine4.freq"].reshape(-1), [0.2, 0.4, 0.6, 0.8])
    assert max_err(ko.positional_embeddings(50, 64), T(m["pos_emb_50_64"])) < 1e-6
    assert max_err(ko.patchify(T(m["patchify_in"]), 2), T(m["patchify_out"])) == 0.0


@pytest.mark.parametrize("t", ["vanilla", "cheby", "fast", "efficientkan", "sine"])
@pytest.mark.parametrize("loop", [False, True])
def test_msa(t, loop):
    blob = load_npz("msa.npz")
    p = t + "."
    sd = state_dict_from(blob, p)
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k)) for k, v in sd.items()}
    x = T(blob[p + "x"]).clone().requires_grad_(True)
    y = ko.msa_forward(params, "", x, 2, faithful_loop=loop)
    (y * torch.linspace(-1, 1, y.numel()).reshape(y.shape)).sum().backward()
    assert max_err(y, T(blob[p + "y"])) < 1e-5
    assert rel_err(x.grad, T(blob[p + "grad_x"])) < 1e-4
    for k, g in grads_from(blob, p).items():
        assert rel_err(params[k].grad, g) < 1e-3, k


@pytest.mark.parametrize("t", ["vanilla", "cheby", "fast", "efficientkan", "sine"])
def test_msa_headline_head_geometry(t):
    """MSA(128, 2) at N = 197 (dh = 64: the head geometry of the ViT-B / ViT-S headline) from the imported reference."""
    blob = load_npz("msa197.npz")
    p = t + "."
    sd = state_dict_from(blob, p)
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k)) for k, v in sd.items()}
    x = T(blob[p + "x"]).clone().requires_grad_(True)
    y = ko.msa_forward(params, "", x, 2)
    (y * T(blob[p + "wgt"])).sum().backward()
    assert max_err(y, T(blob[p + "y"])) < 2e-5
    assert rel_err(x.grad, T(blob[p + "grad_x"])) < 1e-4
    for k, g in grads_from(blob, p).items():      # the key-bias gradients are mathematically zero (softmax shift invariance)
        assert close(params[k].grad, g, rtol=1e-3, atol=5e-6), (k, rel_err(params[k].grad, g))


def test_flash_attention_function():
    f = load_npz("flash.npz")
    q, k, v, do = (T(f[n]) for n in ("q", "k", "v", "do"))
    o, lse = ko.attention_reference(q, k, v)
    for tag, (qb, kb) in {"small": (64, 128), "big": (512, 1024)}.items():
        ot, lset = ko.flash_attention_tiled(q, k, v, qb, kb)
        assert max_err(ot, T(f[tag + ".o"])) < 2e-6
        assert max_err(o, T(f[tag + ".o"])) < 2e-6
        assert max_err(lse, lset) < 1e-5
        dq, dk, dv = ko.flash_attention_backward(q, k, v, o, lse, do)
        assert max_err(dq, T(f[tag + ".dq"])) < 1e-5
        assert max_err(dk, T(f[tag + ".dk"])) < 1e-5
        assert max_err(dv, T(f[tag + ".dv"])) < 1e-5
    oc, _ = ko.attention_reference(q, k, v, causal=True)
    assert max_err(oc, T(f["causal.o"])) < 2e-6


FLASH_X_CASES = ["keypad", "cross", "cross_mask4", "short_causal", "keypad_causal"]


@pytest.mark.parametrize("tag", FLASH_X_CASES)
def test_flash_attention_function_masks_and_cross_lengths(tag):
    """The oracle's restatement of the mask / q_len != k_len semantics (utils.py:141-195, 229-295) against the reference's own
    outputs and gradients (tests/golden/flash_x.npz, make_golden.py::gen_flash_x)."""
    f = load_npz("flash_x.npz")
    q, k, v, do = (T(f[f"{tag}.{n}"]) for n in ("q", "k", "v", "do"))
    causal = bool(int(f[f"{tag}.causal"]))
    mask = torch.from_numpy(f[f"{tag}.mask"]) if f"{tag}.mask" in f else None
    o, lse = ko.attention_reference(q, k, v, causal=causal, mask=mask)
    assert max_err(o, T(f[f"{tag}.o"])) < 3e-6
    dq, dk, dv = ko.flash_attention_backward(q, k, v, o, lse, do, causal=causal, mask=mask)
    assert max_err(dq, T(f[f"{tag}.dq"])) < 2e-5
    assert max_err(dk, T(f[f"{tag}.dk"])) < 2e-5
    assert max_err(dv, T(f[f"{tag}.dv"])) < 2e-5


def test_flash_attention_function_fully_masked_row():
    f = load_npz("flash_x.npz")
    q, k, v = (T(f[f"allmasked.{n}"]) for n in ("q", "k", "v"))
    mask = torch.from_numpy(f["allmasked.mask"])
    o, lse = ko.attention_reference(q, k, v, mask=mask)
    assert max_err(o, T(f["allmasked.o"])) < 3e-6 and float(o[1].abs().max()) == 0.0
    assert float(T(f["allmasked.o"])[1].abs().max()) == 0.0          # the reference's own answer for the fully masked sample


TYPES = ["vanilla", "flash-attn", "efficientkan", "sine", "fourier", "cheby", "fast"]


@pytest.mark.parametrize("geom", ["T", "C"])
@pytest.mark.parametrize("t", TYPES)
def test_model_logits_loss_grads(geom, t):
    blob = load_npz(f"model_{geom}_{t}.npz")
    cfg = [int(v) for v in blob["cfg"]]
    n_patches, heads = cfg[3], cfg[6]
    sd = state_dict_from(blob)
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k)) for k, v in sd.items()}
    logits = ko.vit_forward(params, T(blob["x"]), n_patches, heads, t)
    loss = torch.nn.functional.cross_entropy(logits, T(blob["labels"]))
    loss.backward()
    assert max_err(logits, T(blob["logits"])) < 5e-5
    assert abs(float(loss.detach()) - float(blob["loss"])) < 2e-5
    names = [str(n) for n in blob["grad_names"]]
    for n, gn in zip(names, blob["grad_norms"]):
        mine = float(params[n].grad.double().norm())
        assert abs(mine - gn) <= 2e-4 * max(gn, 1e-3) + 1e-7, (n, mine, gn)
    for n, g in grads_from(blob).items():
        assert rel_err(params[n].grad, g) < 5e-4, n


@pytest.mark.parametrize("t", ["vanilla", "cheby", "efficientkan"])
def test_three_adam_steps(t):
    blob = load_npz(f"model_T_{t}.npz")
    cfg = [int(v) for v in blob["cfg"]]
    losses, sd2 = ko.train_steps(state_dict_from(blob), T(blob["x"]), T(blob["labels"]), cfg[3], cfg[6], t, steps=3)
    assert np.allclose(losses, blob["adam_losses"], atol=5e-5)
    assert max_err(sd2["v_class"], T(blob["adam_v_class"])) < 2e-5
    w = sd2[str(blob["adam_w_name"])].reshape(-1)[:8192]
    assert max_err(w, T(blob["adam_w"])) < 2e-5
