"""GPU parity of the kernels the HEADLINE actually runs, at the headline's own head geometry (dh = 64, N = 197, several heads
per launch, two 32-feature chunks per head) -- forward AND every gradient:

 (1) MSA(128, 2), N = 197 against tensors produced by the imported reference (tests/golden/msa197.npz);
 (2) the grouped q|k|v launch (one kernel for 3*H per-head layers) against the float64 oracle applied layer by layer, at
     (B, N, d, H) = (2, 197, 128, 2) and the ViT-B geometry (2, 197, 768, 12): y, dx, every parameter gradient, d freq;
 (3) one full-geometry block (224x224, patch 16, N = 197) of every BASELINE config's model -- ViT-S FastKAN, ViT-B
     efficient-KAN / Sine / Fourier / Cheby -- logits, loss and all gradients against oracle.vit_forward in float64.

Tolerance: BASELINE.json's 1e-4 (normwise, relative to the largest reference entry), written in the asserts."""
import pytest
import torch

from oracle import kan_oracle as ko
from tests._util import T, close, grads_from, load_npz, max_err, rel_err, state_dict_from

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


@pytest.mark.parametrize("t", ["vanilla", "cheby", "fast", "efficientkan", "sine"])
def test_msa_headline_geometry_against_reference(t):
    from attention import MSA
    blob = load_npz("msa197.npz")
    p = t + "."
    msa = MSA(128, 2, type=t)
    msa.load_state_dict(state_dict_from(blob, p))
    msa = msa.to(DEV)
    x = T(blob[p + "x"]).to(DEV).requires_grad_(True)
    y = msa(x)
    (y * T(blob[p + "wgt"]).to(DEV)).sum().backward()
    assert max_err(y.cpu(), T(blob[p + "y"])) < 1e-5 * max(1.0, float(T(blob[p + "y"]).abs().max()))
    assert rel_err(x.grad.cpu(), T(blob[p + "grad_x"])) < TOL
    got = {k: v.grad.cpu() for k, v in msa.named_parameters() if v.grad is not None}
    want = grads_from(blob, p)
    assert set(want) == set(got)
    for k, g in want.items():
        # key-bias gradients are mathematically zero (softmax shift invariance); the reference's own value is rounding noise
        # of a few 1e-6 at this loss scale (the other gradients of these layers are O(1..10)), and so is ours (up to 9e-6
        # observed, a different noise realisation per kernel revision): the absolute term covers them and nothing else
        assert close(got[k], g, rtol=TOL, atol=2e-5), (k, rel_err(got[k], g))


def _oracle_qkv(msa, x2d, w, h):
    """q|k|v of every head by the oracle's single-layer functions in float64 + gradients of sum(y * w)."""
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in msa.state_dict().items()}
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k) and v.is_floating_point()) for k, v in sd.items()}
    xd = x2d.double().clone().requires_grad_(True)
    dh = x2d.shape[1] // h
    cols = []
    for name in ("q", "k", "v"):
        for hh in range(h):
            cols.append(ko.layer_forward(params, f"{name}_mappings.{hh}.", xd[:, hh * dh:(hh + 1) * dh]))
    y = torch.cat(cols, dim=1)
    (y * w.double()).sum().backward()
    return y.detach(), xd.grad, {k: v.grad for k, v in params.items() if v.grad is not None}


@pytest.mark.parametrize("fam", ["vanilla", "cheby", "efficientkan", "fast", "sine"])
@pytest.mark.parametrize("geom", [(2, 197, 128, 2), (2, 197, 768, 12)])
def test_grouped_qkv_forward_and_backward_vs_fp64_oracle(fam, geom):
    from attention import MSA
    from kanvit import grouped
    b, n, d, h = geom
    torch.manual_seed(31 + d)
    msa = MSA(d, h, type=fam)
    x = torch.randn(b * n, d)
    w = torch.randn(b * n, 3 * d)
    yo, gxo, gpo = _oracle_qkv(msa, x, w, h)
    msa = msa.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, xg)
    (y * w.to(DEV)).sum().backward()
    assert max_err(y.cpu(), yo) < 2e-5 * max(1.0, float(yo.abs().max())), (fam, geom)
    assert rel_err(xg.grad.cpu(), gxo) < TOL, (fam, geom, rel_err(xg.grad.cpu(), gxo))
    got = {k: v.grad.cpu() for k, v in msa.named_parameters() if v.grad is not None}
    assert set(got) == set(gpo), set(got) ^ set(gpo)
    for k, g in gpo.items():
        assert rel_err(got[k], g) < TOL, (fam, geom, k, rel_err(got[k], g))


CONFIGS = {
    # BASELINE.json configs[2] .. [4] + the north_star's ChebyKAN target, one block each at the full 224x224 / patch-16 geometry
    "vits16-fast": (384, 6, "fast"),
    "vitb16-efficientkan": (768, 12, "efficientkan"),
    "vitb16-sine": (768, 12, "sine"),
    "vitb16-fourier": (768, 12, "fourier"),
    "vitb16-cheby": (768, 12, "cheby"),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_full_geometry_block_vs_fp64_oracle(name):
    from model import VisionTransformer
    d, heads, t = CONFIGS[name]
    torch.manual_seed(5)
    m = VisionTransformer((3, 224, 224), n_patches=14, n_blocks=1, d_hidden=d, n_heads=heads, out_d=100, type=t)
    x = torch.randn(2, 3, 224, 224)
    labels = torch.tensor([3, 71])
    sd = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in m.state_dict().items()}
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k) and v.is_floating_point()) for k, v in sd.items()}
    ref = ko.vit_forward(params, x.double(), 14, heads, t)
    ref_loss = torch.nn.functional.cross_entropy(ref, labels)
    ref_loss.backward()

    m = m.to(DEV)
    logits = m(x.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, labels.to(DEV))
    loss.backward()
    assert max_err(logits.cpu(), ref) < TOL, (name, max_err(logits.cpu(), ref))
    assert abs(float(loss) - float(ref_loss)) < TOL
    for k, p in m.named_parameters():
        if not p.requires_grad:               # FastKAN's frozen rbf.grid (models/fastkan.py:22-23)
            continue
        g = params[k].grad
        assert g is not None and p.grad is not None, k
        assert close(p.grad.cpu(), g, rtol=TOL, atol=2e-7), (name, k, rel_err(p.grad.cpu(), g, floor=1e-6))


def test_mixed_sine_fourier_model_vs_fp64_oracle():
    """BASELINE configs[4] as a MODEL: ViT-B geometry (224x224, patch 16, N = 197, 12 heads), two blocks, type="sine,fourier"
    -- block 0 runs SineKAN per-head q|k|v (grid 4), block 1 FourierKAN's (nn.Linear q|k|v, attention.py:136-142), the patch
    embedding is SineKAN grid 28 (model.split_types; the reference takes one global type, model.py:49,67-80,98-103).  Logits,
    loss and EVERY gradient (d freq included) against oracle.vit_forward in float64, which dispatches each layer from the
    state-dict keys."""
    from model import VisionTransformer
    torch.manual_seed(11)
    m = VisionTransformer((3, 224, 224), 14, 2, 768, 12, 100, type="sine,fourier")
    assert m.block_types == ["sine", "fourier"]
    x = torch.randn(2, 3, 224, 224)
    labels = torch.tensor([17, 4])
    sd = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in m.state_dict().items()}
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k) and v.is_floating_point()) for k, v in sd.items()}
    assert ko.layer_kind(params, "blocks.0.attn.q_mappings.0.") == "sine" and ko.layer_kind(params, "blocks.1.attn.q_mappings.0.") == "linear"
    ref = ko.vit_forward(params, x.double(), 14, 12, "sine")
    ref_loss = torch.nn.functional.cross_entropy(ref, labels)
    ref_loss.backward()
    m = m.to(DEV)
    logits = m(x.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, labels.to(DEV))
    loss.backward()
    assert max_err(logits.cpu(), ref) < TOL, max_err(logits.cpu(), ref)
    assert abs(float(loss) - float(ref_loss)) < TOL
    seen = set()
    for k, p in m.named_parameters():
        g = params[k].grad
        assert g is not None and p.grad is not None, k
        assert close(p.grad.cpu(), g, rtol=TOL, atol=2e-7), (k, rel_err(p.grad.cpu(), g, floor=1e-6))
        seen.add(k.split(".")[-1])
    assert {"amplitudes", "freq"} <= seen            # SineKAN parameters of block 0 and of the patch embedding were compared
