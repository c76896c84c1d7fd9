"""The bf16 matrix-core mode (KANVIT_FLAG_BF16_MFMA, set under torch.autocast(bfloat16); BASELINE configs[2], [4]) checked
against the ORACLE -- never against this repo's own fp32 kernels.

Two kinds of check, both with the tolerance written in the assert:
 * TIGHT (forward): the float64 oracle evaluated with the operands of every contraction (basis values Phi(x), packed
   coefficients; q, k, v and the probabilities for attention) rounded to bf16 exactly as the kernels round them
   (oracle.operand_rounding).  What is left is fp32-vs-fp64 accumulation and the rare rounding flip of a basis value that
   sits on a bf16 rounding boundary: agreement 1e-3 of the largest entry or better, an order of magnitude below bf16
   noise -- a mis-packed weight image or a wrong k-permutation cannot hide under it.
 * LOOSE (backward, whole blocks): against the unrounded float64 oracle / the imported reference's fp32 tensors at
   <= 1e-2 in the Frobenius norm ("bf16 configs should be judged against the fp32 reference with a
   stated tolerance ~1e-2 rel"), plus a check that the bf16 kernel really ran (the result differs from exact fp32)."""
import pytest
import torch

from oracle import kan_oracle as ko
from tests._util import T, grads_from, load_npz, state_dict_from

pytestmark = pytest.mark.gpu
DEV = "cuda"
TIGHT = 1e-3          # max |err| / max |ref|, forward against the bf16-operand oracle
LOOSE = 1e-2          # ||err||_F / ||ref||_F against the unrounded oracle / reference fixtures


def fro(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _oracle_qkv(msa, x2d, h, w=None, rounded=False):
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in msa.state_dict().items()}
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k) and v.is_floating_point()) for k, v in sd.items()}
    xd = x2d.double().clone().requires_grad_(True)
    dh = x2d.shape[1] // h

    def run():
        cols = []
        for name in ("q", "k", "v"):
            for hh in range(h):
                cols.append(ko.layer_forward(params, f"{name}_mappings.{hh}.", xd[:, hh * dh:(hh + 1) * dh]))
        return torch.cat(cols, dim=1)

    if rounded:
        with ko.operand_rounding(ko.bf16_round):
            return run().detach(), None, None
    y = run()
    (y * w.double()).sum().backward()
    return y.detach(), xd.grad, {k: v.grad for k, v in params.items() if v.grad is not None}


@pytest.mark.parametrize("fam", ["vanilla", "cheby", "efficientkan", "fast", "sine"])
@pytest.mark.parametrize("geom", [(4, 197, 128, 2), (22, 197, 384, 6)])      # second: M = 4334 >= 4096 rows -> W-stationary forward
def test_qkv_bf16_forward_tight_and_backward_loose(fam, geom):
    from attention import MSA
    from kanvit import grouped
    b, n, d, h = geom
    torch.manual_seed(77 + d)
    msa = MSA(d, h, type=fam)
    x = torch.randn(b * n, d)
    w = torch.randn(b * n, 3 * d)
    y_tight, _, _ = _oracle_qkv(msa, x, h, rounded=True)
    y_exact, gx, gp = _oracle_qkv(msa, x, h, w=w)
    msa = msa.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, xg)
    assert y.dtype == torch.float32
    (y * w.to(DEV)).sum().backward()
    assert maxrel(y, y_tight) < TIGHT, (fam, geom, maxrel(y, y_tight))
    assert 1e-5 < fro(y, y_exact) < LOOSE, (fam, geom, fro(y, y_exact))          # > 1e-5: the bf16 kernel really ran
    assert fro(xg.grad, gx) < LOOSE, (fam, geom, "dx", fro(xg.grad, gx))
    got = {k: v.grad for k, v in msa.named_parameters() if v.grad is not None}
    assert set(got) == set(gp)
    for k, g in gp.items():
        assert fro(got[k], g) < LOOSE, (fam, geom, k, fro(got[k], g))


@pytest.mark.parametrize("fam,i,o,big", [("cheby", 768, 768, False), ("efficientkan", 768, 384, False), ("fast", 768, 384, False),
                                         ("sine", 192, 128, True), ("fourier", 192, 128, True)])
def test_patch_embedding_layer_bf16(fam, i, o, big):
    """groups = 1 launches (the patch embedding): wide K, G = 28 bases for sine / fourier."""
    from tests.test_layers_gpu import make_layer
    torch.manual_seed(9)
    layer = make_layer(fam, i, o, big)
    x = torch.randn(392, i) * 0.7
    sd = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in layer.state_dict().items()}
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k) and v.is_floating_point()) for k, v in sd.items()}
    with ko.operand_rounding(ko.bf16_round):
        y_tight = ko.layer_forward(params, "", x.double()).detach()
    xd = x.double().requires_grad_(True)
    y_exact = ko.layer_forward(params, "", xd)
    w = torch.randn(392, o)
    (y_exact * w.double()).sum().backward()
    layer = layer.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = layer(xg)
    (y * w.to(DEV)).sum().backward()
    assert maxrel(y, y_tight) < TIGHT, (fam, maxrel(y, y_tight))
    assert 1e-5 < fro(y, y_exact.detach()) < LOOSE, (fam, fro(y, y_exact.detach()))
    assert fro(xg.grad, xd.grad) < LOOSE, (fam, "dx", fro(xg.grad, xd.grad))
    for k, p in layer.named_parameters():
        if params[k].grad is not None:
            assert fro(p.grad, params[k].grad) < LOOSE, (fam, k, fro(p.grad, params[k].grad))


@pytest.mark.parametrize("n,d", [(197, 64), (50, 32), (17, 16)])
def test_attention_bf16(n, d):
    """forward against the oracle with q, k, v and the (unnormalised) probabilities rounded to bf16 as the kernel rounds them;
    backward (dq, dk, dv) against the float64 oracle at the loose bound."""
    from kanvit import ops
    torch.manual_seed(3)
    q, k, v = (torch.randn(2, 3, n, d) for _ in range(3))
    do = torch.randn(2, 3, n, d)
    r = ko.bf16_round
    qd, kd, vd = (t.double() for t in (q, k, v))
    s = (r(qd) @ r(kd).transpose(-1, -2)) * d ** -0.5
    p = torch.exp(s - s.amax(dim=-1, keepdim=True))
    o_tight = (r(p) @ r(vd)) / p.sum(dim=-1, keepdim=True)
    qe, ke, ve = (t.double().requires_grad_(True) for t in (q, k, v))
    o_exact, _ = ko.attention_reference(qe, ke, ve)
    o_exact.backward(do.double())
    qg, kg, vg = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o = ops.attention(qg, kg, vg)
    o.backward(do.to(DEV))
    assert maxrel(o, o_tight) < TIGHT, maxrel(o, o_tight)
    assert 1e-5 < fro(o, o_exact.detach()) < LOOSE
    for name, got, ref in (("dq", qg.grad, qe.grad), ("dk", kg.grad, ke.grad), ("dv", vg.grad, ve.grad)):
        assert fro(got, ref) < 1.5 * LOOSE, (name, fro(got, ref))       # three chained bf16 products (S, dP, dS.K)


@pytest.mark.parametrize("t", ["vanilla", "cheby", "fast", "efficientkan", "sine"])
def test_msa_bf16_against_reference_fixture(t):
    """The whole MSA block (grouped q|k|v + attention) in bf16 mode against the imported reference's fp32 tensors."""
    from attention import MSA
    blob = load_npz("msa197.npz")
    p = t + "."
    msa = MSA(128, 2, type=t)
    msa.load_state_dict(state_dict_from(blob, p))
    msa = msa.to(DEV)
    x = T(blob[p + "x"]).to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = msa(x)
    (y * T(blob[p + "wgt"]).to(DEV)).sum().backward()
    assert 1e-5 < fro(y, T(blob[p + "y"])) < LOOSE, fro(y, T(blob[p + "y"]))
    assert fro(x.grad, T(blob[p + "grad_x"])) < 2 * LOOSE, fro(x.grad, T(blob[p + "grad_x"]))
    got = {k: v.grad for k, v in msa.named_parameters() if v.grad is not None}
    for k, g in grads_from(blob, p).items():
        if float(g.abs().max()) > 1e-4:            # key-bias gradients are mathematically zero: nothing to compare
            assert fro(got[k], g) < 2 * LOOSE, (k, fro(got[k], g))


def test_vits_fastkan_block_bf16_config2():
    """BASELINE configs[2]: 224x224 patch-16 FastKAN ViT-S under bf16 autocast (stock GEMMs in bf16, kanvit kernels on the
    bf16 matrix cores), one full-geometry block, against oracle.vit_forward in float64."""
    from model import VisionTransformer
    torch.manual_seed(5)
    m = VisionTransformer((3, 224, 224), n_patches=14, n_blocks=1, d_hidden=384, n_heads=6, out_d=100, type="fast")
    x = torch.randn(2, 3, 224, 224)
    labels = torch.tensor([3, 71])
    sd = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in m.state_dict().items()}
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k) and v.is_floating_point()) for k, v in sd.items()}
    ref = ko.vit_forward(params, x.double(), 14, 6, "fast")
    ref_loss = torch.nn.functional.cross_entropy(ref, labels)
    ref_loss.backward()
    m = m.to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x.to(DEV))
        loss = torch.nn.functional.cross_entropy(logits.float(), labels.to(DEV))
    loss.backward()
    assert fro(logits.float(), ref) < 2 * LOOSE, fro(logits.float(), ref)
    assert abs(float(loss) - float(ref_loss)) < 2e-2
    worst = max((fro(p.grad, params[k].grad), k) for k, p in m.named_parameters() if float(params[k].grad.abs().max()) > 1e-6)
    assert worst[0] < 5 * LOOSE, worst                                     # gradients through LN + softmax + two bf16 GEMM stacks
