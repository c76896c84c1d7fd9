"""The bf16 matrix-core mode (KANVIT_FLAG_BF16_MFMA, set under torch.autocast(bfloat16); BASELINE configs[2], [4]) checked
against the ORACLE -- never against this repo's own fp32 kernels.

 * TIGHT: the float64 oracle with the operands of every contraction rounded to bf16 at exactly the points where the kernels
   round them (oracle.operand_rounding: basis values Phi(x) and packed coefficients forward; dY, W^T, Phi in the backward
   products; q, k, v, the probabilities, dO and dS for attention).  What is left is fp32-vs-fp64 accumulation and the rare
   flip of a value that sits on a bf16 rounding boundary, so forward AND every gradient must agree to 2e-3 of the largest
   entry -- an order of magnitude below bf16 noise, and meaningful also for gradients that are cancelling sums (where a
   comparison with unrounded arithmetic is not): a mis-packed weight image or a wrong k-permutation cannot hide under it.
 * LOOSE: against the UNROUNDED float64 oracle / the imported reference's fp32 tensors: <= 1e-2 in the Frobenius norm for
   forward outputs and for gradients that are not cancellation dominated ("bf16 configs should be
   judged against the fp32 reference with a stated tolerance ~1e-2 rel"), and strictly > 0 (the bf16 kernel really ran)."""
import pytest
import torch

from oracle import kan_oracle as ko
from tests._util import T, grads_from, load_npz, state_dict_from

pytestmark = pytest.mark.gpu
DEV = "cuda"
TIGHT = 2e-3          # max |err| / max |ref| against the bf16-operand oracle (forward and gradients)
LOOSE = 1e-2          # ||err||_F / ||ref||_F against the unrounded oracle / reference fixtures


def fro(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def maxrel(a, b, floor=1e-30):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(floor))


def _params64(module):
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in module.state_dict().items()}
    return {k: v.clone().requires_grad_(not ko.is_buffer_key(k) and v.is_floating_point()) for k, v in sd.items()}


def _oracle_qkv(msa, x2d, h, w, rounded):
    params = _params64(msa)
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
            y = run()
    else:
        y = run()
    (y * w.double()).sum().backward()
    return y.detach(), xd.grad, {k: v.grad for k, v in params.items() if v.grad is not None}


@pytest.mark.parametrize("fam", ["vanilla", "cheby", "efficientkan", "fast", "sine"])
@pytest.mark.parametrize("geom", [(4, 197, 128, 2), (22, 197, 384, 6)])      # second: M = 4334 >= 4096 rows -> W-stationary forward
def test_qkv_bf16_against_rounded_and_exact_oracle(fam, geom):
    from attention import MSA
    from kanvit import grouped
    b, n, d, h = geom
    torch.manual_seed(77 + d)
    msa = MSA(d, h, type=fam)
    x = torch.randn(b * n, d)
    w = torch.randn(b * n, 3 * d)
    y_t, gx_t, gp_t = _oracle_qkv(msa, x, h, w, rounded=True)
    y_e, gx_e, gp_e = _oracle_qkv(msa, x, h, w, rounded=False)
    msa = msa.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, xg)
    assert y.dtype == torch.float32
    (y * w.to(DEV)).sum().backward()
    got = {k: v.grad for k, v in msa.named_parameters() if v.grad is not None}
    assert set(got) == set(gp_t)
    # tight: same rounding points -> same numbers
    assert maxrel(y, y_t) < TIGHT, (fam, geom, maxrel(y, y_t))
    # SineKAN's input gradient (and d freq, which the same kernel forms) stays on the EXACT fp32 register kernel under the bf16
    # flag (include/kanvit.h: the flag allows bf16 products, exact ones are always valid; the bf16 tile kernel is slower there):
    # those two must match one of the two specified arithmetics
    exact_ok = fam == "sine"
    err = min(maxrel(xg.grad, gx_t), maxrel(xg.grad, gx_e)) if exact_ok else maxrel(xg.grad, gx_t)
    assert err < TIGHT, (fam, geom, "dx", err)
    for k, g in gp_t.items():
        err = min(maxrel(got[k], g), maxrel(got[k], gp_e[k])) if (exact_ok and k.endswith("freq")) else maxrel(got[k], g)
        assert err < TIGHT, (fam, geom, k, err)
    # loose: bf16 noise against the unrounded oracle (d freq is one cancelling sum over all rows, features and outputs: 3x)
    assert 1e-5 < fro(y, y_e) < LOOSE, (fam, geom, fro(y, y_e))
    assert fro(xg.grad, gx_e) < LOOSE, (fam, geom, "dx", fro(xg.grad, gx_e))
    for k, g in gp_e.items():
        assert fro(got[k], g) < (3 * LOOSE if k.endswith("freq") else LOOSE), (fam, geom, k, fro(got[k], g))


@pytest.mark.parametrize("fam,i,o,big", [("cheby", 768, 768, False), ("efficientkan", 768, 384, False), ("fast", 768, 384, False),
                                         ("sine", 192, 128, True), ("fourier", 192, 128, True)])
def test_patch_embedding_layer_bf16(fam, i, o, big):
    """groups = 1 launches (the patch embedding): wide K, G = 28 bases for sine / fourier."""
    from tests.test_layers_gpu import make_layer
    torch.manual_seed(9)
    layer = make_layer(fam, i, o, big)
    x = torch.randn(392, i) * 0.7
    w = torch.randn(392, o)
    res = {}
    for rounded in (True, False):
        params = _params64(layer)
        xd = x.double().requires_grad_(True)
        if rounded:
            with ko.operand_rounding(ko.bf16_round):
                y = ko.layer_forward(params, "", xd)
        else:
            y = ko.layer_forward(params, "", xd)
        (y * w.double()).sum().backward()
        res[rounded] = (y.detach(), xd.grad, {k: v.grad for k, v in params.items() if v.grad is not None})
    layer = layer.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = layer(xg)
    (y * w.to(DEV)).sum().backward()
    got = {k: p.grad for k, p in layer.named_parameters() if p.grad is not None}
    y_t, gx_t, gp_t = res[True]
    y_e, gx_e, gp_e = res[False]
    assert maxrel(y, y_t) < TIGHT, (fam, maxrel(y, y_t))
    # wide layers (O > 64) have no bf16 input-gradient kernel: the flag falls back to the EXACT fp32 kernel (include/kanvit.h),
    # so dx must match one of the two specified arithmetics -- bf16 operands, or exact
    assert min(maxrel(xg.grad, gx_t), maxrel(xg.grad, gx_e)) < TIGHT, (fam, "dx", maxrel(xg.grad, gx_t), maxrel(xg.grad, gx_e))
    for k, g in gp_t.items():
        # FastKAN's layernorm.* gradients are column sums of du, i.e. products of the INPUT-gradient kernel: same rule as dx
        err = min(maxrel(got[k], g), maxrel(got[k], gp_e[k])) if k.startswith("layernorm.") else maxrel(got[k], g)
        assert err < TIGHT, (fam, k, err)
    assert 1e-5 < fro(y, y_e) < LOOSE, (fam, fro(y, y_e))
    assert fro(xg.grad, gx_e) < LOOSE, (fam, "dx", fro(xg.grad, gx_e))
    for k, g in gp_e.items():
        assert fro(got[k], g) < (3 * LOOSE if k.endswith("freq") else LOOSE), (fam, k, fro(got[k], g))


@pytest.mark.parametrize("n,d", [(197, 64), (50, 32), (17, 16)])
def test_attention_bf16(n, d):
    from kanvit import ops
    torch.manual_seed(3)
    q, k, v = (torch.randn(2, 3, n, d) for _ in range(3))
    do = torch.randn(2, 3, n, d)
    res = {}
    for rounded in (True, False):
        qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
        if rounded:
            with ko.operand_rounding(ko.bf16_round):
                o = ko._attention_core(qd, kd, vd)
        else:
            o = ko._attention_core(qd, kd, vd)
        o.backward(do.double())
        res[rounded] = (o.detach(), qd.grad, kd.grad, vd.grad)
    qg, kg, vg = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o = ops.attention(qg, kg, vg)
    o.backward(do.to(DEV))
    got = (o, qg.grad, kg.grad, vg.grad)
    for name, a, b_t, b_e in zip(("o", "dq", "dk", "dv"), got, res[True], res[False]):
        if n <= 32 and d <= 32:      # the one-wave-per-head kernels are exact fp32 in either mode (the flag allows bf16, never requires it)
            assert maxrel(a, b_e) < 1e-5, (name, maxrel(a, b_e))
            continue
        assert maxrel(a, b_t) < TIGHT, (name, maxrel(a, b_t))
        assert 1e-5 < fro(a, b_e) < 1.5 * LOOSE, (name, fro(a, b_e))           # gradients: three chained bf16 products


@pytest.mark.parametrize("t", ["vanilla", "cheby", "fast", "efficientkan", "sine"])
def test_msa_bf16_headline_geometry(t):
    """The whole MSA block (grouped q|k|v + attention; dh = 64, N = 197) in bf16 mode, with the parameters and input of the
    imported reference's fixture: forward against the reference's fp32 output (loose) and against the bf16-operand oracle
    (tight); gradients against the bf16-operand oracle at COMPOSITE = 2.5e-2 of the block's largest gradient entry.
    Why not TIGHT for the gradients of the composite: softmax is shift invariant, so sum_n dS[q, n] = 0 in exact arithmetic
    and the parts of dq, dk that multiply the (large) row-constant component of k, q -- the key-bias gradients entirely --
    are sums of bf16 ROUNDING ERRORS of dS; a handful of values that round differently in fp32 and fp64 (rounding flips)
    move such a sum by percents.  The kernels themselves are checked at TIGHT above, each on well-conditioned inputs; an
    unrounded comparison of these gradients is off by 5-10 % and would measure conditioning, not the kernels."""
    from attention import MSA
    blob = load_npz("msa197.npz")
    p = t + "."
    msa = MSA(128, 2, type=t)
    msa.load_state_dict(state_dict_from(blob, p))
    params = _params64(msa)
    xd = T(blob[p + "x"]).double().requires_grad_(True)
    with ko.operand_rounding(ko.bf16_round):
        y_t = ko.msa_forward(params, "", xd, 2)
    (y_t * T(blob[p + "wgt"]).double()).sum().backward()
    msa = msa.to(DEV)
    x = T(blob[p + "x"]).to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = msa(x)
    (y * T(blob[p + "wgt"]).to(DEV)).sum().backward()
    assert 1e-5 < fro(y, T(blob[p + "y"])) < LOOSE, fro(y, T(blob[p + "y"]))
    assert maxrel(y, y_t) < TIGHT, maxrel(y, y_t)
    COMPOSITE = 2.5e-2
    assert maxrel(x.grad, xd.grad) < COMPOSITE, maxrel(x.grad, xd.grad)
    got = {k: v.grad for k, v in msa.named_parameters() if v.grad is not None}
    scale = max(float(v.grad.abs().max()) for v in params.values() if v.grad is not None)
    for k, v in params.items():
        if v.grad is not None:
            assert maxrel(got[k], v.grad, floor=scale) < COMPOSITE, (k, maxrel(got[k], v.grad, floor=scale))
            if ".v_mappings." in k:            # the value path has no such cancellation: tight
                assert maxrel(got[k], v.grad) < 2 * TIGHT, (k, maxrel(got[k], v.grad))


def test_vits_fastkan_block_bf16_config2():
    """BASELINE configs[2]: 224x224 patch-16 FastKAN ViT-S under bf16 autocast (stock GEMMs in bf16, kanvit kernels on the
    bf16 matrix cores), one full-geometry block, against oracle.vit_forward in float64 (unrounded: torch's autocast casts of
    the stock ops are not modelled).  Forward <= 2e-2; gradients by direction and size (cosine >= 0.99, norm within 5 %)
    for every tensor with a non-negligible gradient."""
    from model import VisionTransformer
    torch.manual_seed(5)
    m = VisionTransformer((3, 224, 224), n_patches=14, n_blocks=1, d_hidden=384, n_heads=6, out_d=100, type="fast")
    x = torch.randn(2, 3, 224, 224)
    labels = torch.tensor([3, 71])
    params = _params64(m)
    ref = ko.vit_forward(params, x.double(), 14, 6, "fast")
    ref_loss = torch.nn.functional.cross_entropy(ref, labels)
    ref_loss.backward()
    m = m.to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x.to(DEV))
        loss = torch.nn.functional.cross_entropy(logits.float(), labels.to(DEV))
    loss.backward()
    assert fro(logits.float(), ref.detach()) < 2 * LOOSE, fro(logits.float(), ref.detach())
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 2e-2
    gmax = max(float(v.grad.norm()) for v in params.values() if v.grad is not None)
    report = []
    for k, p in m.named_parameters():
        g = params[k].grad
        if g is None or p.grad is None or float(g.norm()) < 1e-3 * gmax:
            continue
        a, b = p.grad.detach().double().cpu().flatten(), g.flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        report.append((cos, float(a.norm() / b.norm()), k))
    worst = min(report)
    assert worst[0] > 0.99, worst
    assert all(0.95 < r < 1.05 for _, r, _ in report), [x for x in report if not 0.95 < x[1] < 1.05]


def test_vitb_mixed_sine_fourier_model_bf16_config4():
    """BASELINE configs[4] in its own dtype: the two-block type="sine,fourier" ViT-B model (see test_headline_parity_gpu.py's
    fp32 twin) under bf16 autocast, against oracle.vit_forward in float64 at the loose bounds of the ViT-S FastKAN test above
    (forward <= 2e-2; gradients by direction and size)."""
    from model import VisionTransformer
    torch.manual_seed(11)
    m = VisionTransformer((3, 224, 224), 14, 2, 768, 12, 100, type="sine,fourier")
    x = torch.randn(2, 3, 224, 224)
    labels = torch.tensor([17, 4])
    params = _params64(m)
    ref = ko.vit_forward(params, x.double(), 14, 12, "sine")
    ref_loss = torch.nn.functional.cross_entropy(ref, labels)
    ref_loss.backward()
    m = m.to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x.to(DEV))
        loss = torch.nn.functional.cross_entropy(logits.float(), labels.to(DEV))
    loss.backward()
    assert 1e-6 < fro(logits.float(), ref.detach()) < 2 * LOOSE, fro(logits.float(), ref.detach())
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 2e-2
    gmax = max(float(v.grad.norm()) for v in params.values() if v.grad is not None)
    report = []
    for k, p in m.named_parameters():
        g = params[k].grad
        if g is None or p.grad is None or float(g.norm()) < 1e-3 * gmax:
            continue
        a, b = p.grad.detach().double().cpu().flatten(), g.flatten()
        report.append((float(torch.dot(a, b) / (a.norm() * b.norm())), float(a.norm() / b.norm()), k))
    worst = min(report)
    assert worst[0] > 0.99, worst
    assert all(0.95 < r < 1.05 for _, r, _ in report), [x for x in report if not 0.95 < x[1] < 1.05]
