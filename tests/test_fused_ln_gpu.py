"""FastKAN with its LayerNorm formed inside the kernels (KANVIT_FLAG_FUSED_LN, SURVEY.md section 8(f) "fuse the FastKAN
LayerNorm"): models/fastkan.py:66-76 is y = spline(rbf(layernorm(x))) + base(silu(x)).

 (1) the fused launch against the float64 oracle -- forward, dx, every parameter gradient INCLUDING layernorm.weight / .bias
     (randomised, so a wrong gamma/beta index shows) -- for a single layer at the ViT-S patch-embedding shape and for the
     grouped q|k|v launch at the headline head geometry;
 (2) the fused launch against the unfused one (KANVIT_NO_FUSED_LN=1): same numbers to fp32 rounding, so the two routes stay
     interchangeable, and the normalised tensor really is not materialised (peak-memory check);
 (3) shapes the register kernels do not cover keep the separate LayerNorm and still match the oracle;
 (4) misuse of the flag through the C ABI is refused, not mis-executed.
Tolerance: BASELINE.json's 1e-4 normwise for fp32, the bf16 bounds of tests/test_bf16_oracle_gpu.py under autocast."""
import ctypes as C

import pytest
import torch

from oracle import kan_oracle as ko
from tests._util import close, max_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _randomise_ln(mods):
    g = torch.Generator().manual_seed(77)
    for m in mods:
        with torch.no_grad():
            m.layernorm.weight.copy_(1.0 + 0.3 * torch.randn(m.layernorm.weight.shape, generator=g))
            m.layernorm.bias.copy_(0.2 * torch.randn(m.layernorm.bias.shape, generator=g))


def _oracle_layer(layer, x, w):
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in layer.state_dict().items()}
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k) and v.is_floating_point()) for k, v in sd.items()}
    xd = x.double().clone().requires_grad_(True)
    y = ko.layer_forward(params, "", xd)
    (y * w.double()).sum().backward()
    return y.detach(), xd.grad, {k: v.grad for k, v in params.items() if v.grad is not None}


@pytest.mark.parametrize("shape", [(392, 768, 384), (512, 64, 64), (300, 96, 128)])
def test_single_layer_fused_ln_vs_fp64_oracle(shape):
    from kanvit import ops
    from models.fastkan import FastKANLayer
    m_rows, i, o = shape
    torch.manual_seed(3)
    layer = FastKANLayer(i, o)
    _randomise_ln([layer])
    x = 1.5 * torch.randn(m_rows, i) + 0.3
    w = torch.randn(m_rows, o)
    yo, gxo, gpo = _oracle_layer(layer, x, w)
    layer = layer.to(DEV)
    assert ops.ln_fusable(layer.kan_cfg(), m_rows), "this shape is meant to take the fused route"
    xg = x.to(DEV).requires_grad_(True)
    y = layer(xg)
    (y * w.to(DEV)).sum().backward()
    assert max_err(y.cpu(), yo) < 2e-5 * max(1.0, float(yo.abs().max()))
    assert rel_err(xg.grad.cpu(), gxo) < TOL, rel_err(xg.grad.cpu(), gxo)
    got = {k: v.grad.cpu() for k, v in layer.named_parameters() if v.grad is not None}
    assert set(got) == set(gpo), set(got) ^ set(gpo)
    for k, g in gpo.items():
        assert rel_err(got[k], g) < TOL, (k, rel_err(got[k], g))


def _msa_pair(d, h):
    from attention import MSA
    torch.manual_seed(11)
    msa = MSA(d, h, type="fast")
    _randomise_ln(list(msa.q_mappings) + list(msa.k_mappings) + list(msa.v_mappings))
    return msa


@pytest.mark.parametrize("geom", [(2, 197, 128, 2), (2, 197, 384, 6)])
def test_grouped_qkv_fused_equals_unfused(geom, monkeypatch):
    """Same launch with and without the fusion: forward and every gradient agree to fp32 rounding (the two routes differ only
    in where (x - mean) * rstd * gamma + beta is evaluated)."""
    from kanvit import _lib, grouped, ops
    b, n, d, h = geom
    msa = _msa_pair(d, h).to(DEV)
    x = torch.randn(b * n, d, device=DEV)
    w = torch.randn(b * n, 3 * d, device=DEV)

    def run():
        for p in msa.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        y = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, xg)
        (y * w).sum().backward()
        return y.detach(), xg.grad, {k: v.grad.clone() for k, v in msa.named_parameters() if v.grad is not None}

    cfg = ops.LayerCfg(family=ops.RBF, I=d // h, O=d // h, G=8, groups=3 * h, x_group_mod=h, has_base=1, rbf_inv_h=1.75,
                       flags=_lib.FLAG_UNIFORM_KNOTS)
    assert ops.ln_fusable(cfg, b * n)
    y1, gx1, gp1 = run()
    monkeypatch.setenv("KANVIT_NO_FUSED_LN", "1")
    _lib.reload_config()
    try:
        assert not ops.ln_fusable(cfg, b * n)
        y0, gx0, gp0 = run()
    finally:
        monkeypatch.delenv("KANVIT_NO_FUSED_LN")
        _lib.reload_config()
    assert max_err(y1, y0) < 1e-5 * max(1.0, float(y0.abs().max()))
    assert rel_err(gx1, gx0) < 2e-5
    assert set(gp1) == set(gp0)
    for k in gp0:
        assert close(gp1[k], gp0[k], rtol=2e-5, atol=1e-6), (k, rel_err(gp1[k], gp0[k]))


def test_fused_ln_does_not_materialise_u():
    """The fused forward allocates y and the [M, H, 2] statistics only: no [M, 3*d] normalised tensor."""
    from kanvit import grouped
    b, n, d, h = 8, 197, 384, 6
    msa = _msa_pair(d, h).to(DEV)
    x = torch.randn(b * n, d, device=DEV)
    with torch.no_grad():
        grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)        # warm (packing caches, module load)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        y = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x)
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
    u_bytes = b * n * 3 * d * 4
    assert peak < y.numel() * 4 + u_bytes // 2, (peak, y.numel() * 4, u_bytes)


@pytest.mark.parametrize("shape", [(64, 64, 64), (300, 50, 30)])
def test_uncovered_shapes_keep_the_separate_layernorm(shape):
    from kanvit import ops
    from models.fastkan import FastKANLayer
    m_rows, i, o = shape
    torch.manual_seed(4)
    layer = FastKANLayer(i, o)
    _randomise_ln([layer])
    x = torch.randn(m_rows, i)
    w = torch.randn(m_rows, o)
    yo, gxo, gpo = _oracle_layer(layer, x, w)
    layer = layer.to(DEV)
    assert not ops.ln_fusable(layer.kan_cfg(), m_rows)
    xg = x.to(DEV).requires_grad_(True)
    y = layer(xg)
    (y * w.to(DEV)).sum().backward()
    assert max_err(y.cpu(), yo) < 2e-5 * max(1.0, float(yo.abs().max()))
    assert rel_err(xg.grad.cpu(), gxo) < TOL
    for k, g in gpo.items():
        assert rel_err(dict(layer.named_parameters())[k].grad.cpu(), g) < TOL, k


def test_time_benchmark_skips_the_layernorm():
    """forward(x, time_benchmark=True) is rbf(x) without the LayerNorm (models/fastkan.py:67-70): the fused route must not apply it."""
    from models.fastkan import FastKANLayer
    torch.manual_seed(9)
    layer = FastKANLayer(64, 64)
    _randomise_ln([layer])
    x = torch.randn(300, 64)
    sd = {k: v.detach().double() for k, v in layer.state_dict().items()}
    want = ko.fastkan_forward(x.double(), sd["layernorm.weight"], sd["layernorm.bias"], sd["rbf.grid"], sd["spline_linear.weight"],
                              sd["base_linear.weight"], sd["base_linear.bias"], use_layernorm=False)
    layer = layer.to(DEV)
    y_skip = layer(x.to(DEV), time_benchmark=True).cpu()
    y_ln = layer(x.to(DEV)).cpu()
    assert float((y_skip - y_ln).abs().max()) > 1e-2          # gamma / beta are far from identity: the two must differ
    assert max_err(y_skip, want) < 2e-5 * max(1.0, float(want.abs().max()))


def test_bf16_fused_ln_against_rounded_oracle():
    """bf16 MFMA mode with the fusion: against the float64 oracle with bf16 operand rounding (the kernels round the basis values
    and the weights to bf16 and accumulate in fp32), as tests/test_bf16_oracle_gpu.py does for the unfused route."""
    from kanvit import grouped
    b, n, d, h = 2, 197, 384, 6
    msa = _msa_pair(d, h)
    x = torch.randn(b * n, d)
    w = torch.randn(b * n, 3 * d)
    sd = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in msa.state_dict().items()}
    params = {k: v.clone().requires_grad_(not ko.is_buffer_key(k) and v.is_floating_point()) for k, v in sd.items()}
    xd = x.double().clone().requires_grad_(True)
    dh = d // h
    with ko.operand_rounding(ko.bf16_round):
        cols = [ko.layer_forward(params, f"{nm}_mappings.{hh}.", xd[:, hh * dh:(hh + 1) * dh]) for nm in ("q", "k", "v") for hh in range(h)]
        yo = torch.cat(cols, dim=1)
        (yo * w.double()).sum().backward()
    msa = msa.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, xg)
    (y.float() * w.to(DEV)).sum().backward()
    assert rel_err(y.float().cpu(), yo.detach()) < 2e-3
    assert rel_err(xg.grad.cpu(), xd.grad) < 1e-2
    for k, p in msa.named_parameters():
        if p.grad is None or params[k].grad is None:
            continue
        assert rel_err(p.grad.cpu(), params[k].grad) < 1e-2, (k, rel_err(p.grad.cpu(), params[k].grad))


def test_flag_misuse_is_refused():
    from kanvit import _lib
    L = _lib.lib()
    x = torch.randn(512, 64, device=DEV)
    w = torch.randn(1, 64 * 9, 64, device=DEV)
    bp = torch.zeros(1, 8 + 128, device=DEV)
    y = torch.empty(512, 64, device=DEV)
    d = _lib.LayerDesc(family=_lib.RBF, groups=1, x_group_mod=1, I=64, O=64, G=8, has_base=1, rbf_inv_h=1.75,
                       flags=_lib.FLAG_FUSED_LN | _lib.FLAG_UNIFORM_KNOTS, M=512, ldx=64, ldu=64, ldy=64, bparam_stride=136, ln_eps=1e-5)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = L.kanvit_layer_fwd(C.byref(d), p(x), None, p(w), p(bp), None, p(y), None, 0, None)      # no statistics buffer
    assert rc == -22 and b"statistics" in L.kanvit_last_error()
    pd = _lib.PatchDesc(1, 64, 64, 8, 1, 0)
    rc = L.kanvit_patch_embed_fwd(C.byref(d), C.byref(pd), p(x), p(w), p(bp), None, p(y), p(y), p(y), None)
    assert rc == -22


def test_non_default_rbf_grid_takes_the_general_kernels():
    """The register kernels evaluate the eight Gaussians by a recurrence that assumes FastKAN's own grid (centres c0 + j*h, h =
    denominator).  A layer built with another denominator must not be vouched uniform: it runs the LDS-tile kernels (direct
    exp per centre, separate LayerNorm) and still matches the oracle."""
    from kanvit import _lib, ops
    from models.fastkan import FastKANLayer
    torch.manual_seed(12)
    layer = FastKANLayer(64, 64)
    layer.rbf.denominator = 0.9                       # spacing is 4/7: no longer the default layout
    x = torch.randn(300, 64)
    w = torch.randn(300, 64)
    sd = {k: v.detach().double() for k, v in layer.state_dict().items()}
    xd = x.double().requires_grad_(True)
    want = ko.fastkan_forward(xd, sd["layernorm.weight"], sd["layernorm.bias"], sd["rbf.grid"], sd["spline_linear.weight"],
                              sd["base_linear.weight"], sd["base_linear.bias"], denominator=0.9)
    (want * w.double()).sum().backward()
    layer = layer.to(DEV)
    assert not (layer.kan_cfg().flags & _lib.FLAG_UNIFORM_KNOTS) and not ops.ln_fusable(layer.kan_cfg(), 300)
    xg = x.to(DEV).requires_grad_(True)
    y = layer(xg)
    (y * w.to(DEV)).sum().backward()
    assert max_err(y.cpu(), want) < 2e-5 * max(1.0, float(want.abs().max()))
    assert rel_err(xg.grad.cpu(), xd.grad) < TOL
    default = FastKANLayer(64, 64).to(DEV)
    assert default.kan_cfg().flags & _lib.FLAG_UNIFORM_KNOTS and ops.ln_fusable(default.kan_cfg(), 300)


@pytest.mark.parametrize("scale,shift", [(1.0, 0.0), (8.0, 0.0), (3.0, 6.0), (3.0, -9.0), (40.0, 0.0)])
def test_rbf_recurrence_over_the_whole_argument_range(scale, shift):
    """The register kernels form FastKAN's eight Gaussians from two exp anchors and a recurrence (kan_basis.h::kv_rbf8).  Drive the
    normalised input far outside the grid ([-2, 2]): u = scale * xhat + shift reaches |u| ~ 150 at scale 40, where every basis
    value underflows, and sits on the grid's edges for the shifted cases.  Forward and all gradients against the fp64 oracle."""
    from kanvit import ops
    from models.fastkan import FastKANLayer
    torch.manual_seed(5)
    layer = FastKANLayer(64, 64)
    with torch.no_grad():
        layer.layernorm.weight.fill_(scale)
        layer.layernorm.bias.fill_(shift)
    x = torch.randn(512, 64) * 2.0
    w = torch.randn(512, 64)
    yo, gxo, gpo = _oracle_layer(layer, x, w)
    layer = layer.to(DEV)
    assert ops.ln_fusable(layer.kan_cfg(), 512)
    xg = x.to(DEV).requires_grad_(True)
    y = layer(xg)
    (y * w.to(DEV)).sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(xg.grad).all()
    assert max_err(y.cpu(), yo) < 2e-5 * max(1.0, float(yo.abs().max()))
    assert close(xg.grad.cpu(), gxo, rtol=TOL, atol=1e-6), rel_err(xg.grad.cpu(), gxo)
    for k, g in gpo.items():
        got = dict(layer.named_parameters())[k].grad.cpu()
        assert close(got, g, rtol=TOL, atol=1e-6), (k, rel_err(got, g))


def test_rbf_recurrence_keeps_nan_a_nan():
    from models.fastkan import FastKANLayer
    torch.manual_seed(6)
    layer = FastKANLayer(64, 64).to(DEV)
    x = torch.randn(300, 64, device=DEV)
    x[17, 5] = float("nan")
    y = layer(x)
    assert torch.isnan(y[17]).all()                        # the LayerNorm spreads a NaN over its row; the clamp in kv_rbf8 must not hide it
    assert torch.isfinite(y[:17]).all() and torch.isfinite(y[18:]).all()
