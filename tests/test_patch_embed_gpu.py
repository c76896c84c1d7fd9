"""patchify + patch-embedding KAN layer + class token + position embedding in one kernel launch
(kanvit_patch_embed_fwd; kanvit_patch_embed_fwd_ws on the bf16 matrix cores) and its weight gradient gathering x rows from
the images and dY rows from the token-sequence gradient (kanvit_patch_embed_bwd_weight).  The fused launch performs the same fp32 operations in the same order as the three-step path
(gather is pure addressing; the epilogue adds bias, then pos), so outputs must be BITWISE equal to it whenever both run
the same kernel instantiation (patch width a multiple of the 8-feature chunk), and it is checked
against the float64 oracle's VisionTransformer prologue (oracle.patchify / positional_embeddings, pinned to the reference
by tests/test_oracle_golden.py) as well."""
import pytest
import torch

from oracle import kan_oracle as ko
from tests._util import max_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mode():
    return bool(torch.is_autocast_enabled("cuda"))


def _tokens(m, images, fused):
    m._fused_embed = None if fused else {_mode(): False}
    out = m._embed_fused(images)
    if not fused:
        assert out is None
        patches = m.patchify(images, m.n_patches)
        b, p, _ = patches.shape
        tok = m.linear_mapper(patches).reshape(b, p, m.d_hidden)
        out = torch.cat((m.v_class.unsqueeze(0).expand(b, -1, -1), tok), dim=1) + m.pos_embeddings[: p + 1]
    return out


def _gathered_backward(m, chw, npatch, d, b, bf16=False):
    """True when the backward of this launch runs kanvit_patch_embed_bwd_weight (asked of the library, as ops does)."""
    import ctypes as C
    from kanvit import _lib, ops
    cfg = m.linear_mapper.kan_cfg()
    if bf16:
        from dataclasses import replace
        cfg = replace(cfg, flags=cfg.flags | _lib.FLAG_BF16_MFMA)
    _, bp, _ = m.linear_mapper.kan_pack()
    desc = ops._desc(cfg, b * npatch * npatch, cfg.I, cfg.I, cfg.O, 0 if bp is None else bp.shape[-1])
    pd = _lib.PatchDesc(chw[0], chw[1], chw[2], npatch, 1, 0)
    return bool(_lib.lib().kanvit_patch_embed_bwd_weight_ok(C.byref(desc), C.byref(pd)))


@pytest.mark.parametrize("t", ["cheby", "efficientkan", "sine"])      # (FourierKAN at grid 28: 112 weight rows per 2-feature chunk exceed the fp32 forward's staging registers)
@pytest.mark.parametrize("geom", [((3, 32, 32), 4, 64, 5), ((3, 224, 224), 14, 768, 2), ((1, 64, 64), 4, 128, 3), ((1, 28, 28), 7, 64, 6),
                                  ((3, 32, 32), 4, 64, 21), ((1, 64, 64), 4, 128, 37), ((3, 32, 64), 4, 64, 21)])      # last: non-square images
def test_fused_patch_embedding_equals_three_step_path_and_oracle(t, geom):
    from model import VisionTransformer
    chw, npatch, d, b = geom
    torch.manual_seed(1)
    m = VisionTransformer(chw, n_patches=npatch, n_blocks=1, d_hidden=d, n_heads=2, out_d=10, type=t).to(DEV)
    x = torch.randn(b, *chw, device=DEV)
    wgt = torch.randn(b, npatch * npatch + 1, d, device=DEV)
    res = []
    for fused in (True, False):
        m.zero_grad()
        out = _tokens(m, x, fused)
        assert m._fused_embed[False] is (True if fused else False)
        (out * wgt).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert set(res[0][1]) == set(res[1][1])
    if b * npatch * npatch >= 256 and (chw[0] * (chw[1] // npatch) * (chw[2] // npatch)) % 32 == 0:
        # the fused backward of these launches gathered (no patch matrix) -- except efficient-KAN, whose weight-gradient kernels
        # have no register left for the row walker (kan_bwd_weight_reg.hip::kv_bwd_weight_reg_pg_ok)
        assert _gathered_backward(m, chw, npatch, d, b) == (t != "efficientkan")
    if (chw[2] // npatch) % 8 == 0:
        # both paths run the same kernel instantiation (8-feature chunks): same operations in the same order -> bitwise equal
        assert torch.equal(res[0][0], res[1][0])
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], res[1][1][k]), k      # the backward runs the same kernels on the same rows
    else:
        # narrow patches: the gather needs a chunk that divides the patch width, i.e. another k order of the same fp32 sums
        assert max_err(res[0][0], res[1][0]) < 2e-6 * max(1.0, float(res[1][0].abs().max()))
        for k in res[0][1]:
            assert rel_err(res[0][1][k], res[1][1][k]) < 1e-5, k
    # float64 oracle of the prologue (model.py:144-152)
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in m.state_dict().items()}
    patches = ko.patchify(x.cpu().double(), npatch)
    tok = ko.layer_forward(sd, "linear_mapper.", patches).reshape(b, npatch * npatch, d)
    ref = torch.cat([sd["v_class"].unsqueeze(0).expand(b, -1, -1), tok], dim=1) + ko.positional_embeddings(npatch * npatch + 1, d).double()
    assert max_err(res[0][0].cpu(), ref) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_unsupported_geometry_falls_back_once():
    """3-pixel-wide patches cannot hold a (2-pixel) feature chunk of the register kernel, and a 96-wide output is not a whole
    number of column tiles -> the C entry point reports EINVAL, the module remembers it and runs the three-step path (the
    same HIP layer kernel behind patchify; nothing is replaced by torch or the CPU)."""
    from model import VisionTransformer
    torch.manual_seed(0)
    for chw, npatch, d in (((1, 21, 21), 7, 64), ((1, 64, 64), 4, 96)):
        m = VisionTransformer(chw, npatch, 1, d, 2, 10, type="cheby").to(DEV)
        y = m(torch.rand(4, *chw, device=DEV))
        assert m._fused_embed == {False: False} and torch.isfinite(y).all()
    m2 = VisionTransformer((3, 32, 32), 4, 1, 64, 2, 10, type="cheby").to(DEV)
    m2(torch.randn(2, 3, 32, 32, device=DEV))
    assert m2._fused_embed == {False: True}
    with torch.autocast("cuda", dtype=torch.bfloat16):               # 8-pixel patches cannot hold the bf16 kernel's 16-feature chunk:
        m2(torch.randn(2, 3, 32, 32, device=DEV))                    # decided per arithmetic mode, the fp32 decision stands
    assert m2._fused_embed == {False: True, True: False}
    m2(torch.randn(2, 3, 32, 32, device=DEV))


@pytest.mark.parametrize("t", ["cheby", "efficientkan", "sine", "fourier"])
@pytest.mark.parametrize("geom", [((3, 224, 224), 14, 768, 2), ((3, 64, 64), 4, 128, 21), ((1, 32, 32), 2, 256, 3)])
def test_fused_patch_embedding_bf16_mode(t, geom):
    """bf16 autocast: the fused launch (gather prologue, pos / cls epilogue) on the bf16 matrix cores against the three-step
    path through the same bf16 kernel on a patch matrix -- same roundings, same operation order, so bitwise equal forward and
    backward -- and loosely against the float64 oracle (the bf16 error bound of tests/test_bf16_oracle_gpu.py)."""
    from model import VisionTransformer
    chw, npatch, d, b = geom
    torch.manual_seed(2)
    m = VisionTransformer(chw, n_patches=npatch, n_blocks=1, d_hidden=d, n_heads=2, out_d=10, type=t).to(DEV)
    x = torch.randn(b, *chw, device=DEV)
    wgt = torch.randn(b, npatch * npatch + 1, d, device=DEV)
    res = []
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for fused in (True, False):
            m.zero_grad()
            out = _tokens(m, x, fused)
            assert m._fused_embed[True] is (True if fused else False)
            assert out.dtype == torch.float32
            (out.float() * wgt).sum().backward()
            res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert set(res[0][1]) == set(res[1][1])
    assert torch.equal(res[0][0], res[1][0])
    assert not _gathered_backward(m, chw, npatch, d, b, bf16=True)      # bf16 mode: fused gather forward, patch-matrix weight gradient
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in m.state_dict().items()}
    patches = ko.patchify(x.cpu().double(), npatch)
    tok = ko.layer_forward(sd, "linear_mapper.", patches).reshape(b, npatch * npatch, d)
    ref = torch.cat([sd["v_class"].unsqueeze(0).expand(b, -1, -1), tok], dim=1) + ko.positional_embeddings(npatch * npatch + 1, d).double()
    assert rel_err(res[0][0].cpu(), ref) < 1e-2
    # and the fp32 launch of the same model really differs (the bf16 kernel ran)
    assert not torch.equal(_tokens(m, x, t != "fourier"), res[0][0])       # (FourierKAN's fp32 forward is not fused, see above)


def test_patch_embed_bwd_weight_entry_point_rejects_what_it_does_not_cover():
    import ctypes as C
    from kanvit import _lib, ops
    L = _lib.lib()
    pd = _lib.PatchDesc(3, 224, 224, 14, 1, 0)
    cheby = ops.LayerCfg(_lib.CHEBY, 768, 768, 5)
    d = ops._desc(cheby, 2 * 196, 768, 768, 768, 0)
    assert L.kanvit_patch_embed_bwd_weight_ok(C.byref(d), C.byref(pd)) == 1
    assert L.kanvit_patch_embed_bwd_weight_workspace(C.byref(d), C.byref(pd)) > 0
    d_small = ops._desc(cheby, 196, 768, 768, 768, 0)                   # fewer than 256 rows: the general kernel's territory
    assert L.kanvit_patch_embed_bwd_weight_ok(C.byref(d_small), C.byref(pd)) == 0
    d_bad = ops._desc(cheby, 2 * 196 + 1, 768, 768, 768, 0)             # not a whole number of images
    assert L.kanvit_patch_embed_bwd_weight_ok(C.byref(d_bad), C.byref(pd)) == 0
    x = torch.randn(2, 3, 224, 224, device=DEV)
    dy = torch.randn(2, 197, 768, device=DEV)
    dw = torch.empty(1, 768 * 5, 768, device=DEV)
    rc = L.kanvit_patch_embed_bwd_weight(C.byref(d_small), C.byref(pd), ops._ptr(x), None, ops._ptr(dy), ops._ptr(dw), None, 0, None)
    assert rc != 0 and b"not covered" in L.kanvit_last_error()
    rc = L.kanvit_patch_embed_bwd_weight(C.byref(d), C.byref(pd), ops._ptr(x), None, ops._ptr(dy), ops._ptr(dw), None, 0, None)
    assert rc != 0 and b"workspace" in L.kanvit_last_error()


@pytest.mark.parametrize("pre", [0, 1])
@pytest.mark.parametrize("bf16", [False, True])
def test_patch_embed_bwd_weight_through_the_c_abi(pre, bf16):
    """kanvit_patch_embed_bwd_weight called directly, with and without class-token rows in dY (prepend_rows = 0 is not reachable
    through the model), on a batch whose slabs end inside an image: bitwise equal to kanvit_layer_bwd_weight on the patch matrix."""
    import ctypes as C
    from kanvit import _lib, ops
    L = _lib.lib()
    torch.manual_seed(4 + pre)
    B, Cc, H, W, n, O = 23, 3, 64, 32, 4, 96
    P, I = n * n, Cc * (H // n) * (W // n)
    cfg = ops.LayerCfg(_lib.CHEBY, I, O, 5, flags=_lib.FLAG_BF16_MFMA if bf16 else 0)
    images = torch.randn(B, Cc, H, W, device=DEV)
    dy = torch.randn(B, P + pre, O, device=DEV)
    d = ops._desc(cfg, B * P, I, I, O, 0)
    pd = _lib.PatchDesc(Cc, H, W, n, pre, 0)
    assert L.kanvit_patch_embed_bwd_weight_ok(C.byref(d), C.byref(pd)) == (0 if bf16 else 1)
    if bf16:                             # exact fp32 only (the bf16 kernels' short MFMA phases cannot hide the row walker): refused, loudly
        dw = torch.empty(1, I * 5, O, device=DEV)
        rc = L.kanvit_patch_embed_bwd_weight(C.byref(d), C.byref(pd), ops._ptr(images), None, ops._ptr(dy), ops._ptr(dw), None, 0, None)
        assert rc != 0 and b"not covered" in L.kanvit_last_error()
        return
    nb = int(L.kanvit_patch_embed_bwd_weight_workspace(C.byref(d), C.byref(pd)))
    ws = torch.empty(max(nb // 4, 1), device=DEV)
    dw = torch.full((1, I * 5, O), float("nan"), device=DEV)
    ops.check(L.kanvit_patch_embed_bwd_weight(C.byref(d), C.byref(pd), ops._ptr(images), None, ops._ptr(dy), ops._ptr(dw), ops._ptr(ws), C.c_size_t(nb), None), "bwd_weight")
    x = ops.patchify(images, n).reshape(-1, I).contiguous()
    dyt = dy[:, pre:, :].reshape(-1, O).contiguous()
    nb2 = int(L.kanvit_layer_bwd_weight_workspace(C.byref(d)))
    ws2 = torch.empty(max(nb2 // 4, 1), device=DEV)
    dw2 = torch.empty_like(dw)
    ops.check(L.kanvit_layer_bwd_weight(C.byref(d), ops._ptr(x), None, None, ops._ptr(dyt), ops._ptr(dw2), ops._ptr(ws2), C.c_size_t(nb2), None), "layer_bwd_weight")
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all() and torch.equal(dw, dw2)        # (the patch-matrix kernel itself is checked against the fp64 oracle in test_layers_gpu.py)
