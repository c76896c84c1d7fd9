"""The fused small-geometry feed-forward (csrc/ff_small.hip) against the float64 statement of the
reference's block feed-forward, model.py:25-29,36: nn.Sequential(Linear(d, 4d), ReLU(inplace), Linear(4d, d)) -- forward,
dx and all four parameter gradients, ragged row counts (tiles of 32 rows with a partial last tile), determinism, and that the
TransformerBlock really takes this route at d = 64.  Tolerance 1e-4 normwise (BASELINE.json), observed ~1e-6."""
import pytest
import torch

from tests._util import max_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _ref64(x, lin1, lin2, w):
    xd = x.double().requires_grad_(True)
    p = [t.detach().double().requires_grad_(True) for t in (lin1.weight, lin1.bias, lin2.weight, lin2.bias)]
    y = torch.relu(xd @ p[0].t() + p[1]) @ p[2].t() + p[3]
    (y * w.double()).sum().backward()
    return y.detach(), xd.grad, [t.grad for t in p]


@pytest.mark.parametrize("rows", [1, 31, 32, 50, 6400, 2176, 777, 16384, 40001])
def test_ff_small_against_fp64(rows):
    from kanvit import dense
    torch.manual_seed(rows)
    lin1, lin2 = torch.nn.Linear(64, 256), torch.nn.Linear(256, 64)
    x = torch.randn(rows, 64)
    w = torch.randn(rows, 64)
    yo, gxo, gpo = _ref64(x, lin1, lin2, w)
    lin1, lin2 = lin1.to(DEV), lin2.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    assert dense._ff_small_ok(xg, lin1, lin2)
    y = dense.feed_forward(xg, lin1, lin2)
    (y * w.to(DEV)).sum().backward()
    assert max_err(y.cpu(), yo) < 2e-5 * max(1.0, float(yo.abs().max()))
    assert rel_err(xg.grad.cpu(), gxo) < TOL
    for got, want, name in zip((lin1.weight, lin1.bias, lin2.weight, lin2.bias), gpo, ("w1", "b1", "w2", "b2")):
        assert rel_err(got.grad.cpu(), want) < TOL, (name, rel_err(got.grad.cpu(), want))


def test_ff_small_matches_stock_path_and_is_deterministic(monkeypatch):
    from kanvit import dense
    torch.manual_seed(0)
    lin1, lin2 = torch.nn.Linear(64, 256).to(DEV), torch.nn.Linear(256, 64).to(DEV)
    x = torch.randn(6400, 64, device=DEV)
    w = torch.randn(6400, 64, device=DEV)

    def run():
        for p in (*lin1.parameters(), *lin2.parameters()):
            p.grad = None
        xg = x.clone().requires_grad_(True)
        y = dense.feed_forward(xg, lin1, lin2)
        (y * w).sum().backward()
        return [y.detach(), xg.grad] + [p.grad.clone() for p in (*lin1.parameters(), *lin2.parameters())]

    a, b = run(), run()
    assert all(torch.equal(s, t) for s, t in zip(a, b))          # fixed summation order, no atomics
    from kanvit import _lib
    monkeypatch.setenv("KANVIT_NO_FF_SMALL", "1")
    assert "py_no_ff_small=1" in _lib.reload_config()      # the Python-side switches are read once and echoed like the library's
    try:
        c = run()
    finally:
        monkeypatch.delenv("KANVIT_NO_FF_SMALL")
        assert "py_no_ff_small=0" in _lib.reload_config()
    for s, t in zip(a, c):
        assert rel_err(s, t) < 2e-5


def test_large_or_other_shapes_keep_the_stock_gemms():
    from kanvit import dense
    lin1, lin2 = torch.nn.Linear(64, 256).to(DEV), torch.nn.Linear(256, 64).to(DEV)
    from kanvit import _lib
    assert not dense._ff_small_ok(torch.randn(int(_lib.lib().kanvit_ff_small_max_rows()) + 1, 64, device=DEV), lin1, lin2)
    big1, big2 = torch.nn.Linear(768, 3072).to(DEV), torch.nn.Linear(3072, 768).to(DEV)
    assert not dense._ff_small_ok(torch.randn(64, 768, device=DEV), big1, big2)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not dense._ff_small_ok(torch.randn(64, 64, device=DEV), lin1, lin2)


def test_c_abi_refuses_unsupported_shapes():
    from kanvit import _lib
    L = _lib.lib()
    assert L.kanvit_ff_small_supported(64, 256) == 1 and L.kanvit_ff_small_supported(128, 512) == 0
    assert L.kanvit_ff_small_fwd(64, 128, 512, None, None, None, None, None, None, None) == -22
    assert L.kanvit_ff_small_fwd(10 ** 6, 64, 256, None, None, None, None, None, None, None) == -22
    assert L.kanvit_ff_small_fwd(64, 64, 256, None, None, None, None, None, None, None) == -22 and b"null" in L.kanvit_last_error()


@pytest.mark.parametrize("rows,with_delta", [(50, True), (6400, True), (777, False), (2176, True)])
def test_ln_feed_forward_against_fp64(rows, with_delta):
    """s = x + delta; y = FF(LayerNorm(s)) (model.py:36 without the final add): s, y, the shared gradient of x / delta -- which
    also carries the gradient arriving on s itself --, d gamma, d beta and the feed-forward parameter gradients."""
    from kanvit import dense
    torch.manual_seed(100 + rows)
    norm = torch.nn.LayerNorm(64)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.3 * torch.randn(64))
        norm.bias.copy_(0.2 * torch.randn(64))
    lin1, lin2 = torch.nn.Linear(64, 256), torch.nn.Linear(256, 64)
    b = 1 if rows % 50 else rows // 50
    shape = (b, rows // b, 64)
    x, delta = torch.randn(shape), (torch.randn(shape) if with_delta else None)
    ws, wy = torch.randn(shape), torch.randn(shape)

    xd = x.double().requires_grad_(True)
    dd = None if delta is None else delta.double().requires_grad_(True)
    p = [t.detach().double().requires_grad_(True) for t in (norm.weight, norm.bias, lin1.weight, lin1.bias, lin2.weight, lin2.bias)]
    s_ref = xd if dd is None else xd + dd
    h = torch.nn.functional.layer_norm(s_ref, (64,), p[0], p[1], norm.eps)
    y_ref = torch.relu(h @ p[2].t() + p[3]) @ p[4].t() + p[5]
    ((s_ref * ws.double()).sum() + (y_ref * wy.double()).sum()).backward()

    norm, lin1, lin2 = norm.to(DEV), lin1.to(DEV), lin2.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    dg = None if delta is None else delta.to(DEV).requires_grad_(True)
    s, y = dense.ln_feed_forward(xg, dg, norm, lin1, lin2)
    assert s.shape == x.shape and y.shape == x.shape
    ((s * ws.to(DEV)).sum() + (y * wy.to(DEV)).sum()).backward()
    assert max_err(s.cpu(), s_ref) < 1e-6 * max(1.0, float(s_ref.abs().max()))
    assert max_err(y.cpu(), y_ref) < 2e-5 * max(1.0, float(y_ref.abs().max()))
    assert rel_err(xg.grad.cpu(), xd.grad) < TOL
    if dg is not None:
        assert torch.equal(dg.grad, xg.grad)
    mods = (norm.weight, norm.bias, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
    for got, want, name in zip(mods, p, ("gamma", "beta", "w1", "b1", "w2", "b2")):
        assert rel_err(got.grad.cpu(), want.grad) < TOL, (name, rel_err(got.grad.cpu(), want.grad))


def test_block_takes_the_fused_route_and_matches_the_unfused_one(monkeypatch):
    from kanvit import ops
    from model import TransformerBlock
    torch.manual_seed(1)
    blk = TransformerBlock(64, 2, feedforward_dim=256, attn_type="cheby").to(DEV)
    x = torch.randn(8, 50, 64, device=DEV)

    def run():
        blk.zero_grad()
        xg = x.clone().requires_grad_(True)
        ops.timer = ops.KernelTimer()
        y = blk(xg)
        y.square().sum().backward()
        tags = set(ops.timer.records)
        ops.timer = None
        return [y.detach(), xg.grad] + [p.grad.clone() for p in blk.parameters()], tags

    fused, tags = run()
    assert "ff_small_fwd" in tags and "ff_small_bwd" in tags
    from kanvit import _lib
    monkeypatch.setenv("KANVIT_NO_FF_SMALL", "1")
    _lib.reload_config()
    try:
        plain, tags0 = run()
    finally:
        monkeypatch.delenv("KANVIT_NO_FF_SMALL")
        _lib.reload_config()
    assert "ff_small_fwd" not in tags0
    for a, b in zip(fused, plain):
        assert rel_err(a, b) < 2e-5
