"""kanvit_addln_fwd / _bwd (residual add + LayerNorm in one pass) against torch.nn.LayerNorm in fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("M,D", [(1, 4), (7, 8), (129, 64), (1000, 384), (2050, 768), (33, 1024), (5, 260)])
@pytest.mark.parametrize("with_delta", [True, False])
def test_add_layernorm_matches_torch(M, D, with_delta):
    from kanvit.ops import add_layernorm
    torch.manual_seed(M * 7 + D)
    norm = torch.nn.LayerNorm(D).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(D) * 0.5 + 1.0)
        norm.bias.copy_(torch.randn(D) * 0.3)
    x = (torch.randn(M, D, device=DEV) * 2.0 + 0.7).requires_grad_(True)
    delta = torch.randn(M, D, device=DEV).requires_grad_(True) if with_delta else None
    ws, wy = torch.randn(M, D, device=DEV), torch.randn(M, D, device=DEV)

    s, y = add_layernorm(x, delta, norm)
    ((s * ws).sum() + (y * wy).sum()).backward()
    got = [s.detach(), y.detach(), x.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()]
    if with_delta:
        got.append(delta.grad.clone())

    n64 = torch.nn.LayerNorm(D, eps=norm.eps).double().to(DEV)
    n64.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    x64 = x.detach().double().requires_grad_(True)
    d64 = delta.detach().double().requires_grad_(True) if with_delta else None
    s64 = x64 + d64 if with_delta else x64
    y64 = n64(s64)
    ((s64 * ws.double()).sum() + (y64 * wy.double()).sum()).backward()
    want = [s64.detach(), y64.detach(), x64.grad, n64.weight.grad, n64.bias.grad]
    if with_delta:
        want.append(d64.grad)
    for g, w in zip(got, want):
        scale = max(1.0, float(w.abs().max()))
        assert float((g.double() - w).abs().max()) < 2e-5 * scale


def test_add_layernorm_3d_block_shapes_and_determinism():
    from kanvit.ops import add_layernorm
    torch.manual_seed(0)
    norm = torch.nn.LayerNorm(768).to(DEV)
    x = torch.randn(4, 197, 768, device=DEV, requires_grad=True)
    d = torch.randn(4, 197, 768, device=DEV, requires_grad=True)
    outs = []
    for _ in range(2):
        x.grad = d.grad = None
        norm.zero_grad()
        s, y = add_layernorm(x, d, norm)
        (y.square().sum() + s.sum()).backward()
        outs.append((y.detach().clone(), x.grad.clone(), norm.weight.grad.clone()))
    assert s.shape == x.shape and y.shape == x.shape
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.equal(x.grad, d.grad)


def test_unsupported_width_falls_back_to_stock_ops():
    from kanvit.ops import add_layernorm
    norm = torch.nn.LayerNorm(10).to(DEV)           # 10 % 4 != 0
    x = torch.randn(3, 10, device=DEV)
    s, y = add_layernorm(x, None, norm)
    assert torch.allclose(y, norm(x))


@pytest.mark.parametrize("M,D", [(7, 8), (129, 64), (1000, 384), (2050, 768), (5, 260)])
@pytest.mark.parametrize("delta_bf16,y_bf16", [(True, True), (True, False), (False, True)])
def test_bf16_tensors_at_the_boundary(M, D, delta_bf16, y_bf16):
    """torch.autocast: the feed-forward's output (delta), its input (y) and the gradient that arrives on y may be bf16 and are read /
    written as they are (kanvit_addln_*_ex).  Reference: the same mathematics in fp64 on the bf16-ROUNDED inputs; a bf16 output is
    the fp32 result rounded to nearest even (exactly what `.to(bfloat16)` of the fp32 kernel's output gives -> compared bitwise
    with that), the gradient returned for a bf16 delta is the rounded dx."""
    from kanvit.ops import add_layernorm
    torch.manual_seed(M + D)
    norm = torch.nn.LayerNorm(D).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(D) * 0.5 + 1.0)
        norm.bias.copy_(torch.randn(D) * 0.3)
    x = (torch.randn(M, D, device=DEV) * 2.0 + 0.7).requires_grad_(True)
    delta = torch.randn(M, D, device=DEV)
    delta = (delta.bfloat16() if delta_bf16 else delta).requires_grad_(True)
    ws = torch.randn(M, D, device=DEV)
    wy = torch.randn(M, D, device=DEV)
    wy = wy.bfloat16() if y_bf16 else wy                     # the gradient on a bf16 y is bf16

    s, y = add_layernorm(x, delta, norm, y_bf16=y_bf16)
    assert y.dtype == (torch.bfloat16 if y_bf16 else torch.float32) and s.dtype == torch.float32
    torch.autograd.backward([s, y], [ws, wy])
    assert delta.grad.dtype == delta.dtype

    # the all-fp32 kernel on the same (rounded) values
    x2 = x.detach().clone().requires_grad_(True)
    d2 = delta.detach().float().requires_grad_(True)
    s2, y2 = add_layernorm(x2, d2, norm)
    norm.zero_grad()
    g_first = (norm.weight.grad, norm.bias.grad)
    torch.autograd.backward([s2, y2], [ws, wy.float()])
    assert torch.equal(s, s2)
    assert torch.equal(y, y2.bfloat16()) if y_bf16 else torch.equal(y, y2)
    assert torch.equal(x.grad, x2.grad)                      # same arithmetic: bf16 -> fp32 widening is exact
    assert torch.equal(delta.grad, d2.grad.bfloat16()) if delta_bf16 else torch.equal(delta.grad, d2.grad)


def test_block_under_autocast_feeds_bf16_through_the_layernorms(monkeypatch):
    """TransformerBlock.run under bf16 autocast: LN2's output is bf16, the feed-forward output stays bf16 into the next block's add --
    same numbers as the route through explicit casts (fp32 LayerNorm output, .to(bfloat16) in front of the GEMM), bitwise."""
    import model
    from kanvit import ops
    torch.manual_seed(0)
    blk = model.TransformerBlock(384, 6, 1536, attn_type="cheby").to(DEV)
    x = torch.randn(4, 197, 384, device=DEV, requires_grad=True)
    pend = torch.randn(4, 197, 384, device=DEV).bfloat16().requires_grad_(True)
    outs = []
    for force_f32 in (False, True):
        if force_f32:
            monkeypatch.setattr(model, "add_layernorm", lambda x_, d_, n_, y_bf16=False: ops.add_layernorm(x_, None if d_ is None else d_.float(), n_, False))
        x.grad = None
        pend.grad = None
        blk.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            xs, f = blk.run(x, pend)
            assert f.dtype == torch.bfloat16
        (xs.float().square().sum() + f.float().square().sum()).backward()
        outs.append((xs.detach().clone(), f.detach().clone(), x.grad.clone(), pend.grad.clone(), blk.norm2.weight.grad.clone(),
                     blk.ff[0].weight.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
