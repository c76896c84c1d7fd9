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
