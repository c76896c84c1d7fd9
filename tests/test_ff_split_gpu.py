"""Opt-in bf16x3 feed-forward (kanvit_split3_bf16 + bf16 GEMMs with fp32 accumulate): the split image is exact to 2^-16,
the block matches the fp32 block to a few 1e-6, and a whole model stays inside BASELINE's 1e-4 budget."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("pattern", [0, 1])
def test_split_image_reconstructs_the_input(pattern):
    from kanvit import dense as D
    torch.manual_seed(0)
    x = torch.randn(37, 64, device=DEV) * torch.logspace(-3, 3, 64, device=DEV)
    img = D._split3(x, pattern).float()
    K = x.shape[1]
    hi = img[:, :K]
    lo = img[:, 2 * K:] if pattern == 0 else img[:, K:2 * K]
    dup = img[:, K:2 * K] if pattern == 0 else img[:, 2 * K:]
    assert torch.equal(dup, hi)
    assert torch.equal(hi, x.bfloat16().float())
    assert float(((hi + lo) - x).abs().max() / x.abs().max()) < 2.0 ** -15
    assert float((((hi + lo) - x).abs() / x.abs().clamp_min(1e-30)).max()) < 2.0 ** -15


def test_split_image_bias_relu_mask():
    from kanvit import dense as D
    torch.manual_seed(1)
    x = torch.randn(9, 16, device=DEV)
    b = torch.randn(16, device=DEV)
    img = D._split3(x, 0, bias=b, relu=True).float()
    want = torch.relu(x + b)
    assert float((img[:, :16] + img[:, 32:] - want).abs().max()) < 1e-4
    g = torch.randn(9, 16, device=DEV)
    m = D._split3(g, 0, mask=D._split3(x, 0, bias=b, relu=True)).float()
    assert float((m[:, :16] + m[:, 32:] - g * (want.bfloat16().float() > 0)).abs().max()) < 1e-4


def test_feed_forward_split_close_to_fp32():
    from kanvit import dense as D
    torch.manual_seed(2)
    l1, l2 = torch.nn.Linear(64, 256).to(DEV), torch.nn.Linear(256, 64).to(DEV)
    x = torch.randn(2048, 64, device=DEV, requires_grad=True)
    dy = torch.randn(2048, 64, device=DEV)
    res = {}
    for mode in ("fp32", "bf16x3"):
        D.FF_MODE = mode
        try:
            x.grad = None
            l1.zero_grad()
            l2.zero_grad()
            y = D.feed_forward(x, l1, l2)
            y.backward(dy)
            res[mode] = [y.detach().clone(), l2.weight.grad.clone(), l2.bias.grad.clone(), x.grad.clone(), l1.weight.grad.clone()]
        finally:
            D.FF_MODE = "fp32"
    errs = [float((a - b).abs().max()) / float(a.abs().max()) for a, b in zip(res["fp32"], res["bf16x3"])]
    assert errs[0] > 0                                  # the split path really ran
    assert max(errs[:3]) < 3e-5, errs                   # y, dW2, db2
    assert max(errs[3:]) < 5e-2, errs                   # dx, dW1: ReLU-mask flips where a pre-activation is ~ 0


def test_model_with_split_feed_forward_within_parity_budget():
    from kanvit import dense as D
    from model import VisionTransformer
    torch.manual_seed(3)
    m = VisionTransformer((3, 32, 32), n_patches=4, n_blocks=4, d_hidden=64, n_heads=8, out_d=100, type="cheby").to(DEV)
    x = torch.randn(16, 3, 32, 32, device=DEV)
    with torch.no_grad():
        a = m(x)
        D.FF_MODE = "bf16x3"
        try:
            b = m(x)
        finally:
            D.FF_MODE = "fp32"
    assert 0 < float((a - b).abs().max()) < 1e-4
