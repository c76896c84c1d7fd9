"""kanvit.dense: the feed-forward Linear with the token-split weight gradient must be the same function as nn.Linear
(reference model.py:25-29 uses stock nn.Linear) -- outputs identical, gradients equal to rounding."""
import pytest
import torch

from kanvit import dense as D


@pytest.mark.parametrize("M,K,N", [(2048, 48, 96), (1536, 64, 32), (7, 16, 8), (1024, 32, 32)])
@pytest.mark.parametrize("bias", [True, False])
def test_dense_matches_linear(M, K, N, bias):
    torch.manual_seed(M + K + N)
    lin = torch.nn.Linear(K, N, bias=bias).double()
    x = torch.randn(M, K, dtype=torch.double, requires_grad=True)
    dy = torch.randn(M, N, dtype=torch.double)
    y = D.dense(x, lin)
    y.backward(dy)
    got = [y.detach().clone(), x.grad.clone(), lin.weight.grad.clone()] + ([lin.bias.grad.clone()] if bias else [])
    x.grad = None
    lin.zero_grad()
    y2 = lin(x)
    y2.backward(dy)
    want = [y2.detach(), x.grad, lin.weight.grad] + ([lin.bias.grad] if bias else [])
    assert torch.equal(got[0], want[0])
    for g, w in zip(got[1:], want[1:]):
        assert float((g - w).abs().max()) <= 1e-11 * (1 + float(w.abs().max()))


def test_slab_count():
    # ViT-B feed-forward at B=128: 36 macro-tiles -> 8 slabs of 3136 tokens; divisibility and a 128-token floor hold
    assert D.wgrad_slabs(25088, 3072, 768) == 8
    assert D.wgrad_slabs(25088, 768, 3072) == 8
    assert D.wgrad_slabs(100, 64, 128) == 1           # too few tokens to split
    assert D.wgrad_slabs(6400, 256, 64) == 32         # MNIST geometry: small output, many short slabs
    assert D.wgrad_slabs(25088, 4096, 4096) == 1      # output alone fills the chip
    for M in (25088, 12544, 6272, 50176, 1000, 1024 * 3):
        s = D.wgrad_slabs(M, 768, 3072)
        assert M % s == 0 and (s == 1 or M // s >= 128)


def test_split_path_taken_and_deterministic():
    torch.manual_seed(1)
    lin = torch.nn.Linear(32, 64)
    x = torch.randn(4096, 32, requires_grad=True)
    assert D.wgrad_slabs(4096, 64, 32) > 1
    outs = []
    for _ in range(2):
        lin.zero_grad()
        D.dense(x, lin).square().sum().backward()
        outs.append(lin.weight.grad.clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("bias", [True, False])
def test_dense_relu_matches_linear_relu(bias):
    """relu=True is Linear -> ReLU (reference model.py:25-29), with the mask applied to the gradient inside the op."""
    torch.manual_seed(3)
    lin = torch.nn.Linear(24, 40, bias=bias).double()
    x = torch.randn(1536, 24, dtype=torch.double, requires_grad=True)
    dy = torch.randn(1536, 40, dtype=torch.double)
    y = D.dense(x, lin, relu=True)
    y.backward(dy)
    got = [y.detach().clone(), x.grad.clone(), lin.weight.grad.clone()] + ([lin.bias.grad.clone()] if bias else [])
    x.grad = None
    lin.zero_grad()
    y2 = torch.relu(lin(x))
    y2.backward(dy)
    want = [y2.detach(), x.grad, lin.weight.grad] + ([lin.bias.grad] if bias else [])
    assert torch.equal(got[0], want[0])
    for g, w in zip(got[1:], want[1:]):
        assert float((g - w).abs().max()) <= 1e-11 * (1 + float(w.abs().max()))
