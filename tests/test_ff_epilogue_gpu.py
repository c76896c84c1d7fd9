"""kanvit_relu_bwd_bias (csrc/ff_epilogue.hip): the ReLU mask of the incoming gradient and the bias gradient of the feed-forward's
first Linear (reference model.py:25-29: nn.Sequential(Linear, ReLU(inplace), Linear)) in one pass -- against what autograd runs for
the reference (threshold_backward on the saved activation, then sum(0)), evaluated in float64; ragged row counts, bitwise run to
run, and through kanvit.dense (the path TransformerBlock takes) against torch's own Linear -> ReLU -> Linear."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _call(dy, y):
    from kanvit import _lib, ops
    L = _lib.lib()
    M, N = dy.shape
    out = torch.empty_like(dy)
    db = torch.empty(N, device=DEV)
    nb = int(L.kanvit_relu_bwd_bias_workspace(M, N))
    ws = ops._workspace(nb, dy.device)
    _lib.check(L.kanvit_relu_bwd_bias(M, N, ops._ptr(dy), ops._ptr(y), ops._ptr(out), ops._ptr(db), ops._ptr(ws), C.c_size_t(nb), ops._stream()),
               "kanvit_relu_bwd_bias")
    return out, db


@pytest.mark.parametrize("m,n", [(1, 4), (7, 256), (100, 1028), (6400, 256), (1333, 3072), (25216, 3072)])
def test_mask_and_column_sums(m, n):
    torch.manual_seed(m + n)
    dy = torch.randn(m, n, device=DEV)
    y = torch.relu(torch.randn(m, n, device=DEV))             # exact zeros where the ReLU clipped
    out, db = _call(dy, y)
    ref = torch.ops.aten.threshold_backward(dy, y, 0)
    assert torch.equal(out, ref)                               # a select, not arithmetic: bitwise
    want = ref.double().sum(0)
    assert float((db.double() - want).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().sum(0).max()))
    out2, db2 = _call(dy, y)
    assert torch.equal(db, db2) and torch.equal(out, out2)     # ordered partial sums: bitwise run to run


@pytest.mark.parametrize("m,n", [(3, 8), (100, 1028), (1333, 3072), (25216, 3072)])
def test_bf16_mask_and_column_sums(m, n):
    """The autocast variant: bf16 dy / y / dh, fp32 column sums (what `threshold_backward` + `sum(0, dtype=float32)` give)."""
    from kanvit import _lib, ops
    L = _lib.lib()
    torch.manual_seed(m * 7 + n)
    dy = torch.randn(m, n, device=DEV).bfloat16()
    y = torch.relu(torch.randn(m, n, device=DEV)).bfloat16()
    y[0, :4] = torch.tensor([float("nan"), -0.0, 0.0, float("-inf")], device=DEV).bfloat16()
    out = torch.empty_like(dy)
    db = torch.empty(n, device=DEV)
    nb = int(L.kanvit_relu_bwd_bias_workspace(m, n))
    ws = ops._workspace(nb, dy.device)
    _lib.check(L.kanvit_relu_bwd_bias_bf16(m, n, ops._ptr(dy), ops._ptr(y), ops._ptr(out), ops._ptr(db), ops._ptr(ws), C.c_size_t(nb),
                                           ops._stream()), "kanvit_relu_bwd_bias_bf16")
    ref = torch.ops.aten.threshold_backward(dy, y, 0)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))          # bit patterns (the NaN-activation column passes dy through)
    want = ref.double().sum(0)
    assert float((db.double() - want).abs().max()) <= 2e-6 * max(1.0, float(ref.double().abs().sum(0).max()))


def test_fp32_nan_activation_passes_the_gradient_like_torch():
    y = torch.tensor([[float("nan"), -1.0, 0.0, 2.0]], device=DEV)
    dy = torch.ones(1, 4, device=DEV)
    out, _ = _call(dy, y)
    assert torch.equal(out, torch.ops.aten.threshold_backward(dy, y, 0))


def test_empty_and_bad_shapes():
    from kanvit import _lib
    L = _lib.lib()
    db = torch.ones(8, device=DEV)
    assert L.kanvit_relu_bwd_bias(0, 8, None, None, None, C.c_void_p(db.data_ptr()), None, 0, None) == 0
    torch.cuda.synchronize()
    assert float(db.abs().max()) == 0.0                        # no rows: the gradient is exactly zero
    assert L.kanvit_relu_bwd_bias(4, 6, None, None, None, C.c_void_p(db.data_ptr()), None, 0, None) != 0
    assert b"multiple of 4" in L.kanvit_last_error()


def test_dense_feed_forward_matches_torch():
    """kanvit.dense.feed_forward at a shape that takes the stock-GEMM path (d = 128): y, dx and all four parameter gradients equal to
    torch's nn.Sequential within fp32 rounding, with the fused epilogue in the backward."""
    from kanvit import dense
    torch.manual_seed(3)
    lin1, lin2 = torch.nn.Linear(128, 512).to(DEV), torch.nn.Linear(512, 128).to(DEV)
    x = torch.randn(777, 128, device=DEV)
    w = torch.randn(777, 128, device=DEV)

    def run(fn):
        for p in (*lin1.parameters(), *lin2.parameters()):
            p.grad = None
        xg = x.clone().requires_grad_(True)
        (fn(xg) * w).sum().backward()
        return [xg.grad] + [p.grad.clone() for p in (*lin1.parameters(), *lin2.parameters())]

    ours = run(lambda t: dense.feed_forward(t, lin1, lin2))
    ref = run(lambda t: lin2(torch.relu(lin1(t))))
    for a, b in zip(ours, ref):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
