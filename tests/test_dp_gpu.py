"""Data-parallel train step with the HIP kernels in the loop: two ranks share cuda:0 (the GPU box has
one card) and exchange gradients over gloo; on the 8-GPU node the same GradReducer runs over RCCL.
DP(2 ranks x B/2) must equal one process on the whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build():
    sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
    from model import VisionTransformer
    torch.manual_seed(0)
    return VisionTransformer((1, 28, 28), n_patches=7, n_blocks=2, d_hidden=64, n_heads=2, out_d=10, type="cheby").cuda()


def _batch():
    g = torch.Generator().manual_seed(9)
    return torch.rand(16, 1, 28, 28, generator=g).cuda(), (torch.arange(16) % 10).cuda()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _build()
    sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
    from kanvit import dp as kdp
    kdp.broadcast_parameters(model)
    red = kdp.GradReducer(model.parameters(), bucket_mib=0.25)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x, y = _batch()
    lo, hi = kdp.shard_batch(16, rank, world)
    grads0 = None
    for it in range(2):
        loss = torch.nn.functional.cross_entropy(model(x[lo:hi]), y[lo:hi])
        red.zero_grad()
        loss.backward()
        red.finish()
        if it == 0:
            grads0 = [p.grad.detach().cpu().clone() for p in model.parameters()]
        opt.step()
    torch.save({"params": [p.detach().cpu() for p in model.parameters()], "grads0": grads0}, out + f".{rank}")
    dist.destroy_process_group()


def test_dp2_on_gpu_matches_single_process(tmp_path):
    out = str(tmp_path / "p")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    p0, p1 = torch.load(out + ".0"), torch.load(out + ".1")
    model = _build()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x, y = _batch()
    loss = torch.nn.functional.cross_entropy(model(x), y)
    loss.backward()
    # gradients of the averaged shards == gradients of the whole batch.  (Parameters after Adam are NOT compared
    # with the single-process run: e.g. the degree-0 Chebyshev coefficients of the key mappings have an exactly
    # zero true gradient -- softmax is shift invariant -- so Adam's first step on them is lr*sign(rounding noise).)
    for a, b, p in zip(p0["grads0"], p1["grads0"], model.parameters()):
        assert torch.equal(a, b)
        g = p.grad.detach().cpu()
        assert float((a - g).abs().max()) <= 2e-6 + 2e-4 * float(g.abs().max())
    for a, b in zip(p0["params"], p1["params"]):
        assert torch.equal(a, b)                    # replicas stay in lock-step through Adam
