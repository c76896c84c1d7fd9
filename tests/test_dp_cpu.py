"""Data-parallel path on CPU: world_size-2 gloo processes must reproduce single-process gradients
and stay in lock-step after optimizer steps.  The reducer is the same object
bench.py / train.py use with backend "nccl" (= RCCL) on the GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                               torch.nn.Linear(32, 5))


def _worker(rank, world, port, overlap, bucket_mib, out, scaled=True):
    sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kanvit import dp as kdp
    model = _make_model()
    if rank == 1:                                    # ranks start different; broadcast must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    kdp.broadcast_parameters(model)
    g = torch.Generator().manual_seed(42)
    X, Y = torch.randn(16, 12, generator=g), torch.randint(0, 5, (16,), generator=g)
    lo, hi = kdp.shard_batch(16, rank, world)
    red = kdp.GradReducer(model.parameters(), bucket_mib=bucket_mib, overlap=overlap, timing=True)
    assert red.average == "loss"                     # gloo has no ReduceOp.AVG: the mean comes from the pre-scaled loss
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    grads0 = None
    for it in range(3):
        loss = torch.nn.functional.cross_entropy(model(X[lo:hi]), Y[lo:hi])
        red.zero_grad()
        # scaled: the documented protocol (no pass over the buckets after the waits); unscaled: a caller that forgot
        # scale_loss() -- finish() must notice and fall back to its one multi-tensor mul_
        (red.scale_loss(loss) if scaled else loss).backward()
        red.finish()
        if it == 0:
            grads0 = [p.grad.clone() for p in model.parameters()]
        opt.step()
    comm = red.comm_summary()
    torch.save({"grads0": grads0, "params": [p.detach().clone() for p in model.parameters()],
                "nbuckets": len(red.buckets), "comm": comm}, out + f".{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,scaled", [(True, True), (False, True), (True, False)])
@pytest.mark.parametrize("bucket_mib", [64.0, 0.001])
def test_dp2_matches_single_process(tmp_path, overlap, bucket_mib, scaled):
    out = str(tmp_path / "r")
    mp.spawn(_worker, args=(2, _free_port(), overlap, bucket_mib, out, scaled), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    # single-process reference: the whole batch of 16
    model = _make_model()
    g = torch.Generator().manual_seed(42)
    X, Y = torch.randn(16, 12, generator=g), torch.randint(0, 5, (16,), generator=g)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for it in range(3):
        loss = torch.nn.functional.cross_entropy(model(X), Y)
        opt.zero_grad()
        loss.backward()
        if it == 0:
            ref_g = [p.grad.clone() for p in model.parameters()]
        opt.step()
    for a, b, c in zip(r0["grads0"], r1["grads0"], ref_g):
        assert torch.equal(a, b)                                     # identical on both ranks
        assert torch.allclose(a, c, rtol=1e-5, atol=1e-7)            # == single-process gradient
    for a, b, c in zip(r0["params"], r1["params"], model.parameters()):
        assert torch.equal(a, b)
        assert torch.allclose(a, c.detach(), rtol=1e-4, atol=1e-6)
    assert r0["nbuckets"] == (1 if bucket_mib > 1 else 4)
    comm = r0["comm"]                                                 # the record bench.py prints as `comm`
    assert comm["buckets"] == r0["nbuckets"] and comm["world"] == 2 and comm["steps"] == 3 and comm["backend"] == "gloo"
    assert comm["bytes_per_step"] == sum(comm["bucket_bytes"]) == 4 * sum(p.numel() for p in model.parameters())
    assert comm["exposed_ms_per_step"] >= 0.0 and comm["average"] == "loss"


def test_shard_batch_covers_everything():
    sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
    from kanvit.dp import shard_batch
    for gb in (1, 7, 8, 128, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_batch(gb, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _avg_worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kanvit import dp as kdp
    try:
        kdp.GradReducer(_make_model().parameters(), average="avg")
        res = "accepted"
    except ValueError as e:
        res = str(e)
    open(out + f".{rank}", "w").write(res)
    dist.destroy_process_group()


def test_collective_average_is_refused_where_the_backend_has_none(tmp_path):
    """ReduceOp.AVG exists on RCCL only; on gloo the reducer must say so at construction instead of failing in a hook."""
    out = str(tmp_path / "a")
    mp.spawn(_avg_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert "ReduceOp.AVG" in open(out + ".0").read()


def test_reducer_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "kan-vit_amd"))
    from kanvit.dp import GradReducer
    model = _make_model()
    red = GradReducer(model.parameters())
    x = torch.randn(4, 12)
    red.zero_grad()
    model(x).sum().backward()
    red.finish()
    g1 = [p.grad.clone() for p in model.parameters()]
    model2 = _make_model()
    model2(x).sum().backward()
    for a, p in zip(g1, model2.parameters()):
        assert torch.equal(a, p.grad)
