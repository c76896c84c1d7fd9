"""CPU-only checks of the drop-in's host side: train.py's command line (the reference's flags verbatim, train.py:87-97),
the reporting helpers against output of the reference's own (tests/golden/metrics.npz, made by make_golden.py from
utils.py:13-47,79-94), the per-block type extension of BASELINE configs[4], and argument validation."""
import os

import numpy as np
import pytest
import torch

from tests._util import load_npz


def test_command_line_is_the_references():
    import train
    a = train.parse([])
    ref = dict(epochs=20, batch_size=128, learning_rate=0.001, model_type="vanilla", n_blocks=8, d_hidden=64, n_heads=8,
               log_dir="logs")                                                    # train.py:88-96 defaults
    for k, v in ref.items():
        assert getattr(a, k) == v, k
    assert a.device in ("cuda", "cpu")
    # the geometry train.py:19 hard-codes is the default of the added flags; the fast paths are opt-in
    assert (a.in_chans, a.image_size, a.n_patches, a.out_d) == (3, 32, 4, 100)
    assert a.amp == "off" and a.graph is False and a.dp is False
    b = train.parse(["--model-type", "cheby", "--amp", "bf16", "--graph", "--epochs", "1"])
    assert (b.model_type, b.amp, b.graph, b.epochs) == ("cheby", "bf16", True, 1)


def test_train_refuses_to_run_the_hot_path_on_the_cpu():
    import train
    with pytest.raises(SystemExit):
        train.main(train.parse(["--device", "cpu", "--synthetic", "--epochs", "1"]))


def test_save_metrics_is_byte_identical_to_the_reference(tmp_path):
    from utils import save_metrics
    fn = str(tmp_path / "logs" / "m.txt")
    save_metrics(fn, 20, "Train", 1.23456789, 0.5, 0.25, 0.125, 0.0625, 0)
    save_metrics(fn, 20, "Test", 4.60517, 0.01, 0.0123456, 0.99995, 0.5, 1)
    want = load_npz("metrics.npz")["save_metrics_text"].tobytes()
    assert open(fn, "rb").read() == want


def test_calculate_metrics_matches_the_reference():
    from utils import calculate_metrics
    m = load_npz("metrics.npz")
    got = calculate_metrics(m["cm.y_true"], m["cm.y_pred"], m["cm.proba"])
    assert np.allclose(got, m["cm.out"], rtol=0, atol=1e-12)


def test_mixed_block_types_keep_the_reference_layout_per_block():
    """BASELINE configs[4] (SineKAN + FourierKAN mixed blocks): block l of type 'a,b' has exactly the parameters of block l
    of the single-type model of its own type; the patch embedding follows the first entry."""
    from model import VisionTransformer, split_types
    assert split_types("sine") == ["sine"] and split_types("sine+fourier") == ["sine", "fourier"]
    geo = dict(chw=(1, 28, 28), n_patches=7, n_blocks=4, d_hidden=64, n_heads=2, out_d=10)
    mixed = VisionTransformer(**geo, type="sine,fourier")
    sine = VisionTransformer(**geo, type="sine")
    four = VisionTransformer(**geo, type="fourier")
    shapes = {k: tuple(v.shape) for k, v in mixed.state_dict().items()}
    want = {}
    for k, v in sine.state_dict().items():
        if not k.startswith("blocks.") or int(k.split(".")[1]) % 2 == 0:
            want[k] = tuple(v.shape)
    for k, v in four.state_dict().items():
        if k.startswith("blocks.") and int(k.split(".")[1]) % 2 == 1:
            want[k] = tuple(v.shape)
    assert shapes == want
    with pytest.raises(ValueError):
        VisionTransformer(**geo, type="sine,flash-attn")
    with pytest.raises(ValueError):
        VisionTransformer(**geo, type="")


def test_attention_shape_checks_happen_before_any_launch():
    """The self-attention binding (kanvit_attn_desc: ONE sequence length) refuses q / k / v of different shapes before any launch;
    FlashAttentionFunction routes q_len != k_len and masks to the general kernels (kanvit_attn_x_*), which -- like every kanvit
    op -- refuse CPU tensors instead of falling back, and refuses `causal` with k_len > q_len (utils.py:169,183: ill-defined)."""
    from utils import FlashAttentionFunction
    from kanvit import KanvitError, ops
    q = torch.randn(1, 2, 8, 16)
    for kv_len in (5, 12):
        k = torch.randn(1, 2, kv_len, 16)
        with pytest.raises(KanvitError):                                # CPU tensors: no fallback
            FlashAttentionFunction.apply(q, k, k, None, False, 512, 1024)
    with pytest.raises(NotImplementedError):
        FlashAttentionFunction.apply(q, torch.randn(1, 2, 12, 16), torch.randn(1, 2, 12, 16), None, True, 512, 1024)
    with pytest.raises(ValueError):                                     # v does not match k
        FlashAttentionFunction.apply(q, torch.randn(1, 2, 12, 16), torch.randn(1, 2, 11, 16), None, False, 512, 1024)
    with pytest.raises(KanvitError):
        ops._attn_desc(q, torch.randn(1, 2, 5, 16), q, q, False, 0.25)
    with pytest.raises(KanvitError):
        ops._attn_desc(q, q, torch.randn(1, 3, 8, 16), q, False, 0.25)


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` invoked PLAINLY (no torch.distributed.run, no RANK): the parent spawns the two ranks as
    children before touching any GPU, relays rank 0's one JSON line and propagates the exit code.  The --rendezvous-only
    hook stops each rank after the process group is up (gloo here: no GPU in this container)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KANVIT_DIST_BACKEND="gloo")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                                  # ONE line on stdout, whatever the launcher printed
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["sum_of_ones"] == 2.0
    # a failing rank's exit code comes back through the parent (a backend that does not exist fails inside every CHILD)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=dict(env, KANVIT_DIST_BACKEND="no-such-backend"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "launching" in r.stderr and not r.stdout.strip()
