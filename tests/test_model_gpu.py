"""GPU parity of MSA, the full VisionTransformer and three Adam steps against the reference's own
outputs (tests/golden/msa.npz, model_*.npz).  BASELINE.json: within 1e-4 fp32."""
import numpy as np
import pytest
import torch

from tests._util import T, close, grads_from, load_npz, max_err, rel_err, state_dict_from

pytestmark = pytest.mark.gpu
DEV = "cuda"
TYPES = ["vanilla", "flash-attn", "efficientkan", "sine", "fourier", "cheby", "fast"]


@pytest.mark.parametrize("t", ["vanilla", "cheby", "fast", "efficientkan", "sine"])
def test_msa_against_reference(t):
    from attention import MSA
    blob = load_npz("msa.npz")
    p = t + "."
    msa = MSA(64, 2, type=t)
    msa.load_state_dict(state_dict_from(blob, p))
    msa = msa.to(DEV)
    x = T(blob[p + "x"]).to(DEV).requires_grad_(True)
    y = msa(x)
    (y * torch.linspace(-1, 1, y.numel(), device=DEV).reshape(y.shape)).sum().backward()
    assert max_err(y.cpu(), T(blob[p + "y"])) < 1e-5
    assert rel_err(x.grad.cpu(), T(blob[p + "grad_x"])) < 1e-4
    got = {k: v.grad.cpu() for k, v in msa.named_parameters() if v.grad is not None}
    for k, g in grads_from(blob, p).items():
        assert close(got[k], g, rtol=1e-4, atol=5e-6), (k, rel_err(got[k], g))


def build(blob, t):
    from model import VisionTransformer
    c, h, w, npatch, nblk, d, heads, out_d = (int(v) for v in blob["cfg"])
    m = VisionTransformer((c, h, w), npatch, nblk, d, heads, out_d, type=t)
    m.load_state_dict(state_dict_from(blob))
    return m.to(DEV)


@pytest.mark.parametrize("geom", ["T", "C"])
@pytest.mark.parametrize("t", TYPES)
def test_model_logits_loss_grads(geom, t):
    blob = load_npz(f"model_{geom}_{t}.npz")
    m = build(blob, t)
    x, labels = T(blob["x"]).to(DEV), T(blob["labels"]).to(DEV)
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    assert max_err(logits.cpu(), T(blob["logits"])) < 1e-4, max_err(logits.cpu(), T(blob["logits"]))
    assert abs(float(loss) - float(blob["loss"])) < 1e-4
    named = dict(m.named_parameters())
    for n, gn in zip([str(s) for s in blob["grad_names"]], blob["grad_norms"]):
        mine = float(named[n].grad.double().norm())
        assert abs(mine - gn) <= 1e-4 * max(gn, 1e-3) + 1e-6, (n, mine, gn)
    for n, g in grads_from(blob).items():
        assert rel_err(named[n].grad.cpu(), g) < 1e-4, (n, rel_err(named[n].grad.cpu(), g))


@pytest.mark.parametrize("t", TYPES)
def test_three_adam_steps(t):
    """train.py:31-40 order: forward, CE loss, zero_grad, backward, Adam(lr=1e-3).step()."""
    blob = load_npz(f"model_T_{t}.npz")
    m = build(blob, t)
    x, labels = T(blob["x"]).to(DEV), T(blob["labels"]).to(DEV)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(3):
        loss = torch.nn.functional.cross_entropy(m(x), labels)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert np.allclose(losses, blob["adam_losses"], atol=1e-4), (losses, blob["adam_losses"])
    assert max_err(m.v_class.detach().cpu(), T(blob["adam_v_class"])) < 1e-4
    w = dict(m.named_parameters())[str(blob["adam_w_name"])].detach().cpu().reshape(-1)[:8192]
    assert max_err(w, T(blob["adam_w"])) < 1e-4


def test_full_size_properties_vit_b_block():
    """BASELINE-size shapes (ViT-B/16 @224: N=197, d=768, H=12), checked through size-independent
    properties: row-permutation equivariance over the batch, and linearity of the grouped launch in
    its packed weights (the kernel is linear in W for fixed x)."""
    from attention import MSA
    from kanvit import grouped
    torch.manual_seed(0)
    msa = MSA(768, 12, type="cheby").to(DEV)
    x = torch.randn(8, 197, 768, device=DEV)
    y = msa(x)
    perm = torch.randperm(8, device=DEV)
    assert torch.equal(msa(x[perm]), y[perm])                    # samples are independent, bitwise
    x2 = x.reshape(-1, 768)
    q1 = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x2)
    with torch.no_grad():
        for p in msa.parameters():
            p.mul_(2.0)
    q2 = grouped.run_qkv(msa.q_mappings, msa.k_mappings, msa.v_mappings, x2)
    assert max_err(q2.cpu(), 2 * q1.cpu()) < 1e-5 * float(q1.abs().max() + 1)
    assert torch.isfinite(y).all()


def test_bf16_autocast_keeps_kanvit_ops_in_fp32():
    """bench.py --amp bf16: stock GEMMs in bf16, kanvit ops cast to fp32 at their boundary; the result stays
    within bf16 rounding of the fp32 run (loose oracle, SURVEY.md section 7 'bf16 parity')."""
    blob = load_npz("model_T_cheby.npz")
    m = build(blob, "cheby")
    x, labels = T(blob["x"]).to(DEV), T(blob["labels"]).to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x)
        loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    assert max_err(logits.float().cpu(), T(blob["logits"])) < 5e-2
    g = dict(m.named_parameters())["blocks.0.attn.q_mappings.0.cheby_coeffs"].grad
    assert g.dtype == torch.float32 and torch.isfinite(g).all()
    ref = grads_from(blob)["blocks.0.attn.q_mappings.0.cheby_coeffs"]
    assert rel_err(g.cpu(), ref) < 5e-2
