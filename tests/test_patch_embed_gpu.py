"""patchify + patch-embedding KAN layer + class token + position embedding in one kernel launch
(kanvit_patch_embed_fwd).  The fused launch performs the same fp32 operations in the same order as the three-step path
(gather is pure addressing; the epilogue adds bias, then pos), so outputs must be BITWISE equal to it whenever both run
the same kernel instantiation (patch width a multiple of the 8-feature chunk), and it is checked
against the float64 oracle's VisionTransformer prologue (oracle.patchify / positional_embeddings, pinned to the reference
by tests/test_oracle_golden.py) as well."""
import pytest
import torch

from oracle import kan_oracle as ko
from tests._util import max_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tokens(m, images, fused):
    m._fused_embed = None if fused else False
    out = m._embed_fused(images)
    if not fused:
        assert out is None
        patches = m.patchify(images, m.n_patches)
        b, p, _ = patches.shape
        tok = m.linear_mapper(patches).reshape(b, p, m.d_hidden)
        out = torch.cat((m.v_class.unsqueeze(0).expand(b, -1, -1), tok), dim=1) + m.pos_embeddings[: p + 1]
    return out


@pytest.mark.parametrize("t", ["cheby", "efficientkan", "sine"])
@pytest.mark.parametrize("geom", [((3, 32, 32), 4, 64, 5), ((3, 224, 224), 14, 768, 2), ((1, 64, 64), 4, 128, 3), ((1, 28, 28), 7, 64, 6)])
def test_fused_patch_embedding_equals_three_step_path_and_oracle(t, geom):
    from model import VisionTransformer
    chw, npatch, d, b = geom
    torch.manual_seed(1)
    m = VisionTransformer(chw, n_patches=npatch, n_blocks=1, d_hidden=d, n_heads=2, out_d=10, type=t).to(DEV)
    x = torch.randn(b, *chw, device=DEV)
    wgt = torch.randn(b, npatch * npatch + 1, d, device=DEV)
    res = []
    for fused in (True, False):
        m.zero_grad()
        out = _tokens(m, x, fused)
        assert m._fused_embed is (True if fused else False)
        (out * wgt).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert set(res[0][1]) == set(res[1][1])
    if (chw[2] // npatch) % 8 == 0:
        # both paths run the same kernel instantiation (8-feature chunks): same operations in the same order -> bitwise equal
        assert torch.equal(res[0][0], res[1][0])
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], res[1][1][k]), k      # the backward runs the same kernels on the same rows
    else:
        # narrow patches: the gather needs a chunk that divides the patch width, i.e. another k order of the same fp32 sums
        assert max_err(res[0][0], res[1][0]) < 2e-6 * max(1.0, float(res[1][0].abs().max()))
        for k in res[0][1]:
            assert rel_err(res[0][1][k], res[1][1][k]) < 1e-5, k
    # float64 oracle of the prologue (model.py:144-152)
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in m.state_dict().items()}
    patches = ko.patchify(x.cpu().double(), npatch)
    tok = ko.layer_forward(sd, "linear_mapper.", patches).reshape(b, npatch * npatch, d)
    ref = torch.cat([sd["v_class"].unsqueeze(0).expand(b, -1, -1), tok], dim=1) + ko.positional_embeddings(npatch * npatch + 1, d).double()
    assert max_err(res[0][0].cpu(), ref) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_unsupported_geometry_falls_back_once():
    """3-pixel-wide patches cannot hold a (2-pixel) feature chunk of the register kernel, and a 96-wide output is not a whole
    number of column tiles -> the C entry point reports EINVAL, the module remembers it and runs the three-step path (the
    same HIP layer kernel behind patchify; nothing is replaced by torch or the CPU)."""
    from model import VisionTransformer
    torch.manual_seed(0)
    for chw, npatch, d in (((1, 21, 21), 7, 64), ((1, 64, 64), 4, 96)):
        m = VisionTransformer(chw, npatch, 1, d, 2, 10, type="cheby").to(DEV)
        y = m(torch.rand(4, *chw, device=DEV))
        assert m._fused_embed is False and torch.isfinite(y).all()
    m2 = VisionTransformer((3, 32, 32), 4, 1, 64, 2, 10, type="cheby").to(DEV)
    m2(torch.randn(2, 3, 32, 32, device=DEV))
    assert m2._fused_embed is True
    with torch.autocast("cuda", dtype=torch.bfloat16):               # bf16 mode keeps the (gather-less) bf16 kernels
        m2(torch.randn(2, 3, 32, 32, device=DEV))
