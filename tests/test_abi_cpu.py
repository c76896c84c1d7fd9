"""CPU-only checks of the boundary: the C-ABI library builds/loads and exports every symbol
include/kanvit.h declares, descriptors have the C layout, the drop-in modules keep the reference's
state_dict layout, and the product path refuses to run without a GPU (no silent fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests._util import T, load_npz, max_err, state_dict_from

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from kanvit import build
    build.build(verbose=False)
    from kanvit import _lib
    return _lib.lib()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "kanvit.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(kanvit_[a-z0-9_]+)\s*\(", src))
    fams = re.findall(r"KANVIT_DECLARE_FAMILY\((\w+)\)", src)
    names = {n for n in names if "##" not in n}
    for f in fams:
        if f == "name":
            continue
        for suf in ("fwd", "bwd_input", "bwd_weight", "qkv_fwd", "qkv_bwd_input", "qkv_bwd_weight"):
            names.add(f"kanvit_{f}_{suf}")
    return names


def test_every_declared_symbol_is_exported(lib):
    from kanvit import _lib
    declared = _declared_functions()
    assert len(declared) >= 11 + 36
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in kanvit.h but not exported"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_fused_layernorm_query_is_host_only(lib):
    """kanvit_layer_ln_fusable (FastKAN LayerNorm fusion, KANVIT_FLAG_FUSED_LN) is a pure host function: the headline
    geometries qualify, ragged / tiny / non-RBF ones do not, and the flag is refused where it cannot apply."""
    from kanvit import _lib

    def desc(**kw):
        base = dict(family=_lib.RBF, groups=18, x_group_mod=6, I=64, O=64, G=8, has_base=1, rbf_inv_h=1.75, M=25216,
                    ldx=384, ldu=1152, ldy=1152, bparam_stride=8 + 128, ln_eps=1e-5, flags=_lib.FLAG_UNIFORM_KNOTS)
        base.update(kw)
        return _lib.LayerDesc(**base)

    assert lib.kanvit_layer_ln_fusable(ctypes.byref(desc())) == 1                         # ViT-S q|k|v
    assert lib.kanvit_layer_ln_fusable(ctypes.byref(desc(groups=1, x_group_mod=1, I=768, O=384, ldx=768, ldu=768,
                                                         ldy=384))) == 1                  # ViT-S patch embedding
    assert lib.kanvit_layer_ln_fusable(ctypes.byref(desc(M=100))) == 0                    # too few rows for the register kernels
    assert lib.kanvit_layer_ln_fusable(ctypes.byref(desc(I=50, ldx=300))) == 0            # ragged feature count
    assert lib.kanvit_layer_ln_fusable(ctypes.byref(desc(has_base=0))) == 0
    assert lib.kanvit_layer_ln_fusable(ctypes.byref(desc(flags=0))) == 0                  # centres not vouched uniform: LDS-tile kernels, no fusion
    assert lib.kanvit_layer_ln_fusable(ctypes.byref(desc(family=_lib.CHEBY, G=5))) == 0
    d = desc(family=_lib.CHEBY, G=5, flags=_lib.FLAG_FUSED_LN)
    assert lib.kanvit_layer_fwd(ctypes.byref(d), None, None, None, None, None, None, None, 0, None) == -22
    assert b"FUSED_LN" in lib.kanvit_last_error()
    d = desc(flags=_lib.FLAG_FUSED_LN | _lib.FLAG_UNIFORM_KNOTS, bparam_stride=8)
    assert lib.kanvit_layer_fwd(ctypes.byref(d), None, None, None, None, None, None, None, 0, None) == -22
    assert b"gamma" in lib.kanvit_last_error()
    d = desc(flags=_lib.FLAG_FUSED_LN | _lib.FLAG_UNIFORM_KNOTS, M=100)
    assert lib.kanvit_layer_bwd_weight(ctypes.byref(d), None, None, None, None, None, None, 0, None) == -22


def test_descriptor_layout_and_errors(lib):
    from kanvit import _lib
    assert ctypes.sizeof(_lib.LayerDesc) == 10 * 4 + 5 * 8 + 2 * 4
    assert ctypes.sizeof(_lib.AttnDesc) == 8 * 4 + 12 * 8
    assert lib.kanvit_abi_version() == 7
    d = _lib.LayerDesc(family=99, groups=1, x_group_mod=1, I=4, O=3, G=1, M=6, ldx=4, ldy=3)
    rc = lib.kanvit_layer_fwd(ctypes.byref(d), None, None, None, None, None, None, None, 0, None)
    assert rc == -22 and b"family" in lib.kanvit_last_error()
    d.family = _lib.CHEBY
    d.G = 5
    rc = lib.kanvit_layer_fwd(ctypes.byref(d), None, None, None, None, None, None, None, 0, None)
    assert rc == -22 and b"null" in lib.kanvit_last_error()
    d.G = 100
    assert lib.kanvit_layer_fwd(ctypes.byref(d), None, None, None, None, None, None, None, 0, None) == -22
    d.G = 5
    assert lib.kanvit_cheby_fwd(ctypes.byref(d), None, None, None, None, None, None, None, 0, None) == -22
    assert lib.kanvit_sine_fwd(ctypes.byref(d), None, None, None, None, None, None, None, 0, None) == -22
    assert b"mismatch" in lib.kanvit_last_error()
    # workspace sizing is a pure host function
    d.M, d.I, d.O, d.groups, d.x_group_mod, d.ldx, d.ldy = 25216, 64, 64, 36, 12, 768, 2304
    ws = lib.kanvit_layer_bwd_weight_workspace(ctypes.byref(d))
    assert ws % (36 * 320 * 64 * 4) == 0 and ws > 0
    # round 4: the ViT-B q|k|v launch takes the LDS-DMA form (five work-group slabs, each the ordered sum of four row ranges:
    # a quarter of the register form's 21 slab partials); KANVIT_BW_NO_DMA gives the register plan back
    assert ws == 5 * 36 * 320 * 64 * 4
    os.environ["KANVIT_BW_NO_DMA"] = "1"
    try:
        lib.kanvit_config_reload()
        assert lib.kanvit_layer_bwd_weight_workspace(ctypes.byref(d)) == 21 * 36 * 320 * 64 * 4
    finally:
        del os.environ["KANVIT_BW_NO_DMA"]
        lib.kanvit_config_reload()
    d.groups, d.x_group_mod, d.I, d.O, d.ldx, d.ldy = 1, 1, 768, 768, 768, 768        # one fp32 layer: the register form (whose sums the patch-gather kernel reproduces)
    ws1 = lib.kanvit_layer_bwd_weight_workspace(ctypes.byref(d))
    d.flags = _lib.FLAG_BF16_MFMA                                                      # ... and under the bf16 flag (192 units cannot fill 256 CUs with whole work-groups)
    assert lib.kanvit_layer_bwd_weight_workspace(ctypes.byref(d)) == ws1
    d.flags = 0
    d.M, d.I, d.O, d.groups, d.x_group_mod, d.ldx, d.ldy = 25216, 64, 64, 36, 12, 768, 2304
    assert lib.kanvit_layer_fwd_workspace(ctypes.byref(d)) == 0            # exact fp32 path needs no scratch
    d.flags = _lib.FLAG_BF16_MFMA
    assert lib.kanvit_layer_fwd_workspace(ctypes.byref(d)) >= 36 * 320 * 64 * 2   # bf16 repack of the weights
    d.flags = 0
    a = _lib.AttnDesc(B=2, H=3, N=300, D=64, scale=0.125)
    assert lib.kanvit_attn_fwd(ctypes.byref(a), None, None, None, None, None, None) == -22
    a.N, a.D = 197, 63
    assert lib.kanvit_attn_fwd(ctypes.byref(a), None, None, None, None, None, None) == -22
    a.D = 64
    # rowsum(dO*O) per row (16-byte rounded); N = 197 at D = 64 runs the one-kernel backward of round 4 (csrc/attention16.hip): no dS
    # hand-off.  Other lengths keep the dS spill of the exact fp32 path: [B*H][NP][NP], NP = N rounded to 32
    delta = (2 * 3 * 197 * 4 + 15) // 16 * 16
    assert lib.kanvit_attn_bwd_workspace(ctypes.byref(a)) == delta
    a.N = 160
    assert lib.kanvit_attn_bwd_workspace(ctypes.byref(a)) == (2 * 3 * 160 * 4 + 15) // 16 * 16 + 2 * 3 * 160 * 160 * 4
    a.N = 197
    a.flags = 1                                   # bf16 matrix-core mode hands dS over as bf16 (round 3): half the spill
    assert lib.kanvit_attn_bwd_workspace(ctypes.byref(a)) == delta + 2 * 3 * 224 * 224 * 2
    a.N = 300                                     # ... for N <= 256 (one query tile per wave of the dQ kernel); beyond, the recompute kernels
    assert lib.kanvit_attn_bwd_workspace(ctypes.byref(a)) == (2 * 3 * 300 * 4 + 15) // 16 * 16
    a.N = 197
    a.flags = 0


def test_switches_are_read_once_and_reported(lib, monkeypatch):
    """KANVIT_* switches: read at load, not per launch; kanvit_config() reports them; only an explicit reload re-reads."""
    from kanvit import _lib
    base = _lib.reload_config()
    assert "no_reg=0" in base and "no_bf16=0" in base and "attn_v1=0" in base
    monkeypatch.setenv("KANVIT_NO_REG", "1")
    assert _lib.active_config() == base                       # a changed environment alone changes nothing
    assert "no_reg=1" in _lib.reload_config()
    monkeypatch.delenv("KANVIT_NO_REG")
    assert _lib.reload_config() == base
    # the Python-side switches (kanvit/dense.py) follow the same rule and are part of the same report
    assert "py_ff=default" in base and "py_no_ff_small=0" in base and "py_no_lnff=0" in base and "py_no_ff_epi=0" in base
    monkeypatch.setenv("KANVIT_FF", "bf16x3")
    monkeypatch.setenv("KANVIT_NO_LNFF", "1")
    assert _lib.active_config() == base
    from kanvit import dense
    assert dense.ff_mode() == "fp32"
    changed = _lib.reload_config()
    assert "py_ff=bf16x3" in changed and "py_no_lnff=1" in changed and dense.ff_mode() == "bf16x3"
    monkeypatch.delenv("KANVIT_FF")
    monkeypatch.delenv("KANVIT_NO_LNFF")
    assert _lib.reload_config() == base
    assert "dbg" not in base.lower()                          # the ablation mask of round 1 is gone from the shipped library


def test_no_cpu_fallback():
    """CPU tensors must raise; nothing may quietly compute on the host."""
    from kanvit import KanvitError
    from models.cheby import ChebyKANLayer
    layer = ChebyKANLayer(4, 3, 4)
    with pytest.raises(KanvitError):
        layer(torch.randn(6, 4))
    from attention import MSA
    with pytest.raises(KanvitError):
        MSA(64, 2, type="vanilla")(torch.randn(2, 5, 64))


TYPES = ["vanilla", "flash-attn", "efficientkan", "sine", "fourier", "cheby", "fast"]


@pytest.mark.parametrize("geom", ["T", "C"])
@pytest.mark.parametrize("t", TYPES)
def test_reference_state_dict_loads(geom, t):
    """Keys / shapes of the drop-in modules equal the reference's (strict load of its state dict)."""
    from model import VisionTransformer
    blob = load_npz(f"model_{geom}_{t}.npz")
    c, h, w, npatch, nblk, d, heads, out_d = (int(v) for v in blob["cfg"])
    m = VisionTransformer((c, h, w), npatch, nblk, d, heads, out_d, type=t)
    sd = state_dict_from(blob)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert "pos_embeddings" not in m.state_dict()           # persistent=False like model.py:86-90


def test_host_helpers_match_reference():
    from model import VisionTransformer
    m = load_npz("misc.npz")
    vt = VisionTransformer((1, 28, 28), 7, 1, 64, 2, 10, type="vanilla")
    assert max_err(vt.pos_embeddings, T(m["pos_emb_50_64"])) < 1e-6
    vt2 = VisionTransformer((3, 8, 8), 2, 1, 16, 2, 10, type="vanilla")
    assert max_err(vt2.patchify(T(m["patchify_in"]), 2), T(m["patchify_out"])) == 0.0
    from models.sinekan import SineKANLayer
    s = SineKANLayer(32, 32, grid_size=4)
    assert max_err(s.phase, T(m["sine4.phase"])) < 1e-6
    assert np.allclose(s.freq.detach().reshape(-1).numpy(), [0.2, 0.4, 0.6, 0.8])
    from models.effkan import KANLinear
    k = KANLinear(1, 1)
    assert max_err(k.grid, T(m["bspline.grid"])) == 0.0
    for name, val in {"x0": 0.0, "x1": 1.0, "xm": -2.3, "xp": 2.2, "xh": 0.37}.items():
        assert max_err(k.b_splines(torch.tensor([[val]]))[0, 0], T(m["bspline." + name])) < 1e-6
    from models.fastkan import FastKANLayer
    f = FastKANLayer(4, 3)
    assert max_err(f.rbf.grid, T(m["rbf.grid"])) == 0.0 and abs(f.rbf.denominator - 4 / 7) < 1e-12


def test_init_statistics_follow_reference():
    """Initial distributions (not values) follow the reference constructors."""
    torch.manual_seed(0)
    from models.cheby import ChebyKANLayer
    from models.effkan import KANLinear
    from models.nfkan import NaiveFourierKANLayer
    c = ChebyKANLayer(64, 64, 4)
    assert abs(float(c.cheby_coeffs.std()) - 1 / (64 * 5)) < 2e-4
    f = NaiveFourierKANLayer(16, 64, grid_size=28)
    assert abs(float(f.fouriercoeffs.std()) - 1 / (4 * 28 ** 0.5)) < 2e-3
    k = KANLinear(32, 32)
    assert k.spline_weight.shape == (32, 32, 8) and float(k.spline_weight.abs().max()) < 0.2
    assert torch.isfinite(k.spline_weight).all()


def test_assembly_helpers_validate_arguments_without_a_gpu(lib):
    """kanvit_addln_* / kanvit_split3_bf16 reject bad shapes before touching the device (error code + message), return 0 for
    empty inputs, and report workspace sizes."""
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    # last dim must be a multiple of 4 in [4, 1024]
    for D in (3, 10, 1028):
        assert lib.kanvit_addln_fwd(4, D, 1e-5, p, None, p, p, None, p, p, p, None) != 0
        assert b"kanvit_addln_fwd" in lib.kanvit_last_error()
    assert lib.kanvit_addln_fwd(0, 64, 1e-5, p, None, p, p, None, p, p, p, None) == 0          # no rows: nothing to do
    assert lib.kanvit_addln_bwd_workspace(0, 64) == 0
    ws = lib.kanvit_addln_bwd_workspace(25216, 768)
    assert ws > 0 and ws % (2 * 768 * 4) == 0                                                    # whole [2][D] partials
    assert lib.kanvit_addln_bwd(4, 10, p, p, p, p, p, None, p, p, p, p, 1 << 20, None) != 0
    # split image: K must be a positive multiple of 8
    for K in (0, 4, 12):
        assert lib.kanvit_split3_bf16(4, K, p, None, 0, None, 0, p, 0, None) != 0
        assert b"kanvit_split3_bf16" in lib.kanvit_last_error()
    assert lib.kanvit_split3_bf16(0, 16, p, None, 0, None, 0, p, 0, None) == 0
    assert lib.kanvit_relu_bwd_bias(4, 6, p, p, p, p, p, 1 << 20, None) != 0          # N % 4
    assert b"kanvit_relu_bwd_bias" in lib.kanvit_last_error()
    assert lib.kanvit_relu_bwd_bias(4, 8, p, p, p, p, None, 0, None) != 0             # workspace missing
    assert lib.kanvit_relu_bwd_bias_workspace(25216, 3072) % 16 == 0 and lib.kanvit_relu_bwd_bias_workspace(0, 8) == 0


def test_patch_embed_weight_gradient_refuses_the_frequency_flag_on_other_families(lib):
    """KANVIT_FLAG_SINE_DFREQ asks for Q = sum_m dY x cos(.) instead of dW; a family that has no such operand must be refused, as
    kanvit_layer_bwd_weight does -- never handed plain dW back as if it were Q.  Host-side refusal: no GPU needed."""
    from kanvit import _lib, ops
    pd = _lib.PatchDesc(3, 224, 224, 14, 1, 0)
    d = ops._desc(ops.LayerCfg(_lib.CHEBY, 768, 768, 5, flags=_lib.FLAG_SINE_DFREQ), 2 * 196, 768, 768, 768, 0)
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.kanvit_patch_embed_bwd_weight(ctypes.byref(d), ctypes.byref(pd), p, None, p, p, None, 0, None)
    assert rc == -22 and b"SINE_DFREQ" in lib.kanvit_last_error()
    assert lib.kanvit_layer_bwd_weight(ctypes.byref(d), p, None, None, p, p, None, 0, None) == -22


def test_general_attention_refuses_a_non_positive_scale(lib):
    from kanvit import _lib
    a = _lib.AttnDesc(B=1, H=1, N=40, D=32, scale=-0.5)
    e = _lib.AttnExt(Nk=50)
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.kanvit_attn_x_fwd(ctypes.byref(a), ctypes.byref(e), p, p, p, p, p, None) == -22
    assert b"scale" in lib.kanvit_last_error()


def test_variant_build_script_has_no_source_list_of_its_own():
    """tools/build_variant.sh (A/B builds loaded through KANVIT_LIB) must compile exactly kanvit/build.py's SOURCES with its FLAGS:
    round 3's copy of the list had lost attention_x.hip and its library failed to load."""
    from kanvit import build
    sh = open(os.path.join(ROOT, "tools", "build_variant.sh")).read()
    assert "build.SOURCES" in sh and "build.FLAGS" in sh
    assert not any(s[:-4] + " " in sh for s in build.SOURCES)          # no hard-coded names left


def test_kernels_with_hand_counted_waits_use_no_scratch(lib):
    """The LDS-DMA weight gradient waits for its fills with explicit `s_waitcnt vmcnt(N)` (hipcc does not track LDS-DMA against ds_read):
    a scratch access inside its loop would count in vmcnt and let a wait pass before the fill has landed -- silent wrong data.  The code
    objects' own metadata (tools/kernel_meta.py: no GPU, no ROCm tool) must show no scratch at all for it; the same file backs DESIGN.md's
    zero-spill statements for the other kernels of round 4 (one SGPR spilled into a VGPR lane is not scratch)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_meta", os.path.join(ROOT, "tools", "kernel_meta.py"))
    km = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(km)
    from kanvit import build
    ks = {km.demangled_short(n): k for n, k in km.kernels(build.LIB).items()}
    assert len(ks) > 300
    dma = [n for n in ks if n.startswith("kan_bwd_weight_dma_kernel")]
    assert len(dma) == 2                                     # fp32 and bf16 ChebyKAN
    for n in dma:
        assert ks[n][".private_segment_fixed_size"] == 0 and ks[n][".vgpr_spill_count"] == 0 and ks[n][".sgpr_spill_count"] == 0, n
    round4 = [n for n in ks if n.startswith(("kan_bwd_input_res_bf16_kernel", "attn16_fwd_kernel", "attn16_bwd_kernel", "kan_fwd_ws_bf16_kernel"))]
    assert len(round4) >= 20
    for n in round4:
        assert ks[n][".private_segment_fixed_size"] == 0 and ks[n][".vgpr_spill_count"] == 0, n
