"""VisionTransformer / TransformerBlock -- drop-ins for the reference's model.py:14-169.

Same constructor, same state_dict keys; the KAN patch embedding and the per-head KAN q/k/v
mappings + attention run in the HIP kernels (through models/*.py and attention.py).  The
feed-forward block and the LayerNorms are stock dense ops (hipBLASLt / MIOpen through torch) --
they are not KAN kernels."""
import torch
import torch.nn as nn

from attention import MSA, FlashAttention
from kanvit import ops
from kanvit.dense import feed_forward, ff_mode, ff_small_supported, ln_feed_forward
from kanvit.ops import add_layernorm
from models.cheby import ChebyKANLayer
from models.effkan import KANLinear
from models.fastkan import FastKANLayer
from models.nfkan import NaiveFourierKANLayer
from models.sinekan import SineKANLayer


class TransformerBlock(nn.Module):
    """x + MSA(LN(x)), then x + FF(LN(x)) (model.py:31-37)."""

    def __init__(self, d_model, n_heads, feedforward_dim=128, attn_type="vanilla"):
        super().__init__()
        self.norm1 = nn.LayerNorm(d_model)
        self.attn = MSA(d_model, n_heads, type=attn_type)
        self.norm2 = nn.LayerNorm(d_model)
        self.ff = nn.Sequential(nn.Linear(d_model, feedforward_dim), nn.ReLU(inplace=True),
                                nn.Linear(feedforward_dim, d_model))

    def forward(self, x):
        x, f = self.run(x, None)
        return x + f

    def run(self, x, pending):
        """The block with its residual adds fused into the LayerNorms (kanvit.ops.add_layernorm: one pass instead of an
        add kernel + a LayerNorm kernel forward, one pass instead of three + an add backward).  `pending` is the previous
        block's feed-forward output that has not been added to the stream yet (None for the first block); returns
        (stream after the attention add, this block's feed-forward output) -- same arithmetic as model.py:31-37:
            x = x + MSA(LN1(x));  x = x + FF(LN2(x))."""
        x, h1 = add_layernorm(x, pending, self.norm1)
        if x.dim() == 3 and ff_small_supported(self.ff[0].in_features, self.ff[0].out_features):
            # small geometries: residual add + LN2 + feed-forward in one launch; ln_feed_forward itself falls
            # back to add_layernorm + feed_forward for inputs the fused kernel refuses (dtype, autocast, row count)
            return ln_feed_forward(x, self.attn(h1), self.norm2, self.ff[0], self.ff[2])
        # under bf16 autocast the stock feed-forward GEMMs take bf16 operands: LN2 writes its output as bf16 directly, and the bf16
        # feed-forward output comes back into the next block's add_layernorm as it is (no cast passes in either direction)
        ybf = (x.is_cuda and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 and ff_mode() != "bf16x3")
        x, h2 = add_layernorm(x, self.attn(h1), self.norm2, y_bf16=ybf)
        # Same three ops as self.ff (Linear -> ReLU(inplace) -> Linear, model.py:25-29), applied to the 2-D
        # (B*N, d) tensor: nn.Linear on 3-D input returns a VIEW, and an in-place ReLU on a view makes autograd
        # insert CopySlices (two full [B*N, 4d] copies per block in backward: 12 ms/step at ViT-B, B=128).
        # The two Linears go through kanvit.dense: stock GEMMs, but with the weight gradient split over tokens
        # (the unsplit library kernel fills 36 of 256 CUs: 2.2 ms -> 0.83 ms per call).
        b, n, d = x.shape
        return x, feed_forward(h2.reshape(b * n, d), self.ff[0], self.ff[2]).view(b, n, d)   # bias + ReLU in the GEMM epilogue


class _ClsToken(torch.autograd.Function):
    """out[:, 0] + pending[:, 0] (model.py:166-169 reads only the class token; the last block's feed-forward output is still
    pending, TransformerBlock.run).  The two inputs receive the SAME gradient -- g in row 0, zeros elsewhere -- so backward builds
    it once, as ONE padding launch (the small geometries count launches)."""

    @staticmethod
    def forward(ctx, out, pending):
        ctx.shape = out.shape
        ctx.pdtype = pending.dtype
        return out[:, 0] + pending[:, 0]

    @staticmethod
    def backward(ctx, g):
        full = torch.nn.functional.pad(g.unsqueeze(1), (0, 0, 0, ctx.shape[1] - 1))      # [B, N, d]: g in row 0, zeros below
        return full, (full if ctx.pdtype == full.dtype else full.to(ctx.pdtype))      # a bf16 feed-forward output under autocast


def _patch_embedding(kind, in_dim, d):
    if kind in ("vanilla", "flash-attn"):
        return nn.Linear(in_dim, d)
    if kind == "efficientkan":
        return KANLinear(in_dim, d)
    if kind == "sine":
        return SineKANLayer(in_dim, d, grid_size=28)
    if kind == "fourier":
        return NaiveFourierKANLayer(in_dim, d, grid_size=28)     # works here; crashes in the reference (D4)
    if kind == "cheby":
        return ChebyKANLayer(in_dim, d, 4)
    if kind == "fast":
        return FastKANLayer(in_dim, d)
    raise ValueError(f"Unknown transformer type: {kind}")


def split_types(kind: str):
    """'sine' -> ['sine'];  'sine,fourier' (or 'sine+fourier') -> ['sine', 'fourier']: a per-block list of layer types.
    The reference takes ONE global string (model.py:67-80); the list form is the build-only extension of SURVEY.md section
    8(d) cfg5 / BASELINE.json configs[4] ("SineKAN + FourierKAN mixed blocks"): block l uses types[l % len(types)] for its
    per-head q|k|v mappings, the patch embedding uses types[0].  Parameter names and shapes per block are exactly those of the
    single-type model of that block's type, so a mixed model's state_dict is a block-wise splice of reference state_dicts."""
    kinds = [k.strip() for k in kind.replace("+", ",").split(",") if k.strip()]
    if not kinds:
        raise ValueError(f"Unknown transformer type: {kind}")
    if len(kinds) > 1 and any(k in ("flash-attn",) for k in kinds):
        raise ValueError("'flash-attn' stacks bare FlashAttention modules (model.py:113-117) and cannot be mixed per block")
    return kinds


class VisionTransformer(nn.Module):
    def __init__(self, chw, n_patches=7, n_blocks=4, d_hidden=64, n_heads=2, out_d=10, type: str = "vanilla"):
        super().__init__()
        kinds = split_types(type)
        self.block_types = [kinds[l % len(kinds)] for l in range(n_blocks)]
        type = kinds[0]                      # the patch embedding (and the flash-attn switch) follow the first entry
        self.chw = chw
        self.n_patches = n_patches
        self.n_blocks = n_blocks
        self.n_heads = n_heads
        self.d_hidden = d_hidden
        assert chw[1] % n_patches == 0
        assert chw[2] % n_patches == 0
        self.patch_size = (chw[1] // n_patches, chw[2] // n_patches)
        self.input_d = int(chw[0] * self.patch_size[0] * self.patch_size[1])

        self.linear_mapper = _patch_embedding(type, self.input_d, d_hidden)
        self.v_class = nn.Parameter(torch.randn(1, d_hidden))
        self.register_buffer('pos_embeddings', self.positional_embeddings(n_patches ** 2 + 1, d_hidden),
                             persistent=False)
        if type == "flash-attn":
            self.blocks = nn.ModuleList([FlashAttention(dim=d_hidden, heads=n_heads) for _ in range(n_blocks)])
        else:
            self.blocks = nn.ModuleList([TransformerBlock(d_hidden, n_heads, feedforward_dim=4 * d_hidden,
                                                          attn_type=self.block_types[l]) for l in range(n_blocks)])
        self.mlp_head = nn.Sequential(nn.LayerNorm(d_hidden), nn.Linear(d_hidden, out_d))
        self._fused_embed = None       # None = not tried yet, else {bf16 autocast?: the fused patch-embedding kernel covers this model}

    def patchify(self, images, n_patches):
        """(B, C, H, W) -> (B, n_patches^2, C*ph*pw): patches row-major, each flattened in (C, ph, pw)
        order -- one strided view + copy instead of the reference's n_patches^2 slice copies
        (model.py:111-126)."""
        b, c, h, w = images.shape
        ph, pw = h // n_patches, w // n_patches
        t = images.reshape(b, c, n_patches, ph, n_patches, pw).permute(0, 2, 4, 1, 3, 5)
        return t.reshape(b, n_patches * n_patches, c * ph * pw)

    def positional_embeddings(self, seq_length, d):
        """pe[i, j] = sin(i / 10000^(j/d)) for even j, cos(...) for odd j, evaluated in float64 and
        stored as float32 (the values of model.py:128-140, without its O(N*d) python loop)."""
        i = torch.arange(seq_length, dtype=torch.float64).unsqueeze(1)
        j = torch.arange(d, dtype=torch.float64).unsqueeze(0)
        ang = i / torch.pow(torch.tensor(10000.0, dtype=torch.float64), j / d)
        even = (torch.arange(d) % 2 == 0).unsqueeze(0)
        return torch.where(even, torch.sin(ang), torch.cos(ang)).to(torch.float32)

    def _embed_fused(self, images):
        """patchify + patch-embedding KAN layer + class token + position embedding in ONE kernel launch (SURVEY.md section
        8(f)2; kanvit.ops.patch_embed), or None when the fused kernel does not cover this layer / geometry (then the
        three-step path below runs: same arithmetic).  Under bf16 autocast the same launch runs on the bf16 matrix cores."""
        lm = self.linear_mapper
        if isinstance(lm, nn.Linear) or not hasattr(lm, "kan_pack") or hasattr(lm, "kan_u"):
            return None
        # the cached per-model decision is per arithmetic mode: the fp32 and the bf16 kernels tile the patch width differently
        mode = bool(torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        if self._fused_embed is None:
            self._fused_embed = {}
        if self._fused_embed.get(mode) is False:
            return None
        # per-call conditions (they depend on the INPUT, so they never touch the cached per-model decision): the kernel wants
        # contiguous-able fp32 NCHW on the GPU at a 16-byte aligned address, and it produces no gradient for the images --
        # a caller that differentiates w.r.t. the input (saliency, adversarial examples) takes the three-step path below
        if (not images.is_cuda or images.dim() != 4 or images.dtype != torch.float32 or images.requires_grad
                or (images.is_contiguous() and images.data_ptr() % 16)):
            return None
        cfg = lm.kan_cfg()
        w, bp, bias = lm.kan_pack()
        try:
            out = ops.patch_embed(images, w.unsqueeze(0), cfg, None if bp is None else bp.unsqueeze(0),
                                  None if bias is None else bias.reshape(1, -1), self.v_class.reshape(-1),
                                  self.pos_embeddings[: self.n_patches ** 2 + 1], self.n_patches)
        except ops.KanvitError:
            if mode not in self._fused_embed:              # layer / geometry not covered: decided once, at the first forward of a mode
                self._fused_embed[mode] = False
                return None
            raise
        self._fused_embed[mode] = True
        return out

    def forward(self, images):
        out = self._embed_fused(images)
        if out is None:
            patches = self.patchify(images, self.n_patches)
            b, p, _ = patches.shape
            if isinstance(self.linear_mapper, nn.Linear):
                tokens = self.linear_mapper(patches)
            else:
                # ChebyKANLayer returns (B*P, d) like the reference's; restore (B, P, d)
                tokens = self.linear_mapper(patches).reshape(b, p, self.d_hidden)
            cls = self.v_class.unsqueeze(0).expand(b, -1, -1)
            out = torch.cat((cls, tokens), dim=1) + self.pos_embeddings[: p + 1]
        pending = None
        for blk in self.blocks:
            if isinstance(blk, TransformerBlock):
                out, pending = blk.run(out, pending)
            else:                                     # 'flash-attn' models stack bare FlashAttention modules (model.py:113-117)
                out = blk(out)
        if pending is not None:
            out = _ClsToken.apply(out, pending)   # only the class token feeds the head (model.py:166-169)
        else:
            out = out[:, 0]
        return self.mlp_head(out)
