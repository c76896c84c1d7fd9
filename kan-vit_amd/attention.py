"""MSA and FlashAttention -- drop-ins for the reference's attention.py:13-202.

MSA keeps the reference's parameter layout ({q,k,v}_mappings.<head>.<layer keys>) but replaces the
python loop over samples x heads (attention.py:188-202) by two kernel launches per block: one fused
KAN launch that evaluates all 3*H per-head mappings on the batch-folded rows, and one attention
launch over every (sample, head)."""
import torch
from torch import nn

from kanvit import grouped, ops
from models.cheby import ChebyKANLayer
from models.effkan import KANLinear
from models.fastkan import FastKANLayer
from models.sinekan import SineKANLayer
from utils import FlashAttentionFunction, default


class FlashAttention(nn.Module):
    """Bias-free q / kv / out projections around the attention core (attention.py:13-109).
    ``parallel`` / ``mixed_precision`` are dead or broken branches in the reference (SURVEY.md
    section 2) and are rejected here instead of being imitated."""

    def __init__(self, *, dim, heads=8, dim_head=64, causal=False, q_bucket_size=512, k_bucket_size=1024,
                 parallel=False, mixed_precision=False):
        super().__init__()
        if parallel or mixed_precision:
            raise NotImplementedError("parallel / mixed_precision are not part of the accelerated path; "
                                      "data parallelism is done per process (see train.py --dp)")
        self.heads = heads
        self.causal = causal
        self.parallel = parallel
        self.mixed_precision = mixed_precision
        inner = heads * dim_head
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)
        self.q_bucket_size = q_bucket_size
        self.k_bucket_size = k_bucket_size

    def forward(self, x, context=None, mask=None, q_bucket_size=None, k_bucket_size=None):
        qb = default(q_bucket_size, self.q_bucket_size)
        kb = default(k_bucket_size, self.k_bucket_size)
        h = self.heads
        context = default(context, x)
        q = self.to_q(x)
        k, v = self.to_kv(context).chunk(2, dim=-1)
        b, n, _ = q.shape
        # 'b n (h d) -> b h n d' as strided views; the kernel takes the strides as they are
        q, k, v = (t.view(b, t.shape[1], h, -1).permute(0, 2, 1, 3) for t in (q, k, v))
        out = FlashAttentionFunction.apply(q, k, v, mask, self.causal, qb, kb)
        return self.to_out(out.permute(0, 2, 1, 3).reshape(b, n, -1))


class MSA(torch.nn.Module):
    """Multi-head self-attention with one small (KAN or linear) mapping per head and per q/k/v,
    no output projection (attention.py:112-202)."""

    def __init__(self, d, n_heads=4, type: str = "vanilla"):
        super().__init__()
        self.d = d
        self.n_heads = n_heads
        assert d % n_heads == 0
        dh = d // n_heads
        makers = {
            "vanilla": lambda: nn.Linear(dh, dh),
            "flash-attn": lambda: nn.Linear(dh, dh),
            "fourier": lambda: nn.Linear(dh, dh),                       # attention.py:136: Fourier is patch-embed only
            "efficientkan": lambda: KANLinear(dh, dh),
            "fast": lambda: FastKANLayer(dh, dh),
            "sine": lambda: SineKANLayer(dh, dh, grid_size=4),
            "cheby": lambda: ChebyKANLayer(dh, dh, 4),
        }
        if type not in makers:
            # the reference prints and carries on half-built (attention.py:174-176); raise instead
            raise ValueError(f"{type} invalid. Please use a different argument.")
        make = makers[type]
        self.q_mappings = nn.ModuleList([make() for _ in range(n_heads)])
        self.k_mappings = nn.ModuleList([make() for _ in range(n_heads)])
        self.v_mappings = nn.ModuleList([make() for _ in range(n_heads)])
        self.d_head = dh
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, sequences):
        b, n, d = sequences.shape
        qkv = grouped.run_qkv(self.q_mappings, self.k_mappings, self.v_mappings, sequences.reshape(b * n, d))
        return ops.attention_packed(qkv.view(b, n, 3, self.n_heads, self.d_head), causal=False,
                                    scale=1.0 / (self.d_head ** 0.5))
