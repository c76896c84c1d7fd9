"""Pure data parallelism for the ViKANformer train step: one process per GPU, one gradient
all-reduce per step over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" on CPU for
the tests).  The reference has no multi-GPU path at all -- this is the exchange
step BASELINE.json's north_star adds.

Design for MI355X (8 GPUs fully connected, 7 xGMI links x ~153 GB/s per GPU): gradients live in a
few large flat fp32 buckets (default 64 MiB) so RCCL moves few, large messages; parameters' .grad
are VIEWS into the buckets (no gather/scatter copies); a bucket's all-reduce is launched from the
autograd hook of its last-ready parameter, so communication of the late layers' gradients overlaps
the backward of the early layers.  Buckets are filled in reverse registration order, which is the
order backward produces gradients in.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mib: float = 64.0,
                 group: Optional[dist.ProcessGroup] = None, overlap: bool = True, always_reduce: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always_reduce: issue the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.active = dist.is_initialized() and (self.world > 1 or always_reduce)
        self.overlap = overlap and self.active
        self.buckets: List[torch.Tensor] = []
        self._bucket_of = {}
        self._pending: List[int] = []
        self._handles = []
        cap = max(int(bucket_mib * (1 << 20)) // 4, 1)
        cur, cur_n = [], 0
        groups = []
        for p in reversed(self.params):                       # backward order
            if cur and cur_n + p.numel() > cap:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        for bi, plist in enumerate(groups):
            flat = torch.zeros(sum(p.numel() for p in plist), device=plist[0].device, dtype=plist[0].dtype)
            off = 0
            for p in plist:
                p.grad = flat[off:off + p.numel()].view_as(p)  # autograd accumulates in place into the view
                off += p.numel()
                self._bucket_of[p] = bi
            self.buckets.append(flat)
            self._pending.append(len(plist))
        self._counts = list(self._pending)
        if self.overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    # ---- per-step protocol: zero_grad() -> backward -> finish() -> optimizer.step() ----
    def zero_grad(self):
        for b in self.buckets:
            b.zero_()
        self._pending = list(self._counts)
        self._handles = []

    def _launch(self, bi: int):
        self._handles.append(dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, p):
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """Wait for (or, without overlap, perform) the all-reduces and turn sums into means."""
        if not self.active:
            return
        if not self.overlap:
            for bi in range(len(self.buckets)):
                self._launch(bi)
        else:
            for bi, left in enumerate(self._pending):          # parameters that got no gradient this step
                if left > 0:
                    self._launch(bi)
        for h in self._handles:
            h.wait()
        inv = 1.0 / self.world
        for b in self.buckets:
            b.mul_(inv)

    @property
    def grad_bytes(self) -> int:
        return sum(b.numel() * b.element_size() for b in self.buckets)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous, even split of a global batch; the first (global_batch % world) ranks get one more."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
