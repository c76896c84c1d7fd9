"""Pure data parallelism for the ViKANformer train step: one process per GPU, one gradient
all-reduce per step over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" on CPU for
the tests).  The reference has no multi-GPU path at all -- this is the exchange
step BASELINE.json's north_star adds.

Design for MI355X (8 GPUs fully connected, 7 xGMI links x ~153 GB/s per GPU): gradients live in a
few large flat fp32 buckets (default 64 MiB) so RCCL moves few, large messages; a bucket is packed
with one multi-tensor copy and its all-reduce launched from the autograd hook of its last-ready
parameter, so communication of the late layers' gradients overlaps the backward of the early
layers; afterwards the parameters' .grad ARE views into the buckets (the optimizer reads the
averaged gradients in place).  Buckets are filled in reverse registration order, which is the
order backward produces gradients in.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    """Flat-bucket gradient all-reduce.

    Per-step protocol: zero_grad() -> backward -> finish() -> optimizer.step().

    Gradients are bound to the buckets LATE: zero_grad() drops every .grad, so autograd adopts each produced gradient
    without an accumulation kernel; when the last parameter of a bucket has its gradient (post-accumulate hook), the
    bucket is packed with one multi-tensor copy, every parameter's .grad is re-pointed at its slice of the bucket and the
    bucket's all-reduce is launched asynchronously.  The optimizer then reads the averaged gradients straight from the
    buckets.  (Binding early -- .grad pre-set to bucket views -- costs one add_ kernel per parameter per step: 470 tiny
    launches for ViT-B with per-head KAN mappings.)"""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mib: float = 64.0,
                 group: Optional[dist.ProcessGroup] = None, overlap: bool = True, always_reduce: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always_reduce: issue the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.active = dist.is_initialized() and (self.world > 1 or always_reduce)
        self.overlap = overlap and self.active
        self.buckets: List[torch.Tensor] = []
        self._plists: List[List[torch.nn.Parameter]] = []
        self._views: List[List[torch.Tensor]] = []
        self._bucket_of = {}
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
            off, views = 0, []
            for p in plist:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                self._bucket_of[p] = bi
            self.buckets.append(flat)
            self._plists.append(plist)
            self._views.append(views)
        self._counts = [len(pl) for pl in self._plists]
        self._pending = list(self._counts)
        self._bound = [False] * len(self.buckets)
        for p in self.params:
            p.grad = None
        if self.overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        self._pending = list(self._counts)
        self._bound = [False] * len(self.buckets)
        self._handles = []

    def _bind(self, bi: int):
        """Pack bucket bi from the parameters' gradients (zeros where a parameter got none) and re-point .grad at it."""
        plist, views = self._plists[bi], self._views[bi]
        with torch.no_grad():
            src, dst = [], []
            for p, v in zip(plist, views):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    src.append(p.grad)
                    dst.append(v)
            if src:
                torch._foreach_copy_(dst, src)
            for p, v in zip(plist, views):
                p.grad = v
        self._bound[bi] = True

    def _launch(self, bi: int):
        self._bind(bi)
        if self.active:
            self._handles.append(dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, p):
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """Bind/launch whatever the hooks have not, wait for the all-reduces and turn sums into means."""
        for bi in range(len(self.buckets)):
            if not self._bound[bi]:                            # no overlap, or parameters that got no gradient this step
                self._launch(bi)
        if not self.active:
            return
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
