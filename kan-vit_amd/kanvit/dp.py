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

import time
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    """Flat-bucket gradient all-reduce.

    Per-step protocol: zero_grad() -> scale_loss(loss).backward() -> finish() -> optimizer.step().

    Gradients are bound to the buckets LATE: zero_grad() drops every .grad, so autograd adopts each produced gradient
    without an accumulation kernel; when the last parameter of a bucket has its gradient (post-accumulate hook), the
    bucket is packed with one multi-tensor copy, every parameter's .grad is re-pointed at its slice of the bucket and the
    bucket's all-reduce is launched asynchronously.  The optimizer then reads the averaged gradients straight from the
    buckets.  (Binding early -- .grad pre-set to bucket views -- costs one add_ kernel per parameter per step: 470 tiny
    launches for ViT-B with per-head KAN mappings.)

    Sums become means WITHOUT a pass over the buckets (round 3 ran one mul_ per bucket after the waits: 5 launches and
    2 x 277 MB of traffic per ViT-B step on the critical path between the last all-reduce and Adam):
      * average="avg"  -- the collective itself averages (ReduceOp.AVG; RCCL only);
      * average="loss" -- the caller back-propagates scale_loss(loss) = loss / world, so the all-reduced SUM is the mean
                          (any backend; what the gloo tests run).  A step whose loss did not go through scale_loss()
                          falls back to ONE multi-tensor mul_ over all buckets in finish(), so a forgotten call costs
                          time, never correctness;
      * average="auto" (default) -- "avg" on the nccl (= RCCL) backend, "loss" elsewhere.

    Timing (`timing=True`, what bench.py's `comm` record reads): on RCCL every bucket's all-reduce is issued from a side
    stream that first waits for the bucket's pack, with one event in front of it and one behind it ON THAT STREAM (the
    collective runs on RCCL's own stream; the synchronous-form call makes the side stream wait for it, the host never
    blocks), and finish() brackets its waits with two events on the compute stream: the first pair is the all-reduce's
    own duration, the second the part of it that backward did not hide."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mib: float = 64.0,
                 group: Optional[dist.ProcessGroup] = None, overlap: bool = True, always_reduce: bool = False,
                 average: str = "auto", timing: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always_reduce: issue the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.active = dist.is_initialized() and (self.world > 1 or always_reduce)
        self.overlap = overlap and self.active
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        if average not in ("auto", "avg", "loss"):
            raise ValueError(f"average={average!r}: expected 'auto', 'avg' or 'loss'")
        if average == "auto":
            average = "avg" if self.backend == "nccl" else "loss"
        if average == "avg" and self.active and self.backend != "nccl":
            raise ValueError("average='avg' needs ReduceOp.AVG, which only the nccl (RCCL) backend implements")
        self.average = average
        self.timing = timing
        self.buckets: List[torch.Tensor] = []
        self._plists: List[List[torch.nn.Parameter]] = []
        self._views: List[List[torch.Tensor]] = []
        self._bucket_of = {}
        self._handles = []
        self._loss_scaled = False
        self._steps = 0
        self._ev_buckets = []                 # (bucket index, start event, end event) on the communication stream
        self._ev_exposed = []                 # (start, end) on the compute stream around finish()'s waits
        self._host_exposed_s = 0.0            # host-blocking backends (gloo): wall time finish() spent waiting
        self._host_reduce_s = 0.0
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
        self._on_gpu = bool(self.buckets) and self.buckets[0].is_cuda
        # RCCL: collectives go out from a side stream (see the class docstring); other backends keep async_op handles
        self._comm_stream = torch.cuda.Stream(self.buckets[0].device) if (self.active and self._on_gpu and self.backend == "nccl") else None
        self._done_events = []
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
        self._done_events = []
        self._loss_scaled = False

    def scale_loss(self, loss: torch.Tensor) -> torch.Tensor:
        """The tensor to call .backward() on.  average="loss": loss / world (the all-reduced sum of the ranks' gradients is
        then the global-batch mean); otherwise the loss itself."""
        self._loss_scaled = True
        if self.active and self.average == "loss" and self.world > 1:
            return loss * (1.0 / self.world)
        return loss

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
        if not self.active:
            return
        op = dist.ReduceOp.AVG if self.average == "avg" else dist.ReduceOp.SUM
        if self._comm_stream is not None:
            cs = self._comm_stream
            cs.wait_stream(torch.cuda.current_stream())          # the pack of this bucket
            with torch.cuda.stream(cs):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cs)
                dist.all_reduce(self.buckets[bi], op=op, group=self.group)      # enqueue only: RCCL's stream <-> cs
                if self.timing:
                    e1.record(cs)
                    self._ev_buckets.append((bi, e0, e1))
                done = torch.cuda.Event()
                done.record(cs)
            self._done_events.append(done)
        else:
            t0 = time.perf_counter()
            self._handles.append(dist.all_reduce(self.buckets[bi], op=op, group=self.group, async_op=True))
            self._host_reduce_s += time.perf_counter() - t0

    def _on_grad(self, p):
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """Bind/launch whatever the hooks have not and wait for the all-reduces.  The gradients are means afterwards."""
        for bi in range(len(self.buckets)):
            if not self._bound[bi]:                            # no overlap, or parameters that got no gradient this step
                self._launch(bi)
        if not self.active:
            return
        self._steps += 1
        if self._comm_stream is not None:
            cur = torch.cuda.current_stream()
            if self.timing:
                x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                x0.record(cur)
            for ev in self._done_events:
                cur.wait_event(ev)
            if self.timing:
                x1.record(cur)
                self._ev_exposed.append((x0, x1))
        else:
            t0 = time.perf_counter()
            for h in self._handles:
                h.wait()
            self._host_exposed_s += time.perf_counter() - t0
        if self.average == "loss" and self.world > 1 and not self._loss_scaled:
            torch._foreach_mul_(self.buckets, 1.0 / self.world)       # the loss did not go through scale_loss(): one launch

    @property
    def grad_bytes(self) -> int:
        return sum(b.numel() * b.element_size() for b in self.buckets)

    def reset_timing(self):
        self._ev_buckets, self._ev_exposed = [], []
        self._host_exposed_s = self._host_reduce_s = 0.0
        self._steps = 0

    def comm_summary(self) -> dict:
        """Per-step communication record for bench.py (call after a device synchronize).  all-reduce time is measured on
        the stream the collectives are issued from; exposed time is what the compute stream waited in finish()."""
        steps = max(self._steps, 1)
        out = {"backend": self.backend, "world": self.world, "average": self.average, "buckets": len(self.buckets),
               "bucket_bytes": [b.numel() * b.element_size() for b in self.buckets], "bytes_per_step": self.grad_bytes,
               "overlap": bool(self.overlap), "steps": self._steps}
        if self._comm_stream is not None and self._ev_buckets:
            per = [0.0] * len(self.buckets)
            for bi, e0, e1 in self._ev_buckets:
                per[bi] += e0.elapsed_time(e1)
            out["allreduce_ms_per_step"] = round(sum(per) / steps, 4)
            out["allreduce_ms_per_bucket"] = [round(v / steps, 4) for v in per]
            out["exposed_ms_per_step"] = round(sum(a.elapsed_time(b) for a, b in self._ev_exposed) / steps, 4)
            ar = out["allreduce_ms_per_step"]
            if ar > 0 and self.world > 1:      # ring all-reduce moves 2 (w-1)/w of the bytes over each rank's links
                out["bus_GB/s"] = round(2.0 * (self.world - 1) / self.world * self.grad_bytes / (ar * 1e-3) / 1e9, 1)
            out["timed_on"] = "events on the stream the collectives are issued from (RCCL side stream) / the compute stream"
        else:
            out["allreduce_ms_per_step"] = None
            out["exposed_ms_per_step"] = round(1e3 * (self._host_exposed_s + self._host_reduce_s) / steps, 4)
            out["timed_on"] = "host wall time in all_reduce()/wait() (host-blocking backend)"
        return out


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
