"""Data-parallel plumbing: one process per GPU, RCCL through torch.distributed (backend "nccl" is RCCL
on ROCm; "gloo" on CPU for tests).

Frames are independent and instance-bank state is per rank, so the only
exchange of a training step is the gradient all-reduce (reference: MMDistributedDataParallel,
apis/mmdet_train.py:97-102).  Here the gradients of all parameters live in ONE flat buffer
(``FlatGrads``: every ``p.grad`` is a view into it), so the exchange is a single large collective --
xGMI is point-to-point (7 links x ~153 GB/s per GPU): one 391 MB all-reduce keeps every link above its
latency floor where hundreds of small buckets would not -- optionally in bf16 (half the bytes).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def flat_offsets(params, align=64):
    """Start offset of every tensor in a flat buffer, each aligned to ``align`` elements (256 bytes for fp32:
    the GEMM kernels vector-load weights, and padding keeps gradient and parameter offsets identical)."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + align - 1) // align * align
    return offs, off


class FlatGrads:
    """All gradients of ``params`` as views into one contiguous buffer (offsets: ``flat_offsets``)."""

    def __init__(self, params, comm_dtype=None, align=64):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        ref = self.params[0]
        self.offsets, total = flat_offsets(self.params, align)
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
        self.comm_dtype = comm_dtype
        self._comm = torch.empty(total, dtype=comm_dtype, device=ref.device) if comm_dtype not in (None, ref.dtype) else None
        self._views = [p.grad for p in self.params]
        self._loose = []
        self.split = total     # element offset where the SECOND segment (gradients produced late in the backward) starts

    # ---- gradients autograd accumulates itself -----------------------------------------------------
    # With ``p.grad`` preset to a (zeroed) view, autograd's AccumulateGrad runs one ``grad += new`` kernel per
    # parameter and step: ~500 tiny launches for the parameters no kernel of ours writes in place (BatchNorm,
    # convolutions, embeddings, Scale ...).  For those, ``p.grad`` is set to None before backward (autograd then
    # just keeps the incoming tensor: no kernel) and ONE multi-tensor copy moves them into the flat buffer after.
    def loosen(self, inplace_ids):
        """Parameters whose id is not in ``inplace_ids`` get the None-then-gather treatment from now on."""
        self._loose = [(p, v) for p, v in zip(self.params, self._views) if id(p) not in inplace_ids]

    def set_split(self, first_late_param):
        """Two segments for an all-reduce that overlaps the backward: [0, split) holds the gradients the backward
        finishes first (the decoder's), [split, end) the rest (the image encoder's); ``first_late_param`` is the first
        parameter of the second segment (parameters are laid out in the order given to the constructor)."""
        for p, off in zip(self.params, self.offsets):
            if p is first_late_param:
                self.split = off
                return
        raise ValueError("parameter not in this buffer")

    def _segment(self, which):
        if which is None:
            return self._loose
        off_of = {id(p): off for p, off in zip(self.params, self.offsets)}
        early = [(p, v) for p, v in self._loose if off_of[id(p)] < self.split]
        return early if which == "early" else [(p, v) for p, v in self._loose if off_of[id(p)] >= self.split]

    def before_backward(self):
        for p, _ in self._loose:
            p.grad = None

    def after_backward(self, which=None):
        """Gather the gradients autograd accumulated itself into the flat buffer; ``which`` = "early" / "late" limits
        it to one segment (the early segment is gathered -- and can be reduced -- before the late backward has run)."""
        loose = self._segment(which)
        if not loose:
            return
        dst, src = [], []
        for p, view in loose:
            g = p.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                dst.append(view)
                src.append(g if g.dtype == view.dtype else g.to(view.dtype))
            p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)

    def zero(self):
        self.flat.zero_()

    def check_views(self, segment=None):
        """Autograd may replace .grad when it was set to None in between; re-attach if so.  ``segment`` ("early" / "late")
        limits the walk to one segment: the early one must be put right BEFORE its all-reduce starts on the side stream
        (a copy into a buffer that is being reduced would race with the collective)."""
        for p, off in zip(self.params, self.offsets):
            if segment is not None and (off < self.split) != (segment == "early"):
                continue
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                g = p.grad
                p.grad = self.flat[off:off + n].view_as(p)
                if g is not None:
                    p.grad.copy_(g)

    def all_reduce_mean(self, group=None, segment=None):
        """Average the flat gradient (or one segment of it: "early" = [0, split), "late" = [split, end)) over the ranks
        with one collective.  Runs on the CURRENT stream: the caller puts it on a side stream to overlap it with compute."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        world = dist.get_world_size(group)
        lo, hi = (0, self.flat.numel()) if segment is None else ((0, self.split) if segment == "early" else (self.split, self.flat.numel()))
        if hi <= lo:
            return
        part = self.flat[lo:hi]
        if self._comm is not None:
            wire = self._comm[lo:hi]
            wire.copy_(part)
            dist.all_reduce(wire, group=group)
            part.copy_(wire)
        else:
            dist.all_reduce(part, group=group)
        part.div_(world)


def broadcast_parameters(module, src=0, group=None):
    """Rank ``src``'s parameters and buffers to everyone (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def max_over_ranks(seconds, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
