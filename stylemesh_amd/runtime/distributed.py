"""Multi-GPU protocol of the view-sharded path (SURVEY.md section 8 e): one process per GPU, views of the scene
shard over ranks, ONE exchange per step - a sum all-reduce of the flat texture-gradient arena, restricted to the
chunks the ranks' current views can touch (``SparseGradReducer``; RCCL over xGMI on the GPU box: ``torch.distributed``
backend "nccl"; "gloo" in the CPU tests) - then the identical fused update on every rank with
``grad_scale = 1 / world_size`` (DDP-mean semantics).

Nothing here touches the kernels; it is the host logic around them, testable without a GPU.
"""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def shard_views(indices, rank: int, world_size: int):
    """Rank r takes views r, r + R, r + 2R, ... of the scene's train list (keeps the per-rank consecutive-repeat
    structure of RepeatingSampler, so the per-view constant cache still hits)."""
    return list(indices)[rank::world_size]


def padded_shard(indices, rank: int, world_size: int):
    """``shard_views`` padded (by repeating the last view) to ceil(n / R) entries: every rank then takes the same
    number of steps and changes views at the same steps - every step contains a collective."""
    mine = shard_views(indices, rank, world_size)
    per_rank = -(-len(list(indices)) // world_size)
    if not mine:
        raise ValueError("more ranks than views")
    return mine + [mine[-1]] * (per_rank - len(mine))


def steps_per_epoch(n_views: int, index_repeat: int, world_size: int) -> int:
    """All ranks must take the same number of steps (every step contains a collective): ranks with one view
    fewer repeat their last view."""
    per_rank = -(-n_views // world_size)
    return per_rank * index_repeat


def rank_schedule(indices, rank, world_size, index_repeat):
    """View index per step for this rank, padded so that every rank has ``steps_per_epoch`` entries."""
    return [v for v in padded_shard(indices, rank, world_size) for _ in range(index_repeat)]


class ViewBatch(tuple):
    """The 13-tuple of the input contract (``view_contract.assemble_batch``) plus ``new_view``: True on the first step
    of every ``index_repeat`` block of the rank's schedule. The flag is a function of the schedule POSITION only, so
    it is the same on every rank at every step - a rank whose shard was padded by repeating its last view sees
    ``new_view`` although its view key did not change, and still enters the per-view collective
    (``SparseGradReducer.new_view``) that the other ranks are entering."""
    new_view = None

    def __new__(cls, items, new_view=None):
        self = super().__new__(cls, items)
        self.new_view = new_view
        return self


def scheduled_batches(get_view, indices, rank: int, world_size: int, index_repeat=1, repeat=True):
    """The rank's train schedule as ``ViewBatch``es: views ``padded_shard(indices)``, each ``index_repeat``
    consecutive times (``repeat=False``: once, the 'sequential' sampler mode). Consecutive repeats yield the SAME
    object (one decode / upload per view; callers may cache by identity), ``new_view`` marks the block boundaries."""
    rep = index_repeat if repeat else 1
    for i in padded_shard(indices, rank, world_size):
        n = rep[i] if isinstance(rep, list) else rep
        if n < 1:
            continue
        items = get_view(i)
        yield ViewBatch(items, new_view=True)
        rest = ViewBatch(items, new_view=False)
        for _ in range(n - 1):
            yield rest


class _ReduceOp:
    SUM, MAX = "sum", "max"


class _Work:
    """Handle of an asynchronous collective on the communicator's side stream: ``wait()`` orders the caller's current
    stream behind it (no host block), as torch.distributed's NCCL work objects do."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.event)


class RcclComm:
    """The product's own RCCL communicator (csrc/comm.hip: ``sm_comm_init`` = ncclCommInitRank, one per process /
    GPU), exposing the slice of the ``torch.distributed`` module interface the reducers use (``all_reduce`` with
    ``ReduceOp.SUM`` on fp32 / ``ReduceOp.MAX`` on int32, ``async_op``), so that ``SparseGradReducer`` runs unchanged
    over it. Synchronous collectives are enqueued on the CURRENT HIP stream - ordered with the scatter that wrote the
    gradient and the fused update that reads it, with no stream hop - asynchronous ones on a side stream.
    ``torch.distributed`` (any backend) only carries the 128-byte unique id at start-up."""
    ReduceOp = _ReduceOp

    def __init__(self, dist_module, rank: int, world_size: int, device):
        """COLLECTIVE over ``dist_module``: every rank runs the same sequence of ``torch.distributed`` calls whether or
        not its own part succeeds (the id broadcast carries an empty payload when rank 0 could not make one; the
        outcome of ``ncclCommInitRank`` is agreed on with a MIN all-reduce), so that a failure on SOME ranks ends in
        the same exception on ALL of them instead of mismatched collectives."""
        import ctypes
        import torch
        from . import hip
        self._hip, self._torch = hip, torch
        self.rank, self.world_size, self.device = rank, world_size, device
        self.handle, self._side = None, None
        nbytes = hip.lib.sm_comm_unique_id_bytes()
        buf = ctypes.create_string_buffer(nbytes)
        err = None
        if rank == 0:
            rc = hip.lib.sm_comm_get_unique_id(buf)
            if rc != 0:
                err = f"sm_comm_get_unique_id failed ({rc})"
        box = [bytes(buf.raw) if err is None else b""]
        if world_size > 1:
            dist_module.broadcast_object_list(box, src=0)
        if err is None and len(box[0]) != nbytes:
            err = "rank 0 could not create the RCCL unique id"
        handle = ctypes.c_void_p()
        if err is None:
            import contextlib
            with (torch.cuda.device(device) if device is not None else contextlib.nullcontext()):
                rc = hip.lib.sm_comm_init(ctypes.byref(handle), world_size, box[0], rank)
            if rc != 0:
                err, handle = f"sm_comm_init (ncclCommInitRank) failed ({rc})", ctypes.c_void_p()
        if world_size > 1:   # agree on the outcome
            ok = torch.tensor([0 if err else 1], dtype=torch.int32,
                              device=device if str(dist_module.get_backend()) == "nccl" else "cpu")
            dist_module.all_reduce(ok, op=dist_module.ReduceOp.MIN)
            if int(ok) == 0 and err is None:
                err = "another rank could not join the RCCL communicator"
        if err is not None:
            if handle:
                hip.lib.sm_comm_destroy(handle)
            raise RuntimeError(err)
        self.handle = handle

    def _launch(self, tensor, op):
        hip, torch = self._hip, self._torch
        assert tensor.is_cuda and tensor.is_contiguous()
        if op == _ReduceOp.SUM and tensor.dtype == torch.float32:
            hip.check(hip.lib.sm_allreduce_grad(self.handle, tensor.data_ptr(), tensor.numel(), hip.stream()),
                      "sm_allreduce_grad")
        elif op == _ReduceOp.MAX and tensor.dtype == torch.int32:
            hip.check(hip.lib.sm_allreduce_flags_max(self.handle, tensor.data_ptr(), tensor.numel(), hip.stream()),
                      "sm_allreduce_flags_max")
        else:
            raise ValueError(f"unsupported collective: {op} on {tensor.dtype}")

    def all_reduce(self, tensor, op=_ReduceOp.SUM, async_op=False):
        torch = self._torch
        if not async_op:
            self._launch(tensor, op)
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        ready = torch.cuda.Event()
        ready.record()
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            self._launch(tensor, op)
            done = torch.cuda.Event()
            done.record()
        tensor.record_stream(self._side)
        return _Work(done)

    def destroy(self):
        if self.handle:
            self._hip.lib.sm_comm_destroy(self.handle)
            self.handle = None


def make_comm(dist_module, rank: int, world_size: int, device, kind=None):
    """The collective provider of the gradient exchange. ``kind`` (default: env STYLEMESH_COMM, else 'rccl' when the
    process group's backend is nccl = RCCL): 'rccl' = the product's own communicator (``RcclComm``); 'torch' = the
    ``torch.distributed`` module itself (the only choice over gloo: CPU tests, two ranks sharing one GPU).
    The decision is COLLECTIVE: ``RcclComm`` either comes up on every rank or raises on every rank. When 'rccl' was
    asked for explicitly (argument or STYLEMESH_COMM=rccl) that exception propagates - the job fails loudly; when it
    was only the default, every rank prints the reason and uses ``torch.distributed`` (same interface)."""
    if world_size <= 1:
        return None
    explicit = kind or os.environ.get("STYLEMESH_COMM")
    kind = explicit
    if kind is None:
        kind = "rccl" if str(dist_module.get_backend()) == "nccl" else "torch"
    if kind not in ("rccl", "torch"):
        raise ValueError(f"STYLEMESH_COMM / kind must be 'rccl' or 'torch', not {kind!r}")
    if kind == "rccl":
        try:
            return RcclComm(dist_module, rank, world_size, device)
        except RuntimeError as e:
            if explicit == "rccl":
                raise
            import sys
            print(f"[stylemesh_amd] rank {rank}: own RCCL communicator unavailable on some rank ({e}); every rank uses "
                  "torch.distributed", file=sys.stderr)
    return dist_module


def make_grad_reducer(dist_module, world_size: int):
    """In-place SUM all-reduce of the gradient arena (None for a single rank)."""
    if world_size <= 1:
        return None

    def reduce(flat_grad):
        dist_module.all_reduce(flat_grad, op=dist_module.ReduceOp.SUM)
    return reduce


class SparseGradReducer:
    """SUM all-reduce of the gradient arena restricted to the chunks any rank's current view can touch.

    A view writes gradient only where its UV maps land (a few % - tens of % of a 4096^2 hierarchical texture), and
    which chunks those are depends only on the view, not on the step: ``new_view(flags)`` (a collective, once per
    view change - the ranks' schedules change views in lock-step, ``rank_schedule``) max-reduces the per-rank touch
    flags and keeps the index list; every step then gathers the flagged chunks into a compact buffer, all-reduces
    that, and copies the sums back. Untouched chunks are zero on every rank and stay zero. Falls back to the dense
    all-reduce when most chunks are dirty or the arena is not a whole number of chunks. Chunks are 64 floats (256 B):
    a view's footprint is a 2-D blob of scattered texels in row-major planes, so 4 KB chunks flag 49 % of the arena
    where 256 B chunks flag 14-24 % (exact non-zero fraction 8-11 %; bench views, tools/touch_fraction.py).
    """

    def __init__(self, dist_module, world_size: int, chunk_log2: int = 6, dense_above: float = 0.75, n_pieces: int = 4):
        self.dist, self.world, self.chunk_log2, self.dense_above = dist_module, world_size, chunk_log2, dense_above
        self.n_pieces = n_pieces
        self.idx = None
        self.cuts = None          # piece k exchanges idx[cuts[k]:cuts[k+1]] ...
        self.bounds = None        # ... and finishes the arena range [bounds[k], bounds[k+1])
        self.fraction = 1.0
        self.last_bytes = 0

    @property
    def chunk(self):
        return 1 << self.chunk_log2

    def n_chunks(self, numel: int) -> int:
        return -(-numel // self.chunk)

    def new_view(self, flags):
        """``flags``: int32 [n_chunks], non-zero where THIS rank's view can write. Collective."""
        self.dist.all_reduce(flags, op=self.dist.ReduceOp.MAX)
        self.idx = flags.nonzero().flatten()          # one host sync per view change
        self.fraction = self.idx.numel() / max(flags.numel(), 1)
        n_idx = self.idx.numel()
        k = max(1, min(self.n_pieces, n_idx))
        self.cuts = [n_idx * j // k for j in range(k + 1)]
        # arena range a piece completes: from its first flagged chunk up to the next piece's first flagged chunk
        # (everything between flagged chunks is zero on every rank and needs no exchange)
        firsts = self.idx[self.cuts[1:-1]].tolist() if k > 1 else []
        self.bounds = [0] + [f * self.chunk for f in firsts] + [None]

    def __call__(self, flat_grad):
        n = flat_grad.numel()
        if self.idx is None or self.fraction > self.dense_above or n % self.chunk != 0:
            self.dist.all_reduce(flat_grad, op=self.dist.ReduceOp.SUM)
            self.last_bytes = 4 * n
            return
        if self.idx.numel() == 0:      # no rank's view touches the texture: nothing to exchange (same on every rank)
            self.last_bytes = 0
            return
        g2 = flat_grad.view(-1, self.chunk)
        buf = g2.index_select(0, self.idx)
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
        g2.index_copy_(0, self.idx, buf)
        self.last_bytes = 4 * buf.numel()


    def pipelined(self, flat_grad, update_range):
        """Exchange and update overlapped: the flagged chunks are all-reduced in ``n_pieces`` asynchronous pieces
        and ``update_range(lo, hi)`` - the fused optimizer over arena elements [lo, hi) - is issued for a piece as
        soon as its sums have arrived, while the later pieces are still on the links. The ranges tile [0, n) in
        order; same arithmetic as ``__call__`` followed by one update over the whole arena. OPT-IN
        (``StepEngine.pipeline_exchange``, ``bench.py --pipeline-exchange``): verified functionally (2 ranks over gloo,
        CPU and one shared GPU), not yet timed on RCCL hardware - the 1-GPU boxes of this round cannot."""
        n = flat_grad.numel()
        dist = self.dist
        sparse = not (self.idx is None or self.fraction > self.dense_above or n % self.chunk != 0)
        if sparse and self.idx.numel() == 0:
            self.last_bytes = 0
            update_range(0, n)
            return
        if not sparse:
            k = self.n_pieces
            align = 1024
            bounds = sorted({min(n, (n * j // k) // align * align) for j in range(k)} | {0, n})   # no empty pieces
            works = [dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
                     for lo, hi in zip(bounds, bounds[1:])]
            for w, lo, hi in zip(works, bounds, bounds[1:]):
                w.wait()
                update_range(lo, hi)
            self.last_bytes = 4 * n
            return
        g2 = flat_grad.view(-1, self.chunk)
        buf = g2.index_select(0, self.idx)
        k = len(self.cuts) - 1
        works = [dist.all_reduce(buf[self.cuts[j]:self.cuts[j + 1]], op=dist.ReduceOp.SUM, async_op=True)
                 for j in range(k)]
        for j in range(k):
            works[j].wait()
            g2.index_copy_(0, self.idx[self.cuts[j]:self.cuts[j + 1]], buf[self.cuts[j]:self.cuts[j + 1]])
            update_range(self.bounds[j], n if self.bounds[j + 1] is None else self.bounds[j + 1])
        self.last_bytes = 4 * buf.numel()


def make_sparse_grad_reducer(dist_module, world_size: int, **kw):
    return None if world_size <= 1 else SparseGradReducer(dist_module, world_size, **kw)
