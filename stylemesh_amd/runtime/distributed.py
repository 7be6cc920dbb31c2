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
