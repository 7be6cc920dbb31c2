"""The texture-optimisation step on the GPU: texture sample -> VGG -> style / content losses -> hand-written
backward -> atomic scatter into the texture gradient -> fused regulariser + Adam + clamp.

This is the MI355X restatement of ``TextureOptimizationStyleTransferPipeline.forward_with_loss`` + autograd
backward + ``Adam.step`` (reference model/model.py:143-327,387-401) for batch size 1. Python only sequences
kernel launches of ``libstylemesh_hip.so`` on the current stream; there is no host synchronisation inside a
step (the reference synchronises >= 6 times per step) - the only sync is one read-back
of the per-level mask sums when a NEW view arrives (the reference's empty-level filter, model/model.py:256-257).

Everything that depends only on the view (UV grids, level masks, angle / depth pixel weights, layer masks and
their counts, level factors, the content target's VGG features) is computed once per view in ``set_view`` and
reused for the 20-100 consecutive steps the reference's RepeatingSampler spends on it.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import torch

from . import ops
from .fmap import FMap
from .vgg import PRE_POOL, AmaxBook, LevelBuffers, VGGNet, depth_of, layer_hw

DEFAULT_STYLE_LAYERS = ['r11', 'r21', 'r31', 'r41', 'r51']        # content_and_style_losses.py:222
DEFAULT_CONTENT_LAYERS = ['r42']                                  # :223
DEFAULT_STYLE_WEIGHTS = [1e3 / n ** 2 for n in [64, 128, 256, 512, 512]]  # :226
LOSS_TYPES = ["tex_reg", "content", "style", "total"]             # model/model.py:18-22


@dataclass
class EngineConfig:
    tex_w: int = 512
    tex_h: int = 512
    hierarchical: bool = True
    n_layers: int = 4
    style_layers: list = field(default_factory=lambda: list(DEFAULT_STYLE_LAYERS))
    content_layers: list = field(default_factory=lambda: list(DEFAULT_CONTENT_LAYERS))
    style_weights: list = field(default_factory=lambda: list(DEFAULT_STYLE_WEIGHTS))
    content_weights: list = field(default_factory=lambda: [1])
    angle_threshold: float = 60
    style_pyramid_mode: str = "single"
    gram_mode: str = "current"
    use_angle_weight: bool = True
    use_depth_scaling: bool = True
    loss_weights: dict = field(default_factory=lambda: {"content": 0.0, "style": 0.0, "tex_reg": 0.0})
    tex_reg_weights: list | None = None
    learning_rate: float = 1.0
    decay_gamma: float = 0.1
    decay_step_size: int = 30

    def validate(self):
        if self.style_pyramid_mode not in ("single", "multi"):
            raise ValueError(f"Unsupported style_pyramid_mode: {self.style_pyramid_mode}")
        if self.gram_mode not in ("current", "average"):
            raise ValueError(f"Unsupported gram_mode: {self.gram_mode}")
        for l in self.style_layers + self.content_layers:
            depth_of(l)
            if not l.startswith("r") or l in PRE_POOL:
                raise ValueError(f"unsupported loss layer {l}: must be a conv output that does not feed a pool")
        if set(self.style_layers) & set(self.content_layers):
            raise ValueError("a layer cannot be both a style and a content layer")
        if len(self.style_weights) != len(self.style_layers) or len(self.content_weights) != len(self.content_layers):
            raise ValueError("one weight per style / content layer is required")

    def reg_weights(self):
        n = self.n_layers if self.hierarchical else 1
        if self.tex_reg_weights:
            if len(self.tex_reg_weights) != n:
                raise ValueError(f"Have {n} texture layers, but only {len(self.tex_reg_weights)} weights specified")
            return list(self.tex_reg_weights)
        w = [pow(2, n - i - 1) for i in range(n)]  # model/model.py:86-88
        w[-1] = 0
        return w


def trunk_stream(device):
    """A HIGH-priority HIP stream for the step's trunk (convs, pools, sampling, scatter, the closing update). The
    engine's side streams - style branches with thousands of microseconds of slack, the early half of the split update -
    are created at normal priority, so the hardware queues dispatch the trunk's workgroups first and the side work fills
    what is left: measured +2-4 % on the c3 step (165.8 against 159.6-162.4 views/s, same box, alternating runs).
    ``MiniTrainer.fit`` and ``bench.py`` make it the current stream (STYLEMESH_MAIN_PRIORITY=normal: keep the caller's)."""
    if os.environ.get("STYLEMESH_MAIN_PRIORITY", "high") != "high" or not torch.cuda.is_available():
        return None
    torch.cuda.synchronize(device)
    return torch.cuda.Stream(device=device, priority=-1)


class TextureArena:
    """Texture, gradient and Adam moments of all layers back to back in four flat fp32 arenas; the layers are
    [3,H_l,W_l] views (the reference's Parameter layout, texture.py:29-32)."""

    def __init__(self, W, H, n_layers, device, random_init=False):
        self.shapes = [(3, H // 2 ** i, W // 2 ** i) for i in range(n_layers)]
        sizes = [c * h * w for c, h, w in self.shapes]
        self.seg_end = []
        tot = 0
        for s in sizes:
            tot += s
            self.seg_end.append(tot)
        self.n = tot
        self.p = torch.rand(tot, device=device) if random_init else torch.zeros(tot, device=device)
        self.g = torch.zeros(tot, device=device)
        self.m = torch.zeros(tot, device=device)
        self.v = torch.zeros(tot, device=device)

    def views(self, flat):
        out, start = [], 0
        for shp, end in zip(self.shapes, self.seg_end):
            out.append(flat[start:end].view(shp))
            start = end
        return out


class _ViewLevel:
    """Per-view constants of one UV pyramid level."""
    pass


class _capture:
    """``torch.cuda.graph`` with Python's cyclic garbage collector paused: a collection INSIDE the capture region may run
    the destructor of an unrelated, no longer referenced ``CUDAGraph`` (another engine's), whose ``hipGraphDestroy`` is an
    error while a stream of the process captures in global mode - and ``~CUDAGraph`` turns that error into ``abort()``
    (found when the resident views changed the moment at which collections happen in the test suite)."""

    def __init__(self, g, **kw):
        self.ctx = torch.cuda.graph(g, **kw)

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        self.ctx.__enter__()          # (collects once itself, before the capture begins)
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was:
                gc.enable()


class StepEngine:
    MAX_UV_LEVELS = 8   # per-(level, style layer) bounds of the derivative matrices are laid out for this many levels
    N_SLOTS = 2         # buffer sets of per-view constants: the current view + the views prepared ahead. (3 = TWO views in
                        # preparation, measured on the dip schedule in round 4: 663 views/s either way - that schedule is
                        # bound by the ~210 kernel dispatches of a step + a preparation, not by a wait for the read-back)

    def __init__(self, cfg: EngineConfig, vgg_state: dict, device="cuda", random_init=False):
        cfg.validate()
        self.cfg, self.device = cfg, device
        self.vgg = VGGNet(vgg_state, device)
        n_layers = cfg.n_layers if cfg.hierarchical else 1
        self.arena = TextureArena(cfg.tex_w, cfg.tex_h, n_layers, device, random_init)
        self.layers = self.arena.views(self.arena.p)
        self.grads = self.arena.views(self.arena.g)
        self.loss_layers = list(cfg.style_layers) + list(cfg.content_layers)
        # layers that actually receive a loss gradient (a zero loss weight contributes an exactly-zero gradient:
        # its kernels are skipped); the VGG pass stops at the deepest of them
        w_style = float(cfg.loss_weights.get("style", 0.0))
        w_content = float(cfg.loss_weights.get("content", 0.0))
        self.injected = ([l for l in cfg.style_layers if w_style != 0.0]
                         + [l for l in cfg.content_layers if w_content != 0.0])
        self.deepest = max(self.injected, key=depth_of) if self.injected else None
        self.deepest_content = max(cfg.content_layers, key=depth_of) if cfg.content_layers else None
        self._bufs = {}            # (H, W) -> LevelBuffers with gradients
        self._content_bufs = {}    # (h, w) -> LevelBuffers without gradients (content target pass)
        self.targets = None        # targets[layer_index][pyramid_level] -> [C,C] device tensor
        self.view = None
        self.view_key = None
        self._last_batch = None
        self.step_count = 0
        self.epoch = 0
        # device scalars: [content, style] weighted loss accumulators, per-layer sum of squares of the texture
        # ... and, behind them in the same buffer (one fill zeroes all of it every step), the per-layer max |x| bounds
        # of the step's activations / gradients over all UV levels (operand scales of the fp16x2 conv kernels)
        # ... and one bound per (UV level <= 8, style layer) of the style-loss derivative matrices (fp16x2 Gram backward)
        AW = ops.AMAX_FLOATS   # a bound is 64 slots spaced 256 bytes apart: 16 KB (1.3 MB for all of them)
        self._step_scalars = torch.zeros(AW + (AmaxBook.N + self.MAX_UV_LEVELS * len(cfg.style_layers)) * AW, device=device)
        self.loss_buf = self._step_scalars[0:2]
        self.amax = AmaxBook(device, self._step_scalars[AW:AW + AmaxBook.N * AW])
        self._amax_d = self._step_scalars[AW + AmaxBook.N * AW:]
        self.sumsq = torch.zeros(n_layers, device=device)
        self._pbuf = {}            # persistent per-view buffers (fixed addresses), keyed (slot, name)
        self._slot = 0             # slot of the CURRENT view's constants
        self._wslot = 0            # slot set_view is writing (differs from _slot while the next view is being prepared)
        self._prepared = []        # views prepared ahead, next first: (view key, slot, per-view attributes, done event)
        self._prepare_request = []     # [(batch, ready event)]: prepare_view calls to run after the next begin_step
        self._pending_view = None      # viewplan.PendingView: a view whose read-back has not been waited for yet
        self._view_plans = {}          # (slot, view shape, active levels) -> viewplan.ViewPlan
        self._content_graphs, self._content_warm = {}, {}   # captured content-target passes, per content size
        # step programs (runtime/program.py): a small step's library calls recorded once per view slot and replayed with
        # one call. STYLEMESH_STEP_PROGRAM: 1 (default) / 0 / verify (every step runs recorded and is compared with the table)
        self.step_programs = os.environ.get("STYLEMESH_STEP_PROGRAM", "1")
        self._programs, self._prog_warm = {}, {}
        self.view_serial = 0           # bumped whenever a view becomes current (programs re-patch their list lengths)
        self._prog_run = None          # (program, key) whose update segment the coming optimizer_step replays
        self._prog_rec = None          # (recorder, key, index of the first update call) of a step being recorded
        self.program_replays = 0       # diagnostics
        self.fast_view = os.environ.get("STYLEMESH_FAST_VIEW", "1") != "0"
        self._plans = [None] * self.N_SLOTS       # scatter plan per slot
        self._slot_released = [None] * self.N_SLOTS   # event: the steps that read this slot's constants have been enqueued
        self._prep_stream = None
        self._pending_grad_zero = None
        self.prepare_ahead = os.environ.get("STYLEMESH_PREPARE_AHEAD", "1") != "0"
        self._graphs = {}          # view signature -> captured hipGraph of forward_backward
        self._graph_warm = {}      # view signature -> eager runs so far
        self._opt_graph = None
        self.use_graphs = False    # replay captured hipGraphs instead of re-launching ~190 kernels per step
        self.sparse_tiles = True   # run the VGG convs only on tiles that can influence the loss (runtime/sparsity.py)
        # optional: style branches (Gram -> loss -> Gram backward) of the non-deepest layers on a second HIP stream,
        # filling idle CUs at the tails of the conv launches. Measured +1.7 % on c3 only (the conv grids leave few
        # idle CUs and no LDS for co-resident blocks), so it is off by default.
        self.overlap_style = False
        self.group_losses = True       # fp16x2 mode: the loss phase as grouped launches over all levels and layers
        # style layers whose branch runs on a side stream beside the conv trunk (grouped loss phase only)
        self.side_style_layers = tuple(x for x in os.environ.get("STYLEMESH_SIDE_STYLE", "r11").split(",") if x)
        # (inline: the SAME launch sequence - fused Gram epilogues included - issued on the trunk's stream: what the PMC
        # passes profile, since counter collection serialises dispatches)
        self.side_streams = os.environ.get("STYLEMESH_SIDE_STREAMS", "1") != "0"
        self.side_inline = os.environ.get("STYLEMESH_SIDE_STREAMS", "1") == "inline"
        # WHERE in the forward pass the HBM-bound side work is forked (a conv output's name; 'head' = before the texture
        # sampling). The split update's early half and the early style branches (relu1_1: Gram kernels over the largest
        # planes) stream HBM at 3-4 TB/s: beside the head of the step - texture sampling, conv1_1, conv1_2, all
        # HBM-bound themselves - they doubled those kernels' durations (0.41 instead of 0.22 ms before the first MFMA
        # conv of the round-3 timeline); beside the deep, matrix-core-bound convs (conv3_x ...) they are nearly free.
        self.early_update_at = os.environ.get("STYLEMESH_EARLY_UPDATE_AT", "r31")
        self.early_style_at = os.environ.get("STYLEMESH_EARLY_STYLE_AT", "r31")
        self._adam_early_pending = False
        self._step_zeroed = False      # the step's accumulators were zeroed by _step_begin
        self._loss_tables = None       # (signature, Gram / style-loss / Gram-backward problem tables, slab keys)
        self._gram_bwd_ws = {}         # (C, level, layer) -> scratch of the derivative matrices' operand images
        self._side = None
        # the loss branches of the UV levels (5 x [Gram -> loss -> Gram backward] + content MSE each, ~25 small
        # launches per level) are independent: one HIP stream per level lets the small levels' latency-bound kernels
        # run beside the large level's instead of after them
        self.level_streams = True
        # texture scatter as a sorted gather over a per-view plan (csrc/scatter_plan.hip) instead of the tiled atomic
        # kernel: no atomics, bit-reproducible, ~4x faster per step; the plan (radix sort) costs ~1 ms per view
        self.planned_scatter = True
        self._scatter_plan = None
        self._scatter_levels = None
        self._grad_dirty = False   # does the gradient arena hold anything since the last fused update?
        self._lv_streams = []
        # N > 1, OPT-IN (STYLEMESH_PIPELINE_EXCHANGE=1, or =auto: whenever the ranks' views flag at least
        # STYLEMESH_PIPELINE_MIN_MB megabytes of the arena): all-reduce the gradient in pieces and update each arena range as
        # soon as its sums arrive. Same arithmetic as exchange-then-update (update by ranges:
        # test_adam_fused_by_ranges_equals_one_launch; 2-rank bit-identity: tests/test_round5_gpu.py), but only ever timed
        # over gloo with every rank on ONE GPU - and a pipelined step gives up the early half of the split update, whose gain
        # IS measured. So the default stays the plain exchange until an RCCL run on separate GPUs has timed both.
        self.pipeline_exchange = {"1": True, "0": False, "auto": "auto"}.get(
            os.environ.get("STYLEMESH_PIPELINE_EXCHANGE", "0"), False)
        self.pipeline_min_bytes = int(float(os.environ.get("STYLEMESH_PIPELINE_MIN_MB", "32")) * (1 << 20))
        # N > 1, OPT-IN with an ``OwnerAwareGradReducer``: only the chunks two or more ranks' views touch are exchanged before
        # the update; single-owner chunks are updated by their owner at once and reach the others in the background
        self.deferred_exchange = os.environ.get("STYLEMESH_DEFERRED_EXCHANGE", "0") == "1"
        self._world_size = 1
        self.view_tiles = None
        # Resident views (round 5; viewplan.ResidentView), OPT-IN (STYLEMESH_VIEW_CACHE_GB=<budget>): the per-view state of
        # up to view_cache_gb gigabytes of views stays in HBM after a view's first visit; a revisit copies it back into the
        # slot's buffers instead of recomputing it. One rank, the grouped view path, no hipGraph replay. Measured
        # (profiles/r05/resident_views.txt): +7.5 % where the view changes EVERY step and views recur; neutral on the
        # reference's multi-epoch schedules (index_repeat 20: the preparation already hides on a side stream, and the
        # first epoch pays for the copies) - hence off by default.
        self.view_cache_gb = float(os.environ.get("STYLEMESH_VIEW_CACHE_GB", "0"))
        self._resident, self._resident_bytes = {}, 0
        # Resident views are keyed (scene_id, view index): the index alone says nothing about WHICH dataset it indexes
        # (ADVICE r5). ``set_scene`` names the scene / datamodule the coming batches belong to (the trainer does; a caller
        # that feeds one scene never needs to) - a new identity forgets every kept view; so do ``set_style_image`` and
        # ``load_texture`` (masks and lists depend on neither, but an engine re-targeted that far starts clean).
        self.scene_id = None
        self.view_cache_hits = self.view_cache_misses = 0
        self._gram = {}            # C -> scratch (S0, S1, D0, D1)
        self._gram_clean = set()   # keys of _gram whose S0 / S1 slabs currently hold zeros
        self._gram_need = {}       # key -> slabs per mask
        self._gram_arena = None    # backing store of every S0 / S1
        self._hist = {}            # layer -> (ring [9,C,C], count) for gram_mode 'average'
        numel = [c * h * w for c, h, w in self.arena.shapes]
        lam = cfg.loss_weights.get("tex_reg", 0.0)
        rw = cfg.reg_weights() if cfg.hierarchical else [0.0]
        self.reg_active = lam > 0 and cfg.hierarchical   # model/model.py:163-171,264-267
        # d/dp [lam * w_i * mean(p_i^2)] = (2 lam w_i / N_i) p
        self.reg_coef = [2.0 * lam * w / n if self.reg_active else 0.0 for w, n in zip(rw, numel)]
        self.reg_loss_coef = [lam * w / n if self.reg_active else 0.0 for w, n in zip(rw, numel)]
        self._reg_loss_coef_dev = torch.tensor(self.reg_loss_coef, device=self.device)   # no per-step host copy
        ops.clamp_sumsq(self.arena.p, self.arena.seg_end, self.sumsq)
        # Ever-touched chunks of the arena (64 floats = 256 B, the granularity of the multi-GPU exchange): a texel of a
        # ZERO-initialised texture that no view has reached has p = g = m = v = 0 and an exactly-zero update, so the
        # fused update skips chunks no view has ever touched (sm_adam_fused). None = dense update (textures that start
        # non-zero: random_init, load_texture / from_tensor, a loaded optimizer state).
        self.touched_log2 = 6
        self.touched = None if random_init else torch.zeros(-(-self.arena.n // 64), dtype=torch.int32, device=device)
        self._view_flags = None        # the current view's chunks (same granularity)
        self._other_flags = None       # touched & ~view
        self._adam_early_done = None   # event: the update of the other chunks (issued at the head of the step) is done
        self.split_update = os.environ.get("STYLEMESH_SPLIT_UPDATE", "1") != "0"
        self._gram_fused = {}
        self.overlap_min_pixels = int(os.environ.get("STYLEMESH_OVERLAP_MIN_PIXELS", "400000"))
        self.sparse_update = True   # bench.py --dense-adam / tests switch it off

    def _new_side_stream(self):
        """Side streams carry work with slack (style branches joined many kernels later, the early half of the update), at
        normal priority - the trunk's stream is the high-priority one (``trunk_stream``)."""
        n_cus = int(os.environ.get("STYLEMESH_SIDE_CUS", "0"))
        if n_cus > 0:   # a queue confined to n_cus compute units (n_cus / 8 of every XCD): see sm_stream_create_cu_subset
            import ctypes
            out = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                ops.hip.check(ops.hip.lib.sm_stream_create_cu_subset(n_cus, ctypes.byref(out)), "sm_stream_create_cu_subset")
            return torch.cuda.ExternalStream(out.value, device=self.device)
        return torch.cuda.Stream(device=self.device)

    # ------------------------------------------------------------------ texture access
    def set_scene(self, identity):
        """The batches from now on belong to the scene / datamodule ``identity`` (anything hashable)."""
        if identity != self.scene_id:
            self.scene_id = identity
            self.forget_resident_views()

    def forget_resident_views(self):
        self._resident, self._resident_bytes = {}, 0

    def load_texture(self, layer_tensors):
        """``from_tensor`` semantics (texture.py:34-39,83-94) + the clamp every forward starts with."""
        self.forget_resident_views()
        for dst, src in zip(self.layers, layer_tensors):
            assert tuple(dst.shape) == tuple(src.shape), (dst.shape, src.shape)
            dst.copy_(src.to(self.device, torch.float32))
        self.touched = None   # arbitrary content: every texel takes part in the update from now on
        self.sumsq.zero_()
        ops.clamp_sumsq(self.arena.p, self.arena.seg_end, self.sumsq)

    @property
    def lr(self):
        """StepLR (model/model.py:397-399): lr * gamma ** (epoch // step_size)."""
        return self.cfg.learning_rate * self.cfg.decay_gamma ** (self.epoch // self.cfg.decay_step_size)

    # ------------------------------------------------------------------ buffers
    def _level_bufs(self, H, W) -> LevelBuffers:
        key = (H, W)
        if key not in self._bufs:
            self._bufs[key] = LevelBuffers(H, W, self.deepest, True, self.device)
        return self._bufs[key]

    def _grad_planes(self, H, W, skip):
        """Gradient planes of one level size: re-zeroed when a new view arrives."""
        out = []
        b = self._bufs.get((H, W))
        if b is not None:
            out += [g.buf for name, g in b.grad.items() if name not in skip]
        return out

    def _gram_scratch(self, key, n_slabs=1):
        """(S0, S1, D0, D1): partial-sum slabs [n_slabs, C, C] for both masks and the derivative matrices; one set
        per (C, level, layer) so that style branches can run concurrently. All S slabs live in ONE arena, so that a
        step zeroes them with a single fill."""
        cur = self._gram.get(key)
        if cur is None or cur[0].shape[0] < n_slabs:
            self._gram_need[key] = n_slabs
            self._rebuild_gram_arena()
        return self._gram[key]

    def _rebuild_gram_arena(self):
        total = sum(2 * n * k[0] * k[0] for k, n in self._gram_need.items())
        arena = torch.zeros(total, device=self.device)
        off = 0
        for k, n in self._gram_need.items():
            C = k[0]
            sz = n * C * C
            S0, S1 = arena[off:off + sz].view(n, C, C), arena[off + sz:off + 2 * sz].view(n, C, C)
            off += 2 * sz
            old = self._gram.get(k)
            D = old[2:] if old is not None else (torch.zeros(C, C, device=self.device), torch.zeros(C, C, device=self.device))
            self._gram[k] = (S0, S1) + tuple(D)
        self._gram_arena = arena
        self._gram_clean = set(self._gram)   # a fresh arena holds zeros

    def _reserve_gram_scratch(self, active, bufs):
        """Create the slabs of every (level, style layer) of the step BEFORE its branches fork onto their streams:
        growing the arena later would move slabs that queued kernels of another stream still use."""
        if float(self.cfg.loss_weights.get("style", 0.0)) == 0.0:
            return
        for lv, b in zip(active, bufs):
            for layer in self.cfg.style_layers:
                f = b.act[layer]
                self._gram_scratch((f.C, lv.index, layer), ops.gram_workspace_slabs(f.C, f.H, f.W))

    def _step_begin(self, reg=None):
        """Head of a training step, ONE launch: the regulariser loss of the current texture (from the sums of squares the
        previous update left, before this step's update overwrites them) and the zero fill of everything the step
        accumulates into. Returns ``loss_tensors()`` with this step's ``tex_reg`` (in ``reg``, a device float)."""
        reg = torch.empty(1, device=self.device) if reg is None else reg
        if self._can_graph():   # the captured step carries its own fills
            ops.step_begin(self.sumsq, self._reg_loss_coef_dev, reg, None, None)
            return {"content": self.loss_buf[0:1], "style": self.loss_buf[1:2], "tex_reg": reg}
        dirty = self._gram_arena is not None and len(self._gram_clean) != len(self._gram)
        ops.step_begin(self.sumsq, self._reg_loss_coef_dev, reg, self._step_scalars, self._gram_arena if dirty else None)
        if dirty:
            self._gram_clean = set(self._gram)
        self._step_zeroed = True
        return {"content": self.loss_buf[0:1], "style": self.loss_buf[1:2], "tex_reg": reg}

    def _zero_step_accumulators(self):
        """Everything a step accumulates into: the loss pair and - one fill over the arena - the Gram slabs of every
        (level, layer) the previous step added into (instead of two fills per masked-Gram call)."""
        if self._step_zeroed:      # ``_step_begin`` has just done it (the arena may have been created since: zeros)
            self._step_zeroed = False
            if self._gram_arena is None or len(self._gram_clean) == len(self._gram):
                return
        self._step_scalars.zero_()
        if self._gram_arena is not None and len(self._gram_clean) != len(self._gram):
            self._gram_arena.zero_()
            self._gram_clean = set(self._gram)

    # ------------------------------------------------------------------ style targets
    def set_style_image(self, style_image: torch.Tensor, num_levels=5):
        """``ContentAndStyleLoss.set_style_image`` (content_and_style_losses.py:273-286): Gram matrices of the
        VGG features of the reversed style-image pyramid (``image_pyramid``, :83-133)."""
        from .pyramid import image_pyramid_sizes
        self.forget_resident_views()
        img = style_image[0] if style_image.dim() == 4 else style_image
        img = img.to(self.device, torch.float32).contiguous()
        h, w = img.shape[1:]
        sizes = image_pyramid_sizes(h, w, list(range(num_levels)))
        self.style_pyramid_sizes = sizes
        deepest_style = max(self.cfg.style_layers, key=depth_of)
        cache = {}
        for s in sizes:
            if s in cache:
                continue
            b = LevelBuffers(s[0], s[1], deepest_style, False, self.device)
            ops.image_to_fmap(img, b.act["img"])  # bilinear (align_corners=False) resize of the ORIGINAL image
            self.vgg.forward(b)
            grams = []
            for layer in self.cfg.style_layers:
                f = b.act[layer]
                ones = FMap(1, f.H, f.W, self.device).from_dense(torch.ones(1, f.H, f.W))
                S = torch.zeros(ops.gram_workspace_slabs(f.C, f.H, f.W), f.C, f.C, device=self.device)
                n = ops.gram_masked(f, ones, None, S, None, amax_feat=b.amax["a:" + layer])
                grams.append(_mirror_tiles(S[:n].sum(0)) / float(f.H * f.W))
            cache[s] = grams
            del b
        self.targets = [{lvl: cache[s][li] for lvl, s in enumerate(sizes)} for li in range(len(self.cfg.style_layers))]
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ per-view constants
    def _persist(self, key, factory):
        """Per-view buffers live at FIXED device addresses (allocated once per shape, overwritten by every
        set_view): a captured hipGraph of the step stays valid across views."""
        key = (self._wslot, key)   # two slots: the current view's constants and the ones being prepared for the next view
        if key not in self._pbuf:
            self._pbuf[key] = factory()
        return self._pbuf[key]

    def _mark(self, name):
        """Host-time marker inside ``set_view`` (STYLEMESH_SETVIEW_TIMING=1: seconds between consecutive markers)."""
        if os.environ.get("STYLEMESH_SETVIEW_TIMING") != "1":
            return
        import time
        now = time.perf_counter()
        acc = self.__dict__.setdefault("set_view_marks", {})
        last = self.__dict__.get("_mark_last")
        if last is not None and name != "start":
            acc[name] = acc.get(name, 0.0) + now - last
        self._mark_last = now
        # caching-allocator misses (hipMalloc) per phase: a view whose temporaries do not fit the cached blocks stalls here
        n = torch.cuda.memory_stats().get("num_device_alloc", 0) if torch.cuda.is_available() else 0
        if name != "start" and n != self.__dict__.get("_mark_allocs", n):
            misses = self.__dict__.setdefault("set_view_alloc_misses", {})
            misses[name] = misses.get(name, 0) + n - self._mark_allocs
        self._mark_allocs = n

    VIEW_ATTRS = ("view", "view_consts", "view_tiles", "view_sig", "_scatter_plan", "_scatter_levels", "view_key",
                  "_last_batch", "_view_flags", "_other_flags", "_union_flags", "_pending_grad_zero", "_pending_view")

    @staticmethod
    def _batch_key(batch):
        return int(batch[8][0]) if torch.is_tensor(batch[8]) else batch[8]

    def prepare_view(self, batch, ready_event=None, urgent=False):
        """Compute the per-view constants of the NEXT view ahead of its first step, on a side stream, into the other
        buffer slot, while the current view's steps run: ``set_view`` of that view then only waits for an event and swaps
        the slot in. A view change costs 1.6 ms (single level) to 4.6 ms (four levels) of GPU work plus a host read-back
        that drains the launch queue - 0.08 to 0.23 ms per step at 20 steps per view, and on the CLI path the dominant
        cost of a single-level view. ``ready_event``: recorded after the batch's upload (a copy stream). No-op with
        hipGraph replay (captured pointers) and for a view that is already current / prepared."""
        if not self.prepare_ahead or self.use_graphs or self.targets is None or self.view is None:
            return False
        key = self._batch_key(batch)
        if key == self.view_key or any(p[0] == key for p in self._prepared):
            return False
        # a free slot: not the current view's, not one a prepared view waits in
        busy = {self._slot} | {p[1] for p in self._prepared}
        free = [k for k in range(self.N_SLOTS) if k not in busy]
        if not free:
            return False
        if self._prep_stream is None:
            self._prep_stream = self._new_side_stream()
        slot = free[0]
        st = self._prep_stream
        if urgent:
            # The view is needed by the very NEXT step (index_repeat 1): its read-back must arrive while the current step
            # still runs, or the GPU idles while the host enqueues the next one. On a stream of the trunk's priority the
            # preparation's small kernels are dispatched beside the step's instead of behind them.
            if getattr(self, "_prep_stream_hi", None) is None:
                self._prep_stream_hi = torch.cuda.Stream(device=self.device, priority=-1)
            st = self._prep_stream_hi
            if getattr(self, "_prep_last", None) is not None:
                st.wait_event(self._prep_last)       # (the two preparation streams share scratch)
        if self._slot_released[slot] is not None:      # the steps of the view that used this slot are behind this point
            st.wait_event(self._slot_released[slot])
        if ready_event is not None:
            st.wait_event(ready_event)
        current = {a: getattr(self, a, None) for a in self.VIEW_ATTRS}
        self._wslot = slot
        self._scatter_plan = self._plans[slot]
        try:
            with torch.cuda.stream(st):
                self._set_view_body(batch, None, defer=True)
                done = torch.cuda.Event()
                done.record(st)
            self._plans[slot] = self._scatter_plan
            self._prep_last = done
            self._prepared.append((key, slot, {a: getattr(self, a, None) for a in self.VIEW_ATTRS}, done))
        finally:
            self._wslot = self._slot
            for a, v in current.items():
                setattr(self, a, v)
        return True

    def drop_prepared(self):
        """Forget a view prepared ahead (or asked for) that will not be the next one - an epoch cut short, a schedule that
        changed: its launches are ordered before whatever the current stream does next, its slot is free again."""
        preps, self._prepared, self._prepare_request = self._prepared, [], []
        # the preparation kernels read the batch's own tensors (no staging copies): keep every dropped view referenced
        # until its launches have completed
        self._dropped = [d for d in getattr(self, "_dropped", []) if not d[3].query()] + preps
        for prep in preps:
            torch.cuda.current_stream().wait_event(prep[3])

    def set_view(self, batch, reducer=None):
        """Make ``batch`` the current view: swap in the constants ``prepare_view`` computed ahead if they are this
        view's, else compute them now on the current stream (``_set_view_body``)."""
        main = torch.cuda.current_stream()
        old_slot = self._slot
        if self._prepared and reducer is None and self._prepared[0][0] == self._batch_key(batch):
            key, slot, attrs, done = self._prepared.pop(0)
            self.prepared_swaps = getattr(self, "prepared_swaps", 0) + 1   # diagnostics
            main.wait_event(done)
            for a, v in attrs.items():
                setattr(self, a, v)
            self._last_batch = batch
            self._slot = self._wslot = slot
            self._finish_pending_view(batch)      # (the read-back of a view prepared ahead is waited for HERE)
        else:
            for prep in self._prepared:     # prepared views that are not the one asked for: let their launches finish first
                main.wait_event(prep[3])    # (they share scratch with a build on this stream)
            self._prepared = []
            self._scatter_plan = self._plans[self._slot]
            self._wslot = self._slot
            self._set_view_body(batch, reducer)
            self._plans[self._slot] = self._scatter_plan
        self._activate_view()
        if self._slot != old_slot:          # everything that read the old slot's constants is enqueued before this point
            ev = torch.cuda.Event()
            ev.record(main)
            self._slot_released[old_slot] = ev

    def _activate_view(self):
        """Side effects of a view becoming current that must be ordered with the STEPS (main stream): gradient planes
        zeroed outside the new view's active tiles, its chunks OR-ed into the ever-touched flags."""
        self.view_serial += 1
        if self._pending_grad_zero:
            torch._foreach_zero_(self._pending_grad_zero)
        self._pending_grad_zero = None
        if self.touched is not None and self._view_flags is not None:
            ops.flags_or(self.touched, self._view_flags)
            self._other_flags = None   # ever-touched and not in this view: built on first use

    def _fast_view_ok(self):
        """The grouped per-view path (``runtime/viewplan.py``: two library calls instead of ~60 per UV level) covers the
        default configuration; the A/B switches of the list format and dense passes keep the call-per-layer path."""
        return (self.fast_view and self.sparse_tiles and self.deepest is not None
                and os.environ.get("STYLEMESH_SEGMENT_LISTS", "1") != "0"
                and os.environ.get("STYLEMESH_SEGMENT_STARTS", "free") == "free")

    def _set_view_fast(self, batch, reducer=None, defer=False, active_override=None, collective=True):
        """``_set_view_body`` through ``viewplan.ViewPlan``: same buffers, same results, two grouped library calls.
        ``defer``: leave the read-back pending (``_finish_pending_view`` waits for it - a view prepared one ahead: at the
        swap). ``active_override``: the levels to treat as non-empty (second pass after an empty level was found)."""
        from .viewplan import PendingView, ViewPlan
        cfg, dev = self.cfg, self.device
        self._mark("start")
        self._pending_grad_zero = None
        rgb, _, _, _, _, rounded, other, interp_w, idx, uv_map, mask, angle_guidance, angle_degrees = batch
        if rgb.shape[0] != 1:
            raise ValueError("batch size 1 only (the reference's masked_features indexing requires it too)")
        h, w = rgb.shape[2:]
        n_levels = len(uv_map)
        if n_levels > self.MAX_UV_LEVELS:
            raise ValueError(f"{n_levels} UV levels: at most {self.MAX_UV_LEVELS} are supported (per-level operand bounds)")

        fixed = self.use_graphs     # a captured step reads the view's inputs at fixed addresses

        def stage(name, src, dt=torch.float32):
            """The input where the kernels can read it: the batch's own tensor when it already is a contiguous device
            tensor of the right type (the batch object is kept alive with the view), else a copy in a persistent buffer."""
            if src.dtype == torch.bool and dt == torch.uint8:
                src = src.view(torch.uint8)      # (same bytes: 0 / 1)
            if not fixed and src.is_cuda and src.dtype == dt and src.is_contiguous() and src.data_ptr() % 16 == 0:
                return src
            dst = self._persist((name, tuple(src.shape), dt), lambda: torch.empty(src.shape, dtype=dt, device=dev))
            dst.copy_(src, non_blocking=True)
            return dst
        mask_u8 = stage("mask", mask[0], torch.uint8)
        ag, adeg = stage("ag", angle_guidance[0, 0]), stage("adeg", angle_degrees[0, 0])
        rgb_dev = stage("rgb", rgb[0])
        r64 = o64 = iw = None
        if cfg.use_depth_scaling:
            r64, o64 = stage("rounded", rounded[0, 0], torch.int64), stage("other", other[0, 0], torch.int64)
            iw = stage("interp_w", interp_w[0, 0])
        fixed = True       # the UV grids are read by every STEP (sampling, scatter plan): fixed addresses per slot, so that
        grids = [stage(f"grid{i}", uv[0]) for i, uv in enumerate(uv_map)]   # a recorded step serves every view of the slot
        level_hw = tuple(tuple(g.shape[:2]) for g in grids)
        maps_levels = list(range(n_levels)) if cfg.use_depth_scaling else [n_levels - 1]
        vkey = self._batch_key(batch)
        res = None     # the view's resident state (a revisit), when it was kept and still fits this engine's configuration
        if active_override is None and reducer is None and self.view_cache_gb > 0 and not self.use_graphs:
            res = self._resident.get((self.scene_id, vkey))
        active = list(maps_levels) if active_override is None else list(active_override)
        if res is not None:
            active = list(res.active)
        if len({level_hw[a] for a in active}) != len(active):
            raise ValueError("two UV levels of the same resolution are not supported")
        pk = (self._wslot, h, w, level_hw, tuple(active), ops.CONV_MODE, tuple(self.injected))
        plan = self._view_plans.get(pk)
        if plan is None:
            plan = self._view_plans[pk] = ViewPlan(self, self._wslot, h, w, level_hw, maps_levels, active)
        if res is not None and res.plan_key != plan.cache_key:     # kept under another configuration: forget it
            self._resident_bytes -= res.nbytes
            del self._resident[(self.scene_id, vkey)]
            return self._set_view_fast(batch, reducer, defer, active_override, collective)
        self._mark("stage+plan")
        if res is None:
            # content target: VGG features of the captured image at its own resolution (losses :294); resized per level inside
            # sm_view_masks
            if plan.content_bufs is not None:
                ops.image_to_fmap(rgb_dev, plan.content_bufs.act["img"])
                self._content_pass(plan.content_bufs)
            plan.launch_masks(mask_u8, ag, adeg, r64, o64, iw)
        levels = []
        for i, rec in enumerate(plan.levels):
            lv = _ViewLevel()
            lv.grid, lv.H, lv.W, lv.index = grids[i], rec["H"], rec["W"], i
            lv.active = i in active
            for k in ("M", "pixel_weight", "passed"):
                if k in rec:
                    setattr(lv, k, rec[k])
            if lv.active:
                lv.masks, lv.counts, lv.factor = rec["masks"], rec["counts"], rec["factor"]
                if "content_target" in rec:
                    lv.content_target = rec["content_target"]
            levels.append(lv)
        self._mark("masks")
        if res is None:
            plan.launch_lists()
        self._mark("lists")
        self.view, self.view_consts = levels, plan.consts
        self.view_tiles, self.view_sig = None, None
        act = [lv for lv in levels if lv.active]
        count_dev = None
        if collective:        # (a second pass after an empty level keeps the union the first pass exchanged)
            self._union_flags = None
        if reducer is not None and collective:
            flags = self.touch_flags(reducer.chunk_log2, levels)
            count_dev = reducer.new_view_begin(flags)      # collective; flags = the union over the ranks, in place
            self._union_flags = flags
        if plan.lists_desc is not None and act:
            from . import vgg as _vgg
            skip = set(_vgg.POOL_OUTPUT) if (_vgg.FUSE_POOL_BWD and ops.CONV_MODE == "split2") else set()
            self._pending_grad_zero = [g for lv in act for g in self._grad_planes(lv.H, lv.W, skip)]
        self._scatter_levels = None
        want_scatter = bool(self.planned_scatter and act)
        if want_scatter and self._scatter_plan is None:
            self._scatter_plan = ops.ScatterPlan(self.grads, self.arena.g)
        if want_scatter:
            self._scatter_levels = [lv.index for lv in act]
        if self.touched is not None:
            self._view_flags = self._persist(("view_flags", self.touched.numel()), lambda: torch.zeros_like(self.touched))
        self.view_key = int(idx[0]) if torch.is_tensor(idx) else idx
        self._last_batch = batch
        if res is not None and (res.scatter_meta is not None) == want_scatter and res.has_flags == (self.touched is not None):
            # a resident view: its state copied back (the plan's outputs, the sorted scatter plan, the touch flags)
            from .viewplan import CachedPending
            res.restore(plan, self._scatter_plan if want_scatter else None, self._view_flags)
            self._pending_view = CachedPending(plan, res)
            self.view_cache_hits += 1
            self._mark("resident")
            if not defer:
                self._finish_pending_view(batch)
            return
        if res is not None:     # (kept without / with a scatter plan or flags this engine now wants / does not want)
            self._resident_bytes -= res.nbytes
            del self._resident[(self.scene_id, vkey)]
            return self._set_view_fast(batch, reducer, defer, active_override, collective)
        if want_scatter:
            self._scatter_plan.build([lv.grid for lv in act], [lv.pixel_weight for lv in act])
        self._mark("scatter_plan")
        if self.touched is not None:
            self._view_flags.zero_()
            for lv in act:   # the SAMPLED footprint (no pixel weights), see _set_view_body
                ops.tex_touch_flags(self.grads, self.arena.g, lv.grid, None, self._view_flags, self.touched_log2)
        self._pending_view = PendingView(plan, plan.read_back(), count_dev)
        self._pending_view.reducer = reducer if count_dev is not None else None
        self._mark("flags+readback")
        if not defer:
            self._finish_pending_view(batch)

    def _content_pass(self, cb):
        """The content target's VGG pass (losses :294): a dense forward over buffers that never move - after one eager run
        it is captured as a hipGraph and replayed (one launch instead of ~22: 0.4 ms of host time per view change)."""
        key = (cb.H, cb.W, ops.CONV_MODE)
        g = self._content_graphs.get(key)
        if (g is None and (os.environ.get("STYLEMESH_CONTENT_GRAPH", "1") == "0" or torch.cuda.is_current_stream_capturing()
                           or (ops.CONV_TIMER is not None and ops.CONV_TIMER.enabled))):
            return self.vgg.forward(cb)
        if g is None:
            if self._content_warm.get(key, 0) < 1:
                self._content_warm[key] = 1
                return self.vgg.forward(cb)
            cur = torch.cuda.current_stream()
            g = torch.cuda.CUDAGraph()
            # (the capture stream stays alive with the graph: the captured convs use the split-K scratch keyed by its
            # handle - ops.splitk_workspace -, and a later stream that inherited the handle would share that scratch)
            cap = torch.cuda.Stream(device=self.device)
            self._content_cap_streams = getattr(self, "_content_cap_streams", []) + [cap]
            cap.wait_stream(cur)
            with _capture(g, stream=cap):
                self.vgg.forward(cb)
            cur.wait_stream(cap)
            self._content_graphs[key] = g
        g.replay()

    def _finish_pending_view(self, batch):
        """Wait for the read-back of the current view's preparation and apply what depends on it: the lists' lengths
        (grid sizes), the reducer's chunk count, and the empty-level filter (model/model.py:256-257) - a level whose mask
        turned out empty (rare) is dropped by a second, synchronous pass without it."""
        pend, self._pending_view = self._pending_view, None
        if pend is None:
            return
        tiles, msums = pend.finish()
        plan = pend.plan
        if pend.reducer is not None:
            pend.reducer.new_view_end(int(pend.reducer_count))
        empty = [a for a in plan.active if not msums[a] > 0]
        if empty:
            keep = [a for a in plan.active if a not in empty]
            self._wslot = self._slot
            self._set_view_fast(batch, pend.reducer, defer=False, active_override=keep, collective=False)
            return
        self.view_tiles = tiles
        act = [lv for lv in self.view if lv.active]
        self.view_sig = (tuple((lv.index, lv.H, lv.W) for lv in act),
                         None if tiles is None else tuple(v[0].numel() for v in tiles.values()))
        self._keep_resident(pend, batch)
        self._mark("finish")

    def _keep_resident(self, pend, batch):
        """After a view's first (computed) preparation: keep its state in HBM for the next visit, budget permitting."""
        from .viewplan import PendingView, ResidentView
        if (not isinstance(pend, PendingView) or pend.reducer is not None or self.view_cache_gb <= 0 or self.use_graphs
                or self._union_flags is not None):
            return
        key = (self.scene_id, self._batch_key(batch))
        left = self.view_cache_gb * (1 << 30) - self._resident_bytes
        if key in self._resident or left <= 0:
            return
        self.view_cache_misses += 1
        try:
            r = ResidentView(pend.plan, self._scatter_plan if self._scatter_levels is not None else None,
                             self._view_flags if self.touched is not None else None, pend.plan.active, max_bytes=left)
        except MemoryError:
            return             # (would overshoot the budget: recomputed at its next visit)
        self._resident[key] = r
        self._resident_bytes += r.nbytes

    def _set_view_body(self, batch, reducer=None, defer=False):
        """Per-view constants of ``batch``. ``reducer`` (multi-GPU, a ``SparseGradReducer``; only when the per-view
        collective is due at this schedule position, see ``begin_step``): the max-all-reduce of the touch flags and the
        device-side compaction of the exchange's chunk list are enqueued here, and the list's length rides the ONE host
        read-back of this function - a view change costs no additional synchronisation on N > 1."""
        if self._fast_view_ok():
            return self._set_view_fast(batch, reducer, defer)
        cfg = self.cfg
        self._pending_view = None
        self._mark("start")
        self._pending_grad_zero = None
        rgb, _, _, _, _, rounded, other, interp_w, idx, uv_map, mask, angle_guidance, angle_degrees = batch
        dev = self.device
        if rgb.shape[0] != 1:
            raise ValueError("batch size 1 only (the reference's masked_features indexing requires it too)")
        h, w = rgb.shape[2:]
        n_levels = len(uv_map)
        if n_levels > self.MAX_UV_LEVELS:
            raise ValueError(f"{n_levels} UV levels: at most {self.MAX_UV_LEVELS} are supported (per-level operand bounds)")

        def stage(name, src, dt=torch.float32):
            dst = self._persist((name, tuple(src.shape), dt), lambda: torch.empty(src.shape, dtype=dt, device=dev))
            dst.copy_(src, non_blocking=True)
            return dst
        mask_u8 = stage("mask", mask[0], torch.uint8)
        ag, adeg = stage("ag", angle_guidance[0, 0]), stage("adeg", angle_degrees[0, 0])
        rgb_dev = stage("rgb", rgb[0])
        if cfg.use_depth_scaling:
            E = self._persist(("E", n_levels, h, w), lambda: torch.empty(n_levels, h, w, device=dev))
            Wt = self._persist(("Wt", n_levels, h, w), lambda: torch.empty(n_levels, h, w, device=dev))
            ops.level_masks(stage("rounded", rounded[0, 0], torch.int64), stage("other", other[0, 0], torch.int64),
                            stage("interp_w", interp_w[0, 0]), mask_u8, n_levels, E, Wt)
        else:
            maskf = self._persist(("maskf", h, w), lambda: torch.empty(h, w, device=dev))
            maskf.copy_(mask_u8)
        levels = []
        msums = self._persist(("msums", n_levels), lambda: torch.zeros(n_levels, device=dev))
        msums.zero_()
        for i, uv in enumerate(uv_map):
            lv = _ViewLevel()
            lv.grid = stage(f"grid{i}", uv[0])
            lv.H, lv.W = lv.grid.shape[:2]
            lv.index = i
            if not cfg.use_depth_scaling and i != n_levels - 1:
                lv.active = False   # all-zero mask (model/model.py:253-254)
                levels.append(lv)
                continue
            hw = (i, lv.H, lv.W)
            lv.M = self._persist(("M",) + hw, lambda: torch.empty(lv.H, lv.W, device=dev))
            want_pw = cfg.use_angle_weight or cfg.use_depth_scaling
            lv.pixel_weight = self._persist(("pw",) + hw, lambda: torch.empty(lv.H, lv.W, device=dev)) if want_pw else None
            lv.passed = self._persist(("passed",) + hw, lambda: torch.empty(lv.H, lv.W, dtype=torch.uint8, device=dev))
            ops.level_maps(E[i] if cfg.use_depth_scaling else maskf, Wt[i] if cfg.use_depth_scaling else None,
                           ag if cfg.use_angle_weight else None, adeg, float(cfg.angle_threshold), h, w, lv.H, lv.W,
                           lv.M, lv.pixel_weight, lv.passed, msums[i:i + 1])
            levels.append(lv)
        # Which levels are non-empty (model/model.py:256-257) is known only on the device. Everything below is
        # launched on the assumption that all of them are, and the mask sums ride along in the ONE host read-back at
        # the end (the tile-list lengths): the host reaches that sync with all of set_view's launches already queued
        # behind the previous view's steps, instead of draining the queue first and then launching into an idle GPU.
        for lv in levels:
            if hasattr(lv, "M"):
                lv.active = True
        self._mark("stage+level_maps")
        self._union_flags = None
        if reducer is not None:
            flags = self.touch_flags(reducer.chunk_log2, levels)
            count_dev = reducer.new_view_begin(flags)      # collective; flags = the union over the ranks, in place
            msums = torch.cat([msums, count_dev.to(torch.float32)])    # (a chunk count < 2^24 is exact in fp32)
            self._union_flags = flags
        sums = self._finish_view(levels, rgb_dev, msums)
        if reducer is not None:
            reducer.new_view_end(int(sums[-1]))
        if any(hasattr(lv, "M") and not sums[lv.index] > 0 for lv in levels):   # rare: an empty level -> redo without it
            for lv in levels:
                if hasattr(lv, "M"):
                    lv.active = bool(sums[lv.index] > 0)
            self._finish_view(levels, rgb_dev)
        self._mark("finish_view")
        self.view_key = int(idx[0]) if torch.is_tensor(idx) else idx
        self._last_batch = batch
        if self.touched is not None:
            # chunks this view's scatter (and its texture sampling) can reach: OR-ed into the ever-touched flags; the
            # split update (``_adam_early``) treats them apart from the rest
            self._view_flags = self._persist(("view_flags", self.touched.numel()), lambda: torch.zeros_like(self.touched))
            self._view_flags.zero_()
            # (the SAMPLED footprint - no pixel weights: the forward pass samples every pixel of an active level, also
            # the ones whose backward weight is zero, so the early half of the split update must not rewrite p there
            # while the sampling kernel reads it; the weighted flags - ``touch_flags`` - only size the exchange)
            for lv in self.view:
                if lv.active:
                    ops.tex_touch_flags(self.grads, self.arena.g, lv.grid, None, self._view_flags, self.touched_log2)
            # (OR-ed into the ever-touched flags when the view becomes current: ``_activate_view``)

    def _finish_view(self, levels, rgb_dev, msums=None):
        """Layer-resolution masks + counts + level factors (calculate_pyramid, losses :146-217) and the content
        target features, for levels whose ``M`` / ``passed`` maps are already on the device. ``msums`` (device, one
        mask sum per level): returned as a host list, read together with the tile-list lengths."""
        cfg, dev = self.cfg, self.device
        h, w = rgb_dev.shape[1:]
        active = [lv for lv in levels if lv.active]
        n_act = len(active)
        n_lv, n_ll = len(levels), len(self.loss_layers)
        consts = self._persist(("consts", n_lv, n_ll), lambda: torch.zeros(n_lv, n_ll, 4, device=dev))
        consts.zero_()   # [N_all, N_pass, N_fail, factor] per (level, loss layer)
        for lv in active:
            a = lv.index
            lv.masks, lv.counts, lv.factor = {}, {}, {}
            for k, layer in enumerate(self.loss_layers):
                hl, wl = layer_hw(layer, lv.H, lv.W)
                m = self._persist(("lmask", a, layer, hl, wl), lambda: FMap(3, hl, wl, dev))   # all, passed, failed
                ops.layer_masks(lv.M, lv.passed, lv.H, lv.W, hl, wl, m.channel_ptr(0), m.channel_ptr(1),
                                m.channel_ptr(2), consts[a, k, 0:3])
                lv.masks[layer], lv.counts[layer], lv.factor[layer] = m, consts[a, k, 0:3], consts[a, k, 3:4]
        for k, layer in enumerate(self.loss_layers):
            if n_act:
                sizes = [float(math.prod(layer_hw(layer, lv.H, lv.W))) for lv in active]
                ops.level_factors([lv.counts[layer] for lv in active], sizes, [lv.factor[layer] for lv in active])
        self._mark("fv:layer_masks")
        # content target: VGG features of the captured image, resized per level (losses :294, :176-177)
        if cfg.content_layers and n_act:
            key = (h, w)
            if key not in self._content_bufs:
                self._content_bufs[key] = LevelBuffers(h, w, self.deepest_content, False, dev)
            cb = self._content_bufs[key]
            ops.image_to_fmap(rgb_dev, cb.act["img"])
            self.vgg.forward(cb)
            for lv in active:
                lv.content_target = {}
                for layer in cfg.content_layers:
                    src = cb.act[layer]
                    hl, wl = layer_hw(layer, lv.H, lv.W)
                    dst = self._persist(("ctarget", lv.index, layer, hl, wl), lambda: FMap(src.C, hl, wl, dev))
                    ops.fmap_resize_bilinear(src, dst)
                    lv.content_target[layer] = dst
        self._mark("fv:content_target")
        self.view = levels
        self.view_consts = consts
        self.view_tiles = None
        sums_host = None
        if self.sparse_tiles and active and self.deepest is not None and all(lv.grid is not None for lv in active):
            from .sparsity import build_tile_lists, need_maps
            needs = [need_maps(lv.M, lv.H, lv.W, set(self.injected), self.deepest) for lv in active]
            shapes = tuple((lv.H, lv.W) for lv in active)
            from .viewplan import TileLists, resident_lists
            self.view_tiles = TileLists()
            dsts, srcs = [], []
            lists = build_tile_lists(needs, self.deepest, msums, resident=resident_lists())
            if msums is not None:
                lists, sums_host = lists
            self.view_tiles.quads = lists.quads
            for key, (lst, frac, cap) in lists.items():
                # fixed-address storage (a captured graph keeps the pointer); capacity = all tiles of the launch
                buf = self._persist(("tiles", key, shapes), lambda: torch.zeros(max(cap, 1), dtype=torch.int32, device=dev))
                self.view_tiles[key] = (buf[:lst.numel()], frac)
                if lst.numel():
                    dsts.append(buf[:lst.numel()])
                    srcs.append(lst)
            if dsts:
                torch._foreach_copy_(dsts, srcs)
            # gradient planes must be zero outside this view's active tiles: the previous view wrote elsewhere
            # (with the fused pool backward the gradients of the pools' input layers are never materialised)
            from . import vgg as _vgg
            skip = set(_vgg.POOL_OUTPUT) if (_vgg.FUSE_POOL_BWD and ops.CONV_MODE == "split2") else set()
            self._pending_grad_zero = [g for lv in active for g in self._grad_planes(lv.H, lv.W, skip)]
        self._mark("fv:tile_lists(sync)")
        # identifies the step's launch sequence (grid sizes depend on the tile lists)
        self.view_sig = (tuple((lv.index, lv.H, lv.W) for lv in active),
                         None if self.view_tiles is None else tuple(v[0].numel() for v in self.view_tiles.values()))
        self._scatter_levels = None
        if self.planned_scatter and active and all(lv.grid is not None for lv in active):
            if self._scatter_plan is None:
                self._scatter_plan = ops.ScatterPlan(self.grads, self.arena.g)
            self._scatter_plan.build([lv.grid for lv in active], [lv.pixel_weight for lv in active])
            self._scatter_levels = [lv.index for lv in active]
        if msums is not None and sums_host is None:
            sums_host = msums.tolist()   # no tile lists to read along with
        self._mark("fv:scatter_plan")
        return sums_host

    # ------------------------------------------------------------------ the step
    def forward_backward(self, accumulate_grad=True):
        """Forward + backward of the current view; the data-term gradient ACCUMULATES into the gradient arena
        (zeroed by the fused update), the weighted content / style losses into ``loss_buf``.
        ``accumulate_grad=False`` (validation): everything but the final texture scatter - the arena is untouched."""
        if self.view is None or self.targets is None:
            raise RuntimeError("set_style_image() and set_view() must be called first")
        cfg = self.cfg
        from .vgg import OUT_NAMES as vgg_names
        w_style = float(cfg.loss_weights.get("style", 0.0))
        w_content = float(cfg.loss_weights.get("content", 0.0))
        active = [lv for lv in self.view if lv.active]
        if not active or self.deepest is None:
            if self._adam_early_pending:
                self._adam_early()
            self._zero_step_accumulators()
            return
        # layer-major over the active UV levels: every conv layer is ONE grouped launch over all levels
        bufs = [self._level_bufs(lv.H, lv.W) for lv in active]
        self._reserve_gram_scratch(active, bufs)
        self._zero_step_accumulators()
        if len({(lv.H, lv.W) for lv in active}) != len(active):
            raise ValueError("two UV levels of the same resolution are not supported")
        if len(active) <= 8:
            ops.tex_sample_fwd_grouped(self.layers, [lv.grid for lv in active], [b.act["img"] for b in bufs])
        else:
            for lv, b in zip(active, bufs):
                ops.tex_sample_fwd(self.layers, lv.grid, b.act["img"])
        style_on = w_style != 0.0
        side_layers = [l for l in cfg.style_layers if l != self.deepest] if (style_on and self.overlap_style) else []
        if side_layers:
            if self._adam_early_pending:
                self._adam_early()
            # A style layer's branch (masked Gram -> loss + derivative matrices -> Gram backward into grad[layer])
            # only has to finish before the backward pass reaches that layer: fork it onto a side stream right
            # after the layer's forward conv and join right before the dgrad that consumes grad[layer].
            main = torch.cuda.current_stream()
            while len(self._lv_streams) < len(active):
                self._lv_streams.append(self._new_side_stream())
            done = {}

            def fork(layer):   # one side stream per UV level
                if layer not in side_layers:
                    return
                ev = torch.cuda.Event()
                ev.record(main)
                done[layer] = []
                for k, (lv, b) in enumerate(zip(active, bufs)):
                    st = self._lv_streams[k]
                    st.wait_event(ev)
                    with torch.cuda.stream(st):
                        self._style_terms(lv, b, cfg.style_layers.index(layer), layer, w_style)
                        e2 = torch.cuda.Event()
                        e2.record(st)
                        done[layer].append(e2)

            def join(layer):
                for e2 in done.pop(layer, []):
                    main.wait_event(e2)
            self.vgg.forward_group(bufs, self.view_tiles, on_layer=fork, amax=self.amax)
            injected = set(side_layers)
            for lv, b in zip(active, bufs):
                injected |= self._inject_losses(lv, b, w_style, w_content, only_layers={self.deepest} | set(cfg.content_layers))
            self.vgg.backward_group(bufs, injected - {self.deepest}, self.deepest, self.view_tiles, before_layer=join,
                                    amax=self.amax)
            for layer in list(done):
                join(layer)
        else:
            # Grouped loss phase (fp16x2 mode). Only the deepest layer's branch sits between the forward and the backward
            # pass; every other branch runs on a side stream beside the power-limited conv trunk and is joined right
            # before the data-gradient conv that consumes its gradient plane:
            #   * ``side_style_layers`` (relu1_1: HBM-bound Gram kernels over the largest planes) and the content terms fork
            #     right after their layer's forward conv;
            #   * the other style layers fork when the forward pass is done, deepest first - the order the backward pass
            #     needs them in.
            # (gram_mode 'average' keeps ONE history per style layer that the levels' Grams enter one after the other,
            # content_and_style_losses.py:319-323: the grouped launch serves it for a single level - every dip script)
            grouped = (self.group_losses and style_on and ops.GRAM_MODE == "split2"
                       and (cfg.gram_mode != "average" or len(active) == 1))
            self._gram_fused = {}   # style layers whose Gram backward this step's data-gradient convs take (EPI_GRAM)
            # (only for steps long enough to pay for the extra events and stream switches on the host: a single
            # 256 x 341 level is host-bound, its step 3 % slower with them)
            use_side = (grouped and self.side_streams and self._overlap_pays(active)
                        and not torch.cuda.is_current_stream_capturing())
            early = tuple(l for l in self.side_style_layers if l in cfg.style_layers and l != self.deepest) if use_side else ()
            late = tuple(sorted((l for l in cfg.style_layers if l not in early and l != self.deepest),
                                key=depth_of, reverse=True)) if use_side else ()
            side_content = tuple(l for l in cfg.content_layers if l != self.deepest) if (use_side and w_content != 0.0) else ()
            side_done = {}   # layer -> event: its gradient plane is complete
            side = bool(early or late or side_content)
            if side:
                main = torch.cuda.current_stream()
                if not self._lv_streams:
                    self._lv_streams.append(self._new_side_stream())
                st = self._lv_streams[0]
                last_early = max(early, key=depth_of) if early else None

                def on_side(work, done_layers):
                    if self.side_inline:
                        work()
                        return
                    ev = torch.cuda.Event()
                    ev.record(main)
                    st.wait_event(ev)
                    with torch.cuda.stream(st):
                        work()
                        done = torch.cuda.Event()
                        done.record(st)
                    for l in done_layers:
                        side_done[l] = done

                def content_terms(layer):
                    for lv, b in zip(active, bufs):
                        self._content_term(lv, b, cfg.content_layers.index(layer), layer, w_content)

                # the early style branches fork at ``early_style_at`` (never before their own layer exists)
                fork_at = last_early
                if early and self.early_style_at in vgg_names and depth_of(self.early_style_at) > depth_of(last_early) \
                        and depth_of(self.early_style_at) <= depth_of(self.deepest):
                    fork_at = self.early_style_at

                # relu1_1's Gram backward moves into the epilogue of conv1_2's data gradient (EPI_GRAM): its branch only
                # packs the derivative matrices
                fuse = tuple(l for l in early if self._can_fuse_gram_bwd(l, bufs))

                def fork(layer):
                    if layer == fork_at and early:
                        on_side(lambda: self._style_group(active, bufs, w_style, early, fuse), early)
                    if layer in side_content:
                        on_side(lambda: content_terms(layer), (layer,))

                def join(layer):
                    done = side_done.pop(layer, None)
                    if done is not None:
                        main.wait_event(done)
            upd_at = self.early_update_at if (self.early_update_at in vgg_names
                                              and depth_of(self.early_update_at) <= depth_of(self.deepest)) else None

            def on_layer(layer):
                if self._adam_early_pending and layer == upd_at:
                    self._adam_early()
                if side:
                    fork(layer)
            if self._adam_early_pending and upd_at is None:
                self._adam_early()
            self.vgg.forward_group(bufs, self.view_tiles, on_layer=on_layer if (side or self._adam_early_pending) else None,
                                   amax=self.amax)
            injected = set()
            concurrent = (self.level_streams and len(active) > 1 and cfg.gram_mode != "average"
                          and not torch.cuda.is_current_stream_capturing())
            start_bound = False
            if grouped:
                for l in late:   # deepest first: each branch has its own event
                    on_side(lambda l=l: self._style_group(active, bufs, w_style, (l,),
                                                          (l,) if self._can_fuse_gram_bwd(l, bufs) else ()), (l,))
                on_main = [l for l in cfg.style_layers if l not in early and l not in late]
                self._style_group(active, bufs, w_style, on_main)
                injected = set(cfg.style_layers)
                if w_content != 0.0:
                    for layer in cfg.content_layers:
                        if layer not in side_content:
                            for lv, b in zip(active, bufs):
                                self._content_term(lv, b, cfg.content_layers.index(layer), layer, w_content)
                        injected.add(layer)
                start_bound = (self.deepest in on_main and self.deepest not in cfg.content_layers
                               and ops.CONV_MODE == "split2")
            elif concurrent:
                main = torch.cuda.current_stream()
                while len(self._lv_streams) < len(active) - 1:
                    self._lv_streams.append(self._new_side_stream())
                fork = torch.cuda.Event()
                fork.record(main)
                joins = []
                # largest level stays on the main stream, the others fork
                order = sorted(range(len(active)), key=lambda k: -active[k].H * active[k].W)
                for n, k in enumerate(order[1:]):
                    st = self._lv_streams[n]
                    st.wait_event(fork)
                    with torch.cuda.stream(st):
                        injected = self._inject_losses(active[k], bufs[k], w_style, w_content)
                        ev = torch.cuda.Event()
                        ev.record(st)
                        joins.append(ev)
                injected = self._inject_losses(active[order[0]], bufs[order[0]], w_style, w_content)
                for ev in joins:
                    main.wait_event(ev)
            else:
                for lv, b in zip(active, bufs):
                    injected = self._inject_losses(lv, b, w_style, w_content)
            self.vgg.backward_group(bufs, injected - {self.deepest}, self.deepest, self.view_tiles, amax=self.amax,
                                    start_bound_recorded=start_bound, before_layer=join if side else None,
                                    gram_terms=self._gram_fused)
            for done in side_done.values():
                torch.cuda.current_stream().wait_event(done)
        if not accumulate_grad:
            return
        if self.planned_scatter and self._scatter_plan is not None and self._scatter_levels == [lv.index for lv in active]:
            # sorted gather over the plan of this view (built by set_view); texels are stored without being read
            # while the arena is known to be zero (the fused update zeroes it)
            self._scatter_plan.scatter([b.grad["img"] for b in bufs], accumulate=self._grad_dirty)
        else:
            for lv, b in zip(active, bufs):
                ops.tex_sample_bwd(self.grads, lv.grid, b.grad["img"], lv.pixel_weight)
        self._grad_dirty = True

    def _style_group(self, active, bufs, w_style, layers, fuse_layers=()):
        """The style branches (masked Gram -> loss value + derivative matrices -> Gram backward into ``grad[layer]``) of
        the given style ``layers`` over ALL active levels, as grouped launches on the current stream (fp16x2 mode): masked
        Grams (one launch per tile class), loss + derivative matrices (one), their operand images (one), Gram backward
        (one per tile class). The problem tables hold raw pointers: they are rebuilt whenever the level buffers, the
        per-view mask / count buffers or the weights change (those buffers are persistent, so consecutive views of the
        same level set reuse the tables)."""
        cfg = self.cfg
        layers = tuple(layers)
        if not layers:
            return
        l0 = cfg.style_layers[0]
        sig = (tuple((lv.index, b.act[l0].ptr, b.grad[l0].ptr, lv.masks[l0].ptr, lv.counts[l0].data_ptr())
                     for lv, b in zip(active, bufs)),
               w_style, ops.CONV_MODE, tuple(cfg.style_weights),
               tuple(t.data_ptr() for tl in self.targets for t in tl.values()),
               None if self._gram_arena is None else self._gram_arena.data_ptr(), cfg.style_pyramid_mode)
        if self._loss_tables is None:
            self._loss_tables = {}
        if sig not in self._loss_tables:
            if len(self._loss_tables) >= 4:          # (two view slots x the level sets in use; drop the oldest)
                self._loss_tables.pop(next(iter(self._loss_tables)))
            self._loss_tables[sig] = {}
        tables = self._loss_tables[sig]
        fuse_layers = tuple(l for l in fuse_layers if l in layers)
        layers_key = (layers, fuse_layers)
        tab = tables.get(layers_key)
        if tab is None:
            from . import hip
            multi = cfg.style_pyramid_mode == "multi"
            fwd, sty, bwd, keys = [], [], [], []
            fused = {l: [] for l in fuse_layers}   # layer -> per level (ws, mask0, mask1, amax_feat, amax_d) for EPI_GRAM
            for lv, b in zip(active, bufs):
                for layer in layers:
                    li = cfg.style_layers.index(layer)
                    f = b.act[layer]
                    key = (f.C, lv.index, layer)
                    S0, S1, D0, D1 = self._gram_scratch(key, ops.gram_workspace_slabs(f.C, f.H, f.W))
                    k = (lv.index * len(cfg.style_layers) + li) * ops.AMAX_FLOATS
                    ad = self._amax_d[k:k + ops.AMAX_FLOATS]
                    af = self.amax["a:" + layer]
                    m0, m1 = self._style_masks(lv, layer)
                    weight = w_style * float(cfg.style_weights[li])
                    if multi:
                        targets, term_mask = [self.targets[li][2], self.targets[li][2]], [0, 1]
                        if li > 2:   # content_and_style_losses.py:335-338
                            targets.append(self.targets[li][0])
                            term_mask.append(0)
                        counts, skip = lv.counts[layer][1:3], [0, 1]
                    else:
                        S1 = D1 = None
                        targets, term_mask, counts, skip = [self.targets[li][0]], [0], lv.counts[layer][0:1], [0, 0]
                    ws = self._gram_bwd_ws.get(key)
                    if ws is None:
                        ws = self._gram_bwd_ws[key] = torch.empty(ops.gram_backward_ws_bytes(f.C), dtype=torch.uint8,
                                                                  device=self.device)
                    fwd.append(ops.gram_problem(f, m0, m1, S0, S1, af))
                    sp = ops.style_problem(S0, S1, counts, lv.factor[layer], targets, term_mask, skip, weight, f.C, D0, D1, ad)
                    if cfg.gram_mode == "average":     # the layer's ring of 9 detached previous Grams (position set per step)
                        if layer not in self._hist:
                            self._hist[layer] = [torch.zeros(9, f.C, f.C, device=self.device), 0]
                        sp.history = self._hist[layer][0].data_ptr()
                    sty.append(sp)
                    # the deepest layer's gradient starts the backward pass: its bound is recorded here (unless a
                    # content term also writes that buffer)
                    rec = layer == self.deepest and layer not in cfg.content_layers and ops.CONV_MODE == "split2"
                    # a fused layer: only the operand images of D are packed here; the data-gradient conv that produces
                    # this layer's gradient adds the Gram backward in its epilogue (vgg.backward_group, EPI_GRAM)
                    bwd.append(ops.gram_bwd_problem(f, m0, m1, D0, D1, None if layer in fused else b.grad[layer], ws, af, ad,
                                                    relu_gate=(layer == self.deepest),
                                                    amax_out=self.amax["g:" + layer] if rec else None))
                    if layer in fused:
                        fused[layer].append((ws, m0, m1 if D1 is not None else None, af, ad))
                    keys.append(key)
            tab = tables[layers_key] = (ops.struct_array(hip.GramProblem, fwd),
                                        ops.struct_array(hip.StyleProblem, sty),
                                        ops.struct_array(hip.GramBwdProblem, bwd), keys, fused)
        fwd, sty, bwd, keys, fused = tab
        self._gram_fused.update(fused)
        assert all(k in self._gram_clean for k in keys), "Gram slabs must be zero on entry"
        ops.gram_masked_grouped(fwd)
        self._gram_clean.difference_update(keys)
        if cfg.gram_mode == "average":
            self._advance_history(sty, layers)
        ops.style_loss_grouped(sty, self.loss_buf[1:2])
        ops.gram_backward_grouped(bwd)

    def _advance_history(self, sty, layers, advance=True):
        """gram_mode 'average', one level: this step's position in every style layer's history ring, written into the
        (cached, host-side) problem table the grouped style-loss launch reads."""
        for p, layer in zip(sty, layers):
            cnt = self._hist[layer][1] - (0 if advance else 1)
            p.hist_len, p.hist_slot = min(cnt, 9), cnt % 9
            if advance:
                self._hist[layer][1] = cnt + 1

    def _inject_losses(self, lv, b, w_style, w_content, keep=None, only_layers=None, am=None):
        """Loss values into ``loss_buf`` and loss gradients w.r.t. the VGG activations into ``b.grad`` (the deepest
        layer's gradient already ReLU-gated). ``keep`` (dict) receives clones of the style derivative matrices."""
        cfg = self.cfg
        injected = set()
        if w_style != 0.0:
            for li, layer in enumerate(cfg.style_layers):
                if only_layers is not None and layer not in only_layers:
                    continue
                D = self._style_terms(lv, b, li, layer, w_style, am)
                if keep is not None:
                    keep[(lv.index, layer)] = tuple(None if d is None else d.clone() for d in D)
                injected.add(layer)
        if w_content != 0.0:
            for li, layer in enumerate(cfg.content_layers):
                if only_layers is not None and layer not in only_layers:
                    continue
                self._content_term(lv, b, li, layer, w_content)
                injected.add(layer)
        return injected

    def _content_term(self, lv, b, li, layer, w_content, loss_out=None):
        m = lv.masks[layer]
        ops.mse_masked(b.act[layer], lv.content_target[layer], m.channel_ptr(0), lv.counts[layer][0:1],
                       lv.factor[layer], w_content * float(self.cfg.content_weights[li]), b.grad[layer],
                       self.loss_buf[0:1] if loss_out is None else loss_out, relu_gate=(layer == self.deepest))

    def _style_masks(self, lv, layer):
        m = lv.masks[layer]
        if self.cfg.style_pyramid_mode == "multi":
            return m.channel_ptr(1), m.channel_ptr(2)
        return m.channel_ptr(0), None

    def _style_terms(self, lv, b, li, layer, w_style, am=None):
        """``am``: the AmaxBook that holds the bound of ``b.act[layer]`` (default: the step's group book)."""
        cfg = self.cfg
        f = b.act[layer]
        two = ops.GRAM_MODE == "split2"
        af = (self.amax if am is None else am)["a:" + layer] if two else None
        k = (lv.index * len(cfg.style_layers) + li) * ops.AMAX_FLOATS
        ad = self._amax_d[k:k + ops.AMAX_FLOATS] if two else None
        n_slabs = ops.gram_num_slabs(f.C, f.H, f.W)
        key = (f.C, lv.index, layer)
        S0, S1, D0, D1 = self._gram_scratch(key, ops.gram_workspace_slabs(f.C, f.H, f.W))
        pre = ops.GRAM_MODE in ("split", "split2") and key in self._gram_clean   # zeroed by _zero_step_accumulators
        self._gram_clean.discard(key)
        multi = cfg.style_pyramid_mode == "multi"
        weight = w_style * float(cfg.style_weights[li])
        m0, m1 = self._style_masks(lv, layer)
        hist, hist_len, hist_slot = None, 0, 0
        if cfg.gram_mode == "average":
            if layer not in self._hist:
                self._hist[layer] = [torch.zeros(9, f.C, f.C, device=self.device), 0]
            hist, cnt = self._hist[layer]
            hist_len, hist_slot = min(cnt, 9), cnt % 9
            self._hist[layer][1] = cnt + 1
        if multi:
            ops.gram_masked(f, m0, m1, S0, S1, prezeroed=pre, amax_feat=af)
            targets = [self.targets[li][2], self.targets[li][2]]
            term_mask = [0, 1]
            if li > 2:   # content_and_style_losses.py:335-338
                targets.append(self.targets[li][0])
                term_mask.append(0)
            ops.style_loss(S0, S1, lv.counts[layer][1:3], lv.factor[layer], targets, term_mask, [0, 1], weight, f.C,
                           D0, D1, self.loss_buf[1:2], hist, hist_len, hist_slot, n_slabs, amax_d_out=ad)
        else:
            D1 = None
            ops.gram_masked(f, m0, None, S0, None, prezeroed=pre, amax_feat=af)
            ops.style_loss(S0, None, lv.counts[layer][0:1], lv.factor[layer], [self.targets[li][0]], [0], [0, 0],
                           weight, f.C, D0, None, self.loss_buf[1:2], hist, hist_len, hist_slot, n_slabs, amax_d_out=ad)
        ops.gram_backward(f, m0, m1, D0, D1, b.grad[layer], relu_gate=(layer == self.deepest), amax_feat=af, amax_d=ad)
        return D0, D1

    # ------------------------------------------------------------------ explicit-image interface (class mirror)
    def set_external_levels(self, masks, angle_degrees, target_content):
        """Per-level constants from explicit ``pyramid_masks`` (the ``ContentAndStyleLoss.forward`` arguments,
        content_and_style_losses.py:288) instead of a dataset batch."""
        import torch.nn.functional as F
        levels = []
        for i, m in enumerate(masks):
            lv = _ViewLevel()
            lv.index, lv.active, lv.grid, lv.pixel_weight = i, True, None, None
            lv.H, lv.W = m.shape[2:]
            lv.M = (m[0, 0] > 0).to(self.device, torch.float32).contiguous()
            if angle_degrees is not None:
                a = F.interpolate(angle_degrees.to(self.device, torch.float32), (lv.H, lv.W), mode="bilinear")
                lv.passed = (a[0, 0] < self.cfg.angle_threshold).to(torch.uint8).contiguous()
            else:
                lv.passed = torch.ones(lv.H, lv.W, dtype=torch.uint8, device=self.device)
            levels.append(lv)
        self._finish_view(levels, target_content[0].to(self.device, torch.float32).contiguous())

    def images_forward(self, images, w_style=1.0, w_content=1.0):
        """VGG + losses of explicit images (one per level of ``set_external_levels``). Returns
        ``(loss_buf clone [content, style], kept derivative matrices)``; activations stay in the level buffers."""
        self._step_scalars.zero_()
        keep = {}
        for lv, img in zip(self.view, images):
            b = self._level_bufs(lv.H, lv.W)
            ops.image_to_fmap(img[0].detach().to(self.device, torch.float32).contiguous(), b.act["img"])
            self.vgg.forward(b)
            self._inject_losses(lv, b, w_style, w_content, keep, am=b.amax)
        return self.loss_buf.clone(), keep

    def images_backward(self, keep, g_style: float, g_content: float):
        """d(g_style * style + g_content * content) / d image for every level (dense [1,3,H,W] tensors)."""
        out = []
        scratch = torch.zeros(1, device=self.device)
        for lv in self.view:
            b = self._level_bufs(lv.H, lv.W)
            injected = set()
            if g_style != 0.0:
                for layer in self.cfg.style_layers:
                    D0, D1 = keep[(lv.index, layer)]
                    m0, m1 = self._style_masks(lv, layer)
                    d0, d1 = D0 * g_style, (None if D1 is None else D1 * g_style)
                    ad = None
                    if ops.GRAM_MODE == "split2":   # bound of the rescaled derivative matrices (not a hot path)
                        ad = ops.new_amax(self.device)
                        ad[0] = d0.abs().max() if d1 is None else torch.maximum(d0.abs().max(), d1.abs().max())
                    ops.gram_backward(b.act[layer], m0, m1, d0, d1, b.grad[layer], relu_gate=(layer == self.deepest),
                                      amax_feat=b.amax["a:" + layer], amax_d=ad)
                    injected.add(layer)
            if g_content != 0.0:
                for li, layer in enumerate(self.cfg.content_layers):
                    self._content_term(lv, b, li, layer, g_content, loss_out=scratch)
                    injected.add(layer)
            if not injected:
                out.append(torch.zeros(1, 3, lv.H, lv.W, device=self.device))
                continue
            start = max(injected, key=depth_of)
            if start != self.deepest:
                raise ValueError("a zero upstream gradient for the deepest loss layer is not supported")
            self.vgg.backward(b, injected - {start}, start)
            out.append(ops.fmap_to_image(b.grad["img"], 3)[None])
        return out

    def optimizer_step(self, world_size: int = 1):
        """Fused regulariser-gradient + Adam + clamp + zero-grad over the whole arena (one launch)."""
        self.step_count += 1
        run, self._prog_run = self._prog_run, None
        if run is not None and world_size == 1:          # the update segment of the program step_compute replayed
            prog = run[0]
            self._program_patch_update(prog)
            import ctypes as C
            from . import hip as _hip
            rc = _hip.lib.sm_call_replay(C.byref(prog.arr, prog.n_compute * C.sizeof(_hip.Call)), prog.n - prog.n_compute,
                                         _hip.stream(), C.byref(prog._failed))
            if rc != 0:
                raise RuntimeError(f"libstylemesh_hip: replayed update call failed with HIP error code {rc}")
            self._grad_dirty = False
            return
        from . import hip as _hip
        if self._prog_rec is not None:
            if world_size == 1 and ops.lib is _hip.lib:
                ops.lib = self._prog_rec[0]         # the update segment of the step being recorded
            else:
                self._prog_end_recording(discard=True)
        try:
            self._optimizer_step_eager(world_size)
        finally:
            self._prog_end_recording(discard=(world_size != 1))

    def _optimizer_step_eager(self, world_size):
        if self._can_graph():
            # The step-dependent scalars live ON THE DEVICE: a captured one-thread kernel advances {lr, step} and
            # writes {lr / bc1, 1 / sqrt(bc2)} for the update that follows it in the same graph. (Sending them through
            # a pinned host buffer raced: the host runs many steps ahead and overwrote the buffer before an earlier
            # step's async copy had executed.) The host only pushes lr when StepLR changes it - as a fill kernel,
            # whose value travels in the launch itself - and the step when it has advanced outside the graph.
            if not hasattr(self, "_hyper_dev"):
                self._hyper_dev = torch.zeros(3, device=self.device)
                self._hyper_state = torch.zeros(2, dtype=torch.float64, device=self.device)
                self._hyper_lr, self._hyper_step = None, None
            if self._hyper_lr != self.lr:
                self._hyper_state[0:1].fill_(self.lr)
                self._hyper_lr = self.lr
            if self._hyper_step != self.step_count - 1:
                self._hyper_state[1:2].fill_(float(self.step_count - 1))
            self._hyper_step = self.step_count
            # the capture bakes in the flags pointer and the sparse-vs-dense choice: part of the key
            touched, tl2 = self._touched_arg()
            key = (world_size, None if touched is None else touched.data_ptr(), tl2)
            if self._opt_graph is None or self._opt_graph[0] != key:
                if getattr(self, "_opt_warm", 0) < 1:      # one eager run before capturing
                    self._opt_warm = 1
                    self._optimizer_launch(world_size, self._hyper_dev)
                    return
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    self._optimizer_launch(world_size, self._hyper_dev)
                self._opt_graph = (key, g)
            self._opt_graph[1].replay()
            return
        self._optimizer_launch(world_size, None)

    def _touched_arg(self):
        return (self.touched, self.touched_log2) if (self.sparse_update and self.touched is not None) else (None, 0)

    def _can_fuse_gram_bwd(self, layer, bufs) -> bool:
        """The Gram backward of ``layer`` can ride in the epilogue of the data-gradient conv that produces the layer's
        gradient: a 64- / 128-channel style layer right below a conv that itself sits below a pool (relu1_1 / conv1_2,
        relu2_1 / conv2_2: the fp16x2 kernel's 64 x 256 / 128 x 128 tile holds all channels of a position), fp16x2
        arithmetic on both sides, fused pool backward. STYLEMESH_FUSE_GRAM_BWD=<layers> (default r11,r21; 0 = none)."""
        from . import vgg as _vgg
        which = os.environ.get("STYLEMESH_FUSE_GRAM_BWD", "r11,r21")
        if layer not in which.split(",") or layer not in ("r11", "r21") or layer in self.cfg.content_layers:
            return False
        pool_above = {"r11": "p1", "r21": "p2"}[layer]   # conv1_2 / conv2_2 sit right below these pools
        return (_vgg.FUSE_POOL_BWD and ops.CONV_MODE == "split2" and ops.GRAM_MODE == "split2"
                and self.cfg.style_pyramid_mode in ("multi", "single") and all(b.code for b in bufs)
                and bufs[0].act[layer].C in (64, 128) and depth_of(self.deepest) > depth_of(pool_above))

    def _overlap_pays(self, active) -> bool:
        """Side-stream overlap costs the host a few events and stream switches per step: worth it when the GPU step is
        several milliseconds (c3 / c5: 1.6 M pixels over the levels), not for a single small level (c2: 87 k)."""
        return sum(lv.H * lv.W for lv in active) >= self.overlap_min_pixels

    def _adam_early(self, dev_hyper=None, zero_sumsq=True):
        """Split update, first half, at the HEAD of a step: the ever-touched chunks the current view (all ranks' views)
        cannot reach have a zero gradient whatever this step computes, and the step neither samples nor scatters there -
        their update (step count of the update that closes this step) runs on a side stream beside the forward pass;
        ``_optimizer_launch`` then only walks the view's own chunks. Call after ``loss_tensors()`` (which reads the
        sum of squares this zeroes)."""
        self._adam_early_pending = False
        if not (self.split_update and self.sparse_update and self.touched is not None and self._view_flags is not None
                and not self._can_graph() and not torch.cuda.is_current_stream_capturing()
                and self.view is not None and self._overlap_pays([lv for lv in self.view if lv.active])):
            return
        if self._other_flags is None or self._other_flags[0] is not self.touched:   # (a loaded state replaces `touched`)
            self._other_flags = (self.touched, ((self.touched != 0) & (self._view_flags == 0)).to(torch.int32))
        main = torch.cuda.current_stream()
        while len(self._lv_streams) < 2:
            self._lv_streams.append(self._new_side_stream())
        st = self._lv_streams[1]
        ev = torch.cuda.Event()
        ev.record(main)
        st.wait_event(ev)
        with torch.cuda.stream(st):
            if zero_sumsq:
                ops.zero_floats(self.sumsq)
            # (no gradient pointer: these chunks hold a zero data-term gradient - nothing scatters into them during
            # this view and their last update zeroed them - so the launch moves 6 instead of 8 streams)
            ops.adam_fused(self.arena.p, None, self.arena.m,
                           self.arena.v, self.arena.seg_end, self.reg_coef,
                           self.lr, self.step_count + 1, grad_scale=1.0, sumsq_out=self.sumsq, dev_hyper=dev_hyper,
                           touched=self._other_flags[1], touched_log2=self.touched_log2)
            self._adam_early_done = torch.cuda.Event()
            self._adam_early_done.record(st)

    def _optimizer_launch(self, world_size, dev_hyper):
        self._grad_dirty = False   # the fused update zeroes the gradient arena
        if self._adam_early_done is not None:   # second half of the split update: the view's own chunks
            torch.cuda.current_stream().wait_event(self._adam_early_done)
            self._adam_early_done = None
            ops.adam_fused(self.arena.p, self.arena.g, self.arena.m, self.arena.v, self.arena.seg_end, self.reg_coef,
                           self.lr, self.step_count, grad_scale=1.0 / world_size, sumsq_out=self.sumsq,
                           dev_hyper=dev_hyper, touched=self._view_flags, touched_log2=self.touched_log2)
            return
        ops.zero_floats(self.sumsq)
        if dev_hyper is not None:
            ops.adam_hyper_step(self._hyper_state, dev_hyper)
        touched, tl2 = self._touched_arg()
        ops.adam_fused(self.arena.p, self.arena.g, self.arena.m, self.arena.v, self.arena.seg_end, self.reg_coef,
                       self.lr, self.step_count, grad_scale=1.0 / world_size, sumsq_out=self.sumsq,
                       dev_hyper=dev_hyper, touched=touched, touched_log2=tl2)

    def use_pipelined_exchange(self, reducer) -> bool:
        """Multi-GPU tail of the step as pipelined exchange + update? (identical on every rank: it depends on the
        reducer's chunk count - the all-reduced union of the ranks' footprints - and on configuration only)"""
        if reducer is None or not hasattr(reducer, "pipelined") or self._can_graph() or self.pipeline_exchange is False:
            return False
        if self.pipeline_exchange is True:
            return True
        n_idx = getattr(reducer, "n_idx", None)
        if n_idx is None:
            return False
        dense = getattr(reducer, "fraction", 0.0) > getattr(reducer, "dense_above", 1.0)
        nbytes = 4 * self.arena.n if dense else 4 * n_idx * reducer.chunk
        return nbytes >= self.pipeline_min_bytes

    def use_deferred_exchange(self, reducer) -> bool:
        """Multi-GPU tail of the step with the single-owner chunks off the critical path (``OwnerAwareGradReducer``)?
        OPT-IN (STYLEMESH_DEFERRED_EXCHANGE=1; identical on every rank: configuration and the all-reduced chunk lists only):
        functional and bit-identical to exchange-then-update over gloo (tests/test_distributed_cpu.py), never timed over
        RCCL on separate GPUs - the same status as the pipelined exchange."""
        return (self.deferred_exchange and reducer is not None and getattr(reducer, "owner_aware", False)
                and self.touched is not None and self.sparse_update and not self._can_graph() and reducer.ready)

    def _chunk_update(self, world_size, with_sumsq=True):
        a = self.arena

        def update(flags, lr, step):
            ops.adam_fused(a.p, a.g, a.m, a.v, a.seg_end, self.reg_coef, lr, step, grad_scale=1.0 / world_size,
                           sumsq_out=self.sumsq if with_sumsq else None, touched=flags, touched_log2=self.touched_log2)
        return update

    def exchange_and_update_deferred(self, world_size: int, reducer):
        """The step's tail under ``use_deferred_exchange``: ever-touched chunks outside the ranks' views (the early half of the
        split update, or inline), then ``reducer.step``: the previous step's deferred sums, this rank's own single-owner
        chunks, the critical exchange + update of the shared chunks, the deferred exchange in the background."""
        assert reducer.chunk_log2 == self.touched_log2
        self.step_count += 1
        a = self.arena
        if self._adam_early_done is not None:
            torch.cuda.current_stream().wait_event(self._adam_early_done)
            self._adam_early_done = None
        else:
            ops.zero_floats(self.sumsq)
            if self._other_flags is None or self._other_flags[0] is not self.touched:
                self._other_flags = (self.touched, ((self.touched != 0) & (self._view_flags == 0)).to(torch.int32))
            ops.adam_fused(a.p, None, a.m, a.v, a.seg_end, self.reg_coef, self.lr, self.step_count, grad_scale=1.0,
                           sumsq_out=self.sumsq, touched=self._other_flags[1], touched_log2=self.touched_log2)
        update = self._chunk_update(world_size)
        # Chunks this rank SAMPLES without writing a gradient (pixels of weight zero: ``_view_flags`` is the sampled footprint,
        # the reducer's lists the weighted ones of all ranks): outside the early half (which excludes the view's flags) and
        # outside every exchange list - their regulariser / momentum update is this rank's own business, every step
        ex = getattr(self, "_deferred_extra", None)
        if ex is None or ex[0] is not reducer.f_shared:
            union = (reducer.f_shared != 0) | (reducer.f_single_mine != 0) | (reducer.f_single_others != 0)
            ex = self._deferred_extra = (reducer.f_shared, ((self._view_flags != 0) & ~union).to(torch.int32))
        update(ex[1], self.lr, self.step_count)
        reducer.step(a.g, update, self.lr, self.step_count)
        self._grad_dirty = False

    def finish_exchange(self, world_size: int, reducer):
        """Apply the deferred sums still outstanding (end of training, before the texture is read or saved; the engine does
        it itself before every per-view collective)."""
        if reducer is not None and getattr(reducer, "owner_aware", False):
            # (outside a step: sum(p^2) already holds these chunks once - their p^2 of one update ago; the regulariser loss
            # VALUE of a non-owner lags by that one update on them, the gradient never does)
            reducer.drain(self.arena.g, self._chunk_update(world_size, with_sumsq=False))

    def exchange_and_update(self, world_size: int, reducer):
        """Multi-GPU tail of the step: gradient exchange overlapped with the fused update (``reducer.pipelined``:
        the update of an arena range is issued as soon as its sums arrived, later pieces still on the links)."""
        self.step_count += 1
        self.sumsq.zero_()
        self._grad_dirty = False   # every range is zeroed by its update
        a = self.arena

        touched, tl2 = self._touched_arg()

        def update_range(lo, hi):
            ops.adam_fused(a.p, a.g, a.m, a.v, a.seg_end, self.reg_coef, self.lr, self.step_count,
                           grad_scale=1.0 / world_size, sumsq_out=self.sumsq, lo=lo, hi=hi, touched=touched,
                           touched_log2=tl2)
        reducer.pipelined(a.g, update_range)

    def _can_graph(self):
        # 'average' changes launch arguments (history length / slot) every step; the conv timer records events
        return (self.use_graphs and self.cfg.gram_mode == "current"
                and not (ops.CONV_TIMER is not None and ops.CONV_TIMER.enabled))

    def step_forward_backward(self):
        """forward_backward(), replayed from a hipGraph when enabled (captured on the 2nd step of each distinct
        set of active levels; per-view buffers keep their addresses, so the graph survives view changes)."""
        if not self._can_graph():
            return self.forward_backward()
        sig = self.view_sig
        g = self._graphs.get(sig)
        if g is None:
            if self._graph_warm.get(sig, 0) < 1:
                self._graph_warm[sig] = 1
                return self.forward_backward()
            g = torch.cuda.CUDAGraph()
            with _capture(g):
                self.forward_backward()
            self._graphs[sig] = g
        g.replay()

    def end_epoch(self):
        self.epoch += 1

    def request_prepare(self, batch, ready_event=None):
        """Ask for ``prepare_view(batch, ready_event)`` to run right after the NEXT ``begin_step`` has made its own batch
        current. ``prepare_view`` writes the buffer slot the current view does not use, so it may only run once the view
        before it has been swapped in: a caller that learns about view i + 1 BEFORE it steps view i (index_repeat 1: every
        step is the first of its view - ``RepeatingSampler`` of data/abstract_dataset.py:498-512 with the dip scripts'
        ``--index_repeat 1``) leaves the request here instead of calling ``prepare_view`` too early."""
        self._prepare_request.append((batch, ready_event))

    def step_compute(self, batch, reducer=None, new_view=None, exchange=True):
        """Everything of a training step BEFORE the optimizer: per-view work, the step head (regulariser loss of the
        current texture, zero fills), the early half of the split update beside the forward pass, forward + backward
        into the gradient arena and - ``exchange`` - the multi-GPU gradient exchange. Returns this step's losses as
        device tensors that stay valid. ``optimizer_step`` closes the step (a Lightning-style caller does that from its
        optimizer facade: ``model.FusedTextureAdam.step``)."""
        if self._adam_early_done is not None:
            # the previous step's early half of the split update was never closed by ``optimizer_step`` (a foreign
            # optimizer, a skipped step): the texels outside the view carry an update the view's own texels lack
            raise RuntimeError("step_compute() was called again before optimizer_step() closed the previous step: the "
                               "split update needs exactly one optimizer_step per step (STYLEMESH_SPLIT_UPDATE=0 "
                               "disables the split)")
        self.begin_step(batch, reducer, new_view)
        reqs, self._prepare_request = self._prepare_request, []
        self._steps_on_view = getattr(self, "_steps_on_view", 0) + 1
        if reducer is None:
            for req in reqs:
                # (urgent: the schedule changes the view every step - the view before this one lasted a single step)
                self.prepare_view(*req, urgent=getattr(self, "_last_view_steps", 0) == 1)
        out = torch.empty(3, device=self.device)    # this step's [content, style, tex_reg]: stays valid for the caller
        self._prog_end_recording(discard=True)      # (a recording whose optimizer_step never came)
        self._prog_run = None
        key = self._program_key(reducer)
        if key is not None:
            prog = self._programs.get(key)
            if prog is not None and self.step_programs != "verify":
                return self._program_compute(prog, key, out)
            from . import hip as _hip
            if (prog is not None or self._prog_warm.get(key, 0) >= 2) and ops.lib is _hip.lib:
                # steady state: record this step. The recorder stands in for the module-global ``ops.lib`` ONLY while this
                # engine's own calls are being issued: from here to the end of ``step_compute`` and again inside
                # ``optimizer_step`` - never across the caller's code in between (a hook that renders the texture, a
                # validation pass, a second engine's step would otherwise land in this engine's program). An
                # engine that finds another recorder installed does not record.
                from .program import Recorder
                rec = Recorder()
                self._prog_rec = [rec, key, None, out]
                ops.lib = rec
            elif prog is None:
                self._prog_warm[key] = self._prog_warm.get(key, 0) + 1
        try:
            return self._step_compute_eager(out, reducer, exchange)
        except BaseException:
            self._prog_end_recording(discard=True)
            raise
        finally:
            if self._prog_rec is not None:          # paused until optimizer_step
                from . import hip as _hip
                ops.lib = _hip.lib

    def _step_compute_eager(self, out, reducer, exchange):
        losses = self._step_begin(out[2:3])    # tex_reg of the CURRENT (pre-update) texture, device tensors, no sync
        deferred = self.use_deferred_exchange(reducer)
        pipelined = self.use_pipelined_exchange(reducer) and not deferred
        if not pipelined:
            if self._can_graph() or self.early_update_at == "head":
                self._adam_early()
            else:
                self._adam_early_pending = True   # forked inside the forward pass (``early_update_at``)
        self.step_forward_backward()
        if self._adam_early_pending:              # (a forward pass that never reached the fork point)
            self._adam_early()
        # content / style of loss_tensors() are views of the accumulators the NEXT step zeroes: hand out this step's
        # values (one 2-float copy), so that a caller may read them any number of steps later
        ops.copy_floats(out, self.loss_buf, 2)
        losses["content"], losses["style"] = out[0:1], out[1:2]
        if exchange and reducer is not None and not pipelined and not deferred:
            self._timed("exchange", lambda: reducer(self.arena.g))
        if self._prog_rec is not None:
            self._prog_rec[2] = len(self._prog_rec[0].calls)      # the optimizer's calls follow
        return losses

    # ------------------------------------------------------------------ step programs
    def _program_key(self, reducer):
        """None when this step cannot be a replayed program; else what a program's validity depends on. Programs serve the
        SMALL steps (one 256 x 341 level: 1.1 ms of GPU work against 0.65 - 1.1 ms of interpreter + ctypes time for its
        ~70 launches), which run on one stream: no side streams, no split update (``_overlap_pays``), no graphs, no
        per-launch timers, one rank."""
        if self.step_programs == "0" or reducer is not None or self.use_graphs or self.view is None or self.overlap_style:
            return None
        if (ops.CONV_TIMER is not None and ops.CONV_TIMER.enabled) or getattr(self, "phase_timer", None) is not None:
            return None
        if torch.cuda.is_current_stream_capturing() or self._pending_view is not None:
            return None
        active = [lv for lv in self.view if lv.active]
        if not active or self.deepest is None or self._overlap_pays(active):
            return None
        grouped = (self.group_losses and float(self.cfg.loss_weights.get("style", 0.0)) != 0.0 and ops.GRAM_MODE == "split2"
                   and (self.cfg.gram_mode != "average" or len(active) == 1))
        if len(active) > 1 and not grouped:      # (per-level loss streams)
            return None
        empties = None if self.view_tiles is None else frozenset(k for k, v in self.view_tiles.items() if v[0].numel() == 0)
        plan = self._scatter_plan if (self.planned_scatter and self._scatter_levels == [lv.index for lv in active]) else None
        # (the launch stream: recorded words hold per-stream scratch - the split-K slabs, the Gram-backward workspace -
        # and every buffer a recorded call touches must be grow-never or part of this key)
        return (self._slot, ops.hip.stream(), tuple((lv.index, lv.H, lv.W) for lv in active), empties, ops.CONV_MODE, ops.GRAM_MODE,
                None if plan is None else (id(plan), plan.sorted_in, plan.generation, plan.n_entries), self.sparse_update,
                None if self._gram_arena is None else self._gram_arena.data_ptr(),
                None if self.touched is None else self.touched.data_ptr(), self.cfg.gram_mode, self.sparse_tiles,
                tuple(sorted(self.cfg.loss_weights.items())),
                # host-side state the step's first launches depend on (is there anything to zero, does the scatter add or
                # store): a replayed step leaves it as an eager one does, so in a steady run these never change
                self._gram_arena is not None and len(self._gram_clean) != len(self._gram), self._grad_dirty)

    def _prog_end_recording(self, discard=False):
        """Take the recorder off ``ops.lib``; unless ``discard``, turn what it noted into this key's program (or, in
        verify mode, compare it with the program that exists)."""
        rec_state, self._prog_rec = self._prog_rec, None
        if rec_state is None:
            return
        from . import hip as _hip
        rec, key, n_compute, out = rec_state
        if ops.lib is rec:
            ops.lib = _hip.lib
        if discard or n_compute is None:
            return
        if not any(c[0] == "sm_adam_fused" for c in rec.calls[n_compute:]):
            # an update segment without the fused update would be replayed as "no update at all": never store it
            self._prog_warm[key] = -(1 << 30)
            return
        if rec.problem is not None:
            self._prog_warm[key] = -(1 << 30)          # never again for this key
            return
        from .program import StepProgram
        prog = StepProgram(rec)
        prog.n_compute = n_compute
        self._program_patch_points(prog)
        old = self._programs.get(key)
        if old is not None and self.step_programs == "verify":
            # the table as it would have been replayed for THIS step against what the step really issued
            self._program_patch_step(old, out, advance=False)
            self._program_patch_update(old)
            a, b = old.words(), prog.words()
            if a != b:
                bad = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
                raise RuntimeError(f"step program diverged from the step at call {bad}: "
                                   f"{a[bad] if bad < len(a) else None} vs {b[bad] if bad < len(b) else None}")
            self.program_verified = getattr(self, "program_verified", 0) + 1
            return
        prog.view_id = self.view_serial
        if len(self._programs) >= 32:            # (slots x level sets x empty-list sets of a scene: a handful; bounded anyway)
            self._programs.pop(next(iter(self._programs)))
        self._programs[key] = prog

    def _program_patch_points(self, prog):
        """Which argument words of a recorded step change from step to step / view to view."""
        sb = prog.find("sm_step_begin")
        cp = [i for i in prog.find("sm_copy_floats") if i < prog.n_compute]
        if len(sb) != 1 or not cp:
            raise RuntimeError("a recorded step must hold exactly one sm_step_begin and the copy of the loss pair")
        prog.i_begin, prog.i_copy = sb[0], cp[-1]
        prog.i_adam = [i for i in prog.find("sm_adam_fused") if i >= prog.n_compute]
        prog.i_hist = prog.find("sm_style_loss") if self.cfg.gram_mode == "average" else []
        # (grouped loss phase: the ring positions are fields of the style-loss launch's problem table - a host array the
        # program keeps and the library reads at replay time)
        prog.hist_tables = [(prog.host[(i, 0)], prog.word(i, 1)) for i in prog.find("sm_style_loss_grouped")
                            if self.cfg.gram_mode == "average" and (i, 0) in prog.host]
        # active lists: (call, index of the length word, list key) wherever a list's pointer is followed by its length
        prog.lists = []
        if self.view_tiles is not None:
            by_ptr = {}
            for k, (lst, _) in self.view_tiles.items():
                if lst.numel():
                    by_ptr.setdefault(lst.data_ptr(), (k, lst.numel()))
            for i in range(prog.n):
                c = prog.arr[i]
                for j in range(c.n_args - 2):
                    hit = by_ptr.get(c.args[j])
                    if hit is not None and c.args[j + 1] == hit[1]:
                        prog.lists.append((i, j + 1, hit[0]))

    def _program_patch_step(self, prog, out, advance=True):
        """The words of the compute segment that belong to THIS step."""
        prog.patch(prog.i_begin, 3, out.data_ptr() + 8)
        prog.patch(prog.i_copy, 0, out.data_ptr())
        if prog.view_id != self.view_serial:             # a new view in this slot: its lists' lengths
            for i, j, k in prog.lists:
                prog.patch(i, j, self.view_tiles[k][0].numel())
            prog.view_id = self.view_serial
        for table, n_prob in prog.hist_tables:
            # one level: problem k of a launch = style layer k of the launch's layers; every style layer is in one launch
            done = getattr(prog, "_hist_layers", None)
            if done is None:
                starts, k0 = [], 0
                for t, n in prog.hist_tables:
                    starts.append(k0)
                    k0 += n
                order = self._program_hist_order()
                prog._hist_layers = done = {id(t): order[s0:s0 + n] for (t, n), s0 in zip(prog.hist_tables, starts)}
            self._advance_history([table[k] for k in range(n_prob)], done[id(table)], advance)
        for n, i in enumerate(prog.i_hist):               # gram_mode 'average': the history ring's position
            layer = self.cfg.style_layers[n]
            cnt = self._hist[layer][1] - (0 if advance else 1)
            prog.patch(i, 14, min(cnt, 9))
            prog.patch(i, 15, cnt % 9)
            if advance:
                self._hist[layer][1] = cnt + 1

    def _program_hist_order(self):
        """Style layers in the order the grouped loss phase of a single-stream step launches their branches."""
        cfg = self.cfg
        return [l for l in cfg.style_layers]

    def _program_patch_update(self, prog):
        from .program import double_word, float_word
        step = self.step_count
        for i in prog.i_adam:
            prog.patch(i, 8, float_word(self.lr))
            prog.patch(i, 12, double_word(1.0 - 0.9 ** step))
            prog.patch(i, 13, double_word(1.0 - 0.999 ** step))

    def _program_compute(self, prog, key, out):
        """``step_compute`` of a step whose program exists: patch, replay the compute segment with one library call."""
        self._program_patch_step(prog, out)
        from . import hip as _hip
        import ctypes as C
        rc = _hip.lib.sm_call_replay(prog.arr, prog.n_compute, _hip.stream(), C.byref(prog._failed))
        if rc != 0:
            raise RuntimeError(f"libstylemesh_hip: replayed call {prog._failed.value} ({prog.names[prog._failed.value]}) "
                               f"failed with HIP error code {rc}")
        self._grad_dirty = True
        self._prog_run = (prog, key)
        self.program_replays += 1
        return {"content": out[0:1], "style": out[1:2], "tex_reg": out[2:3]}

    def training_step(self, batch, world_size: int = 1, reducer=None, new_view=None, next_batch=None):
        """zero_grad -> forward_with_loss -> backward -> Adam, Lightning's automatic-optimisation order.
        ``new_view``: see ``begin_step``. ``next_batch``: the view of the NEXT step when it differs from this one - its
        per-view constants are prepared beside this step (``request_prepare``). Returns this step's losses as device
        tensors that stay valid."""
        if next_batch is not None:
            # one upcoming view, or a list of them in schedule order (index_repeat 1: the next two)
            for nb in (next_batch if isinstance(next_batch, list) else [next_batch]):
                self.request_prepare(nb)
        self._world_size = world_size
        losses = self.step_compute(batch, reducer, new_view)
        if self.use_deferred_exchange(reducer):
            self._timed("exchange+update", lambda: self.exchange_and_update_deferred(world_size, reducer))
            return losses
        if self.use_pipelined_exchange(reducer):
            self._timed("exchange+update", lambda: self.exchange_and_update(world_size, reducer))
            return losses
        self._timed("update", lambda: self.optimizer_step(world_size))
        return losses

    def _timed(self, tag, fn):
        """Run ``fn``; when ``self.phase_timer`` is set (bench.py --gpus N: an ``ops.KernelTimer``), bracket it with HIP
        events on the current stream so that the exchange and the update of a multi-GPU step can be told apart."""
        t = getattr(self, "phase_timer", None)
        if t is None or not t.enabled:
            return fn()
        return t.launch(fn, 0.0, tag)

    def begin_step(self, batch, reducer=None, new_view=None):
        """Per-view work at the head of a step. ``set_view`` runs when the batch's view KEY differs from the current
        one. The per-view COLLECTIVE (``reducer.new_view``: max-all-reduce of the touch flags) is driven by the
        schedule POSITION instead - ``new_view`` / ``batch.new_view`` (``distributed.ViewBatch``), identical on every
        rank at every step - because a rank whose shard was padded by repeating its last view keeps its key while the
        other ranks change theirs, and all of them must enter the collective. Without a schedule flag the key decides
        (single rank, or callers whose ranks all change views together)."""
        # the same batch object again = the same view (RepeatingSampler schedules): no read of the view index, which
        # would be a device-to-host sync per step when the batch lives on the GPU
        if self.view is None or batch is not self._last_batch:
            key = int(batch[8][0]) if torch.is_tensor(batch[8]) else batch[8]
        else:
            key = self.view_key
        changed = self.view is None or key != self.view_key
        if new_view is None:
            new_view = getattr(batch, "new_view", None)
        # is the per-view collective due at this schedule position?
        due = reducer is not None and hasattr(reducer, "new_view") and (changed if new_view is None else new_view)
        in_set_view = due and changed and hasattr(reducer, "new_view_begin")
        if due:
            self.finish_exchange(getattr(self, "_world_size", 1), reducer)   # (deferred sums of the last step: before the new view's collective)
        if changed:
            import time
            t0 = time.perf_counter()
            self._last_view_steps, self._steps_on_view = getattr(self, "_steps_on_view", 0), 0
            self.set_view(batch, reducer if in_set_view else None)
            self.set_view_host_s = getattr(self, "set_view_host_s", 0.0) + time.perf_counter() - t0   # diagnostics
            self.set_view_calls = getattr(self, "set_view_calls", 0) + 1
        if reducer is not None and not hasattr(reducer, "new_view") and self.touched is not None:
            # A reducer that cannot union the ranks' footprints (the plain dense all-reduce, any callable): the other
            # ranks' gradients arrive in chunks this rank's views never flagged, which the sparse update would skip
            # without updating or zeroing them - every texel takes part in the update from now on
            self.touched = None
            self._other_flags = None
        if due:
            if in_set_view:
                flags = self._union_flags     # set_view entered the collective and read the count with its own read-back
            else:                             # a padded rank that keeps its view (or a reducer without the split form)
                flags = self.touch_flags(reducer.chunk_log2)
                reducer.new_view(flags)       # in place: now the union over the ranks' views
            if self.touched is not None:
                if reducer.chunk_log2 == self.touched_log2:
                    ops.flags_or(self.touched, flags)   # the other ranks' gradients arrive with the exchange
                    if self._view_flags is not None:
                        ops.flags_or(self._view_flags, flags)
                        self._other_flags = None
                else:
                    self.touched = None

    def touch_flags(self, chunk_log2: int, levels=None):
        """int32 flag per 2^chunk_log2 floats of the gradient arena: can the current view's scatter write there?
        (``levels``: the levels of a view that is still being set up.)"""
        n_chunks = -(-self.arena.n // (1 << chunk_log2))
        flags = torch.zeros(n_chunks, dtype=torch.int32, device=self.device)
        for lv in (self.view if levels is None else levels):
            if lv.active:
                ops.tex_touch_flags(self.grads, self.arena.g, lv.grid, lv.pixel_weight, flags, chunk_log2)
        return flags

    def loss_tensors(self):
        """Weighted losses as device tensors, names as the reference logs them (model/model.py:261-270).
        content / style are filled by the forward_backward() that follows."""
        reg = (self.sumsq * self._reg_loss_coef_dev).sum().reshape(1)
        return {"content": self.loss_buf[0:1], "style": self.loss_buf[1:2], "tex_reg": reg}

    def losses(self, lt=None):
        """Host floats (synchronises). Pass the dict returned by ``training_step`` to read that step's values."""
        lt = lt or self.loss_tensors()
        out = {k: float(v) for k, v in lt.items()}
        out["total"] = out["content"] + out["style"] + out["tex_reg"]
        return out


def _mirror_tiles(S):
    """Full symmetric matrix from the upper-triangular 64x64 tiles sm_gram_masked fills."""
    C = S.shape[0]
    T = C // 64
    keep = torch.ones(T, T, device=S.device).triu().repeat_interleave(64, 0).repeat_interleave(64, 1)
    strict = torch.ones(T, T, device=S.device).triu(1).repeat_interleave(64, 0).repeat_interleave(64, 1)
    return S * keep + (S * strict).T
