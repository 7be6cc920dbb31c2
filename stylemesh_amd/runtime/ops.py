"""One thin Python function per C-ABI entry point (argument marshalling only, no arithmetic)."""
from __future__ import annotations

import os

import ctypes as C

import numpy as np
import torch

from . import hip
from .fmap import FMap
from .hip import lib, ptr

CLAMP_LO, CLAMP_HI = -123.6800, 151.0610  # reference model/texture/texture.py:43


class KernelTimer:
    """HIP-event timer around individual launches of one kernel family, on the stream the kernels are launched
    on (torch's current stream). bench.py uses it to price the dominant kernel against its roofline."""

    def __init__(self):
        self.records = []   # (start_event, end_event, algorithmic_work, tag)
        self.bytes = {}     # tag -> algorithmic HBM bytes of the timed launches
        self.info = []      # per record: free-form description of the launch (diagnostics)
        self.enabled = True

    def launch(self, fn, work, tag="f32", nbytes=0.0, info=""):
        if not self.enabled:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.records.append((a, b, work, tag))
        self.info.append(info)
        self.bytes[tag] = self.bytes.get(tag, 0.0) + nbytes

    def summary(self, tag=None):
        """-> (n_launches, total_ms, total_work) of the launches with this tag (None: all); synchronises."""
        torch.cuda.synchronize()
        rec = [r for r in self.records if tag is None or r[3] == tag]
        ms = sum(a.elapsed_time(b) for a, b, _, _ in rec)
        return len(rec), ms, sum(r[2] for r in rec)


CONV_TIMER: KernelTimer | None = None   # set by bench.py; None = no instrumentation


def _hbm_timed(name, nbytes, fn):
    """The step's HBM-bound kernels under the same per-launch events as the convs (tag 'hbm:<name>', work = algorithmic
    bytes): bench.py prices them against the measured copy rate (roofline_hbm)."""
    t = CONV_TIMER
    if t is None or not t.enabled:
        return fn()
    return t.launch(fn, float(nbytes), "hbm:" + name, float(nbytes))


# ---- weight packing (one-time setup; layout transforms only) ------------------------------------------------
def pack_conv_fwd(weight: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] -> [9][Cin_pad][Cout] (tap-major, Cout fastest), Cin padded with zeros to 4 / 8k."""
    cout, cin = weight.shape[:2]
    cin_pad = 4 if cin <= 4 else (cin + 7) // 8 * 8
    w = weight.permute(2, 3, 1, 0).reshape(9, cin, cout)
    out = torch.zeros(9, cin_pad, cout, dtype=torch.float32, device=weight.device)
    out[:, :cin] = w
    return out.contiguous()


def pack_conv_dgrad(weight: torch.Tensor) -> torch.Tensor:
    """Weights of the data-gradient conv: dx[ci][p] = sum W[co][ci][2-ky][2-kx] dy[co][p + (ky-1,kx-1)]
    -> [9][Cout][Cin_pad4] (the roles of the channel dimensions swap)."""
    cout, cin = weight.shape[:2]
    cin_pad = (cin + 3) // 4 * 4
    w = weight.flip(2, 3).permute(2, 3, 0, 1).reshape(9, cout, cin)
    out = torch.zeros(9, cout, cin_pad, dtype=torch.float32, device=weight.device)
    out[:, :, :cin] = w
    return out.contiguous()


def split_eligible(cin_pad: int, cout: int) -> bool:
    """Shapes the fp16x2-split conv kernel takes (sm_conv3x3_grouped_split2)."""
    return cin_pad % 16 == 0 and cout % 64 == 0


def pack_conv_split2(wt: torch.Tensor):
    """fp32 tap-major pack [9][Cin][Cout] -> (wt2, w_scale_inv) of sm_conv3x3_grouped_split2: the weights times the
    power of two s_w that puts max |w| into [2^14, 2^15), as [9][Cin/16][2 parts][2 k-groups][Cout][8 ci] fp16 bit
    patterns (int16) with w s_w = part0 + part1 to 22 significand bits (round-to-nearest); w_scale_inv = 1 / s_w."""
    import math
    taps, cin, cout = wt.shape
    assert taps == 9 and split_eligible(cin, cout)
    amax = float(wt.abs().max())
    k = 14 - math.floor(math.log2(amax)) if amax > 0 else 0
    w = (wt * (2.0 ** k)).view(9, cin // 16, 2, 8, cout).permute(0, 1, 2, 4, 3)     # [9][chunk][kgroup][cout][8]
    h = w.half()
    l = (w - h.float()).half()
    return torch.stack([h, l], dim=2).contiguous().view(torch.int16), 2.0 ** -k    # [9][chunk][2][2][cout][8]


# 'f32' = v_mfma_f32_32x32x2_f32 everywhere (bit-exact fmaf chains); 'split2' = fp16x2-split MFMA (3 partial products,
# fp32 accumulate, operands scaled by powers of two from recorded maxima) wherever a layer's shape allows.
# (round 1's bf16x3 arithmetic - 'split', 6 partial products - was removed in round 6: profiles/r02 keeps its measurements)
CONV_MODE = os.environ.get("STYLEMESH_CONV_MODE", "split2")
if CONV_MODE not in ("split2", "f32"):
    raise ValueError(f"STYLEMESH_CONV_MODE={CONV_MODE!r}: 'split2' (default) or 'f32'")


# ---- texture -------------------------------------------------------------------------------------------------
def tex_sample_fwd(layers, grid: torch.Tensor, out: FMap):
    h, w = grid.shape[-3], grid.shape[-2]
    assert out.H == h and out.W == w and out.C >= 3 and grid.shape[-1] == 2
    hip.check(lib.sm_tex_sample_fwd(hip.ptr_array(layers), hip.int_array([l.shape[2] for l in layers]),
                                    hip.int_array([l.shape[1] for l in layers]), len(layers), ptr(grid), h, w,
                                    out.ptr, hip.stream()), "sm_tex_sample_fwd")


def tex_sample_fwd_grouped(layers, grids, outs):
    """``tex_sample_fwd`` for the UV levels of a view in one launch: ``grids`` [h,w,2] tensors, ``outs`` FMaps."""
    hs, ws = [g.shape[-3] for g in grids], [g.shape[-2] for g in grids]
    assert all(o.H == h and o.W == w and o.C >= 3 for o, h, w in zip(outs, hs, ws))
    out_ptrs = (hip.C.c_void_p * len(outs))(*[o.ptr for o in outs])
    # algorithmic bytes: per pixel the grid (8 B), 4 taps x 3 channels per texture layer gathered, 3 floats written
    nbytes = sum(h * w for h, w in zip(hs, ws)) * (8 + 48 * len(layers) + 12)
    _hbm_timed("tex_sample_fwd", nbytes, lambda: hip.check(
        lib.sm_tex_sample_fwd_grouped(hip.ptr_array(layers), hip.int_array([l.shape[2] for l in layers]),
                                      hip.int_array([l.shape[1] for l in layers]), len(layers), hip.ptr_array(grids),
                                      hip.int_array(hs), hip.int_array(ws), out_ptrs, len(grids), hip.stream()),
        "sm_tex_sample_fwd_grouped"))


def tex_sample_bwd(grad_layers, grid: torch.Tensor, grad_img: FMap, pixel_weight=None):
    h, w = grid.shape[-3], grid.shape[-2]
    assert grad_img.H == h and grad_img.W == w
    hip.check(lib.sm_tex_sample_bwd(hip.ptr_array(grad_layers), hip.int_array([l.shape[2] for l in grad_layers]),
                                    hip.int_array([l.shape[1] for l in grad_layers]), len(grad_layers), ptr(grid),
                                    h, w, grad_img.ptr, ptr(pixel_weight), hip.stream()), "sm_tex_sample_bwd")


class ScatterPlan:
    """Sorted (texel, pixel, weight) list of one view over all its UV levels (``sm_tex_scatter_plan``): built once per
    view, used by every step's ``tex_scatter_planned``. Buffers are allocated once per (entry count) and reused."""

    def __init__(self, grad_layers, arena_grad: torch.Tensor):
        self.grad_layers, self.arena = grad_layers, arena_grad
        self.key_bits = max(1, int(arena_grad.numel()).bit_length())   # all-ones key > every arena offset
        self.lw = hip.int_array([l.shape[2] for l in grad_layers])
        self.lh = hip.int_array([l.shape[1] for l in grad_layers])
        self.n_entries = 0
        self.bufs = None
        self.capacity = 0
        self.packed = None
        self.generation = 0      # bumped whenever a buffer moves
        self.sorted_in = 0
        self.level_hw = None

    def build(self, grids, pixel_weights):
        """``grids``: [h,w,2] tensors (one per level), ``pixel_weights``: [h,w] tensors or None entries."""
        hw = [(g.shape[-3], g.shape[-2]) for g in grids]
        n = 4 * len(self.grad_layers) * sum(h * w for h, w in hw)
        self._ensure(n, hw)
        k0, k1, v0, v1, tmp, cross = self.bufs
        which = C.c_int(0)
        if getattr(self, "_static_key", None) != (tuple(hw), k0.data_ptr()):   # the call's view-independent arguments, once
            self._static_key = (tuple(hw), k0.data_ptr())
            self._static = (hip.ptr_array(self.grad_layers), hip.int_array([h for h, _ in hw]),
                            hip.int_array([w for _, w in hw]), ptr(k0), ptr(k1), ptr(v0), ptr(v1), ptr(tmp), tmp.numel(), ptr(cross))
        gl, hs, ws_, pk0, pk1, pv0, pv1, ptmp, ntmp, pcross = self._static
        hip.check(lib.sm_tex_scatter_plan(gl, self.lw, self.lh, len(self.grad_layers), ptr(self.arena), hip.ptr_array(grids),
                                          hip.ptr_array(pixel_weights), hs, ws_, len(hw), pk0, pk1, pv0, pv1, ptmp, ntmp,
                                          pcross, self.key_bits, C.byref(which), hip.stream()), "sm_tex_scatter_plan")
        self.sorted_in = which.value

    def _ensure(self, n, hw):
        """Buffers for ``n`` entries over levels ``hw``."""
        dev = self.arena.device
        # GROW-ONLY buffers: the views of a scene do not all populate the same UV levels, so n changes from view to view.
        # Re-allocating on every change would move the buffers under everything that holds their addresses (a recorded
        # step program, runtime/program.py) - and did: a replayed scatter wrote its chunk sums into freed memory.
        if self.bufs is None or self.capacity < n:
            tb = lib.sm_tex_scatter_plan_temp_bytes(n, self.key_bits)
            self.bufs = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
                         torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int64, device=dev),
                         torch.empty(max(tb, 16), dtype=torch.uint8, device=dev),
                         torch.empty(lib.sm_tex_scatter_plan_cross_bytes(n), dtype=torch.uint8, device=dev))
            self.capacity = n
            self.generation += 1
        self.n_entries = n
        planes = 4 * sum(hip.plane(h, w) for h, w in hw)
        if self.packed is None or self.packed.numel() < planes:
            self.packed = torch.empty(planes, device=dev)
            self.generation += 1
        self.level_hw = hw

    def live_share(self) -> float:
        """Share of the current view's entries that carry weight (the others sort to the tail and are skipped). Synchronises:
        for reports, not for the step."""
        invalid = 0xffffffff if self.key_bits == 32 else (1 << self.key_bits) - 1
        keys = self.bufs[self.sorted_in][:self.n_entries]
        signed = invalid - (1 << 32) if invalid >= (1 << 31) else invalid        # (the key buffers are int32 tensors)
        return float((keys != signed).float().mean())

    def scatter(self, grad_imgs, accumulate=True):
        """``grad_imgs``: the levels' image-gradient FMaps, in the order of ``build``. Adds into the gradient arena;
        ``accumulate=False`` when the arena is known to be zero (texels are then stored without being read)."""
        assert self.level_hw is not None and [(g.H, g.W) for g in grad_imgs] == self.level_hw
        keys, vals = self.bufs[self.sorted_in], self.bufs[2 + self.sorted_in]
        # algorithmic bytes (upper bound: every entry live - bench.py scales the entry term by ``live_share()``): key + value
        # + the pixel's packed gradient per entry, the image gradients packed once (3 planes read, 16 B per pixel written)
        nbytes = self.n_entries * (4 + 8 + 16) + sum(h * w for h, w in self.level_hw) * 28
        _hbm_timed("scatter", nbytes, lambda: hip.check(
            lib.sm_tex_scatter_planned(ptr(keys), ptr(vals), self.n_entries, hip.ptr_array(grad_imgs),
                                       hip.int_array([h for h, _ in self.level_hw]),
                                       hip.int_array([w for _, w in self.level_hw]), len(self.level_hw),
                                       hip.ptr_array(self.grad_layers), self.lw, self.lh, len(self.grad_layers),
                                       ptr(self.arena), self.key_bits, ptr(self.packed), ptr(self.bufs[5]),
                                       int(accumulate), hip.stream()), "sm_tex_scatter_planned"))


def tex_touch_flags(grad_layers, arena_grad: torch.Tensor, grid: torch.Tensor, pixel_weight, flags: torch.Tensor,
                    chunk_log2: int):
    """ORs into ``flags`` (int32, one per 2^chunk_log2 floats of the flat gradient arena) the chunks that
    ``tex_sample_bwd`` can write for this view level."""
    h, w = grid.shape[-3], grid.shape[-2]
    assert flags.dtype == torch.int32 and flags.numel() * (1 << chunk_log2) >= arena_grad.numel()
    hip.check(lib.sm_tex_touch_flags(hip.ptr_array(grad_layers), hip.int_array([l.shape[2] for l in grad_layers]),
                                     hip.int_array([l.shape[1] for l in grad_layers]), len(grad_layers),
                                     ptr(arena_grad), ptr(grid), h, w, ptr(pixel_weight), ptr(flags), chunk_log2,
                                     hip.stream()), "sm_tex_touch_flags")


def adam_hyper(lr, step, beta1=0.9, beta2=0.999):
    """The two step-dependent fp32 scalars of the update, computed in double as torch does:
    (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t))."""
    return lr / (1.0 - beta1 ** step), 1.0 / (1.0 - beta2 ** step) ** 0.5


def adam_fused(p, g, m, v, seg_end, reg_coef, lr, step, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-8,
               zero_grad=True, sumsq_out=None, dev_hyper=None, lo=0, hi=None, touched=None, touched_log2=0):
    """``lo`` / ``hi``: update only arena elements [lo, hi) (multiples of 4; the pipelined multi-GPU exchange updates
    the arena range by range) - same kernel on offset pointers, segment ends shifted and clipped to the range.
    ``touched``: int32 flag per 2^touched_log2 floats of the WHOLE arena; chunks with flag 0 are skipped (exact only for
    elements with p = g = m = v = 0, see sm_adam_fused). ``g`` may be None: the data-term gradient is zero wherever
    the launch walks (neither read nor zeroed)."""
    if lo != 0 or (hi is not None and hi != p.numel()):
        hi = p.numel() if hi is None else hi
        assert lo % 4 == 0 and 0 <= lo <= hi <= p.numel()
        if hi == lo:
            return
        if touched is not None:
            assert lo % (1 << touched_log2) == 0, "range start must be chunk-aligned"
            touched = touched[lo >> touched_log2:]
        p, g, m, v = p[lo:hi], (None if g is None else g[lo:hi]), m[lo:hi], v[lo:hi]
        seg_end = [min(max(int(e) - lo, 0), hi - lo) for e in seg_end]
    n = p.numel()
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    if touched is not None:
        assert touched.dtype == torch.int32 and touched.numel() * (1 << touched_log2) >= n
    # algorithmic bytes: 7 (6 without the gradient arena) streams x 4 B per element of the FLAGGED chunks; the flagged share
    # is a device quantity: the launch is recorded with the whole range's bytes and bench.py scales by the share it reads
    tag = "adam_dense" if touched is None else ("adam_early(x flagged share)" if g is None else "adam_closing(x flagged share)")
    _hbm_timed(tag, n * (28 if g is not None else 24), lambda: hip.check(
        lib.sm_adam_fused(ptr(p), ptr(g), ptr(m), ptr(v), n, hip.size_array(seg_end), hip.float_array(reg_coef),
                          len(seg_end), lr, beta1, beta2, eps, bc1, bc2, grad_scale, CLAMP_LO, CLAMP_HI, int(zero_grad),
                          ptr(sumsq_out), ptr(dev_hyper), ptr(touched), touched_log2, hip.stream()), "sm_adam_fused"))


def adam_hyper_step(state, dev_hyper, beta1=0.9, beta2=0.999):
    """``state``: device float64 [lr, step]; advances the step and writes the step-dependent scalars of the fused update
    into ``dev_hyper`` (device float32 [2]: lr / bc1, 1 / sqrt(bc2)) - on the device, so the launch can live in a hipGraph."""
    assert state.dtype == torch.float64 and state.numel() == 2 and dev_hyper.dtype == torch.float32 and dev_hyper.numel() >= 2
    hip.check(lib.sm_adam_hyper_step(ptr(state), beta1, beta2, ptr(dev_hyper), hip.stream()), "sm_adam_hyper_step")


def step_begin(sumsq, coef, reg_out, zero_a, zero_b=None):
    """Regulariser loss of the current texture into ``reg_out`` + zero fill of the step's accumulators, one launch."""
    assert (zero_a is None or zero_a.numel() % 4 == 0) and (zero_b is None or zero_b.numel() % 4 == 0)
    hip.check(lib.sm_step_begin(ptr(sumsq), ptr(coef), sumsq.numel(), ptr(reg_out), ptr(zero_a),
                                0 if zero_a is None else zero_a.numel(), ptr(zero_b),
                                0 if zero_b is None else zero_b.numel(), hip.stream()), "sm_step_begin")


def copy_floats(dst, src, n):
    """dst[0:n] = src[0:n] (device fp32) as a library call: a recorded step has no torch launch in it."""
    hip.check(lib.sm_copy_floats(ptr(dst), ptr(src), int(n), hip.stream()), "sm_copy_floats")


def zero_floats(t):
    hip.check(lib.sm_zero_floats(ptr(t), t.numel(), hip.stream()), "sm_zero_floats")


def flags_or(dst, src):
    assert dst.dtype == src.dtype == torch.int32 and dst.numel() == src.numel()
    hip.check(lib.sm_flags_or(ptr(dst), ptr(src), dst.numel(), hip.stream()), "sm_flags_or")


def flags_compact(flags, idx_out, count_out, ws):
    """Ascending indices of the non-zero ``flags`` (int32) into ``idx_out`` (int32, same length), their number into
    ``count_out`` (int32 [1], device); ``ws``: int32 scratch of ``flags_compact_ws_ints(n)``. Deterministic, no sync."""
    assert flags.dtype == idx_out.dtype == count_out.dtype == ws.dtype == torch.int32
    assert idx_out.numel() >= flags.numel() and ws.numel() >= lib.sm_flags_compact_ws_ints(flags.numel())
    hip.check(lib.sm_flags_compact(ptr(flags), flags.numel(), ptr(idx_out), ptr(count_out), ptr(ws), hip.stream()),
              "sm_flags_compact")


def flags_compact_ws_ints(n: int) -> int:
    return lib.sm_flags_compact_ws_ints(n)


def chunks_gather(arena, idx, n_idx: int, chunk_log2: int, compact, n_idx_dev=None):
    """compact[j] = chunk idx[j] of ``arena`` (2^chunk_log2 floats each) for j < n (``n_idx_dev`` if given, else ``n_idx``)."""
    assert compact.numel() >= n_idx << chunk_log2 and arena.data_ptr() % 16 == 0 and compact.data_ptr() % 16 == 0
    hip.check(lib.sm_chunks_gather(ptr(arena), ptr(idx), ptr(n_idx_dev), n_idx, chunk_log2, ptr(compact), hip.stream()),
              "sm_chunks_gather")


def chunks_scatter(arena, idx, n_idx: int, chunk_log2: int, compact, scale: float = 1.0, n_idx_dev=None):
    """The inverse of ``chunks_gather``: chunk idx[j] of ``arena`` = scale * compact[j]."""
    assert compact.numel() >= n_idx << chunk_log2 and arena.data_ptr() % 16 == 0 and compact.data_ptr() % 16 == 0
    hip.check(lib.sm_chunks_scatter(ptr(arena), ptr(idx), ptr(n_idx_dev), n_idx, chunk_log2, ptr(compact), scale,
                                    hip.stream()), "sm_chunks_scatter")


def clamp_sumsq(p, seg_end, sumsq_out=None):
    hip.check(lib.sm_clamp_sumsq(ptr(p), p.numel(), hip.size_array(seg_end), len(seg_end), CLAMP_LO, CLAMP_HI,
                                 ptr(sumsq_out), hip.stream()), "sm_clamp_sumsq")


# ---- VGG -----------------------------------------------------------------------------------------------------
_SPLITK_WS = {}   # device -> scratch for split-K partial tiles (caller-owned, see sm_conv3x3)
SPLITK_WS_FLOATS = 16 << 20


def splitk_workspace(device) -> torch.Tensor:
    # one scratch per (device, launch stream): conv launches of two streams (the step's trunk, a view being prepared
    # ahead on a side stream) must not share their split-K partial slabs
    key = (str(device), hip.stream())
    if key not in _SPLITK_WS:
        _SPLITK_WS[key] = torch.empty(SPLITK_WS_FLOATS, dtype=torch.float32, device=device)
    return _SPLITK_WS[key]


def conv3x3(inp: FMap, wt: torch.Tensor, bias, out: FMap, flags: int, gate: FMap | None = None, wt2=None,
            amax_in=None, amax_out=None):
    cin_pad, cout = wt.shape[1], wt.shape[2]
    assert inp.C >= cin_pad and out.C == cout and (inp.H, inp.W) == (out.H, out.W)
    if (wt2 is not None and CONV_MODE == "split2") or amax_out is not None:
        return conv3x3_grouped([(inp, out, gate)], wt, bias, flags, None, 1.0, wt2, amax_in, amax_out)

    def run():
        ws = splitk_workspace(wt.device)
        hip.check(lib.sm_conv3x3(inp.ptr, ptr(wt), ptr(bias), out.ptr, ptr(gate), cin_pad, cout, inp.H, inp.W,
                                 flags, ptr(ws), ws.numel(), hip.stream()), "sm_conv3x3")
    if CONV_TIMER is None:
        run()
    else:   # algorithmic FLOPs: true channel counts (the first layer has 3, not its padded 4) and true pixels
        cin_true = 3 if cin_pad == 4 else cin_pad
        CONV_TIMER.launch(run, 2.0 * 9 * cin_true * cout * inp.H * inp.W)


AMAX_FLOATS = lib.sm_amax_floats()   # floats of one "amax" bound (64 slots, 256 bytes apart; value = max over them)


def new_amax(device, value: float = 0.0) -> torch.Tensor:
    """A zeroed amax bound (``sm_amax_floats`` floats), optionally preset to ``value`` (tests / one-off callers)."""
    t = torch.zeros(AMAX_FLOATS, dtype=torch.float32, device=device)
    if value:
        t[0] = float(value)
    return t


def fmap_amax(f: FMap, amax_out: torch.Tensor):
    """max |x| of ``f`` max-ed into the (caller-zeroed) amax bound ``amax_out``."""
    assert amax_out.numel() == AMAX_FLOATS
    hip.check(lib.sm_fmap_amax(f.ptr, f.C, f.H, f.W, ptr(amax_out), hip.stream()), "sm_fmap_amax")


def conv_tile_positions(cin_pad: int, cout: int) -> int:
    """Positions per tile of the kernel that ``conv3x3_grouped`` will pick for this layer shape (``CONV_MODE``)."""
    if CONV_MODE == "split2" and split_eligible(cin_pad, cout):
        return lib.sm_conv_split2_tile_positions(cout)
    return lib.sm_conv_tile_positions(cin_pad, cout)


def conv_list_format(cin_pad: int, cout: int):
    """(positions per active-list entry, entries per tile) of the kernel ``conv3x3_grouped`` will pick: the split kernels
    take 32-position SEGMENTS, tile positions / 32 of them per tile (any live segments of one level, padded per level with
    (level << 24) | 0xFFFFFF: ``sparsity.build_tile_lists``); the fp32 kernel whole tiles (entries per tile 0)."""
    bn = conv_tile_positions(cin_pad, cout)
    if CONV_MODE == "split2" and split_eligible(cin_pad, cout):
        return 32, bn // 32
    return bn, 0


def conv3x3_grouped(problems, wt: torch.Tensor, bias, flags: int, tile_list=None, active_fraction=1.0,
                    wt2=None, amax_in=None, amax_out=None, quads=False):
    """One launch over several feature maps: ``problems`` = [(inp, out, gate-or-None[, code[, pooled, pool_code]]), ...]
    (FMaps; with a ``code`` tensor - 'split2' mode only - ``inp`` is the gradient of the 2x2-pooled map and the kernel
    takes the pool's backward on the fly, see ``maxpool_fwd_grouped``; with ``flags & EPI_POOL`` - 'split2' mode, a
    PAIR list from ``sparsity.build_tile_lists`` - the launch writes the pooled map ``pooled`` and its argmax codes
    ``pool_code`` instead of ``out``; with ``flags & EPI_GRAM`` a 7th element (ws, mask0, mask1-or-None, amax_feat,
    amax_d) - the operand images ``gram_backward_grouped`` left in ``ws`` for a problem without ``dfeat`` and the style
    layer's masks / bounds - makes the epilogue add the layer's masked Gram backward).
    ``tile_list``: optional int32 device tensor of active tiles ((problem << 24) | tile).
    ``wt2``: ``pack_conv_split2`` result (pack, w_scale_inv); used when ``CONV_MODE == 'split2'`` together with
    ``amax_in`` (device float: upper bound of max |input|). ``amax_out`` (device float, caller-zeroed, any mode):
    receives max |output| of the launch.
    ``quads``: ``tile_list`` holds vertical QUADS of segments (``cover_segments`` quad modes; 'split2' mode, 64 output
    channels, fp32 planes): the launch takes the resident-input kernel (``SM_LIST_QUADS``)."""
    cin_pad, cout = wt.shape[1], wt.shape[2]
    if tile_list is not None and tile_list.numel() == 0:
        return
    arr = (hip.ConvProblem * len(problems))()
    flops = nbytes = 0.0
    for i, prob in enumerate(problems):
        inp, out, gate = prob[:3]
        code = prob[3] if len(prob) > 3 else None   # "unpool" input: inp = pooled gradient, code = the pool's argmax codes
        if code is None:
            assert inp.C >= cin_pad and out.C == cout and (inp.H, inp.W) == (out.H, out.W)
        else:
            assert CONV_MODE == "split2" and inp.C >= cin_pad and out.C == cout
            assert (inp.H, inp.W) == (out.H // 2, out.W // 2) and code.dtype == torch.int32
            assert code.numel() >= inp.C // 8 * inp.plane
        pooled, pool_code = (prob[4], prob[5]) if len(prob) > 5 else (None, None)
        if flags & hip.EPI_POOL:
            assert CONV_MODE == "split2" and tile_list is not None and pooled is not None and pool_code.dtype == torch.int32
            assert (pooled.C, pooled.H, pooled.W) == (cout, out.H // 2, out.W // 2) and pool_code.numel() >= cout // 8 * pooled.plane
        gram = prob[6] if len(prob) > 6 else None
        if flags & hip.EPI_GRAM:
            assert CONV_MODE == "split2" and gram is not None and code is not None and cout in (64, 128) and gate is not None
        gws, gm0, gm1, gaf, gad = gram if gram is not None else (None,) * 5
        arr[i] = hip.ConvProblem(inp.ptr, out.ptr, ptr(gate), out.H, out.W, ptr(code),
                                 None if pooled is None else pooled.ptr, ptr(pool_code),
                                 ptr(gws), ptr(gm0), ptr(gm1), ptr(gaf), ptr(gad))
        cin_true = 3 if cin_pad == 4 else cin_pad
        flops += 2.0 * 9 * cin_true * cout * out.H * out.W
        # algorithmic HBM bytes: input read once, output written once (pooled: a quarter + 1/2 byte of codes per element),
        # + the epilogue's gate / addend reads
        streams = cout + (cout if flags & hip.EPI_RELU_MASK else 0) + (cout if flags & hip.EPI_ADD else 0)
        if flags & hip.EPI_POOL:
            streams = cout * (0.25 + 0.125 / 4)
        nbytes += 4.0 * streams * out.H * out.W + (4.0 if code is None else 4.5) * cin_true * inp.H * inp.W

    use_split2 = wt2 is not None and amax_in is not None and CONV_MODE == "split2"
    assert not quads or (use_split2 and tile_list is not None and cout == 64), "quad lists: fp16x2 kernel, 64 output channels"
    if quads and os.environ.get("STYLEMESH_VALIDATE_LISTS", "0") == "1":
        check_quad_list(tile_list, [(p[1].H, p[1].W) for p in problems], unpool=len(problems[0]) > 3 and problems[0][3] is not None)

    def run():
        ws = splitk_workspace(wt.device)
        n_list = 0 if tile_list is None else tile_list.numel()
        if use_split2:
            hip.check(lib.sm_conv3x3_grouped_split2(arr, len(problems), ptr(wt2[0]), wt2[1], ptr(bias), cin_pad, cout,
                                                    flags | (hip.LIST_QUADS if quads else 0), ptr(tile_list), n_list,
                                                    ptr(ws), ws.numel(), ptr(amax_in),
                                                    ptr(amax_out), hip.stream()), "sm_conv3x3_grouped_split2")
            return
        hip.check(lib.sm_conv3x3_grouped(arr, len(problems), ptr(wt), ptr(bias), cin_pad, cout, flags, ptr(tile_list), n_list,
                                         ptr(ws), ws.numel(), ptr(amax_out), hip.stream()), "sm_conv3x3_grouped")
    if CONV_TIMER is None:
        run()
    else:   # algorithmic FLOPs of the tiles actually required
        tag = "split2" if use_split2 else "f32"
        wbytes = wt2[0].numel() * 2 if use_split2 else wt.numel() * 4
        CONV_TIMER.launch(run, flops * active_fraction, tag, nbytes * active_fraction + wbytes,
                          f"{cin_pad:3d}->{cout:3d} flags {flags} levels {len(problems)} active {active_fraction:.2f}")


def check_quad_list(tile_list, hws, unpool=False):
    """The preconditions of ``SM_LIST_QUADS`` (include/stylemesh_hip.h), checked on the HOST (a device-to-host copy and a
    sync: debugging only, ``STYLEMESH_VALIDATE_LISTS=1``): four entries per quad, one problem per quad, the first entry a
    live segment inside the plane's interior rows, entry i either padding or the first one + i rows; un-pooling launches:
    first column even, first row a multiple of four. Raises ``ValueError`` - the library itself does not look at the
    entries and would compute wrong outputs for a list that is not made of quads."""
    e = tile_list.detach().cpu().numpy().astype("int64")
    if e.size % 4:
        raise ValueError(f"quad list of {e.size} entries: not a multiple of four")
    e = e.reshape(-1, 4)
    prob, seg = e >> 24, e & 0xFFFFFF
    if (prob < 0).any() or (prob >= len(hws)).any() or (prob != prob[:, :1]).any():
        raise ValueError("quad list: a quad's entries name several problems or a problem the launch does not have")
    pad = seg == 0xFFFFFF
    if pad[:, 0].any():
        raise ValueError("quad list: a quad's first entry is padding")
    for g, (H, W) in enumerate(hws):
        s = seg[prob[:, 0] == g]
        if s.size == 0:
            continue
        Wp = hip.row_stride(W)
        live = s != 0xFFFFFF
        rows = s[:, :1] // Wp - 1
        if (rows < 0).any() or (rows >= H).any() or (s[:, :1] % Wp < 1).any():
            raise ValueError(f"quad list, problem {g}: a first segment outside the image rows / left of column 0")
        want = s[:, :1] + Wp * np.arange(4)[None, :]
        if ((s != want) & live).any():
            raise ValueError(f"quad list, problem {g}: a quad's entries are not vertically adjacent segments")
        if unpool and (((s[:, 0] % Wp - 1) % 2 != 0).any() or (rows[:, 0] % 4 != 0).any()):
            raise ValueError(f"quad list, problem {g}: un-pooling quads start on even columns of rows 4 Y")


def conv3x3_dgrad_c3(dz: FMap, wd: torch.Tensor, out: FMap):
    assert wd.shape[2] == 4 and wd.shape[1] == dz.C and out.C >= 3
    hip.check(lib.sm_conv3x3_dgrad_c3(dz.ptr, ptr(wd), out.ptr, dz.C, dz.H, dz.W, hip.stream()), "sm_conv3x3_dgrad_c3")


def maxpool_fwd(inp: FMap, out: FMap):
    assert (out.H, out.W, out.C) == (inp.H // 2, inp.W // 2, inp.C)
    hip.check(lib.sm_maxpool2x2_fwd(inp.ptr, out.ptr, inp.C, inp.H, inp.W, hip.stream()), "sm_maxpool2x2_fwd")


def maxpool_bwd_relu(act: FMap, pooled: FMap, dpooled: FMap, dact: FMap):
    hip.check(lib.sm_maxpool2x2_bwd_relu(act.ptr, pooled.ptr, dpooled.ptr, dact.ptr, act.C, act.H, act.W,
                                         hip.stream()), "sm_maxpool2x2_bwd_relu")


def _plane_problems(items):
    arr = (hip.PlaneProblem * len(items))()
    for i, (a, b, c, out) in enumerate(items):
        arr[i] = hip.PlaneProblem(a.ptr, None if b is None else b.ptr, None if c is None else c.ptr, out.ptr, a.H, a.W)
    return arr


def plane_tile_positions(kind: int) -> int:
    """Positions per block of the grouped plane kernels (0: conv1_1 dgrad, image plane; 1 / 2: pools, pooled plane)."""
    return lib.sm_plane_tile_positions(kind)


def _tile_args(tile_list):
    return (None, 0) if tile_list is None else (ptr(tile_list), tile_list.numel())


def conv3x3_dgrad_c3_grouped(problems, wd: torch.Tensor, tile_list=None):
    """``problems``: [(dz, out), ...] FMaps of the UV levels - one launch. ``tile_list``: optional int32 device tensor
    of active blocks ((problem << 24) | block of 1024 image positions); the others are neither read nor written."""
    assert wd.shape[2] == 4 and all(dz.C == wd.shape[1] and out.C >= 3 for dz, out in problems)
    if tile_list is not None and tile_list.numel() == 0:
        return
    arr = _plane_problems([(dz, None, None, out) for dz, out in problems])
    # algorithmic bytes: the listed blocks' 64 gradient planes read once, 3 image planes written (dense: every position)
    pos = (tile_list.numel() * plane_tile_positions(0)) if tile_list is not None else sum(dz.H * dz.Wp for dz, _ in problems)
    _hbm_timed("conv1_1_dgrad", 4 * pos * (wd.shape[1] + 3), lambda: hip.check(
        lib.sm_conv3x3_dgrad_c3_tiles(arr, len(problems), ptr(wd), wd.shape[1], *_tile_args(tile_list), hip.stream()),
        "sm_conv3x3_dgrad_c3_tiles"))


def maxpool_fwd_grouped(problems, tile_list=None, codes=None):
    """``problems``: [(inp, out), ...] with equal channel counts - one launch. ``tile_list``: active blocks of 256
    positions of the pooled planes. ``codes``: optional int32 tensors [C / 8 * plane(out)], one per problem, that
    receive the argmax codes (a nibble per channel: which window element held the maximum; 4: none > 0) for
    ``conv3x3_grouped``'s unpool input."""
    C = problems[0][0].C
    assert all((o.H, o.W, o.C) == (i.H // 2, i.W // 2, C) and i.C == C for i, o in problems)
    if tile_list is not None and tile_list.numel() == 0:
        return
    arr = _plane_problems([(i, None, None, o) for i, o in problems])
    if codes is not None:
        assert all(c.dtype == torch.int32 and c.numel() >= C // 8 * o.plane for c, (_, o) in zip(codes, problems))
    hip.check(lib.sm_maxpool2x2_fwd_codes_tiles(arr, None if codes is None else hip.ptr_array(codes), len(problems), C,
                                                *_tile_args(tile_list), hip.stream()), "sm_maxpool2x2_fwd_codes_tiles")


def maxpool_bwd_relu_grouped(problems, tile_list=None):
    """``problems``: [(act, pooled, dpooled, dact), ...] - one launch; ``tile_list`` as for the forward."""
    C = problems[0][0].C
    if tile_list is not None and tile_list.numel() == 0:
        return
    arr = _plane_problems(problems)
    hip.check(lib.sm_maxpool2x2_bwd_relu_tiles(arr, len(problems), C, *_tile_args(tile_list), hip.stream()),
              "sm_maxpool2x2_bwd_relu_tiles")


# ---- losses --------------------------------------------------------------------------------------------------
def gram_num_slabs(C: int, H: int, W: int) -> int:
    """How many leading slabs of the Gram workspace sum to S (mode dependent: the split kernel accumulates into one)."""
    if GRAM_MODE == "split2":
        return lib.sm_gram_split_num_slabs()
    return lib.sm_gram_num_slabs(C, H, W)


def gram_workspace_slabs(C: int, H: int, W: int) -> int:
    """Slabs the workspace must hold in the current mode (split: the one slab every position range adds into)."""
    if GRAM_MODE == "split2":
        return lib.sm_gram_split_num_slabs()
    return lib.sm_gram_workspace_slabs(C, H, W)


# 'split2' = the Gram contraction and its backward GEMM on the fp16 matrix cores with fp16x2-split operands scaled by
# powers of two from recorded maxima (3 partial products; see CONV_MODE 'split2'); 'f32' = v_mfma_f32_32x32x2_f32
# kernels. Default: follows the conv mode.
GRAM_MODE = os.environ.get("STYLEMESH_GRAM_MODE", CONV_MODE)
if GRAM_MODE not in ("split2", "f32"):
    raise ValueError(f"STYLEMESH_GRAM_MODE={GRAM_MODE!r}: 'split2' or 'f32'")


def gram_masked(feat: FMap, mask0, mask1, S0, S1, prezeroed=False, amax_feat=None):
    """S0 / S1: [gram_workspace_slabs(C,H,W), C, C]; returns how many leading slabs sum to S. ``prezeroed`` (split
    modes only): the caller has zeroed the slabs, the kernel adds into them without a fill of its own.
    ``amax_feat`` (device float; 'split2' mode): bound of max |feat|."""
    n = gram_num_slabs(feat.C, feat.H, feat.W)
    na = gram_workspace_slabs(feat.C, feat.H, feat.W)
    assert S0.numel() >= na * feat.C * feat.C and (S1 is None or S1.numel() >= na * feat.C * feat.C)
    if GRAM_MODE == "split2":
        assert amax_feat is not None, "GRAM_MODE 'split2' needs the feature map's bound"
        fn = lib.sm_gram_masked_split_acc if prezeroed else lib.sm_gram_masked_split
        hip.check(fn(feat.ptr, ptr(mask0), ptr(mask1), ptr(S0), ptr(S1), feat.C, feat.H, feat.W,
                     ptr(amax_feat), hip.stream()), "sm_gram_masked_split")
        return n
    hip.check(lib.sm_gram_masked(feat.ptr, ptr(mask0), ptr(mask1), ptr(S0), ptr(S1), feat.C, feat.H, feat.W,
                                 hip.stream()), "sm_gram_masked")
    return n


def gram_problem(feat: FMap, mask0, mask1, S0, S1, amax_feat) -> "hip.GramProblem":
    """One entry of ``gram_masked_grouped`` (pointers only: valid while the tensors live)."""
    return hip.GramProblem(feat.ptr, ptr(mask0), ptr(mask1), ptr(S0), ptr(S1), ptr(amax_feat), feat.C, feat.H, feat.W)


def gram_problem_array(problems):
    arr = (hip.GramProblem * len(problems))()
    for i, p in enumerate(problems):
        arr[i] = p
    return arr


def gram_masked_grouped(arr):
    """fp16x2 Gram forward of many (level, layer) problems (``gram_problem_array``) in at most two launches; every
    S0 / S1 must be zero on entry (the position ranges add into them)."""
    hip.check(lib.sm_gram_masked_split2_grouped(arr, len(arr), hip.stream()), "sm_gram_masked_split2_grouped")


def style_problem(S0, S1, counts, factor, targets, term_mask, skip_if_empty, weight, C, D0, D1, amax_d_out,
                  n_slabs=1) -> "hip.StyleProblem":
    """One entry of ``style_loss_grouped`` (same meanings as the arguments of ``style_loss``; no Gram history)."""
    p = hip.StyleProblem()
    p.S0, p.S1, p.counts, p.factor = ptr(S0), ptr(S1), ptr(counts), ptr(factor)
    for i, t in enumerate(targets):
        p.targets[i] = ptr(t)
        p.term_mask[i] = int(term_mask[i])
    p.n_terms = len(targets)
    p.skip_if_empty[0], p.skip_if_empty[1] = int(skip_if_empty[0]), int(skip_if_empty[1])
    p.weight, p.C, p.D0, p.D1 = float(weight), int(C), ptr(D0), ptr(D1)
    p.history, p.hist_len, p.hist_slot, p.n_slabs, p.amax_d_out = None, 0, 0, int(n_slabs), ptr(amax_d_out)
    return p


def gram_bwd_problem(feat: FMap, mask0, mask1, D0, D1, dfeat: FMap, ws, amax_feat, amax_d, relu_gate,
                     amax_out=None) -> "hip.GramBwdProblem":
    """One entry of ``gram_backward_grouped``; ``ws``: uint8 scratch of ``gram_backward_ws_bytes(C)`` bytes of its own;
    ``amax_out`` (optional amax bound): max |dfeat| is max-ed into it. ``dfeat`` None: only the operand images of D0 / D1
    are written into ``ws`` (a conv launch with ``EPI_GRAM`` consumes them)."""
    assert ws.numel() >= lib.sm_gram_backward_split_ws_bytes(feat.C)
    return hip.GramBwdProblem(feat.ptr, ptr(mask0), ptr(mask1), ptr(D0), ptr(D1), None if dfeat is None else dfeat.ptr,
                              ptr(ws), ptr(amax_feat),
                              ptr(amax_d), ptr(amax_out), feat.C, feat.H, feat.W, int(relu_gate))


def gram_backward_ws_bytes(C: int) -> int:
    return lib.sm_gram_backward_split_ws_bytes(C)


def struct_array(kind, problems):
    arr = (kind * len(problems))()
    for i, p in enumerate(problems):
        arr[i] = p
    return arr


def style_loss_grouped(arr, loss_out):
    """Loss value (added into ``loss_out``) and derivative matrices of many (level, layer) problems in one launch."""
    hip.check(lib.sm_style_loss_grouped(arr, len(arr), ptr(loss_out), hip.stream()), "sm_style_loss_grouped")


def gram_backward_grouped(arr):
    """fp16x2 Gram backward of many (level, layer) problems: one packing launch + one GEMM launch per tile class."""
    hip.check(lib.sm_gram_backward_split2_grouped(arr, len(arr), hip.stream()), "sm_gram_backward_split2_grouped")


def style_loss(S0, S1, counts, factor, targets, term_mask, skip_if_empty, weight, C, D0, D1, loss_out,
               history=None, hist_len=0, hist_slot=0, n_slabs=1, amax_d_out=None):
    hip.check(lib.sm_style_loss(ptr(S0), ptr(S1), ptr(counts), ptr(factor), hip.ptr_array(targets),
                                hip.int_array(term_mask), len(targets), hip.int_array(skip_if_empty), weight, C,
                                ptr(D0), ptr(D1), ptr(loss_out), ptr(history), hist_len, hist_slot, n_slabs,
                                ptr(amax_d_out), hip.stream()), "sm_style_loss")


_GRAM_BWD_WS = {}   # device -> scratch for the fp16x2 operand image of D0 / D1 (largest C = 512)


def gram_backward(feat: FMap, mask0, mask1, D0, D1, dfeat: FMap, relu_gate: bool, amax_feat=None, amax_d=None):
    """``amax_feat`` / ``amax_d`` (device floats; 'split2' mode): bounds of max |feat| and max(|D0|, |D1|)."""
    if GRAM_MODE == "split2":
        assert amax_feat is not None and amax_d is not None, "GRAM_MODE 'split2' needs the operand bounds"
        key = (str(D0.device), hip.stream())   # one scratch per launch stream
        need = lib.sm_gram_backward_split_ws_bytes(feat.C)
        if key not in _GRAM_BWD_WS or _GRAM_BWD_WS[key].numel() < need:
            _GRAM_BWD_WS[key] = torch.empty(max(need, lib.sm_gram_backward_split_ws_bytes(512)), dtype=torch.uint8,
                                            device=D0.device)
        hip.check(lib.sm_gram_backward_split(feat.ptr, ptr(mask0), ptr(mask1), ptr(D0), ptr(D1), dfeat.ptr, feat.C,
                                             feat.H, feat.W, int(relu_gate), ptr(_GRAM_BWD_WS[key]),
                                             ptr(amax_feat), ptr(amax_d),
                                             hip.stream()), "sm_gram_backward_split")
        return
    hip.check(lib.sm_gram_backward(feat.ptr, ptr(mask0), ptr(mask1), ptr(D0), ptr(D1), dfeat.ptr, feat.C, feat.H,
                                   feat.W, int(relu_gate), hip.stream()), "sm_gram_backward")


def mse_masked(pred: FMap, target: FMap, mask, count, factor, weight, dpred: FMap, loss_out, relu_gate=False):
    hip.check(lib.sm_mse_masked(pred.ptr, target.ptr, ptr(mask), ptr(count), ptr(factor), weight, dpred.ptr,
                                ptr(loss_out), pred.C, pred.H, pred.W, int(relu_gate), hip.stream()),
              "sm_mse_masked")


# ---- per-view constants ----------------------------------------------------------------------------------------
def level_masks(rounded, other, interp_w, mask_u8, n_levels, E, Wt):
    h, w = mask_u8.shape[-2:]
    hip.check(lib.sm_level_masks(ptr(rounded), ptr(other), ptr(interp_w), ptr(mask_u8), h, w, n_levels, ptr(E),
                                 ptr(Wt), hip.stream()), "sm_level_masks")


def level_maps(E, Wt, angle_guidance, angle_deg, thr, h, w, H, W, M, pixel_weight, passed, m_sum):
    hip.check(lib.sm_level_maps(ptr(E), ptr(Wt), ptr(angle_guidance), ptr(angle_deg), thr, h, w, H, W, ptr(M),
                                ptr(pixel_weight), ptr(passed), ptr(m_sum), hip.stream()), "sm_level_maps")


def layer_masks(M, passed, H, W, hl, wl, m_all, m_pass, m_fail, counts):
    hip.check(lib.sm_layer_masks(ptr(M), ptr(passed), H, W, hl, wl, ptr(m_all), ptr(m_pass), ptr(m_fail),
                                 ptr(counts), hip.stream()), "sm_layer_masks")


def level_factors(counts_all, sizes, factors):
    hip.check(lib.sm_level_factors(hip.ptr_array(counts_all), hip.float_array(sizes), len(sizes),
                                   hip.ptr_array(factors), hip.stream()), "sm_level_factors")


def need_step(need_out, mode, M, need_src):
    """mode 0: mask only, 1: through a 3x3 conv, 2: through a 2x2 pool (see sm_need_step)."""
    ho, wo = need_out.shape if need_out is not None else (0, 0)
    H, W = M.shape if M is not None else (0, 0)
    hs, ws = need_src.shape
    hip.check(lib.sm_need_step(ptr(need_out), ho, wo, mode, ptr(M), H, W, ptr(need_src), hs, ws, hip.stream()),
              "sm_need_step")


def cover_segments(problems):
    """``problems``: [(need [h,w] float tensor, starts int32 tensor, count int32 [1] tensor, tag[, pair_w[, quad]]), ...] (<= 64):
    disjoint 32-position segments covering the needed positions of every plane, starts as (tag << 24) | q
    (``sm_cover_segments``). ``quad`` = 1: vertical QUADS of segments (four entries per run; the resident-input conv kernel)."""
    arr = (hip.CoverProblem * len(problems))()
    for i, prob in enumerate(problems):
        need, starts, count, tag = prob[:4]
        pair_w = prob[4] if len(prob) > 4 else 0   # > 0: PAIR mode (``need`` = need map of the pooled plane, see the header)
        quad = prob[5] if len(prob) > 5 else 0
        h, w = need.shape
        rows, width = (2 * h + 2, pair_w) if pair_w else (h + 2, w)
        if rows * hip.row_stride(width) >= 0xFFFFFF:
            raise ValueError(f"segment list of a {rows - 2} x {width} plane: a list entry holds the position in 24 bits "
                             f"(planes up to ~16.7 M padded positions); STYLEMESH_SEGMENT_LISTS=0 lists whole tiles instead")
        arr[i] = hip.CoverProblem(ptr(need), ptr(starts), ptr(count), h, w, int(tag), starts.numel(), int(pair_w), int(quad))
    nbytes = lib.sm_cover_segments_ws_bytes(arr, len(problems))
    dev = problems[0][0].device
    # grow-only scratch (bit images + chunk tables), one per (device, launch stream) - as the split-K slabs: the view being
    # prepared ahead on a side stream, the trunk's own view change and a second engine of the process (tests, bench legs) build
    # lists concurrently, and a scratch shared by two streams gave one of them another plane's bit image (a two-engine test
    # failed once in nine suite runs before round 6 keyed it by stream)
    key = (str(dev), hip.stream())
    ws = _COVER_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            torch.cuda.synchronize(dev)      # (enqueued launches may still be using the old one; rare)
        ws = _COVER_WS[key] = torch.empty(max(2 * nbytes, 8 << 20), dtype=torch.uint8, device=dev)
    hip.check(lib.sm_cover_segments(arr, len(problems), ptr(ws), ws.numel(), hip.stream()), "sm_cover_segments")


_COVER_WS = {}


def tile_flags(need, bn, flags):
    h, w = need.shape
    hip.check(lib.sm_tile_flags(ptr(need), h, w, bn, ptr(flags), hip.stream()), "sm_tile_flags")


def fmap_resize_bilinear(inp: FMap, out: FMap):
    hip.check(lib.sm_fmap_resize_bilinear(inp.ptr, inp.C, inp.H, inp.W, out.ptr, out.H, out.W, hip.stream()),
              "sm_fmap_resize_bilinear")


def image_to_fmap(img: torch.Tensor, out: FMap):
    c, h, w = img.shape[-3:]
    assert out.C >= c
    hip.check(lib.sm_image_to_fmap(ptr(img), c, h, w, out.ptr, out.H, out.W, hip.stream()), "sm_image_to_fmap")


def fmap_to_image(inp: FMap, channels=None) -> torch.Tensor:
    c = inp.C if channels is None else channels
    out = torch.empty(c, inp.H, inp.W, dtype=torch.float32, device=inp.buf.device)
    hip.check(lib.sm_fmap_to_image(inp.ptr, c, inp.H, inp.W, ptr(out), hip.stream()), "sm_fmap_to_image")
    return out
