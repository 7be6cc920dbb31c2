"""The VGG-19 conv stack of the loss (reference model/losses/content_and_style_losses.py:7-70) as a plan of
HIP kernel launches over padded-planar feature maps: forward (conv + bias + ReLU, 2x2 max-pool) and the
hand-derived backward (data gradients only - the weights are frozen, :43-45).

Only the layers that feed a requested output are ever computed (the reference runs all 16 convs and discards
conv5_2..5_4).
"""
from __future__ import annotations

import os

import torch

from . import hip, ops
from .fmap import FMap

# network order; ("pool", k) = MaxPool2d(2,2) after block k
NODES = [
    ("conv1_1", "img", "r11", 3, 64), ("conv1_2", "r11", "r12", 64, 64), ("pool", "r12", "p1", 64, 64),
    ("conv2_1", "p1", "r21", 64, 128), ("conv2_2", "r21", "r22", 128, 128), ("pool", "r22", "p2", 128, 128),
    ("conv3_1", "p2", "r31", 128, 256), ("conv3_2", "r31", "r32", 256, 256), ("conv3_3", "r32", "r33", 256, 256),
    ("conv3_4", "r33", "r34", 256, 256), ("pool", "r34", "p3", 256, 256),
    ("conv4_1", "p3", "r41", 256, 512), ("conv4_2", "r41", "r42", 512, 512), ("conv4_3", "r42", "r43", 512, 512),
    ("conv4_4", "r43", "r44", 512, 512), ("pool", "r44", "p4", 512, 512),
    ("conv5_1", "p4", "r51", 512, 512), ("conv5_2", "r51", "r52", 512, 512), ("conv5_3", "r52", "r53", 512, 512),
    ("conv5_4", "r53", "r54", 512, 512), ("pool", "r54", "p5", 512, 512),
]
OUT_NAMES = [n[2] for n in NODES]
PRE_POOL = {"r12", "r22", "r34", "r44", "r54"}


def depth_of(layer: str) -> int:
    if layer not in OUT_NAMES:
        raise ValueError(f"unknown VGG output key: {layer}")
    return OUT_NAMES.index(layer)


def layer_hw(layer: str, H: int, W: int):
    """Spatial size of a VGG output for an H x W input (floor halving per pool)."""
    h, w = H, W
    for kind, _, out, _, _ in NODES:
        if kind == "pool":
            h, w = h // 2, w // 2
        if out == layer:
            return h, w
    raise ValueError(layer)


POOL_INPUT = {out: src for kind, src, out, _, _ in NODES if kind == "pool"}     # p1 -> r12, ...
POOL_OUTPUT = {src: out for kind, src, out, _, _ in NODES if kind == "pool"}    # r12 -> p1, ...
NODE_BELOW = {out: (kind, src) for kind, src, out, _, _ in NODES}                # r12 -> ("conv1_2", "r11"), ...
# fp16x2 mode, grouped passes: the pool forward records argmax codes and the data-gradient conv below a pool takes the
# pool's backward from them while it stages its operand - no pool-backward pass (2.75 plane sizes of HBM traffic per
# pool) and a quarter-size operand read
FUSE_POOL_BWD = os.environ.get("STYLEMESH_FUSE_POOL_BWD", "1") != "0"


def fuse_pool_fwd() -> bool:
    """fp16x2 mode, grouped passes with active lists: the forward conv BELOW a pool takes the 2x2 maxima (and writes the
    argmax codes) in its epilogue (``hip.EPI_POOL``; its list holds vertical segment pairs, ``sparsity.build_tile_lists``
    key (conv, 'fp')) - no pool pass, and the pre-pool map, which nothing else reads once the pool backward is fused
    too, is never written."""
    return FUSE_POOL_BWD and ops.CONV_MODE == "split2" and os.environ.get("STYLEMESH_FUSE_POOL_FWD", "1") != "0"


class AmaxBook:
    """One device float per VGG tensor: an upper bound of max |x| of the activation ('a:<layer>') or gradient
    ('g:<layer>') planes, recorded by the kernel that writes them (``amax_out``) and read by the fp16x2-split conv that
    consumes them as its operand scale (``amax_in``; ``ops.CONV_MODE == 'split2'``). Max-pooling passes bounds through:
    a pooled activation is bounded by the pool's input, a pool-backward gradient by the pooled gradient. Zeroed once per
    pass (``zero()``), before the first kernel that records into it."""

    N = 2 * len(OUT_NAMES)
    W = ops.AMAX_FLOATS          # floats per bound (64 slots, 256 bytes apart: include/stylemesh_hip.h)

    def __init__(self, device, storage=None):
        names = ["a:" + n for n in OUT_NAMES] + ["g:" + n for n in OUT_NAMES]
        self.idx = {n: i for i, n in enumerate(names)}
        self.buf = torch.zeros(len(names) * self.W, dtype=torch.float32, device=device) if storage is None else storage
        assert self.buf.numel() == len(names) * self.W

    def __getitem__(self, name):
        i = self.idx[name]
        return self.buf[i * self.W:(i + 1) * self.W]

    def act_bound(self, layer):
        """bound of the activation planes named ``layer`` (a conv output or a pool output)"""
        return self["a:" + POOL_INPUT.get(layer, layer)]

    def grad_bound(self, layer):
        """bound of the gradient planes of ``layer`` (written by a dgrad conv / the loss kernels, or by a pool backward)"""
        return self["g:" + POOL_OUTPUT.get(layer, layer)]

    def zero(self):
        self.buf.zero_()


def _amax_on():
    return ops.CONV_MODE == "split2" or ops.GRAM_MODE == "split2"


class LevelBuffers:
    """All activation (and, if ``with_grad``, gradient) feature maps of one VGG pass at one input size.
    Dedicated per size so that the zero borders written at allocation stay valid forever."""

    def __init__(self, H: int, W: int, last_layer: str, with_grad: bool, device="cuda"):
        self.H, self.W, self.last = H, W, depth_of(last_layer)
        self.act = {"img": FMap(4, H, W, device)}  # 3 channels + 1 zero plane (K-chunk of 4)
        self.grad = {"img": FMap(3, H, W, device)} if with_grad else {}
        self.code = {}
        h, w = H, W
        for kind, _, out, _, cout in NODES[:self.last + 1]:
            if kind == "pool":
                h, w = h // 2, w // 2
            if h < 1 or w < 1:
                raise ValueError(f"input {H}x{W} too small for VGG layer {out}")
            self.act[out] = FMap(cout, h, w, device)
            if with_grad:
                self.grad[out] = FMap(cout, h, w, device)
                if kind == "pool":   # argmax codes of the pool (fused pool backward of the fp16x2 data-gradient convs)
                    self.code[out] = torch.zeros(cout // 8 * self.act[out].plane, dtype=torch.int32, device=device)
        self.amax = AmaxBook(device)   # bounds of this buffer set's tensors (single-level passes)

    def nbytes(self):
        return sum(f.buf.numel() * 4 for f in list(self.act.values()) + list(self.grad.values()))


class VGGNet:
    def __init__(self, state_dict: dict, device="cuda"):
        self.device = device
        self.wf, self.wd, self.bias = {}, {}, {}
        self.wf2, self.wd2 = {}, {}     # fp16x2-split packs (pack, 1 / weight scale) of the layers the split kernel takes
        for kind, _, _, _, _ in NODES:
            if kind == "pool":
                continue
            w = state_dict[kind + ".weight"].detach().to(device=device, dtype=torch.float32)
            self.wf[kind] = ops.pack_conv_fwd(w)
            self.wd[kind] = ops.pack_conv_dgrad(w)
            for packs, splits2 in ((self.wf, self.wf2), (self.wd, self.wd2)):
                p = packs[kind]
                splits2[kind] = ops.pack_conv_split2(p) if ops.split_eligible(p.shape[1], p.shape[2]) else None
            self.bias[kind] = state_dict[kind + ".bias"].detach().to(device=device, dtype=torch.float32).contiguous()

    def forward(self, b: LevelBuffers):
        """conv + bias + ReLU / max-pool chain from ``b.act['img']`` through the last layer of ``b``."""
        am = b.amax if _amax_on() else None
        if am is not None:
            am.zero()
        for kind, src, out, _, _ in NODES[:b.last + 1]:
            if kind == "pool":
                ops.maxpool_fwd(b.act[src], b.act[out])
            else:
                ops.conv3x3(b.act[src], self.wf[kind], self.bias[kind], b.act[out], hip.EPI_BIAS_RELU, wt2=self.wf2[kind],
                            amax_in=None if am is None or src == "img" else am.act_bound(src),
                            amax_out=None if am is None else am["a:" + out])

    def forward_group(self, bufs, tiles=None, on_layer=None, amax: AmaxBook | None = None):
        """``forward`` for several levels at once: one grouped conv launch per layer (all levels share the
        weights), which fills the chip where a single small level cannot. ``tiles``: optional active-tile
        lists from ``sparsity.build_tile_lists`` (only tiles that can influence the loss are computed).
        ``amax``: the group's bounds (one per layer over all levels; zeroed by the caller), fp16x2 mode only."""
        last = bufs[0].last
        assert all(b.last == last for b in bufs)
        am = amax if _amax_on() else None
        assert am is not None or not _amax_on(), "CONV_MODE 'split2' needs the group's AmaxBook"
        pooled_by_conv = set()   # pools whose output the conv below them has already written (EPI_POOL)
        quads = getattr(tiles, "quads", ())   # lists of vertical segment quads: the resident-input kernel (viewplan.TileLists)
        for kind, src, out, _, _ in NODES[:last + 1]:
            if kind == "pool":
                if out in pooled_by_conv:
                    continue
                fused = FUSE_POOL_BWD and ops.CONV_MODE == "split2" and all(out in b.code for b in bufs)
                ops.maxpool_fwd_grouped([(b.act[src], b.act[out]) for b in bufs],
                                        tiles[("pool", out)][0] if tiles else None,
                                        [b.code[out] for b in bufs] if fused else None)
            elif tiles and (kind, "fp") in tiles and fuse_pool_fwd() and all(POOL_OUTPUT[out] in b.code for b in bufs):
                po = POOL_OUTPUT[out]
                tl, frac = tiles[(kind, "fp")]
                ops.conv3x3_grouped([(b.act[src], b.act[out], None, None, b.act[po], b.code[po]) for b in bufs],
                                    self.wf[kind], self.bias[kind], hip.EPI_BIAS_RELU | hip.EPI_POOL, tl, frac,
                                    self.wf2[kind],
                                    None if am is None or src == "img" else am.act_bound(src),
                                    None if am is None else am["a:" + out], quads=(kind, "fp") in quads)
                pooled_by_conv.add(po)
                if on_layer is not None:
                    on_layer(out)
            else:
                if tiles and (kind, "f") not in tiles:
                    raise RuntimeError(f"{kind}: the tile lists were built for the pooling epilogue (STYLEMESH_FUSE_POOL_FWD) "
                                       "but this pass cannot use it (no code buffers / mode changed since set_view)")
                tl, frac = tiles[(kind, "f")] if tiles else (None, 1.0)
                ops.conv3x3_grouped([(b.act[src], b.act[out], None) for b in bufs], self.wf[kind], self.bias[kind],
                                    hip.EPI_BIAS_RELU, tl, frac, self.wf2[kind],
                                    None if am is None or src == "img" else am.act_bound(src),
                                    None if am is None else am["a:" + out], quads=(kind, "f") in quads)
                if on_layer is not None:
                    on_layer(out)      # the layer's activation is enqueued: side work may branch off here

    def backward_group(self, bufs, injected: set, start_layer: str, tiles=None, before_layer=None,
                       amax: AmaxBook | None = None, start_bound_recorded=False, gram_terms=None):
        """``backward`` for several levels at once (same injected layers on every level). ``amax``: as in
        ``forward_group``; the bound of the start layer's gradient (written by the loss kernels) is taken here unless
        the loss kernels recorded it themselves (``start_bound_recorded``)."""
        am = amax if _amax_on() else None
        assert am is not None or not _amax_on(), "CONV_MODE 'split2' needs the group's AmaxBook"
        if am is not None and not start_bound_recorded:
            for b in bufs:
                ops.fmap_amax(b.grad[start_layer], am["g:" + start_layer])
        fuse = FUSE_POOL_BWD and ops.CONV_MODE == "split2" and all(b.code for b in bufs)
        quads = getattr(tiles, "quads", ())   # (as in forward_group)
        unpool = None   # fused pool backward: name of the pooled map whose gradient the next conv un-pools on the fly

        for kind, src, out, _, _ in reversed(NODES[:depth_of(start_layer) + 1]):
            if before_layer is not None and kind != "pool":
                before_layer(src)      # the injected gradient of ``src`` is about to be consumed
            if kind == "pool":
                if src in injected:
                    raise ValueError(f"style/content layer {src} directly below a pool is not supported")
                if fuse and src != "img" and not NODE_BELOW[src][1].startswith("p") and NODE_BELOW[src][1] != "img":
                    unpool = out       # the conv that produced ``src`` reads grad[out] + code[out] instead of grad[src]
                    continue
                ops.maxpool_bwd_relu_grouped([(b.act[src], b.act[out], b.grad[out], b.grad[src]) for b in bufs],
                                             tiles[("pool", out)][0] if tiles else None)
            elif src == "img":
                ops.conv3x3_dgrad_c3_grouped([(b.grad[out], b.grad["img"]) for b in bufs], self.wd[kind],
                                             tiles[("img", "d")][0] if tiles else None)
            elif src.startswith("p"):
                tl, frac = tiles[(kind, "b")] if tiles else (None, 1.0)
                ops.conv3x3_grouped([(b.grad[out], b.grad[src], None) for b in bufs], self.wd[kind], None, 0, tl, frac,
                                    self.wd2[kind], None if am is None else am.grad_bound(out),
                                    None if am is None else am["g:" + src], quads=(kind, "b") in quads)
            else:
                tl, frac = tiles[(kind, "b")] if tiles else (None, 1.0)
                flags = hip.EPI_RELU_MASK | (hip.EPI_ADD if src in injected else 0)
                gt = gram_terms.get(src) if gram_terms else None
                if gt is not None:
                    # the style layer's Gram backward is computed in this launch's epilogue instead of being read from
                    # grad[src] (``gram_terms[src]``: per level (ws, mask0, mask1, amax_feat, amax_d), engine._style_group)
                    if unpool is None or len(gt) != len(bufs) or ops.CONV_MODE != "split2":
                        raise RuntimeError(f"{kind}: a fused Gram backward needs the un-pooling fp16x2 data gradient")
                    flags = hip.EPI_RELU_MASK | hip.EPI_GRAM
                if unpool is not None:
                    probs = [(b.grad[unpool], b.grad[src], b.act[src], b.code[unpool]) for b in bufs]
                    if gt is not None:
                        probs = [p + (None, None, g) for p, g in zip(probs, gt)]
                    unpool = None
                else:
                    probs = [(b.grad[out], b.grad[src], b.act[src]) for b in bufs]
                ops.conv3x3_grouped(probs, self.wd[kind], None, flags,
                                    tl, frac, self.wd2[kind], None if am is None else am.grad_bound(out),
                                    None if am is None else am["g:" + src], quads=(kind, "b") in quads)
            assert unpool is None or kind == "pool", "a fused pool backward must be consumed by the conv below it"

    def backward(self, b: LevelBuffers, injected: set, start_layer: str):
        """Back-propagate to ``b.grad['img']``.

        On entry ``b.grad[start_layer]`` holds the gradient w.r.t. the PRE-ReLU output of that (deepest) layer,
        and for every other layer in ``injected`` the buffer holds the loss gradient w.r.t. its activation;
        it is added to the gradient arriving from above and gated by the layer's ReLU in the data-gradient
        kernel's epilogue."""
        am = b.amax if _amax_on() else None
        if am is not None:   # gradient bounds of this pass (the activation bounds of the forward pass stay)
            am.buf[len(OUT_NAMES) * am.W:].zero_()
            ops.fmap_amax(b.grad[start_layer], am["g:" + start_layer])
        for kind, src, out, _, _ in reversed(NODES[:depth_of(start_layer) + 1]):
            if kind == "pool":
                if src in injected:
                    raise ValueError(f"style/content layer {src} directly below a pool is not supported")
                ops.maxpool_bwd_relu(b.act[src], b.act[out], b.grad[out], b.grad[src])
            elif src == "img":
                ops.conv3x3_dgrad_c3(b.grad[out], self.wd[kind], b.grad["img"])
            elif src.startswith("p"):
                ops.conv3x3(b.grad[out], self.wd[kind], None, b.grad[src], 0, wt2=self.wd2[kind],
                            amax_in=None if am is None else am.grad_bound(out),
                            amax_out=None if am is None else am["g:" + src])
            else:
                flags = hip.EPI_RELU_MASK | (hip.EPI_ADD if src in injected else 0)
                ops.conv3x3(b.grad[out], self.wd[kind], None, b.grad[src], flags, gate=b.act[src],                             wt2=self.wd2[kind], amax_in=None if am is None else am.grad_bound(out),
                            amax_out=None if am is None else am["g:" + src])
