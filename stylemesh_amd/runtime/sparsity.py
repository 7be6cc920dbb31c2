"""Which tiles of which VGG layer can influence the loss of a UV level?

The losses read a layer's features only where the level's mask is set (masked Gram, masked content MSE,
reference content_and_style_losses.py:136-143,301-348), and the level masks partition the view by depth
(model/model.py:210-221): a level typically covers 20-60 % of the image. Everything outside the mask's
receptive field is dead computation in the reference. ``need[layer]`` is the exact set of positions of a layer
whose value can reach the loss (and, equivalently, whose gradient can be non-zero):

    need[loss layer]        |= mask at that layer's resolution
    need[conv input]        |= dilate3x3(need[conv output])
    need[pool input]        |= 2x2 up-sampling of need[pool output]

The conv kernels then run only on the position tiles that intersect ``need`` (forward: of the conv's output;
backward: of the conv's input, whose gradient it produces). Results in the needed region are identical to the
dense computation; positions outside it are never read by anything that reaches the loss.

Everything runs on the device (``sm_need_step`` / ``sm_tile_flags`` + one ``nonzero``); one host read-back of the
list lengths per view (they are grid sizes).
"""
from __future__ import annotations

import torch

from . import ops
from .vgg import NODES, POOL_OUTPUT, PRE_POOL, depth_of, fuse_pool_fwd, layer_hw


def need_maps(M: torch.Tensor, H: int, W: int, injected, last_layer: str) -> dict:
    """``M``: level mask [H,W] (0/1 float, device). Returns {layer or 'img': [h,w] 0/1 float tensor}."""
    nodes = NODES[:depth_of(last_layer) + 1]
    sizes = {"img": (H, W)}
    for _, _, out, _, _ in nodes:
        sizes[out] = layer_hw(out, H, W)
    need = {last_layer: torch.empty(sizes[last_layer], device=M.device)}
    ops.need_step(None, 0, M if last_layer in injected else None, need[last_layer])
    for kind, src, out, _, _ in reversed(nodes):
        need[src] = torch.empty(sizes[src], device=M.device)
        ops.need_step(need[out], 2 if kind == "pool" else 1, M if src in injected else None, need[src])
    return need


def build_tile_lists(needs, last_layer: str, extra=None, resident: bool = False):
    """``needs``: one ``need_maps`` dict per level of the grouped launch (in problem order).
    ``resident``: the launches with 64 output channels take vertical QUADS of segments (``viewplan.list_jobs``); the keys
    of those lists are the ``quads`` attribute of the returned ``viewplan.TileLists``.
    Returns {(conv name, 'f' | 'b'): (int32 device tensor of (problem << 24) | tile, active fraction, list capacity)};
    for the split conv kernels the entries are 32-position SEGMENTS, ``ops.conv_list_format`` of them per tile.
    ``extra``: optional 1-D float device tensor that rides along in the one read-back; then returns (dict, list)."""
    from . import hip
    import numpy as np
    dev = next(iter(needs[0].values())).device
    import os
    coarse = os.environ.get("STYLEMESH_SEGMENT_LISTS", "1") == "0"   # A/B: flag whole tiles, as round 2 did

    def fmt(cin_pad, cout):
        bn, group = ops.conv_list_format(cin_pad, cout)
        return (bn * group, group) if (coarse and group > 0) else (bn, group)
    free = not coarse and os.environ.get("STYLEMESH_SEGMENT_STARTS", "free") == "free"   # A/B: 'grid' = aligned segments
    pairs = free and fuse_pool_fwd()

    def cover_src(g, layer):
        """(need map the cover runs on, pair_w, (h, w) of the plane the segments index, quad) of a cover key: a layer name
        (free segments), ('pair', layer) / ('quadp', layer): pairs / quads over the need map of the layer's POOLED plane,
        ('quad', layer): quads over the layer's own need map."""
        if isinstance(layer, tuple) and layer[0] == "quad":
            return needs[g][layer[1]], 0, tuple(needs[g][layer[1]].shape), 1
        if isinstance(layer, tuple):
            h, w = needs[g][layer[1]].shape
            return needs[g][POOL_OUTPUT[layer[1]]], w, (h, w), int(layer[0] == "quadp")
        return needs[g][layer], 0, tuple(needs[g][layer].shape), 0
    jobs = []            # (key, layer, bn, group): group > 0 = a SEGMENT list (bn = 32) consumed `group` entries per tile
    for kind, src, dst, cin, cout in NODES[:depth_of(last_layer) + 1]:
        if kind == "pool":
            # forward and backward are both indexed by blocks of the POOLED plane: one list, key ('pool', output layer)
            jobs.append((("pool", dst), dst, ops.plane_tile_positions(1), 0))
            continue
        f = fmt(4 if cin == 3 else cin, cout)
        quads_f = resident and free and f[1] > 0 and cout == 64 and cin % 64 == 0   # (viewplan.list_jobs: the same table)
        if pairs and f[1] > 0 and dst in PRE_POOL and POOL_OUTPUT[dst] in needs[0]:
            # the conv below a pool takes the maxima in its epilogue (EPI_POOL): its segments come in vertical PAIRS
            # that cover the need map of the POOLED plane; key (conv, 'fp'), cover key ('pair', layer)
            jobs.append(((kind, "fp"), ("quadp", dst), 32, 4) if quads_f else ((kind, "fp"), ("pair", dst)) + f)
        else:
            jobs.append(((kind, "f"), ("quad", dst), 32, 4) if quads_f else ((kind, "f"), dst) + f)
        if src != "img":
            fb = fmt(cout, cin)
            if resident and free and fb[1] > 0 and cin == 64 and cout % 64 == 0:
                jobs.append(((kind, "b"), ("quad", src), 32, 4))
            else:
                jobs.append(((kind, "b"), src) + fb)
        else:
            jobs.append((("img", "d"), "img", ops.plane_tile_positions(0), 0))   # conv1_1's data gradient
    # --- segment jobs with FREE starts: one greedy cover per (layer, level) need map, all in one launch
    cover = {}           # layer -> (starts tensor [levels, cap], offsets, counts tensor)
    cover_problems = []
    if free:
        seg_layers = sorted({layer for _, layer, _, group in jobs if group > 0}, key=str)

        def cap_of(g, layer):
            nd, pair_w, (h, w), quad = cover_src(g, layer)
            if quad:
                return (4 * ((nd.shape[0] + 1) // 2) * ((nd.shape[1] + 15) // 16 + 1) + 4 if pair_w else
                        4 * ((nd.shape[0] + 3) // 4) * ((nd.shape[1] + 31) // 32 + 1) + 4)
            if pair_w:
                return 2 * nd.shape[0] * ((nd.shape[1] + 15) // 16 + 1) + 2
            return h * hip.row_stride(w) // 32 + 2
        caps = {layer: max(cap_of(g, layer) for g in range(len(needs))) for layer in seg_layers}
        n_prob = len(seg_layers) * len(needs)
        counts_dev = torch.zeros(max(n_prob, 1), dtype=torch.int32, device=dev)
        k = 0
        for layer in seg_layers:
            starts = torch.empty(len(needs), caps[layer], dtype=torch.int32, device=dev)
            cover[layer] = (starts, k)
            for g in range(len(needs)):
                nd, pair_w, _, quad = cover_src(g, layer)
                cover_problems.append((nd, starts[g], counts_dev[k:k + 1], g, pair_w, quad))
                k += 1
        for i in range(0, len(cover_problems), 64):
            ops.cover_segments(cover_problems[i:i + 64])
    # flags of a (layer, tile size) pair are shared by the conv that produces the layer and the dgrad that produces
    # its gradient; all flags go into ONE buffer, the levels of a pair next to each other: every list is then one
    # contiguous slice of the compacted buffer, and a single nonzero + a single read-back serve all of them
    flag_jobs = [j for j in jobs if not (free and j[3] > 0)]
    seg = {}             # (layer, bn) -> [offset of level 0, ..., offset of level n-1, end]
    starts, shifts = [], []
    total = 0
    for _, layer, bn, _ in flag_jobs:
        if (layer, bn) in seg:
            continue
        offs = []
        for g, nd in enumerate(needs):
            h, w = nd[layer].shape
            offs.append(total)
            starts.append(total)
            shifts.append((g << 24) - total)          # global flag index -> (problem << 24) | tile-in-problem
            total += (h * hip.row_stride(w) + bn - 1) // bn
        seg[(layer, bn)] = offs + [total]
    flags = torch.empty(total, dtype=torch.uint8, device=dev)
    for (layer, bn), offs in seg.items():
        for g in range(len(needs)):
            ops.tile_flags(needs[g][layer], bn, flags[offs[g]:offs[g + 1]])
    active = torch.nonzero(flags).flatten()                            # ascending global flag indices
    starts_d = torch.tensor(starts, device=dev)
    which = torch.searchsorted(starts_d, active, right=True) - 1
    entries = (active + torch.tensor(shifts, device=dev)[which]).to(torch.int32)
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(flags.to(torch.int64), 0)])
    bounds = sorted({b for o in seg.values() for b in o})     # every (layer, tile size, level) boundary
    picked = csum[torch.tensor(bounds, device=dev)]
    parts = [picked.to(torch.float64)]
    if free:
        parts.append(counts_dev.to(torch.float64))
    if extra is not None:   # counts < 2^53 and the extras (mask sums) are exact in float64
        parts.append(extra.to(torch.float64))
    both = torch.cat(parts).tolist()                                   # the one read-back
    host = [int(v) for v in both[:len(bounds)]]
    n_cov = counts_dev.numel() if free else 0
    cov_counts = [int(v) for v in both[len(bounds):len(bounds) + n_cov]]
    extra_host = both[len(bounds) + n_cov:] if extra is not None else None
    cnt = dict(zip(bounds, host))
    from .viewplan import TileLists
    out = TileLists()
    out.quads = frozenset(j[0] for j in jobs if isinstance(j[1], tuple) and j[1][0] in ("quad", "quadp"))
    if coarse:   # A/B: whole tiles - every live tile contributes all of its `group` segments
        for key, layer, bn, group in jobs:
            if group > 0:
                offs = seg[(layer, bn)]
                lst = entries[cnt[offs[0]]:cnt[offs[-1]]]
                n_all = offs[-1] - offs[0]
                g_of = lst >> 24
                t_of = lst & 0xFFFFFF
                wp = torch.tensor([hip.row_stride(nd[layer].shape[1]) for nd in needs], dtype=torch.int32, device=dev)
                q = (t_of[:, None] * group + torch.arange(group, device=dev, dtype=torch.int32)[None, :]) * 32 + wp[g_of.long()][:, None]
                out[key] = (((g_of[:, None] << 24) | q).reshape(-1).to(torch.int32), lst.numel() / max(n_all, 1),
                            n_all * group)
        jobs = [j for j in jobs if j[3] <= 0]
    # Segment lists (the split conv kernels: any `group` live 32-position segments of ONE level form a tile): per level the
    # run of live entries padded to a multiple of `group` with (level << 24) | 0xFFFFFF. One gather builds all of them: the
    # index list is assembled on the host from the counts just read (behind the sources: one padding entry per level).
    idx_parts, spans = [], {}
    pos = 0
    if free:
        src_parts, src_base = [], {}
        base = 0
        for layer, (st, k0) in cover.items():
            src_base[layer] = base
            src_parts.append(st.reshape(-1))
            base += st.numel()
        n_src = base
    else:
        n_src = int(entries.numel())
    for key, layer, bn, group in jobs:
        if group <= 0 or (layer, bn, group) in spans:
            continue
        begin, live, n_all = pos, 0, 0
        for g in range(len(needs)):
            if free:
                st, k0 = cover[layer]
                n_g = cov_counts[k0 + g]
                if n_g > st.shape[1]:
                    raise RuntimeError(f"segment cover of {layer}: {n_g} entries exceed the buffer's {st.shape[1]}")
                a = src_base[layer] + g * st.shape[1]
                idx_parts.append(np.arange(a, a + n_g, dtype=np.int64))
                h, w = cover_src(g, layer)[2]
                n_all += (h * hip.row_stride(w) + 31) // 32
            else:
                offs = seg[(layer, bn)]
                a, b_ = cnt[offs[g]], cnt[offs[g + 1]]
                n_g = b_ - a
                idx_parts.append(np.arange(a, b_, dtype=np.int64))
                n_all = offs[-1] - offs[0]
            pad = (-n_g) % group
            if pad:
                idx_parts.append(np.full(pad, n_src + g, dtype=np.int64))
            pos += n_g + pad
            live += n_g
        spans[(layer, bn, group)] = (begin, pos, live, n_all)
    if spans:
        pads = torch.tensor([(g << 24) | 0xFFFFFF for g in range(len(needs))], dtype=torch.int32, device=dev)
        idx_dev = torch.from_numpy(np.concatenate(idx_parts) if idx_parts else np.zeros(0, np.int64)).to(dev, non_blocking=True)
        source = torch.cat(src_parts + [pads]) if free else torch.cat([entries, pads])
        seg_entries = source[idx_dev]
    for key, layer, bn, group in jobs:
        if group > 0:
            begin, end, live, n_all = spans[(layer, bn, group)]
            lst = seg_entries[begin:end]
            if not free:   # aligned segments from the flags: (level << 24) | segment -> (level << 24) | first position
                wp = torch.tensor([hip.row_stride(nd[layer].shape[1]) for nd in needs], dtype=torch.int32, device=dev)
                g_of, s_of = lst >> 24, lst & 0xFFFFFF
                lst = torch.where(s_of == 0xFFFFFF, lst, (g_of << 24) | (s_of * 32 + wp[g_of.long()]))
            # capacity of the persistent list buffer: every segment of every level (pair lists: every run the cover can
            # emit - a short row still takes a whole pair) + one padded tile per level
            cap = sum(cap_of(g, layer) for g in range(len(needs))) if (free and isinstance(layer, tuple)) else n_all
            out[key] = (lst, live / max(n_all, 1), cap + group * len(needs))
        else:
            offs = seg[(layer, bn)]
            n_all = offs[-1] - offs[0]
            lst = entries[cnt[offs[0]]:cnt[offs[-1]]]
            out[key] = (lst, lst.numel() / max(n_all, 1), n_all)
    return out if extra is None else (out, extra_host)
