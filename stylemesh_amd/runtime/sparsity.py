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
from .vgg import NODES, depth_of, layer_hw


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


def build_tile_lists(needs, last_layer: str, extra=None):
    """``needs``: one ``need_maps`` dict per level of the grouped launch (in problem order).
    Returns {(conv name, 'f' | 'b'): (int32 device tensor of (problem << 24) | tile, active fraction, n_all_tiles)}.
    ``extra``: optional 1-D float device tensor that rides along in the one read-back; then returns (dict, list)."""
    from . import hip
    dev = next(iter(needs[0].values())).device
    jobs = []            # (key, layer, bn)
    for kind, src, dst, cin, cout in NODES[:depth_of(last_layer) + 1]:
        if kind == "pool":
            # forward and backward are both indexed by blocks of the POOLED plane: one list, key ('pool', output layer)
            jobs.append((("pool", dst), dst, ops.plane_tile_positions(1)))
            continue
        jobs.append(((kind, "f"), dst, ops.conv_tile_positions(4 if cin == 3 else cin, cout)))
        if src != "img":
            jobs.append(((kind, "b"), src, ops.conv_tile_positions(cout, cin)))
        else:
            jobs.append((("img", "d"), "img", ops.plane_tile_positions(0)))   # conv1_1's data gradient
    # flags of a (layer, tile size) pair are shared by the conv that produces the layer and the dgrad that produces
    # its gradient; all flags go into ONE buffer, the levels of a pair next to each other: every list is then one
    # contiguous slice of the compacted buffer, and a single nonzero + a single read-back serve all of them
    seg = {}             # (layer, bn) -> [offset of level 0, ..., offset of level n-1, end]
    starts, shifts = [], []
    total = 0
    for _, layer, bn in jobs:
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
    bounds = sorted({o[0] for o in seg.values()} | {o[-1] for o in seg.values()})
    picked = csum[torch.tensor(bounds, device=dev)]
    if extra is None:
        host, extra_host = picked.tolist(), None                              # the one read-back
    else:   # counts < 2^53 and the extras (mask sums) are exact in float64
        both = torch.cat([picked.to(torch.float64), extra.to(torch.float64)]).tolist()
        host, extra_host = [int(v) for v in both[:len(bounds)]], both[len(bounds):]
    cnt = dict(zip(bounds, host))
    out = {}
    for key, layer, bn in jobs:
        offs = seg[(layer, bn)]
        lst = entries[cnt[offs[0]]:cnt[offs[-1]]]
        n_all = offs[-1] - offs[0]
        out[key] = (lst, lst.numel() / max(n_all, 1), n_all)
    return out if extra is None else (out, extra_host)
