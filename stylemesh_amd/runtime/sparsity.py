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
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import hip, ops
from .vgg import NODES, depth_of, layer_hw


def need_maps(M: torch.Tensor, H: int, W: int, injected, last_layer: str) -> dict:
    """``M``: level mask [H,W] (0/1 float, device). Returns {layer or 'img': [h,w] 0/1 float tensor}."""
    sizes = {"img": (H, W)}
    for _, _, out, _, _ in NODES[:depth_of(last_layer) + 1]:
        sizes[out] = layer_hw(out, H, W)
    M4 = M[None, None]
    need = {}
    for kind, src, out, _, _ in reversed(NODES[:depth_of(last_layer) + 1]):
        cur = need.get(out)
        if cur is None:
            cur = torch.zeros(1, 1, *sizes[out], device=M.device)
        if out in injected:
            cur = torch.maximum(cur, F.interpolate(M4, sizes[out], mode="nearest"))   # the layer mask (losses :172)
        need[out] = cur
        if kind == "pool":
            up = cur.repeat_interleave(2, 2).repeat_interleave(2, 3)
            add = torch.zeros(1, 1, *sizes[src], device=M.device)
            add[:, :, :up.shape[2], :up.shape[3]] = up
        else:
            add = F.max_pool2d(cur, 3, 1, 1)
        need[src] = add if src not in need else torch.maximum(need[src], add)
    return {k: v[0, 0] for k, v in need.items()}


def tile_flags(need_hw: torch.Tensor, bn: int) -> torch.Tensor:
    """bool[n_tiles]: does tile t (positions q in [Wp + t*bn, Wp + (t+1)*bn) of the padded plane) hold a needed
    position?"""
    h, w = need_hw.shape
    Wp = hip.row_stride(w)
    plane = torch.zeros(h, Wp, device=need_hw.device)
    plane[:, 1:w + 1] = need_hw            # rows 1..H of the padded plane, starting at q = Wp
    q = plane.reshape(-1)
    nt = (q.numel() + bn - 1) // bn
    q = F.pad(q, (0, nt * bn - q.numel())).view(nt, bn)
    return q.amax(1) > 0


def build_tile_lists(needs, last_layer: str):
    """``needs``: one ``need_maps`` dict per level of the grouped launch (in problem order).
    Returns {(conv name, 'f' | 'b'): (int32 device tensor of (problem << 24) | tile, active fraction, n_all_tiles)}."""
    out = {}
    for kind, src, dst, cin, cout in NODES[:depth_of(last_layer) + 1]:
        if kind == "pool":
            continue
        jobs = [("f", dst, ops.conv_tile_positions(4 if cin == 3 else cin, cout))]
        if src != "img":
            jobs.append(("b", src, ops.conv_tile_positions(cout, cin)))
        for direction, layer, bn in jobs:
            entries, total = [], 0
            for g, nd in enumerate(needs):
                fl = tile_flags(nd[layer], bn)
                total += fl.numel()
                idx = torch.nonzero(fl).flatten().to(torch.int32)
                entries.append(idx + (g << 24))
            lst = torch.cat(entries).contiguous()
            out[(kind, direction)] = (lst, lst.numel() / max(total, 1), total)
    return out
