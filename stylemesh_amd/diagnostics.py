"""Dynamic range of the operands the fp16x2 kernels are fed.

The fp16x2 split (csrc/conv_split_kernel.h) scales every operand TENSOR by one power of two taken from the tensor's recorded
max |x|: an element more than 2^18 below that maximum loses low bits of its second part - an absolute error
<= 2^-40 max|x|, which an fp32 chain would not make. Harmless while the small elements are summed with large ones; the
question is how the operand tensors of a LATE step look - after thousands of Adam steps and two learning-rate decays
(scripts/train/optimize_texture_scannet_with_angle_and_depth.sh:11-15), when a few outlier pixels may own the maximum.

``operand_census(eng)`` looks at the step the engine has just run (``forward_backward`` of the current view, dense tiles so
that every stored position is this step's): for every tensor a split kernel reads - the activations ``a:<layer>`` (conv
forward and Gram operands), the gradients ``g:<layer>`` (data-gradient operands; below a fused pool backward the POOLED
gradient is the operand) and the style-loss derivative matrices ``D:<level>:<layer>`` - the histogram of
log2(bound / |x|) over the non-zero elements of all UV levels, under the bound the kernels used (one per layer over all
levels; ``AmaxBook``), and the share of elements beyond 2^12 ... 2^24.
"""
import torch

from .runtime import ops
from .runtime.vgg import OUT_NAMES

THRESHOLDS = (12, 16, 18, 20, 22, 24)


def _entry(parts, bound):
    """parts: list of tensors (any shape); bound: the operand bound the kernels derived their scale from"""
    vals = [p.reshape(-1).abs() for p in parts]
    vals = [v[v != 0] for v in vals]
    n = sum(int(v.numel()) for v in vals)
    if n == 0 or not bound > 0:
        return None
    hist = torch.zeros(48, dtype=torch.int64, device=vals[0].device)
    true_max = 0.0
    for v in vals:
        if v.numel() == 0:
            continue
        true_max = max(true_max, float(v.max()))
        r = torch.log2(bound / v).clamp_(0, 47.99).to(torch.int64)
        hist += torch.bincount(r, minlength=48)
    hist = hist.cpu()
    cum = torch.flip(torch.cumsum(torch.flip(hist, [0]), 0), [0])       # cum[k] = elements with log2(bound / |x|) >= k
    return {"nonzero_elements": n, "bound": bound, "true_max": true_max,
            "median_log2_bound_over_x": int(torch.searchsorted(torch.cumsum(hist, 0), torch.tensor((n + 1) // 2))),
            "share_beyond_2^k": {str(k): float(cum[k]) / n for k in THRESHOLDS},
            "histogram_log2_bound_over_x": hist.tolist()}


def operand_census(eng):
    """-> {tensor name: entry}; call right after ``eng.forward_backward()`` with ``eng.sparse_tiles = False``."""
    assert ops.CONV_MODE == "split2", "the census is about the fp16x2 operands"
    torch.cuda.synchronize()
    active = [lv for lv in eng.view if lv.active]
    bufs = [eng._level_bufs(lv.H, lv.W) for lv in active]
    out = {}
    for name in OUT_NAMES:
        if name == "img" or not all(name in b.act for b in bufs):
            continue
        for kind, planes in (("a", [b.act[name] for b in bufs]), ("g", [b.grad[name] for b in bufs if name in b.grad])):
            key = f"{kind}:{name}"
            if key not in eng.amax.idx or not planes:
                continue
            bound = float(eng.amax[key].max())
            e = _entry([p.to_dense() for p in planes], bound)
            if e is not None:
                out[key] = e
    live_levels = {lv.index for lv in active}
    for key, (S0, S1, D0, D1) in eng._gram.items():
        C, level, layer = key
        if level not in live_levels:
            continue                       # (scratch of a level an earlier view had)
        li = eng.cfg.style_layers.index(layer)
        k = (level * len(eng.cfg.style_layers) + li) * ops.AMAX_FLOATS
        bound = float(eng._amax_d[k:k + ops.AMAX_FLOATS].max())
        e = _entry([d for d in (D0, D1) if d is not None], bound)
        if e is not None:
            out[f"D:{level}:{layer}"] = e
    return out


def summarize(census):
    """worst share beyond 2^18 over the tensors, and which tensor shows it"""
    worst = max(census.items(), key=lambda kv: kv[1]["share_beyond_2^k"]["18"])
    return {"tensors": len(census), "worst_share_beyond_2^18": worst[1]["share_beyond_2^k"]["18"], "worst_tensor": worst[0],
            "bound_over_true_max_max": max(e["bound"] / e["true_max"] for e in census.values() if e["true_max"] > 0)}
