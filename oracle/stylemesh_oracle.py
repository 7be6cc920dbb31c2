"""ORACLE - test infrastructure, NOT product code.

CPU restatement (torch fp32 CPU ops + explicit index arithmetic) of the StyleMesh texture-optimisation
inner loop, written from the reference's behaviour; every function cites the reference file:line it
follows (paths relative to lukasHoel/stylemesh). Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``stylemesh_amd`` + ``libstylemesh_hip.so``) never does and fails loudly without its HIP library.

Parity pin: the reference has no tests or golden vectors of its own. This oracle
is pinned against outputs of the reference's own Python, generated in the build container by
``tests/golden/make_goldens.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_vs_golden.py`` is the check). All arithmetic of the reference lives in PyTorch ATen
(pinned ``torch==1.9.1+cu111``, reference requirements.txt:98); the goldens were produced with
torch 2.10 CPU kernels, whose semantics for the modes used are unchanged.

Two kinds of functions live here:
* ``*_explicit``: index-arithmetic restatements of the ATen conventions the HIP kernels must
  reproduce (grid_sample, nearest / bilinear resize, erode, max-pool, Adam) - checked against ATen and
  the goldens in the CPU test-suite, and used by the GPU tests for per-kernel comparisons.
* the end-to-end step (``forward_with_loss`` / ``OraclePipeline``): same op sequence as the reference on
  torch CPU kernels with autograd, exposing the intermediates the GPU tests compare against.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

CLAMP_LO, CLAMP_HI = -123.6800, 151.0610  # reference model/texture/texture.py:43

VGG_ORDER = ["conv1_1", "conv1_2", "P", "conv2_1", "conv2_2", "P", "conv3_1", "conv3_2", "conv3_3", "conv3_4", "P",
             "conv4_1", "conv4_2", "conv4_3", "conv4_4", "P", "conv5_1", "conv5_2", "conv5_3", "conv5_4", "P"]


# ---------------------------------------------------------------------------------------------------
# explicit restatements of the ATen conventions (SURVEY.md section 7.2 "interpolation conventions")
# ---------------------------------------------------------------------------------------------------
def grid_sample_border_explicit(tex: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """``F.grid_sample(tex[None], grid, 'bilinear', 'border', align_corners=True)`` by index arithmetic.

    Reference call site model/texture/texture.py:49-53. ``tex`` [C,H,W], ``grid`` [1,h,w,2] with
    ``grid[...,0]`` = x (column) and ``grid[...,1]`` = y (row) in [-1,1]:
    ``ix = (x+1)/2*(W-1)`` clipped to [0,W-1]; 4 taps at floor / floor+1, taps outside the texture dropped.
    """
    C, H, W = tex.shape
    gx, gy = grid[0, ..., 0], grid[0, ..., 1]
    ix = ((gx + 1) / 2 * (W - 1)).clamp(0, W - 1)
    iy = ((gy + 1) / 2 * (H - 1)).clamp(0, H - 1)
    x0, y0 = ix.floor(), iy.floor()
    x1, y1 = x0 + 1, y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    out = torch.zeros(C, *gx.shape, dtype=tex.dtype)
    for xx, yy, ww in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
        inb = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        xi = xx.clamp(0, W - 1).long()
        yi = yy.clamp(0, H - 1).long()
        out += tex[:, yi, xi] * (ww * inb)
    return out[None]


def grid_sample_border_backward_explicit(shape_chw, grid: torch.Tensor, grad_out: torch.Tensor) -> torch.Tensor:
    """Gradient of :func:`grid_sample_border_explicit` w.r.t. the texture: 4-tap scatter-add."""
    C, H, W = shape_chw
    gx, gy = grid[0, ..., 0], grid[0, ..., 1]
    ix = ((gx + 1) / 2 * (W - 1)).clamp(0, W - 1)
    iy = ((gy + 1) / 2 * (H - 1)).clamp(0, H - 1)
    x0, y0 = ix.floor(), iy.floor()
    x1, y1 = x0 + 1, y0 + 1
    g = torch.zeros(C, H * W, dtype=grad_out.dtype)
    go = grad_out[0].reshape(C, -1)
    for xx, yy, ww in ((x0, y0, (x1 - ix) * (y1 - iy)), (x1, y0, (ix - x0) * (y1 - iy)),
                       (x0, y1, (x1 - ix) * (iy - y0)), (x1, y1, (ix - x0) * (iy - y0))):
        inb = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        idx = (yy.clamp(0, H - 1).long() * W + xx.clamp(0, W - 1).long()).reshape(-1)
        g.index_add_(1, idx, go * (ww * inb).reshape(-1))
    return g.view(C, H, W)


def nearest_index(out_size: int, in_size: int) -> torch.Tensor:
    """Legacy ``mode='nearest'`` source index: ``min(floor(dst * (in/out) in fp32), in-1)``.
    (ATen upsample_nearest; call sites model/model.py:219,238, content_and_style_losses.py:172-174)."""
    scale = torch.tensor(in_size / out_size, dtype=torch.float32)
    return (torch.arange(out_size, dtype=torch.float32) * scale).floor().long().clamp(max=in_size - 1)


def resize_nearest_explicit(x: torch.Tensor, out_hw) -> torch.Tensor:
    """``F.interpolate(x, out_hw, mode='nearest')`` for [B,C,H,W]."""
    iy = nearest_index(out_hw[0], x.shape[2])
    ix = nearest_index(out_hw[1], x.shape[3])
    return x[:, :, iy][:, :, :, ix]


def bilinear_coords(out_size: int, in_size: int):
    """``align_corners=False`` source coordinates: ``src = max((dst+0.5)*in/out - 0.5, 0)``; returns
    (i0, i1, lambda1) with ``i1 = min(i0+1, in-1)`` (ATen upsample_bilinear2d)."""
    scale = in_size / out_size
    src = ((torch.arange(out_size, dtype=torch.float32) + 0.5) * scale - 0.5).clamp(min=0)
    i0 = src.floor().long().clamp(max=in_size - 1)
    i1 = (i0 + 1).clamp(max=in_size - 1)
    return i0, i1, src - i0.float()


def resize_bilinear_explicit(x: torch.Tensor, out_hw) -> torch.Tensor:
    """``F.interpolate(x, out_hw, mode='bilinear')`` (align_corners=False) for [B,C,H,W].
    Call sites model/model.py:199, content_and_style_losses.py:94,120,161,176."""
    y0, y1, ly = bilinear_coords(out_hw[0], x.shape[2])
    x0, x1, lx = bilinear_coords(out_hw[1], x.shape[3])
    ly = ly.view(1, 1, -1, 1)
    lx = lx.view(1, 1, 1, -1)
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly) + bot * ly


def erode_explicit(x: torch.Tensor) -> torch.Tensor:
    """Reference ``erode`` (model/model.py:204-208): keep x where the zero-padded 3x3 box mean equals 1."""
    p = F.pad(x, (1, 1, 1, 1))
    s = torch.zeros_like(x)
    H, W = x.shape[2:]
    for dy in range(3):
        for dx in range(3):
            s = s + p[:, :, dy:dy + H, dx:dx + W]
    return x * ((s / 9).clamp(0, 1) == 1)


def maxpool2x2_explicit(x: torch.Tensor) -> torch.Tensor:
    """``MaxPool2d(2, 2)`` with floor output size (content_and_style_losses.py:27-32)."""
    H, W = x.shape[2] // 2 * 2, x.shape[3] // 2 * 2
    x = x[:, :, :H, :W]
    return torch.maximum(torch.maximum(x[:, :, 0::2, 0::2], x[:, :, 0::2, 1::2]),
                         torch.maximum(x[:, :, 1::2, 0::2], x[:, :, 1::2, 1::2]))


def adam_step_explicit(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
    """One ``torch.optim.Adam`` update (weight_decay 0, amsgrad off) in fp32 tensors + double scalars, as
    torch 2.x's single-tensor path computes it (reference model/model.py:395). Returns new (p, m, v)."""
    m = m + (g - m) * (1 - beta1)                       # exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + (g * g) * (1 - beta2)               # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# ---------------------------------------------------------------------------------------------------
# texture (reference model/texture/texture.py)
# ---------------------------------------------------------------------------------------------------
def sample_texture(layers, grid):
    """``HierarchicalNeuralTexture.forward`` (texture.py:96-100) / ``NeuralTexture.forward`` (:46-54):
    sum over layers of bilinear border grid_sample on the same normalised grid. Layers are expected to be
    already clamped (``normalize()``, :41-44)."""
    return sum(F.grid_sample(l[None], grid, mode="bilinear", padding_mode="border", align_corners=True)
               for l in layers)


def tex_regularizer(layers, weights):
    """``HierarchicalNeuralTexture.regularizer`` (texture.py:102-108): sum_i w_i * mean(layer_i^2)."""
    return sum(torch.mean(l ** 2.0) * w for l, w in zip(layers, weights))


def identity_grid(H, W):
    """Sampling grid of ``HierarchicalNeuralTexture.get_image`` (texture.py:110-121)."""
    w_range = torch.arange(0, W, dtype=torch.float) / (W - 1.0) * 2.0 - 1.0
    h_range = torch.arange(0, H, dtype=torch.float) / (H - 1.0) * 2.0 - 1.0
    v, u = torch.meshgrid(h_range, w_range, indexing="ij")
    return torch.stack([u, v], 2)[None]


# ---------------------------------------------------------------------------------------------------
# VGG / Gram / style pyramid (reference model/losses/content_and_style_losses.py)
# ---------------------------------------------------------------------------------------------------
def vgg_forward(state: dict, x: torch.Tensor, out_keys, explicit_pool=False, keep_all: dict | None = None) -> dict:
    """``VGG.forward`` (content_and_style_losses.py:47-70): 3x3 pad-1 conv + ReLU, 2x2 max-pool.
    ``keep_all`` (tests only) receives every intermediate activation with ``retain_grad()`` set."""
    out = {}
    block, idx = 1, 1
    for name in VGG_ORDER:
        if name == "P":
            x = maxpool2x2_explicit(x) if explicit_pool else F.max_pool2d(x, 2, 2)
            out[f"p{block}"] = x
            block, idx = block + 1, 1
        else:
            x = F.relu(F.conv2d(x, state[name + ".weight"], state[name + ".bias"], padding=1))
            out[f"r{block}{idx}"] = x
            idx += 1
        if keep_all is not None and x.requires_grad:
            x.retain_grad()
        if all(k in out for k in out_keys):
            break
    if keep_all is not None:
        keep_all.update(out)
    return {k: out[k] for k in out_keys}


def gram_matrix(feat: torch.Tensor) -> torch.Tensor:
    """``GramMatrix`` (:74-80): F F^T / (h*w) for [b,c,h,w]."""
    b, c, h, w = feat.shape
    Fl = feat.reshape(b, c, h * w)
    return torch.bmm(Fl, Fl.transpose(1, 2)) / (h * w)


def image_pyramid_sizes(h: int, w: int, levels, minimum_size=256):
    """Sizes produced by ``image_pyramid(..., reverse=True)`` (:83-133): returns a list of (h, w)."""
    sizes, min_entry, min_index = [], None, len(levels)
    for i, level in enumerate(levels):
        if level == 0:
            sizes.append((h, w))
            continue
        hd, wd = int(h / 2 ** level), int(w / 2 ** level)
        if hd < minimum_size or wd < minimum_size:
            if min_entry is None:
                min_entry = (minimum_size, int(w * minimum_size / h)) if w > h else (int(h * minimum_size / w), minimum_size)
                min_index = i
            sizes.append(min_entry)
        else:
            sizes.append((hd, wd))
    rev = sizes[:min_index + 1][::-1]
    while len(rev) < len(sizes):
        rev.append((h, w))
    return rev


def image_pyramid(img: torch.Tensor, levels, minimum_size=256):
    """``image_pyramid(img, levels, reverse=True)`` (:83-133): smallest entry first, padded with the
    original image; every entry is a bilinear (align_corners=False) resize of the ORIGINAL image."""
    h, w = img.shape[2:]
    return [img if s == (h, w) else F.interpolate(img, s, mode="bilinear")
            for s in image_pyramid_sizes(h, w, levels, minimum_size)]


def style_targets(state, style_image, style_layers, num_levels=5):
    """``set_style_image`` (:273-286): ``targets[layer_index][level]`` = Gram of VGG(style pyramid[level])."""
    pyr = image_pyramid(style_image, list(range(num_levels)))
    by_size = {}   # the padded entries repeat the original image (:126-131): same input, same features - encoded once
    enc = []
    for p in pyr:
        key = tuple(p.shape[2:])
        if key not in by_size:
            by_size[key] = vgg_forward(state, p, style_layers)
        enc.append(by_size[key])
    return [{lvl: gram_matrix(enc[lvl][layer]).detach() for lvl in range(num_levels)} for layer in style_layers]


def masked_features(features, mask):
    """``masked_features`` (:136-143): gather valid pixels -> [B,C,N_valid,1]; all-zero [B,C,h*w,1] if none."""
    cropped = features[:, :, mask.squeeze() > 0].unsqueeze(3)
    if cropped.shape[2] == 0:
        return torch.zeros_like(features).reshape(features.shape[0], features.shape[1], -1).unsqueeze(3)
    return cropped


# ---------------------------------------------------------------------------------------------------
# configuration + the step (reference model/model.py)
# ---------------------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    hierarchical: bool = True
    style_layers: list = field(default_factory=lambda: ['r11', 'r21', 'r31', 'r41', 'r51'])
    content_layers: list = field(default_factory=lambda: ['r42'])
    style_weights: list = field(default_factory=lambda: [1e3 / n ** 2 for n in [64, 128, 256, 512, 512]])
    content_weights: list = field(default_factory=lambda: [1])
    angle_threshold: float = 60
    style_pyramid_mode: str = "single"
    gram_mode: str = "current"
    use_angle_weight: bool = True
    use_depth_scaling: bool = True
    loss_weights: dict = field(default_factory=lambda: {"content": 0.0, "style": 0.0, "tex_reg": 0.0})
    tex_reg_weights: list | None = None          # default [8,4,2,0] for 4 layers (model/model.py:86-88)
    learning_rate: float = 1.0
    decay_gamma: float = 0.1
    decay_step_size: int = 30

    def reg_weights(self, n_layers):
        if self.tex_reg_weights:
            return list(self.tex_reg_weights)
        w = [pow(2, n_layers - i - 1) for i in range(n_layers)]
        w[-1] = 0
        return w


def level_masks_and_weights(batch, pred_shapes, cfg: OracleConfig):
    """Per-level validity masks and depth-interpolation weights (model/model.py:188-254).

    Returns ``(masks, weights)``: lists over UV levels of [1,1,H_i,W_i] float tensors
    (``weights[i]`` is None without depth scaling)."""
    _, _, _, _, _, rounded, other, interp_w, _, _, mask, _, _ = batch
    dt = interp_w.dtype   # fp32; fp64 only for the "is a mismatch a benign ReLU/pool flip?" diagnostics
    mask = mask.unsqueeze(1).to(dt)

    def erode(x):
        k = torch.ones(1, 1, 3, 3, dtype=dt)
        em = torch.clamp(F.conv2d(x, k, padding=(1, 1)) / 9, 0, 1)
        return x * (em == 1)

    if cfg.use_depth_scaling:
        masks, weights = [], []
        for i, hw in enumerate(pred_shapes):
            m = ((rounded == i) + (other == i)).to(dt) * mask
            masks.append((F.interpolate(erode(m), hw, mode="nearest") > 0).to(dt))
            m1 = erode((rounded == i) * mask) * interp_w
            m2 = erode((other == i) * mask) * (1 - interp_w)
            weights.append(F.interpolate(m1 + m2, hw, mode="nearest"))
    else:
        masks = [torch.zeros(1, 1, *hw, dtype=dt) for hw in pred_shapes]
        masks[-1] = (F.interpolate(mask, pred_shapes[-1], mode="nearest") > 0).to(dt)
        weights = [None] * len(pred_shapes)
    return masks, weights


class GramCache:
    """The 10-deep detached Gram history of ``gram_mode='average'`` (content_and_style_losses.py:319-323)."""

    def __init__(self, style_layers):
        self.cache = {k: [] for k in style_layers}

    def average(self, layer, y_hat):
        c = [g.detach() for g in self.cache[layer][:9]]
        c.insert(0, y_hat)
        self.cache[layer] = c
        return torch.mean(torch.stack(c), dim=0)


def content_and_style_loss(state, preds, target_content, pyramid_masks, angle_degrees, targets, cfg: OracleConfig,
                           gram_cache: GramCache | None = None, content_enc=None, record=None):
    """``ContentAndStyleLoss.forward`` + ``calculate_pyramid`` (content_and_style_losses.py:146-217,288-350).

    ``preds``: active prediction levels; ``pyramid_masks``: their masks. Returns ``(style, content)``.
    ``record`` (dict) receives intermediates keyed by level index for the GPU tests."""
    layers = cfg.style_layers + cfg.content_layers
    all_acts = [dict() for _ in preds] if record is not None else [None] * len(preds)
    enc = [vgg_forward(state, p, layers, keep_all=ka) for p, ka in zip(preds, all_acts)]
    if record is not None:
        record["all_acts"] = all_acts   # every activation, with .grad = dL/d(activation) after backward
    if content_enc is None:
        content_enc = vgg_forward(state, target_content, layers)
    mse = torch.nn.MSELoss()
    n = len(preds)
    factors = [dict() for _ in range(n)]
    info = []
    for pi in range(n):
        mask = pyramid_masks[pi]
        passed = F.interpolate(angle_degrees, mask.shape[2:], mode="bilinear") < cfg.angle_threshold
        lv = {}
        for k, o in enc[pi].items():
            with torch.no_grad():
                m_i = F.interpolate(mask, o.shape[2:], mode="nearest")
                m_pass = F.interpolate(mask * passed, o.shape[2:], mode="nearest")
                m_fail = F.interpolate(mask * (~passed), o.shape[2:], mode="nearest")
                c_t = masked_features(F.interpolate(content_enc[k], o.shape[2:], mode="bilinear"), m_i)
                factors[pi][k] = torch.mean(m_i)
            lv[k] = dict(m=m_i, m_pass=m_pass, m_fail=m_fail, c=c_t, p=masked_features(o, m_i),
                         p_pass=masked_features(o, m_pass), p_fail=masked_features(o, m_fail), feat=o)
        info.append(lv)
    for k in layers:
        s = sum(factors[i][k] for i in range(n))
        for i in range(n):
            factors[i][k] = factors[i][k] / s
    style = torch.zeros(1)
    content = torch.zeros(1)
    for pi in range(n):
        for li, layer in enumerate(cfg.style_layers):
            if cfg.style_pyramid_mode == "single":
                y = targets[li][0]
                y_hat = gram_matrix(info[pi][layer]["p"])
            elif cfg.style_pyramid_mode == "multi":
                y = targets[li][2]
                y_hat = gram_matrix(info[pi][layer]["p_pass"])
            else:
                raise ValueError(f"Unsupported style_pyramid_mode: {cfg.style_pyramid_mode}")
            if cfg.gram_mode == "average":
                y_hat = gram_cache.average(layer, y_hat)
            f = factors[pi][layer]
            l = cfg.style_weights[li] * f * mse(y, y_hat)
            if cfg.style_pyramid_mode == "multi":
                y_hat_fail = gram_matrix(info[pi][layer]["p_fail"])
                if torch.sum(info[pi][layer]["m_fail"]) > 0:
                    l = l + cfg.style_weights[li] * f * mse(y, y_hat_fail)
                if li > 2:
                    l = l + cfg.style_weights[li] * f * mse(targets[li][0], y_hat)
            style = style + l
        for li, layer in enumerate(cfg.content_layers):
            f = factors[pi][layer]
            content = content + cfg.content_weights[li] * f * mse(info[pi][layer]["c"], info[pi][layer]["p"])
    if record is not None:
        record["enc"] = enc
        record["factors"] = factors
        record["info"] = info
        record["content_enc"] = content_enc
    return style, content


def forward_with_loss(state, layers, batch, cfg: OracleConfig, targets, gram_cache=None, record=None):
    """``TextureOptimizationStyleTransferPipeline.forward_with_loss`` (model/model.py:178-327), B = 1.

    ``layers``: list of texture layer tensors [3,H_i,W_i] requiring grad (one entry when not hierarchical);
    they are clamped in place first, as every reference forward does (texture.py:47). Returns a dict of
    weighted losses ``content, style, tex_reg, total``."""
    rgb, _, _, _, _, _, _, _, _, uv_map, mask, angle_guidance, angle_degrees = batch
    with torch.no_grad():
        for l in layers:
            l.clamp_(CLAMP_LO, CLAMP_HI)
    preds = [sample_texture(layers, v) for v in uv_map]
    shapes = [tuple(p.shape[2:]) for p in preds]
    masks, weights = level_masks_and_weights(batch, shapes, cfg)
    if record is not None:
        record["preds"] = [p.detach().clone() for p in preds]
        record["masks"], record["weights"] = masks, weights
        record["pred_grads_raw"] = [None] * len(preds)
        for i, p in enumerate(preds):
            if p.requires_grad:
                p.register_hook(lambda g, i=i: record["pred_grads_raw"].__setitem__(i, g.clone()))
    for i, p in enumerate(preds):
        if not p.requires_grad:
            continue
        if cfg.use_angle_weight:   # model/model.py:195-202
            p.register_hook(lambda g: g * F.interpolate(angle_guidance, g.shape[2:], mode="bilinear"))
        if cfg.use_depth_scaling:  # model/model.py:245-251 (weight looked up by matching height)
            p.register_hook(lambda g, i=i: g * next(w for w in weights if w.shape[2] == g.shape[2]))
    active = [i for i, m in enumerate(masks) if torch.sum(m) > 0]   # model/model.py:256-257
    style, content = content_and_style_loss(state, [preds[i] for i in active], rgb, [masks[i] for i in active],
                                            angle_degrees, targets, cfg, gram_cache, record=record)
    losses = {"content": cfg.loss_weights["content"] * content, "style": cfg.loss_weights["style"] * style}
    if cfg.loss_weights.get("tex_reg", 0) > 0 and cfg.hierarchical:
        losses["tex_reg"] = cfg.loss_weights["tex_reg"] * tex_regularizer(layers, cfg.reg_weights(len(layers)))
    else:
        losses["tex_reg"] = torch.zeros_like(losses["content"])
    losses["total"] = sum(losses.values())
    if record is not None:
        record["active"] = active
    return losses


class OraclePipeline:
    """The reference training loop for one scene on CPU: clamp -> sample -> VGG -> losses -> backward ->
    Adam (+ StepLR per epoch), Lightning's automatic-optimisation order."""

    def __init__(self, vgg_state, style_image, cfg: OracleConfig, tex_wh, n_layers=4, init_layers=None, targets=None):
        self.state, self.cfg = vgg_state, cfg
        W, H = tex_wh
        n = n_layers if cfg.hierarchical else 1
        self.layers = [torch.zeros(3, H // 2 ** i, W // 2 ** i) for i in range(n)]
        if init_layers is not None:
            self.layers = [t.clone().float() for t in init_layers[:n]]
        for l in self.layers:
            l.requires_grad_(True)
        # ``targets``: a previous ``style_targets`` result for the same style image / layers (tests share it)
        self.targets = targets if targets is not None else style_targets(
            vgg_state, style_image[None] if style_image.dim() == 3 else style_image, cfg.style_layers)
        self.gram_cache = GramCache(cfg.style_layers)
        self.m = [torch.zeros_like(l) for l in self.layers]
        self.v = [torch.zeros_like(l) for l in self.layers]
        self.step_count = 0
        self.epoch = 0

    @property
    def lr(self):
        return self.cfg.learning_rate * self.cfg.decay_gamma ** (self.epoch // self.cfg.decay_step_size)

    def grads(self, batch, record=None):
        for l in self.layers:
            l.grad = None
        losses = forward_with_loss(self.state, self.layers, batch, self.cfg, self.targets, self.gram_cache, record)
        losses["total"].backward()
        return losses, [l.grad.clone() for l in self.layers]

    def apply_adam(self, grads):
        self.step_count += 1
        with torch.no_grad():
            for i, g in enumerate(grads):
                p, m, v = adam_step_explicit(self.layers[i].detach(), g, self.m[i], self.v[i], self.step_count, self.lr)
                self.layers[i].copy_(p)
                self.m[i], self.v[i] = m, v

    def training_step(self, batch, record=None):
        losses, grads = self.grads(batch, record)
        self.apply_adam(grads)
        return {k: float(v) for k, v in losses.items()}

    def end_epoch(self):
        self.epoch += 1


# ---------------------------------------------------------------------------------------------------
# Reprojection warp of the evaluation metric (reference data/utils.py:36-194,
# scripts/eval/eval_image_folders.py:286-330). Explicit index arithmetic, one view pair (B = 1).
# ---------------------------------------------------------------------------------------------------
def unproject_explicit(cam2world: torch.Tensor, K: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
    """World-space points [H,W,4] of a depth map [H,W] (reference data/utils.py:36-70; the reference multiplies row
    vectors with the matrix it is given, so its caller passes cam2world transposed - here: the plain matrix)."""
    H, W = depth.shape
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    x = (xx - K[0, 2]) / K[0, 0] * depth
    y = (yy - K[1, 2]) / K[1, 1] * depth
    pts = torch.stack([x, y, depth, torch.ones_like(depth)], -1)
    return (pts.reshape(-1, 4) @ cam2world.T).reshape(H, W, 4)


def _unnormalize_align_corners(g, size):
    return (g + 1.0) / 2.0 * (size - 1)


def grid_sample_nearest_border_explicit(img: torch.Tensor, gx: torch.Tensor, gy: torch.Tensor) -> torch.Tensor:
    """``F.grid_sample(img[None], grid, mode='nearest', padding_mode='border', align_corners=True)`` for img [C,H,W]
    and normalised coordinates gx, gy [h,w]: clip the source coordinate to the border, then round half to even."""
    C, H, W = img.shape
    ix = _unnormalize_align_corners(gx, W).clamp(0, W - 1)
    iy = _unnormalize_align_corners(gy, H).clamp(0, H - 1)
    xi = torch.round(ix).long().clamp(0, W - 1)      # torch.round = nearbyint (half to even), as ATen
    yi = torch.round(iy).long().clamp(0, H - 1)
    return img[:, yi, xi]


def reproject_explicit(c2w_src, c2w_tar, K, depth_src, depth_tar, color_tar, mask_tar, depth_tol=0.1):
    """``reproject`` (reference data/utils.py:73-194) for one pair: warp the target view's image into the source view
    through the source depth. Returns (color [3,H,W], mask bool [H,W]); the colour is zero outside the mask.
    Every source pixel is un-projected with its depth, moved into the target camera, projected, and rejected when
    (0) its depth is 0, (1) it lands outside [0, W-1) x [0, H-1), (2) none of the 4 surrounding target depths is
    within ``depth_tol`` of its own target-space z, or the bilinearly sampled target validity mask is <= 0.99.
    Note the reference's pixel -> grid mapping ``2 x / W - 1`` (not W - 1) under align_corners=True: the sampled
    location is x (W - 1) / W."""
    H, W = depth_src.shape
    f32 = torch.float32
    src2tar = torch.linalg.inv(c2w_tar.to(f32)) @ c2w_src.to(f32)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=f32), torch.arange(W, dtype=f32), indexing="ij")
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x = (xx - cx) / fx * depth_src
    y = (yy - cy) / fy * depth_src
    pts = torch.stack([x, y, depth_src, torch.ones_like(depth_src)], -1).reshape(-1, 4)
    pts = (pts @ src2tar.T).reshape(H, W, 4)
    z = pts[..., 2]
    px = pts[..., 0] / (1e-8 + z) * fx + cx
    py = pts[..., 1] / (1e-8 + z) * fy + cy
    bad = (depth_src == 0) | (px < 0) | (py < 0) | (px >= W - 1) | (py >= H - 1)
    lx, ly = torch.floor(px), torch.floor(py)
    grid = lambda a, b: (2.0 * a / W - 1.0, 2.0 * b / H - 1.0)
    dz = []
    for a, b in ((lx, ly), (lx, ly + 1), (lx + 1, ly), (lx + 1, ly + 1)):
        gx, gy = grid(a, b)
        dz.append((z - grid_sample_nearest_border_explicit(depth_tar[None], gx, gy)[0]).abs())
    bad |= torch.minimum(torch.minimum(dz[0], dz[1]), torch.minimum(dz[2], dz[3])) > depth_tol
    gx, gy = grid(px, py)
    g = torch.stack([gx, gy], -1)[None]
    color = grid_sample_border_explicit(color_tar, g)[0]
    msk = grid_sample_border_explicit(mask_tar.to(f32)[None], g)[0, 0]
    mask = (msk > 0.99) & ~bad
    return color * mask, mask
