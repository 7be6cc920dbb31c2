"""``VGG`` / ``GramMatrix`` / ``image_pyramid`` / ``masked_features`` / ``ContentAndStyleLoss`` with the reference's
surface (model/losses/content_and_style_losses.py) over the HIP kernels.

Same names, constructor arguments, defaults, ``ValueError``s and return structure. What differs by design:
* every heavy operator is a kernel of ``libstylemesh_hip.so`` (conv/pool/Gram/MSE, fwd and hand-written bwd);
* ``masked_features`` gathers are never materialised on the hot path (mask-multiply inside the Gram / MSE kernels
  gives the same sums); the function itself is kept for API compatibility;
* the ``pyramid`` dict returned by ``ContentAndStyleLoss.forward`` carries masks, factors and ``size`` but not
  the gathered feature lists (``'p'``, ``'c'`` ...), which no caller in the reference reads.
Batch size 1 only - as in the reference, whose ``masked_features`` indexing (:137) does not work for B > 1.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ...runtime import hip, ops
from ...runtime.engine import EngineConfig, StepEngine
from ...runtime.fmap import FMap
from ...runtime.pyramid import image_pyramid_sizes
from ...runtime.vgg import NODES, PRE_POOL, LevelBuffers, VGGNet, depth_of


class _VGGFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, keys):
        if x.shape[0] != 1:
            raise ValueError("batch size 1 only")
        net = module._net()
        H, W = x.shape[2:]
        last = max(keys, key=depth_of)
        b = LevelBuffers(H, W, last, True, x.device)
        ops.image_to_fmap(x[0].detach().float().contiguous(), b.act["img"])
        net.forward(b)
        ctx.b, ctx.net, ctx.keys, ctx.last = b, net, list(keys), last
        return tuple(ops.fmap_to_image(b.act[k])[None] for k in keys)

    @staticmethod
    def backward(ctx, *grads):
        b, keys = ctx.b, ctx.keys
        injected = set()
        for k, g in zip(keys, grads):
            if g is None:
                continue
            if not k.startswith("r") or (k in PRE_POOL and k != ctx.last):
                raise ValueError(f"gradient through VGG output {k} is not supported")
            b.grad[k].from_dense(g[0])
            injected.add(k)
        if ctx.last not in injected:
            b.grad[ctx.last].zero_()
        gl = b.grad[ctx.last].planes   # gradient w.r.t. the pre-ReLU output of the deepest layer
        gl.mul_((b.act[ctx.last].planes > 0).to(gl.dtype))
        ctx.net.backward(b, injected - {ctx.last}, ctx.last)
        return ops.fmap_to_image(b.grad["img"], 3)[None], None, None


class VGG(nn.Module):
    """VGG-19 conv stack; ``forward(x, out_keys)`` returns the named activations (:47-70)."""

    def __init__(self, pool='max', model_path=None, freeze=True):
        super().__init__()
        if pool != 'max':
            raise ValueError("only pool='max' is implemented in the HIP path (no reference script uses 'avg')")
        for kind, _, _, cin, cout in NODES:
            if kind != "pool":
                setattr(self, kind, nn.Conv2d(cin, cout, kernel_size=3, padding=1))
        if model_path:
            self.load_state_dict(torch.load(model_path))
        if freeze:
            for param in self.parameters():
                param.requires_grad = False
        self._packed = None

    def _net(self) -> VGGNet:
        dev = self.conv1_1.weight.device
        if dev.type != "cuda":
            raise RuntimeError("the HIP VGG needs its weights on the GPU (call .cuda() / .to('cuda'))")
        if self._packed is None or self._packed.device != dev:
            self._packed = VGGNet(self.state_dict(), dev)
        return self._packed

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def forward(self, x, out_keys):
        outs = _VGGFn.apply(x, self, tuple(out_keys))
        return dict(zip(out_keys, outs))


class _GramFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat):
        b, c, h, w = feat.shape
        if b != 1 or c % 64 != 0:
            raise ValueError("GramMatrix: batch 1 and a multiple of 64 channels only")
        f = FMap(c, h, w, feat.device).from_dense(feat[0].detach())
        ones = FMap(1, h, w, feat.device).from_dense(torch.ones(1, h, w))
        S = torch.zeros(ops.gram_workspace_slabs(c, h, w), c, c, device=feat.device)
        # fp16x2 mode: the operand bound the producing conv would have recorded (class mirror, not a hot path)
        ctx.af = None
        if ops.GRAM_MODE == "split2":
            ctx.af = ops.new_amax(feat.device)
            ctx.af[0] = feat.detach().abs().max()
        n = ops.gram_masked(f, ones, None, S, None, amax_feat=ctx.af)
        ctx.f, ctx.ones = f, ones
        from ...runtime.engine import _mirror_tiles
        return (_mirror_tiles(S[:n].sum(0)) / float(h * w))[None]

    @staticmethod
    def backward(ctx, gG):
        f = ctx.f
        D = ((gG[0] + gG[0].T) / float(f.H * f.W)).contiguous()   # dF = (dG + dG^T) F / (h w)
        df = FMap(f.C, f.H, f.W, f.buf.device)
        ad = None
        if ops.GRAM_MODE == "split2":
            ad = ops.new_amax(D.device)
            ad[0] = D.abs().max()
        ops.gram_backward(f, ctx.ones, None, D, None, df, relu_gate=False, amax_feat=ctx.af, amax_d=ad)
        return df.to_dense()[None]


class GramMatrix(nn.Module):
    def forward(self, input):
        return _GramFn.apply(input)


def image_pyramid(img, levels, reverse=False, minimum_size=256):
    """Reference :83-133. Every entry is a bilinear (align_corners=False) resize of ``img`` [1,C,h,w]."""
    h, w = img.shape[2:]
    sizes_rev = image_pyramid_sizes(h, w, levels, minimum_size)
    if reverse:
        sizes = sizes_rev
    else:   # (orig, halves ..., min_entry, ..., min_entry)
        sizes = []
        min_entry = sizes_rev[0]
        for level in levels:
            if level == 0:
                sizes.append((h, w))
            else:
                hd, wd = int(h / 2 ** level), int(w / 2 ** level)
                sizes.append(min_entry if (hd < minimum_size or wd < minimum_size) else (hd, wd))
    out = []
    src = img[0].float().contiguous()
    for s in sizes:
        if s == (h, w):
            out.append(img)
        else:
            dst = FMap(src.shape[0], s[0], s[1], img.device)
            ops.image_to_fmap(src, dst)
            out.append(ops.fmap_to_image(dst)[None])
    return out


def masked_features(features, mask):
    """Reference :136-143 (API compatibility; the loss kernels multiply by the mask instead)."""
    cropped = features[:, :, mask.squeeze() > 0].unsqueeze(3)
    if cropped.shape[2] == 0:
        return torch.zeros_like(features).reshape(features.shape[0], features.shape[1], -1).unsqueeze(3)
    return cropped


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, *preds):
        losses, keep = owner._engine.images_forward(preds, 1.0, 1.0)
        ctx.owner, ctx.keep = owner, keep
        return losses[1:2].clone(), losses[0:1].clone()   # (style, content)

    @staticmethod
    def backward(ctx, g_style, g_content):
        grads = ctx.owner._engine.images_backward(ctx.keep, float(g_style), float(g_content))
        return (None, *grads)


class ContentAndStyleLoss(nn.Module):
    style_layers = ['r11', 'r21', 'r31', 'r41', 'r51']
    content_layers = ['r42']
    style_weights = [1e3 / n ** 2 for n in [64, 128, 256, 512, 512]]
    content_weights = [1 for i in range(len(content_layers))]
    style_pyramid_modes = ['single', 'multi']
    gram_modes = ['current', 'average']

    def __init__(self, model_path, style_layers=style_layers, content_layers=content_layers,
                 style_weights=style_weights, content_weights=content_weights, angle_threshold=60,
                 style_pyramid_mode='single', gram_mode='current'):
        super().__init__()
        if not model_path:
            raise ValueError("No model_path provided")
        self.vgg = VGG(model_path=model_path)
        self.style_layers, self.content_layers = style_layers, content_layers
        self.layers = style_layers + content_layers
        self.style_weights, self.content_weights = style_weights, content_weights
        self.style_pyramid_mode, self.gram_mode = style_pyramid_mode, gram_mode
        if style_pyramid_mode not in self.style_pyramid_modes:
            raise ValueError(f"Unsupported style_pyramid_mode: {style_pyramid_mode}")
        self.angle_threshold = angle_threshold
        self.style_targets = None
        self._engine = None

    def _get_engine(self, device) -> StepEngine:
        if self._engine is None:
            cfg = EngineConfig(tex_w=8, tex_h=8, hierarchical=False, style_layers=list(self.style_layers),
                               content_layers=list(self.content_layers), style_weights=list(self.style_weights),
                               content_weights=list(self.content_weights), angle_threshold=self.angle_threshold,
                               style_pyramid_mode=self.style_pyramid_mode, gram_mode=self.gram_mode,
                               loss_weights={"style": 1.0, "content": 1.0, "tex_reg": 0.0})
            self._engine = StepEngine(cfg, self.vgg.state_dict(), device)
        return self._engine

    def set_style_image(self, style_image, num_levels=5):
        eng = self._get_engine(style_image.device if style_image.is_cuda else torch.device("cuda"))
        eng.set_style_image(style_image, num_levels)
        print('Use style image pyramid of shapes:')
        for s in eng.style_pyramid_sizes:
            print(torch.Size([1, 3, *s]))
        self.style_targets = [{lvl: g[None] for lvl, g in t.items()} for t in eng.targets]

    def forward(self, pred, target_content, pyramid_masks, angle_unnormalized=None):
        if self.style_targets is None:
            raise RuntimeError("set_style_image() must be called before forward()")
        eng = self._engine
        eng.set_external_levels(pyramid_masks, angle_unnormalized, target_content)
        style_loss, content_loss = _LossFn.apply(self, *pred)
        pyramid = {
            'm': [lv.masks[self.layers[-1]].to_dense()[0:1][None] for lv in eng.view],
            'm_passed_angle_filter': [{k: lv.masks[k].to_dense()[1:2][None] for k in self.layers} for lv in eng.view],
            'm_failed_angle_filter': [{k: lv.masks[k].to_dense()[2:3][None] for k in self.layers} for lv in eng.view],
            'f': [{k: lv.factor[k] for k in self.layers} for lv in eng.view],
            'size': len(eng.view),
        }
        return style_loss, content_loss, pyramid
