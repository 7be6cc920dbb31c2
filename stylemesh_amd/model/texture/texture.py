"""``NeuralTexture`` / ``HierarchicalNeuralTexture`` with the reference's surface (model/texture/texture.py:10-135)
over the HIP texture kernels.

Same constructor arguments, attributes (``.data`` Parameter [C,H,W], ``.layers`` ModuleList), methods
(``forward``, ``normalize``, ``from_tensor``, ``get_image``, ``regularizer``, ``save_image``, ``save_layers``,
``save_texture``) and the same in-place clamp at the start of every forward. ``forward`` is differentiable: its
backward is the atomic scatter-add kernel (K2). C = 3 only (the reference's pipeline hard-codes 3, model.py:79).
The module requires the built HIP library and a GPU; there is no CPU path.
"""
from __future__ import annotations

from os.path import join

import torch
import torch.nn as nn

from ...runtime import ops
from ...runtime.fmap import FMap
from .utils import from_grid_range


def to_image(texture, startIndex=0, padChannels=True, normalize_transform=from_grid_range):
    """Texture tensor -> PIL image (reference texture.py:10-19). The colour transform and the 8-bit quantisation run on
    the tensor's own device; only the uint8 image (48 MB at 4096^2) crosses to the host."""
    from PIL import Image
    texture = texture.detach()[startIndex:(startIndex + 3)]
    if padChannels and texture.shape[0] != 3:
        c, (h, w) = 3 - texture.shape[0], texture.shape[1:]
        texture = torch.cat((texture, torch.zeros(c, h, w).type_as(texture)), dim=0)
    texture = normalize_transform(texture.clone()).clamp(0, 1)
    arr = (texture.permute(1, 2, 0) * 255.0 + 0.5).to(torch.uint8).cpu().numpy()   # ToPILImage of a float tensor
    return Image.fromarray(arr)


class _ImageWriter:
    """JPEG encoding of the per-epoch texture exports off the training thread: ``save_image`` hands the finished PIL
    image (a host copy: the texture keeps training) to ONE writer thread - the encoder runs in C without the
    interpreter lock - and ``wait()`` (end of ``MiniTrainer.fit``, interpreter exit) joins what is pending. The
    reference encodes five 4096^2-class JPEGs synchronously at every epoch end (model/model.py:378-385)."""

    def __init__(self):
        self._pool, self._pending = None, []

    def save(self, image, path):
        import atexit
        from concurrent.futures import ThreadPoolExecutor
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="stylemesh-texture-export")
            atexit.register(self.wait)
        self._pending.append(self._pool.submit(image.save, path))

    def wait(self):
        pending, self._pending = self._pending, []
        for f in pending:
            f.result()


IMAGE_WRITER = _ImageWriter()


class _SampleFn(torch.autograd.Function):
    """sum_l grid_sample(layer_l, grid; bilinear, border, align_corners=True) for a batch of one."""

    @staticmethod
    def forward(ctx, grid, *layers):
        if grid.shape[0] != 1:
            raise ValueError("batch size 1 only")
        for l in layers:
            if l.shape[0] != 3:
                raise ValueError("textures with C = 3 channels only")
        g = grid.detach().contiguous().float()
        h, w = g.shape[1:3]
        out = FMap(3, h, w, g.device)
        ops.tex_sample_fwd([l.detach() for l in layers], g, out)
        ctx.save_for_backward(g)
        ctx.shapes = [tuple(l.shape) for l in layers]
        return out.to_dense()[None]

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        h, w = g.shape[1:3]
        gimg = FMap(3, h, w, g.device).from_dense(grad_out[0])
        grads = [torch.zeros(s, device=g.device) for s in ctx.shapes]
        ops.tex_sample_bwd(grads, g, gimg, None)
        return (None, *grads)


class NeuralTexture(nn.Module):
    def __init__(self, W, H, C, random_init=False):
        super().__init__()
        self.W, self.H, self.C = W, H, C
        init = torch.rand(C, H, W) if random_init else torch.zeros(C, H, W)
        self.data = nn.Parameter(init, requires_grad=True)

    @staticmethod
    def from_tensor(data: torch.Tensor):
        C, H, W = data.shape
        texture = NeuralTexture(W, H, C)
        texture.data = nn.Parameter(data, requires_grad=True)
        return texture

    def normalize(self):
        with torch.no_grad():
            self.data.clamp_(ops.CLAMP_LO, ops.CLAMP_HI)

    def forward(self, x):
        self.normalize()
        return _SampleFn.apply(x, self.data)

    def get_image(self):
        return self.data

    def save_image(self, dir, prefix="", normalize_transform=from_grid_range):
        IMAGE_WRITER.save(to_image(self.get_image(), normalize_transform=normalize_transform), join(dir, f"{prefix}texture.jpg"))

    def save_layers(self, dir, prefix="", normalize_transform=from_grid_range):
        self.save_image(dir, prefix, normalize_transform)

    def save_texture(self, dir, prefix=""):
        torch.save(self.get_image().detach().cpu(), join(dir, f"{prefix}texture.pt"))


class HierarchicalNeuralTexture(nn.Module):
    def __init__(self, W, H, C, num_layers=4, random_init=False):
        super().__init__()
        self.W, self.H, self.C = W, H, C
        # laplacian-style pyramid: layer i has size (W // 2^i, H // 2^i)
        self.layers = nn.ModuleList([NeuralTexture(W // pow(2, i), H // pow(2, i), C, random_init)
                                     for i in range(num_layers)])

    @staticmethod
    def from_tensor(data: list):
        C, H, W = data[0].shape
        textures = []
        for i, d in enumerate(data):
            ci, hi, wi = d.shape
            assert (W // pow(2, i) == wi and H // pow(2, i) == hi and C == ci)
            textures.append(NeuralTexture.from_tensor(d))
        texture = HierarchicalNeuralTexture(W, H, C, num_layers=len(textures))
        texture.layers = nn.ModuleList(textures)
        return texture

    def forward(self, x):
        for layer in self.layers:
            layer.normalize()
        return _SampleFn.apply(x, *[layer.data for layer in self.layers])   # one fused pass over all layers

    def regularizer(self, weights):
        reg = 0.0
        for i, layer in enumerate(self.layers):
            reg += torch.mean(torch.pow(layer.data, 2.0)) * weights[i]
        return reg

    def get_image(self):
        w_range = torch.arange(0, self.W, dtype=torch.float) / (self.W - 1.0) * 2.0 - 1.0
        h_range = torch.arange(0, self.H, dtype=torch.float) / (self.H - 1.0) * 2.0 - 1.0
        v, u = torch.meshgrid(h_range, w_range, indexing="ij")
        uv_id = torch.stack([u, v], 2).unsqueeze(0).type_as(self.layers[0].data)
        return self.forward(uv_id)[0, 0:3, :, :]

    def save_image(self, dir, prefix="", normalize_transform=from_grid_range):
        with torch.no_grad():
            IMAGE_WRITER.save(to_image(self.get_image(), normalize_transform=normalize_transform),
                              join(dir, f"{prefix}texture.jpg"))

    def save_layers(self, dir, prefix="", normalize_transform=from_grid_range):
        for i, l in enumerate(self.layers):
            l.save_image(dir, prefix + f"_layer{str(i)}_", normalize_transform)

    def save_texture(self, dir, prefix=""):
        for i, l in enumerate(self.layers):
            l.save_texture(dir, f"{prefix}layer-{i}-")
