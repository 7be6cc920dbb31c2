"""Input contract of the hot path: the per-view maps the texture optimisation consumes.

Restates, in numpy, the pieces of the reference's dataset layer that *compute* something
(everything else there is file-system crawling, which is out of scope):

* ``calculate_mask``         - reference ``data/scannet_dataset.py:308-328`` (ScanNet: UV != 0 AND
                               resized depth > 0) and ``data/matterport_dataset.py:295-311`` (UV only)
* ``calculate_depth_level``  - reference ``data/scannet_dataset.py:330-366``
                               (= ``data/matterport_dataset.py:313-349``)
* ``uv_to_grid``             - reference ``model/texture/utils.py:56-60,87-91`` (``ToTensor`` + ``to_grid``)
* ``pre`` / ``post``         - reference ``model/losses/rgb_transform.py:5-21``
* ``RepeatingSampler``       - reference ``data/abstract_dataset.py:498-512``
* ``assemble_batch``         - the 13-tuple of ``Abstract_Dataset.__getitem__``
                               (``data/abstract_dataset.py:270-345``) after batch collation (B = 1)

Everything here is host-side preparation that happens once per view; none of it is on the
per-step GPU path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# reference: model/losses/rgb_transform.py:8 (BGR order, [0,1] scale)
IMAGENET_MEAN_BGR = (0.40760392, 0.45795686, 0.48501961)


def pre(rgb01: torch.Tensor) -> torch.Tensor:
    """RGB in [0,1] ``[3,H,W]`` -> BGR, mean-subtracted, x255 (reference rgb_transform.py:5-11)."""
    x = rgb01[[2, 1, 0]].clone().float()
    mean = torch.tensor(IMAGENET_MEAN_BGR, dtype=x.dtype).view(3, 1, 1)
    return (x - mean) * 255.0


def post(x: torch.Tensor) -> torch.Tensor:
    """Inverse of :func:`pre` followed by clamp to [0,1] (reference rgb_transform.py:14-21).

    Unlike the reference this never mutates its argument (the reference's in-place ``mul_`` scales
    the live parameter when run on CPU, SURVEY.md section 4 hazard 2).
    """
    x = x.detach().float() * (1.0 / 255.0)      # on the tensor's own device (a 4096^2 texture: on the GPU)
    mean = torch.tensor(IMAGENET_MEAN_BGR, dtype=x.dtype, device=x.device).view(3, 1, 1)
    x = x + mean
    return x[[2, 1, 0]].clamp(0, 1)


def _cv2_linear_taps(n_in: int, n_out: int):
    """Source index and fraction of cv2's INTER_LINEAR along one axis (imgproc/resize.cpp): the coordinate
    ``(dst + 0.5) * in / out - 0.5`` is formed in double and rounded to float, the fraction is dropped at both borders."""
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo, hi = s < 0, s >= n_in - 1
    s = np.clip(s, 0, n_in - 1)
    f[lo | hi] = 0.0
    return s, np.minimum(s + 1, n_in - 1), f


def resize_bilinear_np(img: np.ndarray, out_hw) -> np.ndarray:
    """``cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR)`` of a 2-D float32 array, as the reference uses it at
    ``data/abstract_dataset.py:301`` and ``data/scannet_dataset.py:323``: half-pixel centres, no anti-aliasing, float
    weights, horizontal pass first (cv2 is absent from the target image; tests/test_reference_conventions.py checks
    this against a loop restatement and hand-worked samples)."""
    a = np.ascontiguousarray(img, dtype=np.float32)
    oh, ow = out_hw
    x0, x1, fx = _cv2_linear_taps(a.shape[1], int(ow))
    y0, y1, fy = _cv2_linear_taps(a.shape[0], int(oh))
    one = np.float32(1.0)
    rows = a[:, x0] * (one - fx) + a[:, x1] * fx                       # float32 throughout, like cv2's WT = float
    return (rows[y0] * (one - fy)[:, None] + rows[y1] * fy[:, None]).astype(np.float32)


def resize_nearest_np(img: np.ndarray, out_hw) -> np.ndarray:
    """Legacy nearest resize ``src = floor(dst * in / out)`` (cv2.INTER_NEAREST as used for the angle map at
    ``data/abstract_dataset.py:308``; the mask goes through Pillow instead: ``resize_mask_pil``)."""
    h, w = img.shape[:2]
    oh, ow = out_hw
    ys = np.minimum((np.arange(oh) * (h / oh)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def resize_mask_pil(mask: np.ndarray, out_hw) -> np.ndarray:
    """The reference's mask resize: ``Image.fromarray(bool mask).resize((w, h), Image.NEAREST)``
    (``data/scannet_dataset.py:325``, ``data/abstract_dataset.py:311``). Pillow's nearest filter samples at pixel
    CENTRES (source index = floor((dst + 0.5) * in / out), accumulated step by step in double) - not cv2's / torch's
    legacy ``floor(dst * in / out)`` - so the same Pillow call is made here rather than restated."""
    from PIL import Image
    h, w = out_hw
    return np.asarray(Image.fromarray(np.ascontiguousarray(mask, dtype=bool)).resize((int(w), int(h)), Image.NEAREST)).astype(bool)


def calculate_mask(uvmap: np.ndarray, depth: np.ndarray | None = None) -> np.ndarray:
    """Valid-pixel mask of a UV map ``(H,W,3)``: u != 0 OR v != 0, AND (ScanNet only) depth > 0 after a
    bilinear resize of the depth image to the UV map's size. Reference ``data/scannet_dataset.py:308-328``.
    Pass ``depth=None`` for the Matterport variant (``data/matterport_dataset.py:295-311``)."""
    mask = (uvmap[:, :, 0] != 0) | (uvmap[:, :, 1] != 0)
    if depth is not None:
        d = np.asarray(depth, dtype=np.float32)
        if d.ndim == 3:
            d = d[:, :, 0]
        if d.shape != mask.shape:
            d = resize_bilinear_np(d, mask.shape)
        mask = mask & (d > 0)
    return mask


def pyramid_heights(pyramid_levels: int, min_pyramid_height: int = 256, max_height: int = 960,
                    n_rendered: int = 5) -> np.ndarray:
    """UV pyramid heights the reference's renderer script produces (``scripts/scannet/render_uvs.py:79-90``:
    ``linspace(256, 960, 5)``), truncated to ``pyramid_levels`` entries."""
    return np.linspace(min_pyramid_height, max_height, n_rendered)[:pyramid_levels]


def calculate_depth_level(depth: np.ndarray, levels, min_pyramid_depth: float):
    """Per-pixel UV-pyramid level selection from metric depth. Reference ``data/scannet_dataset.py:330-366``.

    ``uv_h = 32 * depth / min_pyramid_depth`` is the ideal UV-map height for a pixel; returns
    ``(continuous_level f32, nearest_level i64, second_nearest_level i64, weight_of_nearest f32)``.
    """
    levels = np.asarray(levels, dtype=np.float64)
    n_levels = len(levels)
    depth = np.asarray(depth)
    if depth.ndim == 3:
        depth = depth.squeeze()
    uv_height = 32 * (depth / min_pyramid_depth)
    dist = np.subtract.outer(uv_height, levels)
    rounded = np.argmin(np.abs(dist), axis=2)
    residues = levels[rounded] - uv_height
    step = np.where(residues > 0, -1, 1)
    step[residues == 0] = 0
    other = np.clip(rounded + step, 0, n_levels - 1)
    height_difference = np.abs(levels[rounded] - levels[other])
    w = np.abs(residues / (height_difference + 1e-6))
    w[height_difference == 0] = 0
    w = 1 - w
    cont = np.where(residues > 0, other + w, other - w)
    cont[w == 1] = rounded[w == 1]
    return (cont.astype(np.float32), rounded.astype(np.int64), other.astype(np.int64), w.astype(np.float32))


def uv_to_grid(uv_hw3: np.ndarray) -> torch.Tensor:
    """``(H,W,3)`` UV map in [0,1] -> ``(H,W,2)`` sampling grid in [-1,1] (``x*2-1``, drop the LOD channel).
    Reference ``model/texture/utils.py:6-8,16-18,42-46,56-60``."""
    t = torch.from_numpy(np.ascontiguousarray(uv_hw3[:, :, :2], dtype=np.float32))
    return t * 2.0 - 1


class RepeatingSampler:
    """Yields each index ``index_repeat`` consecutive times (reference ``data/abstract_dataset.py:498-512``)."""

    def __init__(self, indices, index_repeat):
        if isinstance(index_repeat, int):
            self.indices = [i for i in indices for _ in range(index_repeat)]
        elif isinstance(index_repeat, list):
            self.indices = [i for i in indices for _ in range(index_repeat[i])]
        else:
            raise ValueError('unsupported index_repeat type', index_repeat)

    def __iter__(self):
        return iter(self.indices)

    def __len__(self):
        return len(self.indices)


def assemble_batch(rgb01_chw: torch.Tensor, depth_hw: np.ndarray, uv_levels, angle_cos_hw: np.ndarray,
                   levels, min_pyramid_depth: float, idx: int = 0, use_depth_in_mask: bool = True,
                   extrinsics=None, intrinsics=None):
    """Build the collated (B = 1) 13-tuple the pipeline's ``training_step`` consumes.

    Order and shapes follow ``Abstract_Dataset.__getitem__`` (``data/abstract_dataset.py:270-345``):
    ``(rgb[1,3,H,W], extr[1,4,4], intr[1,4,4], depth[1,1,H,W], depth_level f32, rounded_level i64,
    other_level i64, interp_weight f32 (all [1,1,H,W]), idx, [uv_i[1,H_i,W_i,2]], mask bool[1,H,W],
    angle_guidance[1,1,H,W] = cos(theta), angle_degrees[1,1,H,W])``.

    ``depth_hw`` and ``angle_cos_hw`` are at the base view resolution; ``uv_levels`` is the list of
    ``(H_i,W_i,3)`` UV maps (last = largest). The mask is computed on the largest UV map and
    nearest-resized to the base resolution exactly as the reference does (``:288,311``).
    """
    H, W = depth_hw.shape
    mask_big = calculate_mask(uv_levels[-1], depth_hw if use_depth_in_mask else None)
    mask = resize_mask_pil(mask_big, (H, W))
    cont, rounded, other, w = calculate_depth_level(depth_hw, levels, min_pyramid_depth)
    angle = torch.from_numpy(np.ascontiguousarray(angle_cos_hw, dtype=np.float32))[None, None]
    eye = torch.eye(4, dtype=torch.float64)[None]
    return (
        pre(rgb01_chw)[None],
        eye.clone() if extrinsics is None else extrinsics,
        eye.clone() if intrinsics is None else intrinsics,
        torch.from_numpy(np.ascontiguousarray(depth_hw, dtype=np.float32))[None, None],
        torch.from_numpy(cont)[None, None],
        torch.from_numpy(rounded)[None, None],
        torch.from_numpy(other)[None, None],
        torch.from_numpy(w)[None, None],
        torch.tensor([idx]),
        [uv_to_grid(u)[None] for u in uv_levels],
        torch.from_numpy(mask)[None],
        angle,
        torch.rad2deg(torch.acos(angle)),
    )
