"""Reprojection-error evaluation with the reference's surface (data/utils.py:73-194 ``reproject``;
scripts/eval/eval_image_folders.py:185-204 pair sampling, :286-305 masked MSE) over ``sm_reproject``.

``reproject`` keeps the reference signature and return values (warped target colour [B,3,H,W], zero outside the
validity mask, and the mask [B,H,W] bool); ``ReprojectionError`` is the accumulator the evaluation script builds
from ``torchmetrics.MeanSquaredError`` over the masked elements; ``evaluate_sequence`` runs the script's three
pairings (random within +-threshold, fixed short and long offsets) over a list of frames already in memory.
LPIPS / Gram distances of that script need third-party networks and are out of scope. GPU + HIP library only.
"""
from __future__ import annotations

import random

import torch

from ..runtime import hip
from ..runtime.hip import lib, ptr


def _dense(t, shape):
    t = t.detach().to(torch.float32).contiguous()
    assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
    return t


def _src2tar(c2w_src, c2w_tar):
    """inverse(cam2world_tar) @ cam2world_src in fp32 on the host, as the reference computes it (:78-79)."""
    m = torch.linalg.inv(c2w_tar.detach().float().cpu()) @ c2w_src.detach().float().cpu()
    return m.contiguous()


def _launch(c2w_src, c2w_tar, K, depth_src, depth_tar, color_tar, mask_tar, styled_src=None, depth_tol=0.1):
    H, W = depth_src.shape[-2:]
    dev = color_tar.device
    if dev.type != "cuda":
        raise RuntimeError("reprojection evaluation needs the tensors on the GPU (no CPU path)")
    d_s, d_t = _dense(depth_src.reshape(H, W), (H, W)), _dense(depth_tar.reshape(H, W), (H, W))
    col = _dense(color_tar.reshape(3, H, W), (3, H, W))
    msk = _dense(mask_tar.reshape(H, W), (H, W))
    sty = None if styled_src is None else _dense(styled_src.reshape(3, H, W), (3, H, W))
    m = _src2tar(c2w_src, c2w_tar)
    Kc = K.detach().float().cpu()
    intr = torch.tensor([Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2]], dtype=torch.float32)
    out = torch.empty(3, H, W, device=dev)
    mout = torch.empty(H, W, dtype=torch.uint8, device=dev)
    nb = lib.sm_reproject_blocks(H, W)
    partial = torch.zeros(nb, 2, dtype=torch.float64, device=dev)
    hip.check(lib.sm_reproject(m.data_ptr(), intr.data_ptr(), H, W, ptr(d_s), ptr(d_t), ptr(col), ptr(msk), ptr(sty),
                               ptr(out), ptr(mout), ptr(partial), float(depth_tol), hip.stream()), "sm_reproject")
    return out, mout.bool(), partial


def reproject(cam2world_src, cam2world_tar, W, H, intrinsic, depth_src, depth_tar, color_tar, mask_tar):
    """Reference signature (data/utils.py:73): batched [B,...] tensors; returns (color_tar_to_src [B,3,H,W],
    mask [B,H,W] bool)."""
    B = mask_tar.shape[0]
    colors, masks = [], []
    for b in range(B):
        c, m, _ = _launch(cam2world_src[b], cam2world_tar[b], intrinsic[b], depth_src[b], depth_tar[b], color_tar[b],
                          mask_tar[b])
        assert c.shape[-2:] == (H, W)
        colors.append(c)
        masks.append(m)
    return torch.stack(colors), torch.stack(masks)


class ReprojectionError:
    """Masked MSE between a styled source frame and the styled target frame warped into it, accumulated over pairs
    (``torchmetrics.MeanSquaredError(compute_on_step=False)`` on ``masked_select``-ed elements,
    eval_image_folders.py:206,299-302): sum of squared differences / number of compared elements."""

    def __init__(self):
        self.sum = 0.0
        self.count = 0.0
        self._pending = []

    def update(self, styled_src, pose_src, depth_src, styled_tar, pose_tar, depth_tar, intrinsic, depth_tol=0.1):
        """Frames as [3,H,W] / [H,W] / [4,4] device tensors; the target validity mask is ``depth_tar > 0`` (:296)."""
        warped, mask, partial = _launch(pose_src, pose_tar, intrinsic, depth_src, depth_tar, styled_tar,
                                        (depth_tar > 0), styled_src, depth_tol)
        self._pending.append(partial)          # no host sync per pair
        return warped, mask

    def compute(self) -> float:
        if self._pending:
            tot = torch.stack([p.sum(0) for p in self._pending]).sum(0).tolist()
            self.sum += tot[0]
            self.count += tot[1]
            self._pending = []
        return self.sum / self.count if self.count > 0 else float("nan")


def sample_pairs(n, threshold=10, rng=random):
    """Random partner within +-threshold frames (eval_image_folders.py:185-193)."""
    pairs = []
    for i in range(n):
        start, end = max(0, i - threshold), min(n, i + threshold)
        pairs.append(rng.choice([j for j in range(start, end) if j != i]))
    return pairs


def sample_pairs_det(n, threshold=10):
    """Fixed offset: i - threshold, else i + threshold, else i itself (eval_image_folders.py:196-204)."""
    pairs = []
    for i in range(n):
        left, right = i - threshold, i + threshold
        pairs.append(left if left >= 0 else right if right < n else i)
    return pairs


def evaluate_sequence(frames, intrinsic, pair_threshold=10, pair_threshold_short=5, pair_threshold_long=20, seed=0):
    """``frames``: list of dicts with device tensors ``styled`` [3,H,W], ``depth`` [H,W], ``pose`` [4,4] (cam2world).
    Returns {'reprojection_mse', 'reprojection_mse_short', 'reprojection_mse_long'} like the script's report."""
    n = len(frames)
    rng = random.Random(seed)
    plans = {"reprojection_mse": sample_pairs(n, pair_threshold, rng),
             "reprojection_mse_short": sample_pairs_det(n, pair_threshold_short),
             "reprojection_mse_long": sample_pairs_det(n, pair_threshold_long)}
    out = {}
    for name, pairs in plans.items():
        acc = ReprojectionError()
        for i, j in enumerate(pairs):
            a, b = frames[i], frames[j]
            acc.update(a["styled"], a["pose"], a["depth"], b["styled"], b["pose"], b["depth"], intrinsic)
        out[name] = acc.compute()
    return out
