"""the reprojection warp / masked MSE kernel against the reference-generated golden G9 and
against the oracle restatement on larger seeded view pairs."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from gpu_util import require_gpu

sys.path.insert(0, os.path.join(REPO, "oracle"))
import stylemesh_oracle as O  # noqa: E402
from stylemesh_amd.data import synthetic as S  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ev():
    require_gpu()
    from stylemesh_amd import eval as E
    return E


def _pair(seed, hw):
    rng = np.random.default_rng(seed)
    room = S.BoxRoom((6.0, 4.5, 2.8))
    L = np.asarray((6.0, 4.5, 2.8))
    pos = L * np.array([0.4, 0.45, 0.5]) + rng.uniform(-0.3, 0.3, 3)
    yaw, pitch = rng.uniform(0, 2 * np.pi), rng.uniform(-0.15, 0.15)
    pos2 = pos + rng.uniform(-0.3, 0.3, 3) * np.array([1, 1, 0.3])
    yaw2, pitch2 = yaw + rng.uniform(-0.3, 0.3), pitch + rng.uniform(-0.1, 0.1)
    K, c2w_s = S.camera_matrices(pos, yaw, pitch, hw)
    _, c2w_t = S.camera_matrices(pos2, yaw2, pitch2, hw)
    d_s, d_t = room.render(pos, yaw, pitch, hw)[2], room.render(pos2, yaw2, pitch2, hw)[2]
    d_s[7:15, 20:40] = 0
    col_t = S.smooth_noise(rng, 3, *hw) * 100.0
    col_s = S.smooth_noise(rng, 3, *hw) * 100.0
    t = torch.from_numpy
    return t(K), t(c2w_s), t(c2w_t), t(d_s.astype(np.float32)), t(d_t.astype(np.float32)), t(col_t.astype(np.float32)), \
        t(col_s.astype(np.float32))


def test_reproject_matches_reference_golden(ev):
    d = load_golden("g9_reproject")
    for n in range(3):
        t = lambda k: torch.from_numpy(d[f"{k}{n}"]).cuda()
        H, W = d[f"depth_src{n}"].shape
        color, mask = ev.reproject(t("c2w_src")[None], t("c2w_tar")[None], W, H, t("K")[None], t("depth_src")[None, None],
                                   t("depth_tar")[None, None], t("color_tar")[None], (t("depth_tar") > 0)[None])
        ref_mask = t("out_mask")
        assert int((mask[0] != ref_mask).sum()) <= 2
        both = (mask[0] & ref_mask)[None]
        np.testing.assert_allclose((color[0] * both).cpu().numpy(), (t("out_color") * both).cpu().numpy(), rtol=1e-4,
                                   atol=2e-3)
        assert float((color[0] * ~mask[0][None]).abs().max()) == 0.0      # zero outside the mask


@pytest.mark.parametrize("seed,hw", [(1, (96, 128)), (5, (240, 320)), (9, (37, 53))])
def test_reproject_and_mse_match_oracle(ev, seed, hw):
    K, c2w_s, c2w_t, d_s, d_t, col_t, col_s = _pair(seed, hw)
    ref_color, ref_mask = O.reproject_explicit(c2w_s, c2w_t, K, d_s, d_t, col_t, d_t > 0)
    acc = ev.ReprojectionError()
    warped, mask = acc.update(col_s.cuda(), c2w_s.cuda(), d_s.cuda(), col_t.cuda(), c2w_t.cuda(), d_t.cuda(), K.cuda())
    n_diff = int((mask.cpu() != ref_mask).sum())
    assert n_diff <= max(2, 2e-4 * mask.numel()), n_diff              # threshold tests on fp32 coordinates
    both = (mask.cpu() & ref_mask)[None]
    np.testing.assert_allclose((warped.cpu() * both).numpy(), (ref_color * both).numpy(), rtol=1e-4, atol=2e-3)
    assert 0.2 < float(ref_mask.float().mean()) < 1.0
    m3 = ref_mask[None].expand(3, -1, -1)
    ref_mse = float(((col_s - ref_color)[m3] ** 2).double().mean())
    assert abs(acc.compute() - ref_mse) <= 2e-3 * ref_mse
    # accumulating a second, identical pair leaves the mean unchanged
    acc.update(col_s.cuda(), c2w_s.cuda(), d_s.cuda(), col_t.cuda(), c2w_t.cuda(), d_t.cuda(), K.cuda())
    assert abs(acc.compute() - ref_mse) <= 2e-3 * ref_mse


def test_identity_pair_and_pair_sampling(ev):
    """A frame reprojected onto itself: every pixel with depth survives, and - because the reference maps pixel x to
    the grid value 2 x / W - 1 under align_corners=True - is sampled at x (W - 1) / W, a sub-pixel shift the kernel
    must reproduce (the oracle has it). Pair sampling follows the reference's rules."""
    K, c2w_s, _, d_s, _, col, _ = _pair(4, (64, 80))
    acc = ev.ReprojectionError()
    warped, mask = acc.update(col.cuda(), c2w_s.cuda(), d_s.cuda(), col.cuda(), c2w_s.cuda(), d_s.cuda(), K.cuda())
    ref_color, ref_mask = O.reproject_explicit(c2w_s, c2w_s, K, d_s, d_s, col, d_s > 0)
    # the projected coordinates are integers up to rounding, so the bound tests (px < 0, px >= W - 1) are ties on the
    # image border: compare away from it
    inner = torch.zeros_like(ref_mask)
    inner[1:-1, 1:-1] = True
    # (and floor(px) of an integer +- 1 ulp picks either neighbour set for the depth test at depth discontinuities)
    assert int(((mask.cpu() != ref_mask) & inner).sum()) <= 5e-3 * mask.numel() and float(mask.float().mean()) > 0.5
    both = (mask.cpu() & ref_mask & inner)[None]
    np.testing.assert_allclose((warped.cpu() * both).numpy(), (ref_color * both).numpy(), rtol=1e-4, atol=2e-3)
    m3 = mask.cpu()[None].expand(3, -1, -1)
    ref_mse = float(((col - warped.cpu())[m3] ** 2).double().mean())      # the accumulation, on the kernel's own mask
    assert abs(acc.compute() - ref_mse) <= 1e-5 * ref_mse and ref_mse < 0.2 * float(col.var())
    assert ev.sample_pairs_det(6, 2) == [2, 3, 0, 1, 2, 3]
    assert ev.sample_pairs_det(3, 5) == [0, 1, 2]
    p = ev.sample_pairs(20, 3, random.Random(0))
    assert all(abs(j - i) <= 3 and j != i for i, j in enumerate(p))
    frames = [dict(styled=col.cuda(), depth=d_s.cuda(), pose=c2w_s.cuda()) for _ in range(4)]
    rep = ev.evaluate_sequence(frames, K.cuda(), pair_threshold=2, pair_threshold_short=1, pair_threshold_long=3)
    assert set(rep) == {"reprojection_mse", "reprojection_mse_short", "reprojection_mse_long"}
    assert all(abs(v - ref_mse) <= 2e-3 * ref_mse for v in rep.values())


def test_end_to_end_rasterise_optimise_render_evaluate():
    """SURVEY.md section 8 rows f3 -> a* -> f2 -> f4 chained on a synthetic scene (tools/eval_scene.py): the rasteriser's
    maps feed the optimisation, the texture is exported, and the styled frames of neighbouring views agree where they
    see the same surface (one shared texture): reprojection MSE far below the frames' pixel variance."""
    require_gpu()
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import eval_scene
    before, after, var, saved = eval_scene.main(n_views=6, epochs=1, index_repeat=4, tex=256)
    assert saved, "no texture image written"
    assert all(v == 0.0 for v in before.values())              # zero texture: identical (black) frames
    assert var > 1.0
    for v in after.values():
        assert np.isfinite(v) and 0.0 < v < 0.3 * var
