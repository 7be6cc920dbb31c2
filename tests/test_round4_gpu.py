"""Round-4 kernels and engine paths against their restatements (GPU).

* ``sm_cover_segments`` (parallel since ABI 8) against the SEQUENTIAL greedy it replaces, restated on the host: the lists
  must be identical, entry for entry (flat and pair mode; planes of one chunk, many chunks, and larger than the 160 KB of
  LDS the round-3 kernel was limited to - ADVICE r3 medium).
* ``StepEngine.request_prepare`` / ``training_step(next_batch=...)``: a schedule with a NEW view every step
  (index_repeat 1: scripts/train/optimize_texture_scannet_dip.sh:16, data/abstract_dataset.py:498-512) prepared one view
  ahead gives the results of the same schedule without preparation.
* The split update's pairing rule: a second ``step_compute`` without the closing ``optimizer_step`` raises.
"""
import numpy as np
import pytest
import torch

from gpu_util import require_gpu

pytestmark = pytest.mark.gpu


def _row_stride(w):
    return (w + 1 + 3) // 4 * 4


def greedy_flat(need, tag):
    """The sequential cover of csrc/prep.hip's header comment: over the positions of rows 1 .. h of the padded plane, the
    next segment starts at the first needed position not yet covered, rounded down to a multiple of 4, and covers 32."""
    h, w = need.shape
    Wp = _row_stride(w)
    flat = np.zeros((h, Wp), bool)
    flat[:, 1:w + 1] = need > 0
    pos = np.flatnonzero(flat.reshape(-1))
    out, cursor, k = [], 0, 0
    while k < len(pos):
        if pos[k] < cursor:
            k = int(np.searchsorted(pos, cursor))
            continue
        st = int(pos[k]) & ~3
        out.append((tag << 24) | (Wp + st))
        cursor = st + 32
    return np.array(out, np.int32)


def greedy_pairs(need_pooled, tag, full_w):
    """Pair mode: every row Y of the POOLED need map is covered by runs of 16 windows starting at the first needed window
    not yet covered; a run = the segment of image row 2Y at column 2 X0 and the one right below it."""
    Wp = _row_stride(full_w)
    out = []
    for Y, row in enumerate(need_pooled > 0):
        cursor = 0
        for X in np.flatnonzero(row):
            if X < cursor:
                continue
            q = (2 * Y + 1) * Wp + 2 * int(X) + 1
            out += [(tag << 24) | q, (tag << 24) | (q + Wp)]
            cursor = int(X) + 16
    return np.array(out, np.int32)


def _blobs(rng, h, w, density):
    """Need maps like a view's: a few blobs + isolated pixels, or near-dense."""
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w), bool)
    for _ in range(6):
        cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(0.05, 0.3) * min(h, w)
        m |= (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
    m |= rng.random((h, w)) < density
    return m


@pytest.mark.parametrize("hw", [(5, 7), (16, 21), (64, 85), (256, 341), (784, 1045), (1100, 1400)])
def test_cover_segments_equals_the_sequential_greedy(hw):
    require_gpu()
    from stylemesh_amd.runtime import ops
    h, w = hw
    rng = np.random.default_rng(h * 1000 + w)
    maps = [np.zeros((h, w), bool), np.ones((h, w), bool), _blobs(rng, h, w, 0.0), _blobs(rng, h, w, 0.002),
            rng.random((h, w)) < 0.03, rng.random((h, w)) < 0.6]
    edge = np.zeros((h, w), bool)
    edge[:, 0] = edge[:, -1] = True                 # first / last pixel of every row: segments across row ends
    maps.append(edge)
    problems, keep = [], []
    for g, m in enumerate(maps):
        nd = torch.from_numpy(m.astype(np.float32)).cuda()
        cap = h * _row_stride(w) // 32 + 2
        starts = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
        count = torch.full((1,), -7, dtype=torch.int32, device="cuda")
        problems.append((nd, starts, count, g))
        keep.append((m, starts, count))
    ops.cover_segments(problems)                    # all maps of the size in ONE call
    for g, (m, starts, count) in enumerate(keep):
        ref = greedy_flat(m, g)
        n = int(count)
        assert n == len(ref), (hw, g, n, len(ref))
        assert np.array_equal(starts[:n].cpu().numpy(), ref), (hw, g)
        # the property the conv kernels rely on: disjoint, every needed position covered
        q = (ref & 0xFFFFFF).astype(np.int64)
        assert np.all(np.diff(q) >= 32)
    # mixed sizes and modes in one call
    h2, w2 = max(2, h // 2), max(2, w // 2)
    pooled = _blobs(rng, h2, w2, 0.01)
    nd_p = torch.from_numpy(pooled.astype(np.float32)).cuda()
    cap_p = 2 * h2 * ((w2 + 15) // 16 + 1) + 2
    st_p = torch.full((cap_p,), -1, dtype=torch.int32, device="cuda")
    ct_p = torch.zeros(1, dtype=torch.int32, device="cuda")
    small = rng.random((7, 9)) < 0.4
    nd_s = torch.from_numpy(small.astype(np.float32)).cuda()
    st_s = torch.full((16,), -1, dtype=torch.int32, device="cuda")
    ct_s = torch.zeros(1, dtype=torch.int32, device="cuda")
    full_w = 2 * w2 + (w % 2 if w >= 4 else 0)
    ops.cover_segments([(nd_p, st_p, ct_p, 3, full_w), (nd_s, st_s, ct_s, 5), problems[2]])
    ref_p = greedy_pairs(pooled, 3, full_w)
    assert int(ct_p) == len(ref_p) and np.array_equal(st_p[:len(ref_p)].cpu().numpy(), ref_p)
    ref_s = greedy_flat(small, 5)
    assert int(ct_s) == len(ref_s) and np.array_equal(st_s[:len(ref_s)].cpu().numpy(), ref_s)


def test_cover_segments_respects_the_capacity():
    require_gpu()
    from stylemesh_amd.runtime import ops
    m = np.ones((40, 50), bool)
    nd = torch.from_numpy(m.astype(np.float32)).cuda()
    starts = torch.full((10,), -1, dtype=torch.int32, device="cuda")
    guard = starts.clone()
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.cover_segments([(nd, starts, count, 1)])
    ref = greedy_flat(m, 1)
    assert int(count) == len(ref) > 10               # the true count is reported, only `cap` entries are written
    assert np.array_equal(starts.cpu().numpy(), ref[:10]) and guard.numel() == 10


def _dip_engine(n_layers=1):
    from stylemesh_amd.data import synthetic as S
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    cfg = EngineConfig(tex_w=512, tex_h=512, hierarchical=True, n_layers=n_layers, style_weights=[1000., 1000., 10., 10., 1000.],
                       angle_threshold=3000.0, style_pyramid_mode="single", gram_mode="average", use_angle_weight=False,
                       use_depth_scaling=False, loss_weights={"content": 7e1, "style": 1e-3, "tex_reg": 0.0},
                       learning_rate=1.0, decay_step_size=15)
    eng = StepEngine(cfg, S.seeded_vgg_state(0))
    eng.set_style_image(S.style_image(1, 96, 80))
    return eng


def _small_views(seeds, hw=(48, 64)):
    from stylemesh_amd.data import synthetic as S
    room = S.BoxRoom((12.0, 9.0, 3.0))
    dev = torch.device("cuda")
    out = []
    for s in seeds:
        v = S.make_view(s, view_hw=hw, level_hw=[hw], level_heights=[hw[0]], min_pyramid_depth=0.25, room=room)
        out.append(tuple([u.to(dev) for u in x] if isinstance(x, list) else (x.to(dev) if torch.is_tensor(x) and i != 8 else x)
                         for i, x in enumerate(v)))
    return out


def test_a_new_view_every_step_prepared_one_ahead_equals_unprepared():
    """index_repeat 1 (the dip scripts): ``training_step(next_batch=...)`` prepares view i + 1 beside step i; 12 steps so
    that the 10-deep Gram history of gram_mode 'average' wraps. Same losses and textures as without preparation."""
    require_gpu()
    views = _small_views((0, 2, 6, 7, 9, 11, 12, 14, 16, 18, 22, 23))
    a, b = _dip_engine(), _dip_engine()
    b.prepare_ahead = False
    for k, v in enumerate(views):
        nxt = views[k + 1] if k + 1 < len(views) else None
        la = a.losses(a.training_step(v, next_batch=nxt))
        lb = b.losses(b.training_step(v))
        np.testing.assert_allclose(la["total"], lb["total"], rtol=1e-5)
    assert getattr(a, "prepared_swaps", 0) >= len(views) - 2      # every view after the first came prepared
    err = (a.arena.p - b.arena.p).abs()
    assert float((err > 1e-4).float().mean()) < 5e-3, float(err.max())


def test_step_compute_twice_without_optimizer_step_raises():
    """The early half of the split update has already moved the texels outside the view: a step that is never closed by
    ``optimizer_step`` would leave the texture half-updated."""
    require_gpu()
    from stylemesh_amd.data import synthetic as S
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    cfg = EngineConfig(tex_w=256, tex_h=256, hierarchical=True, n_layers=4, style_weights=[1000., 1000., 10., 10., 1000.],
                       angle_threshold=30.0, style_pyramid_mode="multi", loss_weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3},
                       learning_rate=1.0)
    eng = StepEngine(cfg, S.seeded_vgg_state(0))
    eng.set_style_image(S.style_image(1, 96, 80))
    v = _small_views((2,))[0]
    eng.training_step(v)                      # a whole step: fine
    eng.step_compute(v)
    if eng._adam_early_done is None:
        pytest.skip("the split update did not run on this view (nothing outside the view has been touched yet)")
    with pytest.raises(RuntimeError, match="optimizer_step"):
        eng.step_compute(v)
