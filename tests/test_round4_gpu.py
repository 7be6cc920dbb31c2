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


def _views_multi(seeds, view_hw=(48, 64), level_hw=((24, 32), (36, 48), (48, 64))):
    from stylemesh_amd.data import synthetic as S
    room = S.BoxRoom((12.0, 9.0, 3.0))
    dev = torch.device("cuda")
    out = []
    for s in seeds:
        v = S.make_view(s, view_hw=view_hw, level_hw=list(level_hw), level_heights=[h for h, _ in level_hw],
                        min_pyramid_depth=0.25, room=room)
        out.append(tuple([u.to(dev) for u in x] if isinstance(x, list) else (x.to(dev) if torch.is_tensor(x) and i != 8 else x)
                         for i, x in enumerate(v)))
    return out


@pytest.mark.parametrize("flags", [dict(mode="multi", angle=True, depth=True, thr=30.0),
                                   dict(mode="single", angle=False, depth=False, thr=3000.0),
                                   dict(mode="multi", angle=True, depth=False, thr=30.0)])
@pytest.mark.parametrize("conv_mode", ["split2", "f32"])
def test_grouped_view_preparation_equals_the_call_per_layer_path(flags, conv_mode, monkeypatch):
    """``viewplan.ViewPlan`` (sm_view_masks + sm_view_lists) against ``_set_view_body``'s call-per-layer path: the same
    level maps, layer masks, counts, factors, content targets and - entry for entry - the same active lists."""
    require_gpu()
    from stylemesh_amd.data import synthetic as S
    from stylemesh_amd.runtime import ops
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    monkeypatch.setattr(ops, "CONV_MODE", conv_mode)
    monkeypatch.setattr(ops, "GRAM_MODE", conv_mode)
    from stepcmp import assert_same_step, lock
    views = _views_multi((2, 6, 9, 0))

    def engine(fast):
        cfg = EngineConfig(tex_w=512, tex_h=512, hierarchical=True, n_layers=4, style_weights=[1000., 1000., 10., 10., 1000.],
                           angle_threshold=flags["thr"], style_pyramid_mode=flags["mode"], use_angle_weight=flags["angle"],
                           use_depth_scaling=flags["depth"], loss_weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3},
                           learning_rate=1.0)
        e = StepEngine(cfg, S.seeded_vgg_state(0))
        e.set_style_image(S.style_image(1, 96, 80))
        e.fast_view = fast
        return e
    a, b = engine(True), engine(False)
    for v in views:
        a.set_view(v)
        b.set_view(v)
        assert a._pending_view is None and a.view_tiles is not None
        assert [lv.active for lv in a.view] == [lv.active for lv in b.view]
        assert set(a.view_tiles) == set(b.view_tiles)
        for key in b.view_tiles:
            la, fa = a.view_tiles[key]
            lb, fb = b.view_tiles[key]
            assert torch.equal(la, lb), key
            assert abs(fa - fb) < 1e-12, key
        assert torch.equal(a.view_consts, b.view_consts)
        for la, lb in zip(a.view, b.view):
            if not lb.active:
                continue
            assert torch.equal(la.M, lb.M) and torch.equal(la.passed, lb.passed)
            if lb.pixel_weight is not None:
                assert torch.equal(la.pixel_weight, lb.pixel_weight)
            for layer in b.loss_layers:
                assert torch.equal(la.masks[layer].planes, lb.masks[layer].planes), layer
            for layer in b.cfg.content_layers:
                assert torch.equal(la.content_target[layer].planes, lb.content_target[layer].planes)
        # and a training step on top of it, from the same state, is the same step (tests/stepcmp.py: not bit for bit -
        # the Gram kernels add their position ranges with fp32 atomics, two runs of ONE path differ too)
        m0, v0 = lock(a, b)
        np.testing.assert_allclose(a.losses(a.training_step(v))["total"], b.losses(b.training_step(v))["total"], rtol=1e-5)
        assert_same_step(a, b, m0, v0, what="step on the prepared view")


def test_empty_level_found_at_the_read_back_is_dropped(monkeypatch):
    """A level whose mask is empty (model/model.py:256-257) is only known after the read-back: the grouped path redoes
    the view without it, as the call-per-layer path does."""
    require_gpu()
    from stylemesh_amd.data import synthetic as S
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    v = list(_views_multi((2,))[0])
    # send every pixel to depth level 2: levels 0 and 1 end up empty
    v[5] = torch.full_like(v[5], 2)
    v[6] = torch.full_like(v[6], 2)
    v = tuple(v)
    res = []
    for fast in (True, False):
        cfg = EngineConfig(tex_w=256, tex_h=256, hierarchical=True, n_layers=4, style_weights=[1000., 1000., 10., 10., 1000.],
                           angle_threshold=30.0, style_pyramid_mode="multi", loss_weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3})
        e = StepEngine(cfg, S.seeded_vgg_state(0))
        e.set_style_image(S.style_image(1, 96, 80))
        e.fast_view = fast
        res.append((e, e.losses(e.training_step(v))))
    (a, la), (b, lb) = res
    assert [lv.active for lv in a.view] == [lv.active for lv in b.view] == [False, False, True]
    np.testing.assert_allclose(la["total"], lb["total"], rtol=1e-5)
    from stepcmp import assert_same_step
    zero = torch.zeros_like(a.arena.m)       # both engines took their first step from the zero texture
    assert_same_step(a, b, zero, zero, what="first step without the empty levels")


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
    that the 10-deep Gram history of gram_mode 'average' wraps. Every step is the step of the unprepared engine."""
    require_gpu()
    views = _small_views((0, 2, 6, 7, 9, 11, 12, 14, 16, 18, 22, 23))
    a, b = _dip_engine(), _dip_engine()
    b.prepare_ahead = False
    from stepcmp import assert_same_step, lock
    for k, v in enumerate(views):
        nxt = [views[j] for j in (k + 1, k + 2) if j < len(views)] or None      # two views in preparation (three slots)
        m0, v0 = lock(a, b)                      # lock-step, Gram history included (tests/stepcmp.py)
        la = a.losses(a.training_step(v, next_batch=nxt))
        lb = b.losses(b.training_step(v))
        np.testing.assert_allclose(la["total"], lb["total"], rtol=1e-5)
        assert_same_step(a, b, m0, v0, what=f"step {k}")
    assert getattr(a, "prepared_swaps", 0) >= len(views) - 2      # every view after the first came prepared


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


PROGRAM_CASES = {
    "only2D": dict(mode="single", angle=False, depth=False, thr=3000.0, gram_mode="current", n_layers=4, rep=5,
                   weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3}),
    "with_angle": dict(mode="multi", angle=True, depth=False, thr=30.0, gram_mode="current", n_layers=4, rep=5,
                       weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3}),
    "dip": dict(mode="single", angle=False, depth=False, thr=3000.0, gram_mode="average", n_layers=1, rep=1,
                weights={"content": 7e1, "style": 1e-3, "tex_reg": 0.0}),
    # several UV levels whose ACTIVE set changes from view to view (an empty level is dropped): programs of different level
    # sets alternate in one slot, and every buffer whose size follows the level set must stay where it was recorded
    "multi_level": dict(mode="multi", angle=True, depth=True, thr=30.0, gram_mode="current", n_layers=4, rep=3, levels=True,
                        weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3}),
}


def _program_engine(c, programs):
    from stylemesh_amd.data import synthetic as S
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    cfg = EngineConfig(tex_w=512, tex_h=512, hierarchical=True, n_layers=c["n_layers"], style_weights=[1000., 1000., 10., 10., 1000.],
                       angle_threshold=c["thr"], style_pyramid_mode=c["mode"], gram_mode=c["gram_mode"],
                       use_angle_weight=c["angle"], use_depth_scaling=c["depth"], loss_weights=dict(c["weights"]),
                       learning_rate=1.0, decay_step_size=3)
    e = StepEngine(cfg, S.seeded_vgg_state(0))
    e.set_style_image(S.style_image(1, 96, 80))
    e.overlap_min_pixels = 1 << 30       # a small step: one stream, no split update - the regime programs serve
    e.step_programs = programs
    return e


@pytest.mark.parametrize("name", list(PROGRAM_CASES))
@pytest.mark.parametrize("conv_mode", ["split2", "f32"])
def test_replayed_step_program_equals_the_eager_step(name, conv_mode, monkeypatch):
    """``sm_call_replay`` of a recorded step (runtime/program.py) against the same schedule issued call by call: same
    losses at every step, same texture - over view changes (new list lengths in both view slots), the optimizer's
    changing scalars, StepLR epochs and, for the dip flags, the Gram history's ring position. A third engine runs in
    'verify' mode: every step is recorded again and compared WORD BY WORD with what the program would have issued."""
    require_gpu()
    from stepcmp import assert_same_step, lock
    from stylemesh_amd.runtime import ops
    monkeypatch.setattr(ops, "CONV_MODE", conv_mode)
    monkeypatch.setattr(ops, "GRAM_MODE", conv_mode)
    c = PROGRAM_CASES[name]
    if c.get("levels"):
        views = []
        for k, v in enumerate(_views_multi((0, 2, 6, 7, 9, 11, 12, 14))):
            if k % 3 == 1:       # every third view: all pixels on depth level 2 - levels 0 and 1 are empty
                v = list(v)
                v[5], v[6] = torch.full_like(v[5], 2), torch.full_like(v[6], 2)
                v = tuple(v)
            views.append(v)
    else:
        views = _small_views((0, 2, 6, 7, 9, 11, 12, 14))
    rep = c["rep"]
    n_steps = 40 if rep > 1 else 30
    sched = [views[(i // rep) % len(views)] for i in range(n_steps)]
    a, b, v = _program_engine(c, "1"), _program_engine(c, "0"), _program_engine(c, "verify")
    worst = worst_tex = 0.0
    for i, batch in enumerate(sched):
        if i == n_steps // 2:
            for e in (a, b, v):
                e.end_epoch(); e.end_epoch(); e.end_epoch()           # StepLR: the learning rate drops
        nxt = ([sched[j] for j in (i + 1, i + 2) if j < n_steps] or None) if rep == 1 else None
        if rep > 1 and i % rep == 1 and i - 1 + rep < n_steps:
            for e in (a, b, v):
                e.prepare_view(sched[i - 1 + rep])
        # LOCK-STEP: every step starts from the eager engine's state (free-running engines at lr 1 drift apart chaotically
        # - two eager runs of this schedule differ by 5e-3 in the loss after 40 steps), so each step's own result is compared
        for e in (a, v):
            m0, v0 = lock(e, b)
        la = a.losses(a.training_step(batch, next_batch=nxt))
        lb = b.losses(b.training_step(batch, next_batch=nxt))
        v.training_step(batch, next_batch=nxt)
        worst = max(worst, abs(la["total"] - lb["total"]) / abs(lb["total"]))
        np.testing.assert_allclose(la["total"], lb["total"], rtol=1e-5, err_msg=f"step {i}")
        dev = assert_same_step(a, b, m0, v0, what=f"step {i}")      # through Adam's moments (tests/stepcmp.py)
        assert_same_step(v, b, m0, v0, what=f"verify engine, step {i}")
        worst_tex = max(worst_tex, float(dev[0] / (0.1 * dev[1])))
    print(f"\n[{name} {conv_mode}] lock-step: worst loss deviation replayed vs eager {worst:.2e}, worst gradient deviation "
          f"{worst_tex:.2e} max|g|, replays {a.program_replays}, verified {getattr(v, 'program_verified', 0)}")
    assert a.program_replays >= n_steps // 2, a.program_replays
    assert b.program_replays == 0 and getattr(v, "program_verified", 0) >= n_steps // 3
    assert a.step_count == b.step_count == n_steps


def test_step_program_through_step_compute_and_optimizer_step():
    """The Lightning-shaped caller closes the step itself (``FusedTextureAdam.step`` -> ``optimizer_step``): the program's
    two segments are replayed by the two calls."""
    require_gpu()
    from stepcmp import assert_same_step, lock
    c = PROGRAM_CASES["only2D"]
    views = _small_views((0, 2))
    a, b = _program_engine(c, "1"), _program_engine(c, "0")
    for i in range(12):
        batch = views[i // 6]
        m0, v0 = lock(a, b)             # lock-step (see the test above)
        la = a.step_compute(batch)
        a.optimizer_step()
        lb = b.step_compute(batch)
        b.optimizer_step()
        np.testing.assert_allclose(a.losses(la)["total"], b.losses(lb)["total"], rtol=1e-5)
        assert_same_step(a, b, m0, v0, what=f"step {i}")
    assert a.program_replays >= 6


@pytest.mark.parametrize("depth", [True, False])
def test_level_masks_and_weights_match_the_oracle(depth):
    """The per-view mask / weight kernels against the ORACLE's restatement of model/model.py:188-254 (since
    round 3 the class mirror and the fused engine share these kernels, so their agreement proves nothing about them).
    ``M`` must equal the oracle's level mask exactly; ``pixel_weight`` = (depth-interpolation weight) x (bilinear angle
    guidance) within fp32 rounding."""
    require_gpu()
    import torch.nn.functional as F
    import stylemesh_oracle as O
    from stylemesh_amd.data import synthetic as S
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    level_hw = ((24, 32), (36, 48), (48, 64))
    room = S.BoxRoom((12.0, 9.0, 3.0))
    cfg = EngineConfig(tex_w=256, tex_h=256, hierarchical=True, n_layers=4, style_weights=[1000., 1000., 10., 10., 1000.],
                       angle_threshold=30.0, style_pyramid_mode="multi", use_angle_weight=True, use_depth_scaling=depth,
                       loss_weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3})
    eng = StepEngine(cfg, S.seeded_vgg_state(0))
    eng.set_style_image(S.style_image(1, 96, 80))
    ocfg = O.OracleConfig(hierarchical=True, angle_threshold=30.0, style_pyramid_mode="multi", use_angle_weight=True,
                          use_depth_scaling=depth)
    for seed in (2, 6, 9):
        v = S.make_view(seed, view_hw=(48, 64), level_hw=list(level_hw), level_heights=[h for h, _ in level_hw],
                        min_pyramid_depth=0.25, room=room)
        masks, weights = O.level_masks_and_weights(v, list(level_hw), ocfg)
        dev = tuple([u.cuda() for u in x] if isinstance(x, list) else (x.cuda() if torch.is_tensor(x) and i != 8 else x)
                    for i, x in enumerate(v))
        eng.set_view(dev)
        for lv in eng.view:
            if not hasattr(lv, "M"):
                continue
            i = lv.index
            assert torch.equal(lv.M.cpu(), masks[i][0, 0]), (seed, i)
            ag = F.interpolate(v[11], level_hw[i], mode="bilinear")[0, 0]          # model/model.py:195-199
            want = ag * (weights[i][0, 0] if depth else 1.0)
            np.testing.assert_allclose(lv.pixel_weight.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
            deg = F.interpolate(v[12], level_hw[i], mode="bilinear")[0, 0]
            assert torch.equal(lv.passed.cpu() != 0, deg < 30.0) or float(((lv.passed.cpu() != 0) != (deg < 30.0)).float().mean()) < 1e-3
