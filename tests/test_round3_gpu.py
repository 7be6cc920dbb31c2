"""Round-3 GPU tests: ADVICE r2 fixes (dense reducer + ever-touched update, early-update race, graph keys) and the
round-3 kernels."""
import os
import tempfile

import numpy as np
import pytest
import torch

from conftest import REPO
from golden_cases import (FLAGSETS, LOSS_WEIGHTS, MULTIVIEW_SEEDS, SMALL_LEVEL_HW, SMALL_ROOM, SMALL_VIEW_HW, STYLE_HW,
                          STYLE_SEED, STYLE_WEIGHTS, TEX, VGG_SEED)
from gpu_util import require_gpu
from stylemesh_amd.data import synthetic as S
from test_round2_gpu import _launch_ranks

pytestmark = pytest.mark.gpu


def _engine(random_init=False, **kw):
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    cfgd = FLAGSETS["with_angle_and_depth"]
    cfg = EngineConfig(tex_w=TEX, tex_h=TEX, hierarchical=True, n_layers=4, style_weights=STYLE_WEIGHTS,
                       angle_threshold=cfgd["thr"], style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"],
                       use_angle_weight=True, use_depth_scaling=True, loss_weights=dict(LOSS_WEIGHTS),
                       learning_rate=1, decay_gamma=0.1, decay_step_size=1, **kw)
    eng = StepEngine(cfg, S.seeded_vgg_state(VGG_SEED), random_init=random_init)
    eng.set_style_image(S.style_image(STYLE_SEED, *STYLE_HW))
    return eng


def _small_view(seed):
    return S.make_view(seed, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                       min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))


def test_two_rank_dense_reducer_on_zero_initialised_texture():
    """with a reducer that cannot union the ranks' footprints (``make_grad_reducer``) the engine must
    leave the ever-touched sparse update - otherwise the other rank's gradients sit in chunks the update skips, are
    never applied NOR zeroed, and the ranks' textures diverge. Both ranks end with identical textures, a zeroed
    gradient arena, and the same textures as with the sparse reducer."""
    require_gpu()
    with tempfile.TemporaryDirectory() as tmp:
        r = _launch_ranks([os.path.join(REPO, "tests", "two_rank_worker.py"), "dense_vs_sparse", tmp], 2,
                          {"STYLEMESH_TEST_BACKEND": "gloo"}, timeout=1200)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        r0, r1 = (torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in (0, 1))
    for kind in ("dense", "sparse"):
        assert torch.equal(r0[kind]["p"], r1[kind]["p"]) and torch.equal(r0[kind]["m"], r1[kind]["m"]), kind
        assert float(r0[kind]["g"].abs().max()) == 0.0 and float(r1[kind]["g"].abs().max()) == 0.0, kind
        assert float(r0[kind]["p"].abs().max()) > 0
    assert r0["dense"]["dense_update"] and not r0["sparse"]["dense_update"]
    # Against the sparse reducer's run: the same texels moved (the support of p is the union of all views' footprints on
    # both), most of them to the same values. (Not texel by texel: two RUNS differ by the atomic order of the Gram sums,
    # which the exact pool ties of a zero texture and Adam at lr 1 amplify - test_graph_replay_equals_eager.)
    a, b = r0["dense"]["p"], r0["sparse"]["p"]
    assert torch.equal(a != 0, b != 0)
    assert float(((a - b).abs() > 1e-3).float().mean()) < 0.5 and float((a - b).abs().median()) <= 1e-3


def test_split_update_view_flags_cover_every_sampled_texel():
    """the early half of the split update rewrites p in the ever-touched chunks OUTSIDE the view's
    flags while the forward pass samples the texture - so the view's flags must cover every texel the forward SAMPLES
    (pixels whose backward weight is zero included), not only those that receive a gradient."""
    require_gpu()
    from stylemesh_amd.runtime import ops
    eng = _engine()
    view = _small_view(MULTIVIEW_SEEDS[0])
    eng.set_view(view)
    weighted = eng.touch_flags(eng.touched_log2)
    sampled = torch.zeros_like(weighted)
    for lv in eng.view:
        if lv.active:
            assert lv.pixel_weight is not None and float((lv.pixel_weight == 0).float().mean()) > 0
            ops.tex_touch_flags(eng.grads, eng.arena.g, lv.grid, None, sampled, eng.touched_log2)
    assert torch.equal(eng._view_flags != 0, sampled != 0)
    assert bool(((weighted != 0) & (sampled == 0)).sum() == 0)   # (a superset; on a 4096^2 texture a strict one)
    # (exactness of the split update over these flags: test_round2_gpu.py::test_split_update_equals_dense)
    # every pixel of an active level is sampled: the flags cover the support of a scatter of ones without weights
    cover = torch.zeros_like(eng.arena.g)
    for lv in eng.view:
        if lv.active:
            b = eng._level_bufs(lv.H, lv.W)
            ones = type(b.grad["img"])(3, lv.H, lv.W).from_dense(torch.ones(3, lv.H, lv.W))
            ops.tex_sample_bwd(eng.arena.views(cover), lv.grid, ones, None)
    hit = (cover != 0)
    pad = (-hit.numel()) % (1 << eng.touched_log2)
    hit_chunks = torch.nn.functional.pad(hit, (0, pad)).view(-1, 1 << eng.touched_log2).any(1)
    assert bool((hit_chunks & (eng._view_flags == 0)).sum() == 0)


def test_optimizer_graph_is_recaptured_when_the_flags_change():
    """the captured update bakes in the ever-touched flags pointer and the sparse / dense choice. After
    ``load_texture`` (dense update from then on) the replayed graph must update EVERY texel - also those in chunks the
    old flags excluded (the regulariser pulls them towards zero)."""
    require_gpu()
    eng = _engine()
    eng.use_graphs = True
    view = _small_view(MULTIVIEW_SEEDS[0])
    for k in range(3):
        eng.training_step(view)
    key0 = eng._opt_graph[0]
    assert key0[1] == eng.touched.data_ptr()
    untouched = (eng.touched == 0).repeat_interleave(1 << eng.touched_log2)[:eng.arena.n].clone()
    assert bool(untouched.any())
    eng.load_texture([torch.full_like(l, 5.0) for l in eng.layers])
    before = eng.arena.p.clone()
    for k in range(3):
        eng.training_step(view)
    torch.cuda.synchronize()
    assert eng._opt_graph[0] != key0 and eng._opt_graph[0][1] is None
    n0 = eng.arena.seg_end[0]   # layer 0 has a non-zero regulariser weight
    moved = (eng.arena.p != before)[:n0][untouched[:n0]]
    assert bool(moved.all())


def test_more_than_eight_uv_levels_are_rejected():
    require_gpu()
    eng = _engine()
    v = list(_small_view(MULTIVIEW_SEEDS[0]))
    v[9] = [v[9][0]] * 9
    with pytest.raises(ValueError, match="UV levels"):
        eng.set_view(tuple(v))
