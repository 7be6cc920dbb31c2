"""Round 5: resident views (``viewplan.ResidentView``) - the per-view state of a revisited view copied back from HBM
instead of recomputed. Everything ``set_view`` computes is a function of the view alone (reference model/model.py:204-254,
content_and_style_losses.py:146-217, data/abstract_dataset.py:498-512: the same views in every epoch), and the kernels
that compute it are deterministic: the restored buffers must equal the recomputed ones BIT FOR BIT."""
import pytest
import torch

from gpu_util import require_gpu
from test_round4_gpu import _views_multi

pytestmark = pytest.mark.gpu


def _engine(monkeypatch, cache_gb, **cfg_kw):
    from stylemesh_amd.data import synthetic as S
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    monkeypatch.setenv("STYLEMESH_VIEW_CACHE_GB", str(cache_gb))
    kw = dict(tex_w=512, tex_h=512, hierarchical=True, n_layers=4, style_weights=[1000., 1000., 10., 10., 1000.],
              angle_threshold=30.0, style_pyramid_mode="multi", use_angle_weight=True, use_depth_scaling=True,
              loss_weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3}, learning_rate=1.0)
    kw.update(cfg_kw)
    e = StepEngine(EngineConfig(**kw), S.seeded_vgg_state(0))
    e.set_style_image(S.style_image(1, 96, 80))
    return e


def _state(eng):
    """Everything a step reads of the current view's preparation, cloned."""
    out = {}
    for lv in eng.view:
        for k in ("M", "pixel_weight", "passed"):
            t = getattr(lv, k, None)
            if t is not None:
                out[(lv.index, k)] = t.clone()
        if lv.active:
            for layer, m in lv.masks.items():
                out[(lv.index, "mask", layer)] = m.buf.clone()
            for layer, t in getattr(lv, "content_target", {}).items():
                out[(lv.index, "ctarget", layer)] = t.buf.clone()
    out["consts"] = eng.view_consts.clone()
    for key, (lst, frac) in eng.view_tiles.items():
        out[("list", key)] = lst.clone()
        out[("frac", key)] = torch.tensor(frac)
    sp = eng._scatter_plan
    out["scatter_keys"] = sp.bufs[sp.sorted_in][:sp.n_entries].clone()
    out["scatter_vals"] = sp.bufs[2 + sp.sorted_in][:sp.n_entries].clone()
    out["flags"] = eng._view_flags.clone()
    out["sig"] = eng.view_sig
    out["active"] = [lv.active for lv in eng.view]
    return out


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


@pytest.mark.parametrize("cfg_kw", [{}, dict(style_pyramid_mode="single", use_angle_weight=False, use_depth_scaling=False,
                                             angle_threshold=3000.0)])
def test_restored_view_state_equals_the_computed_one(cfg_kw, monkeypatch):
    require_gpu()
    views = _views_multi((2, 6, 9, 0))
    a, b = _engine(monkeypatch, 4, **cfg_kw), _engine(monkeypatch, 0, **cfg_kw)
    first = []
    for v in views:                       # first visits: computed, then kept
        a.set_view(v)
        b.set_view(v)
        torch.cuda.synchronize()
        first.append(_state(a))
        _same(first[-1], _state(b))
    assert a.view_cache_hits == 0 and a.view_cache_misses == len(views) and len(a._resident) == len(views)
    assert b.view_cache_misses == 0 and not b._resident
    for rounds in range(2):               # revisits, in another order: served from HBM
        for i in (2, 0, 3, 1):
            a.set_view(views[i])
            torch.cuda.synchronize()
            _same(first[i], _state(a))
    assert a.view_cache_hits == 8 and a.view_cache_misses == len(views)
    # a step on a restored view runs and gives the loss of the same step on a computed one
    a.set_view(views[1])
    b.set_view(views[1])
    la, lb = a.losses(a.training_step(views[1])), b.losses(b.training_step(views[1]))
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * abs(lb[k]) + 1e-6, (k, la[k], lb[k])


def test_resident_views_under_a_view_change_every_step(monkeypatch):
    """The dip schedule (index_repeat 1, the next view prepared beside the current step): three epochs over five views
    with and without resident views, in LOCK-STEP (tests/stepcmp.py: the engine without resident views hands its state -
    Gram history included - to the other before every step): every step has the same losses and the same gradient up to
    the Gram sums' atomic order, and the same texels moved."""
    require_gpu()
    from stepcmp import assert_same_step, lock
    views = _views_multi((2, 6, 9, 0, 7))
    kw = dict(hierarchical=False, n_layers=1, gram_mode="average", style_pyramid_mode="single", use_angle_weight=False,
              use_depth_scaling=False, angle_threshold=3000.0)
    torch.manual_seed(3)
    a, b = _engine(monkeypatch, 4, **kw), _engine(monkeypatch, 0, **kw)
    sched = [views[i % len(views)] for i in range(3 * len(views))]
    for i, v in enumerate(sched):
        nxt = sched[i + 1] if i + 1 < len(sched) else None
        m0, v0 = lock(a, b)
        la = a.losses(a.training_step(v, new_view=True, next_batch=nxt))
        lb = b.losses(b.training_step(v, new_view=True, next_batch=nxt))
        for k in la:
            assert abs(la[k] - lb[k]) <= 1e-5 * abs(lb[k]) + 1e-6, (i, k, la[k], lb[k])
        assert_same_step(a, b, m0, v0, what=f"step {i}")
    torch.cuda.synchronize()
    assert a.view_cache_hits == 2 * len(views) and b.view_cache_hits == 0
    assert torch.equal(a.arena.p != 0, b.arena.p != 0)


def test_budget_and_configuration_changes(monkeypatch):
    require_gpu()
    views = _views_multi((2, 6, 9))
    eng = _engine(monkeypatch, 4)
    eng.set_view(views[0])
    one = eng._resident_bytes
    assert one > 0
    eng.view_cache_gb = 2.5 * one / 2 ** 30          # room for one more view of this size, not two
    eng.set_view(views[1])
    eng.set_view(views[2])
    # (the budget is checked against what a view WOULD take, before it is allocated - never overshot)
    assert len(eng._resident) == 2 and eng._resident_bytes <= eng.view_cache_gb * 2 ** 30
    eng.set_view(views[2])                           # not kept: computed again, correct
    assert eng.view_tiles is not None
    # the views of ANOTHER scene under the same indices are not these (keyed by the scene's identity)
    hits = eng.view_cache_hits
    eng.set_scene("another scene")
    assert not eng._resident and eng._resident_bytes == 0
    eng.view_cache_gb = 4.0
    eng.set_view(views[0])
    assert eng.view_cache_hits == hits and len(eng._resident) == 1
    eng.set_view(views[1])
    eng.set_view(views[0])
    assert eng.view_cache_hits == hits + 1
    # a view kept under another configuration is forgotten, not restored
    from stylemesh_amd.runtime import ops
    hits = eng.view_cache_hits
    monkeypatch.setattr(ops, "CONV_MODE", "f32")
    monkeypatch.setattr(ops, "GRAM_MODE", "f32")
    eng.set_view(views[0])
    assert eng.view_cache_hits == hits
