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


@pytest.mark.parametrize("n,p", [(1_044_480, 0.2), (1000, 0.5), (1025, 0.0), (70_000, 1.0), (3, 0.7)])
def test_flags_compact_matches_nonzero(n, p):
    """``sm_flags_compact``: the ascending index list of the flagged chunks and its length, without a host sync."""
    require_gpu()
    from stylemesh_amd.runtime import ops
    g = torch.Generator(device="cuda").manual_seed(n)
    flags = (torch.rand(n, device="cuda", generator=g) < p).to(torch.int32) * 7     # any non-zero value flags a chunk
    idx = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    cnt = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    ws = torch.empty(ops.flags_compact_ws_ints(n), dtype=torch.int32, device="cuda")
    ops.flags_compact(flags, idx, cnt, ws)
    want = flags.nonzero().flatten().to(torch.int32)
    assert int(cnt) == want.numel()
    assert torch.equal(idx[:want.numel()], want)
    assert bool((idx[want.numel():] == -1).all())     # nothing written behind the list


@pytest.mark.parametrize("chunk_log2", [2, 4, 6, 10])
def test_chunks_gather_scatter_round_trip(chunk_log2):
    require_gpu()
    from stylemesh_amd.runtime import ops
    chunk = 1 << chunk_log2
    n_chunks = 5000
    arena = torch.randn(n_chunks * chunk, device="cuda")
    idx = torch.randperm(n_chunks, device="cuda")[:1777].sort().values.to(torch.int32)
    buf = torch.full((idx.numel() * chunk + 64,), 3.0, device="cuda")
    ops.chunks_gather(arena, idx, idx.numel(), chunk_log2, buf)
    want = arena.view(-1, chunk).index_select(0, idx.long()).reshape(-1)
    assert torch.equal(buf[:want.numel()], want) and bool((buf[want.numel():] == 3.0).all())
    # device-side count: only the first *count chunks move, the launch is sized for the capacity
    cnt = torch.tensor([1000], dtype=torch.int32, device="cuda")
    buf2 = torch.zeros_like(buf)
    ops.chunks_gather(arena, idx, idx.numel(), chunk_log2, buf2, n_idx_dev=cnt)
    assert torch.equal(buf2[:1000 * chunk], want[:1000 * chunk]) and float(buf2[1000 * chunk:].abs().max()) == 0.0
    out = torch.zeros_like(arena)
    ops.chunks_scatter(out, idx, idx.numel(), chunk_log2, buf, scale=0.5)
    ref = torch.zeros_like(arena)
    ref.view(-1, chunk).index_copy_(0, idx.long(), 0.5 * want.view(-1, chunk))
    assert torch.equal(out, ref)


def test_sparse_reducer_on_the_device_matches_dense_sum():
    """``SparseGradReducer`` with CUDA tensors (product kernels for the compaction / gather / scatter) against the plain
    sums, one rank over a stand-in 'communicator' that doubles what it is given (= two ranks holding the same data)."""
    require_gpu()
    from stylemesh_amd.runtime.distributed import SparseGradReducer

    class Doubler:
        class ReduceOp:
            SUM, MAX = "sum", "max"

        @staticmethod
        def all_reduce(t, op=None, async_op=False):
            if op == "sum":
                t.mul_(2.0)

            class Done:
                def wait(self):
                    pass
            return Done() if async_op else None
    red = SparseGradReducer(Doubler, 2, chunk_log2=6)
    n_chunks = 4096
    flags = (torch.rand(n_chunks, device="cuda") < 0.3).to(torch.int32)
    g = torch.randn(n_chunks * 64, device="cuda") * flags.repeat_interleave(64)
    want = 2.0 * g
    cnt = red.new_view_begin(flags.clone())
    red.new_view_end(int(cnt))
    assert red.n_idx == int(flags.sum()) and abs(red.fraction - red.n_idx / n_chunks) < 1e-9
    red(g)
    assert torch.equal(g, want) and red.last_bytes == red.n_idx * 256
    # the pipelined form: ranges tile the arena in order
    g2 = torch.randn(n_chunks * 64, device="cuda") * flags.repeat_interleave(64)
    want2 = 2.0 * g2
    ranges = []
    red.pipelined(g2, lambda lo, hi: ranges.append((lo, hi)))
    assert torch.equal(g2, want2) and ranges[0][0] == 0 and ranges[-1][1] == g2.numel()
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and len(ranges) == 4


def test_prepare_view_ahead_equals_set_view():
    """``StepEngine.prepare_view``: the next view's constants computed on a side stream into the other buffer slot while
    the current view's steps run; the swap at the view change gives the same training trajectory as computing them at
    the view change. Lock-step (state copied before every step: two RUNS differ by the atomic order of the Gram sums)."""
    require_gpu()
    views = [_small_view(s) for s in MULTIVIEW_SEEDS[:3]]
    a, b = _engine(), _engine()
    b.prepare_ahead = False
    sched = [views[k // 4 % 3] for k in range(20)]            # view changes every 4 steps, views come back
    used = 0
    for k, v in enumerate(sched):
        for name in ("p", "m", "v"):
            getattr(b.arena, name).copy_(getattr(a.arena, name))
        b.sumsq.copy_(a.sumsq)
        if b.touched is not None:
            b.touched.copy_(a.touched)
        if k % 4 == 1 and k + 3 < len(sched):
            used += bool(a.prepare_view(sched[k + 3]))
        la = a.losses(a.training_step(v))
        lb = b.losses(b.training_step(v))
        np.testing.assert_allclose(la["total"], lb["total"], rtol=1e-5)
        err = (a.arena.p - b.arena.p).abs()
        assert float((err > 1e-4).float().mean()) < 5e-3, (k, float(err.max()))
        assert torch.equal(a.touched != 0, b.touched != 0)
    assert used >= 3 and a._slot in (0, 1) and a._prepared is None
    # a prepared view that is NOT the next one is dropped and the asked-for view is built normally
    a.prepare_view(views[0])
    a.training_step(views[1])
    assert a.view_key == MULTIVIEW_SEEDS[1] or a.view_key is not None
