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
    # Against the sparse reducer, in LOCK-STEP (the worker hands the dense engine's state to the sparse one before every
    # step; tests/stepcmp.py): every step's reduced gradient is the same up to summation order, and the same texels moved
    from stepcmp import check_deviation
    assert r0["lockstep"].shape[0] >= 6 and torch.equal(r0["lockstep"], r1["lockstep"])    # (identical on both ranks)
    for k, dev in enumerate(r0["lockstep"]):
        check_deviation(dev, r0["lr"], what=f"dense vs sparse reducer, step {k}")
    a, b = r0["dense"]["p"], r0["sparse"]["p"]
    assert torch.equal(a != 0, b != 0)


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
    from stepcmp import assert_same_step, lock
    views = [_small_view(s) for s in MULTIVIEW_SEEDS[:3]]
    a, b = _engine(), _engine()
    b.prepare_ahead = False
    sched = [views[k // 4 % 3] for k in range(20)]            # view changes every 4 steps, views come back
    used = 0
    for k, v in enumerate(sched):
        m0, v0 = lock(b, a)
        if k % 4 == 1 and k + 3 < len(sched):
            used += bool(a.prepare_view(sched[k + 3]))
        la = a.losses(a.training_step(v))
        lb = b.losses(b.training_step(v))
        np.testing.assert_allclose(la["total"], lb["total"], rtol=1e-5)
        assert_same_step(b, a, m0, v0, what=f"step {k}")
        assert torch.equal(a.touched != 0, b.touched != 0)
    assert used >= 3 and a._slot in range(a.N_SLOTS) and not a._prepared
    # a prepared view that is NOT the next one is dropped and the asked-for view is built normally
    a.prepare_view(views[0])
    a.training_step(views[1])
    assert a.view_key == MULTIVIEW_SEEDS[1] or a.view_key is not None


def _pair_list(ops, hip, need_pooled_list, full_hw_list, group):
    """Segment-pair lists (``sm_cover_segments`` pair mode) of several problems, each padded to whole tiles."""
    parts, covers = [], []
    for g, (nd, (H, W)) in enumerate(zip(need_pooled_list, full_hw_list)):
        cap = 2 * nd.shape[0] * ((nd.shape[1] + 15) // 16 + 1) + 2
        starts = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
        count = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.cover_segments([(nd, starts, count, g, W)])
        n = int(count)
        assert 0 <= n <= cap and n % 2 == 0 and (n > 0 or float(nd.sum()) == 0)
        st = starts[:n]
        pad = (-n) % group
        parts.append(torch.cat([st, torch.full((pad,), (g << 24) | 0xFFFFFF, dtype=torch.int32, device="cuda")]))
        covers.append(st.cpu().numpy())
    return torch.cat(parts), covers


def test_cover_segments_pair_mode():
    """Pair mode: every needed window of the pooled map lies in exactly one run of 16 windows; a run = the segment of
    image row 2Y starting at an even column + the segment right below it."""
    require_gpu()
    from stylemesh_amd.runtime import hip, ops
    g = torch.Generator().manual_seed(5)
    for (H, W) in [(37, 50), (64, 85), (16, 21), (256, 341)]:
        Ho, Wo = H // 2, W // 2
        nd = (torch.rand(Ho, Wo, generator=g) < 0.15).float()
        nd[Ho // 2, :] = 1.0            # a full row
        nd[:, Wo - 1] = 1.0             # the last window of every row
        (lst, covers) = _pair_list(ops, hip, [nd.cuda()], [(H, W)], 4)
        st = covers[0] & 0xFFFFFF
        Wp = hip.row_stride(W)
        top, bot = st[0::2], st[1::2]
        assert (bot == top + Wp).all()
        y, x = top // Wp - 1, top % Wp - 1
        assert (y % 2 == 0).all() and (x % 2 == 0).all() and (y >= 0).all() and (y + 1 < 2 * Ho).all() and (x >= 0).all()
        covered = np.zeros((Ho, Wo), np.int32)
        for yy, xx in zip(y // 2, x // 2):
            covered[yy, xx:min(xx + 16, Wo)] += 1
        assert covered.max() == 1                              # disjoint
        assert (covered[nd.numpy() > 0] == 1).all()            # complete


@pytest.mark.parametrize("C,cout,hws", [(64, 64, [(37, 50)]), (64, 64, [(150, 201), (64, 85)]), (128, 128, [(40, 53), (33, 47)]),
                                        (256, 256, [(21, 30)]), (512, 512, [(12, 17), (16, 21)]), (64, 64, [(256, 341)]),
                                        (64, 64, [(2, 2), (5, 3)]), (128, 128, [(3, 5), (40, 53)])])
def test_conv_pooling_epilogue_matches_conv_then_pool(C, cout, hws, monkeypatch):
    """EPI_POOL (the forward conv below a pool stores the pooled map + argmax codes instead of its output) against the
    same conv with the same segment-pair list followed by the pool pass: bit-identical pooled values, codes and bound,
    whole tiles and K-split tail tiles, several levels in one launch, odd sizes."""
    require_gpu()
    import torch.nn.functional as F
    from stylemesh_amd.runtime import hip, ops
    from stylemesh_amd.runtime.fmap import FMap
    monkeypatch.setattr(ops, "CONV_MODE", "split2")
    torch.manual_seed(C + len(hws))
    wgt = torch.randn(cout, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    b = (torch.randn(cout) * 0.3).cuda()
    w = ops.pack_conv_fwd(wgt).cuda()
    w2 = ops.pack_conv_split2(w)
    _, group = ops.conv_list_format(C, cout)
    xs, needs = [], []
    for (H, W) in hws:
        x = F.relu(torch.randn(C, H, W) * 2)
        x[:, : H // 3] = 0            # whole windows of zeros (code 4) and exact ties
        xs.append(x)
        nd = torch.zeros(H // 2, W // 2)
        nd[: max(1, H // 3), :] = 1
        nd[:, W // 4:] = 1
        if (H, W) == (64, 85):
            nd[:] = 0                 # a level without a single needed window: no list entries, nothing written
        needs.append(nd.cuda())
    lst, _ = _pair_list(ops, hip, needs, hws, group)
    amax_in = ops.new_amax("cuda", max(float(x.abs().max()) for x in xs))
    ins = [FMap(C, H, W).from_dense(x.cuda()) for x, (H, W) in zip(xs, hws)]
    # reference: conv with the same list, then the pool pass over the whole plane
    outs = [FMap(cout, H, W) for (H, W) in hws]
    ops.conv3x3_grouped([(i, o, None) for i, o in zip(ins, outs)], w, b, hip.EPI_BIAS_RELU, lst, 1.0, w2, amax_in,
                        ops.new_amax("cuda"))
    pooled_ref = [FMap(cout, H // 2, W // 2) for (H, W) in hws]
    codes_ref = [torch.zeros(cout // 8 * p.plane, dtype=torch.int32, device="cuda") for p in pooled_ref]
    ops.maxpool_fwd_grouped(list(zip(outs, pooled_ref)), None, codes_ref)
    # fused
    outs2 = [FMap(cout, H, W) for (H, W) in hws]
    pooled = [FMap(cout, H // 2, W // 2) for (H, W) in hws]
    codes = [torch.zeros(cout // 8 * p.plane, dtype=torch.int32, device="cuda") for p in pooled]
    amax_out = ops.new_amax("cuda")
    for rep in range(2):   # (the second launch into the same buffers: nothing depends on their previous content)
        ops.conv3x3_grouped([(i, o, None, None, p, c) for i, o, p, c in zip(ins, outs2, pooled, codes)], w, b,
                            hip.EPI_BIAS_RELU | hip.EPI_POOL, lst, 1.0, w2, amax_in, amax_out)
    true_max = 0.0
    for nd, p, pr, c, cr, o2 in zip(needs, pooled, pooled_ref, codes, codes_ref, outs2):
        assert float(o2.planes.abs().max()) == 0.0            # the full-resolution output is not written
        assert p.border_is_zero()
        m = nd > 0
        got, ref = p.to_dense(), pr.to_dense()
        assert torch.equal(got[:, m], ref[:, m])
        cg = c.view(cout // 8, -1)[:, :(p.H + 2) * p.Wp].view(cout // 8, p.H + 2, p.Wp)[:, 1:p.H + 1, 1:p.W + 1]
        cf = cr.view(cout // 8, -1)[:, :(p.H + 2) * p.Wp].view(cout // 8, p.H + 2, p.Wp)[:, 1:p.H + 1, 1:p.W + 1]
        assert torch.equal(cg[:, m], cf[:, m])
        if bool(m.any()) and p.H * p.W > 4:
            assert int(((cg[:, m] & 0xF) == 4).sum()) > 0      # closed windows occur
        if not bool(m.any()):
            assert float(p.planes.abs().max()) == 0.0          # nothing of an empty level is written
        true_max = max(true_max, float(got.abs().max()))
    assert float(amax_out.max()) == true_max


def test_engine_step_with_and_without_pooling_epilogue(monkeypatch):
    """The step with the pools taken in the conv epilogues (default) against the same step with separate pool passes: the
    same losses and the same GRADIENT after one forward + backward from the same random texture, up to the summation order
    of differently composed tiles (tests/stepcmp.py: measured 2.4e-7 max|g|, the distance of either form from itself)."""
    require_gpu()
    from stepcmp import assert_same_pass, one_pass
    res = {}
    for fuse in ("1", "0", "1"):
        monkeypatch.setenv("STYLEMESH_FUSE_POOL_FWD", fuse)
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        eng = _engine(random_init=True)
        res.setdefault(fuse, []).append(one_pass(eng, _small_view(MULTIVIEW_SEEDS[0])))
        keys = set(eng.view_tiles)
        assert (("conv1_2", "fp") in keys) == (fuse == "1") and (("conv1_2", "f") in keys) == (fuse == "0")
    self_d = assert_same_pass(res["1"][1], res["1"][0], what="fused, run twice")
    cross_d = assert_same_pass(res["1"][0], res["0"][0], what="fused vs separate pool passes")
    print(f"\n[pool epilogue] max|dg| / max|g|: form vs itself {self_d:.2e}, fused vs separate {cross_d:.2e}")


@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("hws,two_masks", [([(37, 50)], True), ([(150, 201), (64, 85)], True), ([(40, 53)], False)])
def test_conv_gram_epilogue_matches_gram_backward_then_add(hws, two_masks, C, monkeypatch):
    """EPI_GRAM (the data-gradient conv that produces a 64- / 128-channel style layer's gradient adds that layer's masked
    Gram backward in its epilogue) against the two-launch form - Gram backward into the gradient plane, then the conv with
    EPI_ADD: the same bits (same operand images, scales, fp16 pairs and product order), one or two masks, several levels,
    with and without an active-segment list."""
    require_gpu()
    import torch.nn.functional as F
    from stylemesh_amd.runtime import hip, ops
    from stylemesh_amd.runtime.fmap import FMap
    monkeypatch.setattr(ops, "CONV_MODE", "split2")
    monkeypatch.setattr(ops, "GRAM_MODE", "split2")
    torch.manual_seed(len(hws) * 7 + two_masks + C)
    wgt = torch.randn(C, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    wd = ops.pack_conv_dgrad(wgt).cuda()
    wd2 = ops.pack_conv_split2(wd)
    D0 = (torch.randn(C, C) * 3e-3).cuda()
    D1 = (torch.randn(C, C) * 1e-3).cuda() if two_masks else None
    feats, masks, dps, codes, covers = [], [], [], [], []
    for g, (H, W) in enumerate(hws):
        f = F.relu(torch.randn(C, H, W) * 2)
        feats.append(FMap(C, H, W).from_dense(f.cuda()))
        mk = torch.zeros(2, H, W)
        sel = torch.rand(H, W)
        mk[0] = (sel < 0.3).float()
        mk[1] = ((sel >= 0.3) & (sel < 0.45)).float()
        mk[:, H // 2:, : W // 3] = 0                 # a region neither mask reaches
        masks.append(FMap(2, H, W).from_dense(mk.cuda()))
        act = F.relu(torch.randn(C, H, W))
        a, pooled = FMap(C, H, W).from_dense(act.cuda()), FMap(C, H // 2, W // 2)
        code = torch.zeros(C // 8 * pooled.plane, dtype=torch.int32, device="cuda")
        ops.maxpool_fwd_grouped([(a, pooled)], None, [code])
        codes.append(code)
        dps.append(FMap(C, H // 2, W // 2).from_dense((torch.randn(C, H // 2, W // 2) * 1e-4).cuda()))
    af = ops.new_amax("cuda", max(float(f.planes.abs().max()) for f in feats))
    ad = ops.new_amax("cuda", max(float(D0.abs().max()), float(D1.abs().max()) if two_masks else 0.0))
    amax_in = ops.new_amax("cuda", max(float(d.planes.abs().max()) for d in dps))

    def mptr(m, k):
        return m.channel_ptr(k)
    for use_list in (False, True):
        lst = None
        if use_list:   # free-start segments over a need map that leaves holes
            _, group = ops.conv_list_format(C, C)
            parts = []
            for g, (H, W) in enumerate(hws):
                nd = (torch.rand(H, W) < 0.4).float().cuda()
                cap = H * hip.row_stride(W) // 32 + 2
                starts = torch.empty(cap, dtype=torch.int32, device="cuda")
                count = torch.zeros(1, dtype=torch.int32, device="cuda")
                ops.cover_segments([(nd, starts, count, g)])
                n = int(count)
                parts.append(torch.cat([starts[:n], torch.full(((-n) % group,), (g << 24) | 0xFFFFFF, dtype=torch.int32,
                                                                  device="cuda")]))
            lst = torch.cat(parts)
        # two launches: Gram backward into the gradient plane, then the conv adds it (no K-split: same sums)
        ref = [FMap(C, H, W) for (H, W) in hws]
        ws1 = [torch.empty(ops.gram_backward_ws_bytes(C), dtype=torch.uint8, device="cuda") for _ in hws]
        ops.gram_backward_grouped(ops.struct_array(hip.GramBwdProblem, [
            ops.gram_bwd_problem(f, mptr(m, 0), mptr(m, 1) if two_masks else None, D0, D1, r, w_, af, ad, relu_gate=False)
            for f, m, r, w_ in zip(feats, masks, ref, ws1)]))
        assert max(float(r.planes.abs().max()) for r in ref) > 0
        tiny = torch.zeros(4, device="cuda")
        with monkeypatch.context() as mp:
            mp.setattr(ops, "splitk_workspace", lambda device: tiny)
            ops.conv3x3_grouped([(d, r, f, c) for d, r, f, c in zip(dps, ref, feats, codes)], wd, None,
                                hip.EPI_RELU_MASK | hip.EPI_ADD, lst, 1.0, wd2, amax_in, ops.new_amax("cuda"))
        # one launch
        out = [FMap(C, H, W) for (H, W) in hws]
        ws2 = [torch.empty(ops.gram_backward_ws_bytes(C), dtype=torch.uint8, device="cuda") for _ in hws]
        ops.gram_backward_grouped(ops.struct_array(hip.GramBwdProblem, [
            ops.gram_bwd_problem(f, mptr(m, 0), mptr(m, 1) if two_masks else None, D0, D1, None, w_, af, ad, relu_gate=False)
            for f, m, w_ in zip(feats, masks, ws2)]))
        amax_out = ops.new_amax("cuda")
        ops.conv3x3_grouped([(d, o, f, c, None, None, (w_, mptr(m, 0), mptr(m, 1) if two_masks else None, af, ad))
                             for d, o, f, c, w_, m in zip(dps, out, feats, codes, ws2, masks)], wd, None,
                            hip.EPI_RELU_MASK | hip.EPI_GRAM, lst, 1.0, wd2, amax_in, amax_out)
        for o, r in zip(out, ref):
            if use_list:   # (the reference plane keeps the bare Gram term where the list has no segment)
                written = o.planes != 0
                assert torch.equal(o.planes[written], r.planes[written]) and int(written.sum()) > 0
            else:
                assert torch.equal(o.planes, r.planes)
            assert o.border_is_zero()
        assert float(amax_out.max()) == max(float(o.planes.abs().max()) for o in out)


def test_engine_step_with_and_without_gram_epilogue(monkeypatch):
    """The step with the Gram backward of relu1_1 / relu2_1 inside the data-gradient launches of conv1_2 / conv2_2 against
    the separate Gram-backward launches: the same losses and the same gradient after one forward + backward from the same
    texture, up to the K-split of tail tiles (the fused launch runs whole tiles) - tests/stepcmp.py."""
    require_gpu()
    from stepcmp import assert_same_pass, one_pass
    res = {}
    for fuse in ("1", "0", "1"):
        monkeypatch.setenv("STYLEMESH_FUSE_GRAM_BWD", "r11,r21" if fuse == "1" else "0")
        monkeypatch.setenv("STYLEMESH_OVERLAP_MIN_PIXELS", "0")     # the small test view takes the side-stream path
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        eng = _engine(random_init=True)
        res.setdefault(fuse, []).append(one_pass(eng, _small_view(MULTIVIEW_SEEDS[0])))
        assert set(eng._gram_fused) == ({"r11", "r21"} if fuse == "1" else set())
    self_d = assert_same_pass(res["1"][1], res["1"][0], what="fused, run twice")
    cross_d = assert_same_pass(res["1"][0], res["0"][0], what="Gram epilogue vs separate launches")
    print(f"\n[gram epilogue] max|dg| / max|g|: form vs itself {self_d:.2e}, fused vs separate {cross_d:.2e}")
