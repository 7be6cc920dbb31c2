"""End-to-end parity of the GPU step (StepEngine over libstylemesh_hip.so) against the committed goldens that
the reference itself produced (tests/golden/g5_*, g6_*, g8_*) and against the oracle on the same seeded inputs.

Stated fp32 tolerances
* losses: rtol 2e-4.
* texture gradients: |err| <= 1e-3*|ref| + 2e-4*max|ref| on >= 97 % of the texels and <= 2e-2*max|ref| everywhere
  (at these tiny test sizes, 40x56 .. 64x88 pixel views, one flipped window's receptive field is a few % of the
  image; the fraction shrinks with the view size).
  The forward activations agree with the reference to ~2e-6 relative (MFMA k-ordered sums vs MKLDNN order), but
  a max-pool window whose two largest activations are closer than that rounding noise can route its gradient
  to the other pixel ("argmax flip"): the gradient then differs inside that pixel's receptive field by a few
  1e-3 of max|g|. test_mismatches_originate_only_at_pool_near_ties checks that this is the ONLY source.
* texture values (range +-150) after k <= 5 Adam steps at lr 1: step 1 exact to 1e-5 (<= 3 sign-flip texels, see
  SURVEY.md section 7.2 hazard); later steps |err| <= 2e-3 on >= 97 % of the texels, <= 2e-2 on >= 99 %, <= 0.3
  everywhere. Adam divides by sqrt(v): a RELATIVE gradient difference d becomes an ABSOLUTE update difference
  ~ lr*d per step, so the 1e-3-level gradient differences above (and the flips) show up at the 1e-3..1e-1 level
  in texels whose gradient is small. The Adam kernel itself is exact to 1e-5 given identical gradients
  (tests/test_kernels_gpu.py::test_adam_fused_matches_oracle)."""
import numpy as np
import pytest
import torch

import stylemesh_oracle as O
from conftest import batch_from_golden, load_golden
from golden_cases import (FLAGSETS, LOSS_WEIGHTS, MULTIVIEW_SEEDS, SMALL_LEVEL_HW, SMALL_ROOM, SMALL_VIEW_HW,
                          STYLE_HW, STYLE_SEED, STYLE_WEIGHTS, TEX, VGG_SEED)
from gpu_util import assert_close, require_gpu
from stylemesh_amd.data import synthetic as S

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def make_engine(cfgd, init=None):
    require_gpu()
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    cfg = EngineConfig(tex_w=TEX, tex_h=TEX, hierarchical=cfgd["hier"], n_layers=4, style_weights=STYLE_WEIGHTS,
                       angle_threshold=cfgd["thr"], style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"],
                       use_angle_weight=cfgd["angle"], use_depth_scaling=cfgd["depth"], loss_weights=dict(LOSS_WEIGHTS),
                       learning_rate=1, decay_gamma=0.1, decay_step_size=1)
    eng = StepEngine(cfg, S.seeded_vgg_state(VGG_SEED))
    if init is not None:
        eng.load_texture(init[:len(eng.layers)])
    eng.set_style_image(S.style_image(STYLE_SEED, *STYLE_HW))
    return eng


def grad_close(mine, ref, what):
    ref = torch.as_tensor(ref)
    mine = mine.detach().cpu()
    mx = float(ref.abs().max())
    err = (mine - ref).abs()
    bad = err > 1e-3 * ref.abs() + 2e-4 * mx
    msg = f"{what}: {int(bad.sum())} / {bad.numel()} beyond tolerance, max err {float(err.max()):.3e} vs max|ref| {mx:.3e}"
    assert float(bad.float().mean()) <= 0.03 and float(err.max()) <= 2e-2 * mx, msg


def texture_close(mine, ref, step, what):
    err = (mine.detach().cpu() - ref).abs()
    n = err.numel()
    if step == 0:
        assert int((err > 1e-5).sum()) <= 3, (what, int((err > 1e-5).sum()))
        return
    f3, f2 = float((err > 2e-3).float().mean()), float((err > 2e-2).float().mean())
    # (the coarsest layer of the test textures has 192 texels: the fractions are floored at a handful of texels)
    assert (f3 <= 0.03 or (err > 2e-3).sum() <= 8) and (f2 <= 0.01 or (err > 2e-2).sum() <= 2) and float(err.max()) <= 0.3, \
        f"{what}: {f3:.4f} of texels beyond 2e-3, {f2:.4f} beyond 2e-2, max {float(err.max()):.3e} (n = {n})"


def full_grads(eng):
    """data-term gradient + analytic regulariser gradient = what autograd gives the reference"""
    return [g + c * p for g, c, p in zip(eng.grads, eng.reg_coef, eng.layers)]


def test_style_targets_match_golden():
    d = load_golden("g4_style")
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    require_gpu()
    eng = StepEngine(EngineConfig(tex_w=64, tex_h=64, loss_weights={"style": 1.0, "content": 1.0}),
                     S.seeded_vgg_state(VGG_SEED))
    eng.set_style_image(S.style_image(int(d["style_seed"]), 600, 520))
    assert eng.style_pyramid_sizes == [tuple(s) for s in d["shapes_600x520"]]
    for li, layer in enumerate(eng.cfg.style_layers):
        for lvl in (0, 1, 2):
            g = eng.targets[li][lvl].cpu()
            if f"target_{layer}_{lvl}" in d.files:
                ref = d[f"target_{layer}_{lvl}"]
                assert_close(g, ref, 2e-4, 2e-5 * float(np.abs(ref).max()), f"{layer}/{lvl}")
            else:
                ref = d[f"target_{layer}_{lvl}_sub"]
                assert_close(g[::5, ::7], ref, 2e-4, 2e-5 * float(np.abs(ref).max()), f"{layer}/{lvl}")
                np.testing.assert_allclose(float(g.double().sum()), float(d[f"target_{layer}_{lvl}_sum"]), rtol=1e-4)


@pytest.mark.parametrize("name", list(FLAGSETS))
def test_forward_backward_matches_reference_golden(name):
    d = load_golden("g5_" + name)
    cfgd = FLAGSETS[name]
    init = [T(d[f"init{i}"]) for i in range(4)]
    eng = make_engine(cfgd, init)
    batch = batch_from_golden(d)
    eng.set_view(batch)
    n_steps = 3 if cfgd["gram"] == "average" else 1
    for s in range(n_steps):
        tag = f"_s{s}" if n_steps > 1 else ""
        eng.arena.g.zero_()
        lt = eng.loss_tensors()
        eng.forward_backward()
        losses = eng.losses(lt)
        for k in ("content", "style", "tex_reg", "total"):
            np.testing.assert_allclose(losses[k], float(d[f"loss_{k}{tag}"].reshape(-1)[0]), rtol=2e-4, err_msg=k)
        for i, g in enumerate(full_grads(eng)):
            grad_close(g, d[f"grad{i}{tag}"], f"{name} grad{i}{tag}")
    # sampled images per level
    for k, lv in enumerate(eng.view):
        if lv.active:
            b = eng._level_bufs(lv.H, lv.W)
            assert_close(b.act["img"].to_dense(3), d[f"pred{k}"][0], 1e-5, 2e-4, f"pred{k}")


@pytest.mark.parametrize("name", ["with_angle_and_depth", "flat_single"])
def test_intermediates_match_oracle(name, monkeypatch):
    """Feature maps, masks, factors and the per-layer gradients against the oracle; mismatches of the
    gradients must originate only at max-pool windows whose top-2 activations are within rounding noise.
    (Run with the two-pass pool backward: the fused form never materialises the pre-pool gradients this test walks
    through; tests/test_kernels_gpu.py::test_conv3x3_split2_fused_pool_backward shows both forms bit-identical.)"""
    from stylemesh_amd.runtime import vgg as _vgg
    monkeypatch.setattr(_vgg, "FUSE_POOL_BWD", False)
    cfgd = FLAGSETS[name]
    d = load_golden("g5_" + name)
    init = [T(d[f"init{i}"]) for i in range(4)]
    eng = make_engine(cfgd, init)
    batch = batch_from_golden(d)
    ocfg = O.OracleConfig(hierarchical=cfgd["hier"], style_weights=STYLE_WEIGHTS, angle_threshold=cfgd["thr"],
                          style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"], use_angle_weight=cfgd["angle"],
                          use_depth_scaling=cfgd["depth"], loss_weights=dict(LOSS_WEIGHTS))
    pipe = O.OraclePipeline(S.seeded_vgg_state(VGG_SEED), S.style_image(STYLE_SEED, *STYLE_HW), ocfg, (TEX, TEX),
                            init_layers=init)
    rec = {}
    pipe.grads(batch, rec)
    # whole planes are compared below, also where the loss cannot see them: every tile is computed (what the active lists
    # leave out - and that the listed part equals the dense computation - is test_sparse_tiles_equal_dense's subject)
    eng.sparse_tiles = False
    eng.set_view(batch)
    eng.forward_backward()
    assert [lv.index for lv in eng.view if lv.active] == rec["active"]
    for a, i in enumerate(rec["active"]):
        lv = eng.view[i]
        b = eng._level_bufs(lv.H, lv.W)
        assert_close(lv.M, rec["masks"][i][0, 0], 0, 0)
        for layer in eng.loss_layers:
            ref = rec["enc"][a][layer].detach()[0]
            assert_close(b.act[layer].to_dense(), ref, 1e-4, 2e-4 * float(ref.abs().max()), f"feat {layer} level {i}")
            np.testing.assert_allclose(float(lv.factor[layer]), float(rec["factors"][a][layer]), rtol=1e-5)
            info = rec["info"][a][layer]
            assert_close(lv.masks[layer].to_dense()[0], info["m"][0, 0], 0, 0)
            # the angle filter compares an interpolated angle with the threshold: allow a pixel or two to flip
            assert (lv.masks[layer].to_dense()[1].cpu() != info["m_pass"][0, 0]).float().mean() < 2e-3
        # per-layer gradients, deepest first. dZ = dL/d(pre-ReLU output) for conv layers, dL/dp for pools.
        clean = True   # no argmax flip met yet on the way down
        order = [n for n in reversed(list(b.grad)) if n != "img"]
        for layer in order:
            t = rec["all_acts"][a][layer]
            ref = (t.grad * (t.detach() > 0))[0] if layer.startswith("r") else t.grad[0]
            mine = b.grad[layer].to_dense().cpu()
            err = (mine - ref).abs()
            n_bad = int((err > 1e-3 * float(ref.abs().max())).sum())
            if clean and n_bad:
                # first mismatch: a handful of elements, every one a rounding-level tie of a non-smooth operator -
                # either a max-pool window (pre-pool layer) whose top-2 activations coincide, or a ReLU gate whose
                # pre-activation is zero to rounding (one side computes +3e-6, the other 0 at max|act| ~ 1e2)
                assert n_bad <= 16, (layer, n_bad)
                act = t.detach()[0]
                mine_act = b.act[layer].to_dense().cpu() if layer.startswith("r") else None
                tie = 2e-5 * float(act.abs().max())
                for c, y, x in torch.nonzero(err > 1e-3 * float(ref.abs().max())):
                    relu_tie = mine_act is not None and max(abs(float(act[c, y, x])), abs(float(mine_act[c, y, x]))) <= tie
                    if relu_tie:
                        continue
                    assert layer in ("r12", "r22", "r34", "r44"), f"level {i}: first gradient mismatch at {layer}"
                    wy, wx = int(y) // 2 * 2, int(x) // 2 * 2
                    top2 = act[c, wy:wy + 2, wx:wx + 2].reshape(-1).topk(2).values
                    assert float(top2[0] - top2[1]) <= tie, (layer, int(c), wy, wx, top2)
                clean = False
            elif clean:
                assert n_bad == 0
        ref = rec["pred_grads_raw"][i][0]
        if clean:
            grad_close(b.grad["img"].to_dense(), ref, f"raw image gradient level {i}")
        else:   # a flipped window: its receptive field (tens of pixels wide at these sizes) carries the difference
            err = (b.grad["img"].to_dense().cpu() - ref).abs()
            assert float(err.max()) <= 0.2 * float(ref.abs().max())
            assert float((err > 1e-3 * float(ref.abs().max())).float().mean()) <= 0.15


@pytest.mark.parametrize("init_name", ["zero", "seeded"])
def test_adam_steps_match_reference_golden(init_name):
    d = load_golden("g6_adam_" + init_name)
    g5 = load_golden("g5_with_angle_and_depth")
    init = [T(g5[f"init{i}"]) for i in range(4)] if init_name == "seeded" else None
    eng = make_engine(FLAGSETS["with_angle_and_depth"], init)
    batch = batch_from_golden(g5)
    for step in range(5):
        lt = eng.training_step(batch)
        np.testing.assert_allclose(eng.losses(lt)["total"], float(d[f"loss_total_step{step}"].reshape(-1)[0]), rtol=5e-4)
        if step % 2 == 1:
            eng.end_epoch()
        if step in (0, 1, 4):
            for i in range(4):
                ref = T(d[f"p{i}_after{step + 1}"]).clamp(O.CLAMP_LO, O.CLAMP_HI)
                texture_close(eng.layers[i], ref, step, f"{init_name} layer {i} after {step + 1} steps")
    assert float(eng.arena.g.abs().max()) == 0.0   # the fused update leaves a zeroed gradient


def test_multiview_mean_gradient_step_matches_golden():
    """The R-GPU step on one GPU: gradients of R views accumulate in the arena, scaled by 1/R in the update."""
    d = load_golden("g8_multiview")
    g5 = load_golden("g5_with_angle_and_depth")
    init = [T(g5[f"init{i}"]) for i in range(4)]
    eng = make_engine(FLAGSETS["with_angle_and_depth"], init)
    for s in MULTIVIEW_SEEDS:
        batch = S.make_view(s, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                            min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))
        eng.set_view(batch)
        lt = eng.loss_tensors()
        eng.forward_backward()
        np.testing.assert_allclose(eng.losses(lt)["total"], float(d[f"loss_total_view{s}"].reshape(-1)[0]), rtol=5e-4)
    R = len(MULTIVIEW_SEEDS)
    for i in range(4):
        grad_close(eng.grads[i] / R + eng.reg_coef[i] * eng.layers[i], d[f"mean_grad{i}"], f"mean grad {i}")
    eng.optimizer_step(world_size=R)
    for i in range(4):
        ref = T(d[f"p{i}_after"]).clamp(O.CLAMP_LO, O.CLAMP_HI)
        texture_close(eng.layers[i], ref, 1, f"multiview layer {i}")


def test_graph_replay_equals_eager():
    """The captured hipGraph of the step (per-view buffers and tile lists at fixed addresses, Adam scalars through
    device memory) reproduces the eager launches, across a view change. Lock-step: before every step the graph
    engine gets the eager engine's texture and Adam state, so that only that ONE step is compared (over several
    steps Adam at lr 1 amplifies the atomic-order noise of the scatter chaotically)."""
    g5 = load_golden("g5_with_angle_and_depth")
    init = [T(g5[f"init{i}"]) for i in range(4)]
    views = [S.make_view(s, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                         min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM)) for s in (3, 4)]
    eager = make_engine(FLAGSETS["with_angle_and_depth"], init)
    graph = make_engine(FLAGSETS["with_angle_and_depth"], init)
    graph.use_graphs = True
    from stepcmp import assert_same_step, lock
    for step in range(8):
        m0, v0 = lock(graph, eager)
        le = eager.losses(eager.training_step(views[step // 4]))
        lg = graph.losses(graph.training_step(views[step // 4]))
        np.testing.assert_allclose(lg["total"], le["total"], rtol=1e-5)
        assert_same_step(graph, eager, m0, v0, what=f"step {step}")       # (tests/stepcmp.py: through Adam's moments)
    assert len(graph._graphs) >= 1 and graph._opt_graph is not None


def test_sparse_tiles_equal_dense():
    """Skipping the conv tiles that cannot influence the loss leaves losses and texture gradients unchanged
    (up to the K-split summation order of the differently sized grids), also after a view change."""
    g5 = load_golden("g5_with_angle_and_depth")
    init = [T(g5[f"init{i}"]) for i in range(4)]
    big_levels = [(96, 128), (160, 214)]
    views = [S.make_view(s, view_hw=(96, 128), level_hw=big_levels, level_heights=[96, 160], min_pyramid_depth=0.75,
                         room=S.BoxRoom((12.0, 9.0, 3.0))) for s in (0, 2)]
    res = []
    for sparse in (False, True):
        eng = make_engine(FLAGSETS["with_angle_and_depth"], init)
        eng.sparse_tiles = sparse
        per_view = []
        for v in views:
            eng.set_view(v)
            if sparse:
                fr = [f for _, f in eng.view_tiles.values()]
                assert min(fr) < 0.9, "the test views should leave some tiles inactive"
            eng.arena.g.zero_()
            lt = eng.loss_tensors()
            eng.forward_backward()
            per_view.append((eng.losses(lt), [g.clone() for g in eng.grads]))
        res.append(per_view)
    for (ld, gd), (ls, gs) in zip(*res):
        for k in ld:
            np.testing.assert_allclose(ls[k], ld[k], rtol=1e-5)
        for a, b in zip(gs, gd):
            grad_close(a, b.cpu(), "sparse vs dense")


def test_content_only_plumbing_config_matches_oracle():
    """BASELINE config 1: one view, 256^2 texture, content loss only (style weight 0): the VGG pass stops at r42
    and only the content term injects a gradient."""
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    require_gpu()
    lw = {"content": 7e1, "style": 0.0, "tex_reg": 0.0}
    batch = S.make_view(5, view_hw=(64, 85), level_hw=[(64, 85)], level_heights=[64], min_pyramid_depth=0.25,
                        room=S.BoxRoom(SMALL_ROOM))
    vgg = S.seeded_vgg_state(VGG_SEED)
    style = S.style_image(STYLE_SEED, *STYLE_HW)
    for hier in (True, False):
        eng = StepEngine(EngineConfig(tex_w=256, tex_h=256, hierarchical=hier, loss_weights=dict(lw), angle_threshold=3000,
                                      use_angle_weight=False, use_depth_scaling=False, learning_rate=1.0), vgg)
        eng.set_style_image(style)
        assert eng.deepest == "r42"
        pipe = O.OraclePipeline(vgg, style, O.OracleConfig(hierarchical=hier, loss_weights=dict(lw), angle_threshold=3000,
                                                           use_angle_weight=False, use_depth_scaling=False), (256, 256))
        rng = np.random.default_rng(1)
        init = [torch.from_numpy(((S.smooth_noise(rng, 3, 256 >> i, 256 >> i) - 0.5) * 80).astype(np.float32))
                for i in range(len(eng.layers))]
        eng.load_texture(init)
        with torch.no_grad():
            for l, t in zip(pipe.layers, init):
                l.copy_(t)
        ref_losses, ref_grads = pipe.grads(batch)
        eng.set_view(batch)
        lt = eng.loss_tensors()
        eng.forward_backward()
        mine = eng.losses(lt)
        np.testing.assert_allclose(mine["content"], float(ref_losses["content"]), rtol=2e-4)
        assert mine["style"] == 0.0 and mine["tex_reg"] == 0.0
        for g, r in zip(eng.grads, ref_grads):
            grad_close(g, r, f"content-only hier={hier}")


def test_side_stream_style_branches_equal_serial():
    g5 = load_golden("g5_with_angle_and_depth")
    init = [T(g5[f"init{i}"]) for i in range(4)]
    batch = batch_from_golden(g5)
    res = []
    for overlap in (False, True):
        eng = make_engine(FLAGSETS["with_angle_and_depth"], init)
        eng.overlap_style = overlap
        eng.set_view(batch)
        for _ in range(2):
            eng.arena.g.zero_()
            lt = eng.loss_tensors()
            eng.forward_backward()
        torch.cuda.synchronize()
        res.append((eng.losses(lt), [g.clone() for g in eng.grads]))
    for k in res[0][0]:
        np.testing.assert_allclose(res[1][0][k], res[0][0][k], rtol=1e-5)
    for a, b in zip(res[1][1], res[0][1]):
        grad_close(a, b.cpu(), "side-stream vs serial")


def test_view_without_valid_pixels_is_a_noop_for_the_data_term():
    """A view whose mask is empty (camera facing the window / all depth dropped out): every level is filtered out
    (reference model/model.py:256-257), the data-term gradient stays exactly zero, content / style losses are 0 and
    the update is driven by the regulariser alone."""
    g5 = load_golden("g5_with_angle_and_depth")
    init = [T(g5[f"init{i}"]) for i in range(4)]
    batch = list(batch_from_golden(g5))
    batch[9] = [torch.zeros_like(u) for u in batch[9]]          # uv = 0 everywhere -> mask false everywhere
    batch[10] = torch.zeros_like(batch[10])
    eng = make_engine(FLAGSETS["with_angle_and_depth"], init)
    eng.set_view(tuple(batch))
    assert not any(lv.active for lv in eng.view)
    before = [l.clone() for l in eng.layers]
    lt = eng.training_step(tuple(batch))
    losses = eng.losses(lt)
    assert losses["content"] == 0.0 and losses["style"] == 0.0 and losses["tex_reg"] > 0.0
    assert float(eng.arena.g.abs().max()) == 0.0
    moved = [float((a - b).abs().max()) for a, b in zip(eng.layers, before)]
    assert moved[0] > 0.0 and all(torch.isfinite(l).all() for l in eng.layers)   # layer 0 has tex_reg weight 8


def test_level_streams_equal_serial():
    """The UV levels' loss branches on separate HIP streams (default) give the serial result."""
    g5 = load_golden("g5_with_angle_and_depth")
    init = [T(g5[f"init{i}"]) for i in range(4)]
    batch = batch_from_golden(g5)
    res = []
    for concurrent in (False, True):
        eng = make_engine(FLAGSETS["with_angle_and_depth"], init)
        eng.level_streams = concurrent
        eng.set_view(batch)
        assert sum(lv.active for lv in eng.view) > 1
        for _ in range(3):
            eng.arena.g.zero_()
            lt = eng.loss_tensors()
            eng.forward_backward()
        torch.cuda.synchronize()
        res.append((eng.losses(lt), [g.clone() for g in eng.grads]))
    for k in res[0][0]:
        np.testing.assert_allclose(res[1][0][k], res[0][0][k], rtol=1e-5)
    for a, b in zip(res[1][1], res[0][1]):
        grad_close(a, b.cpu(), "level streams vs serial")


def test_full_size_properties():
    """BASELINE sizes (4096^2 hier-4 texture, UV levels 256x341 .. 784x1045, multi + angle + depth), checked through
    size-independent properties: (a) dead-tile elimination does not change losses / gradients, (b) the gradient
    arena accumulates linearly over repeated passes, (c) texels no view pixel maps to keep an exactly-zero gradient
    and, under zero init, stay exactly 0 through the fused update, (d) the update leaves a zeroed gradient and a
    clamped texture, (e) everything is finite."""
    require_gpu()
    from stylemesh_amd.runtime import ops
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    cfg = EngineConfig(tex_w=4096, tex_h=4096, style_weights=STYLE_WEIGHTS, angle_threshold=30, style_pyramid_mode="multi",
                       loss_weights=dict(LOSS_WEIGHTS), learning_rate=1.0, decay_step_size=3)
    eng = StepEngine(cfg, S.seeded_vgg_state(0))
    eng.set_style_image(S.style_image(1, 600, 520))
    view = S.make_view(2, level_hw=S.SCANNET_LEVEL_HW, room=S.BoxRoom((12.0, 9.0, 3.0)))
    # a non-constant texture: with the all-zero initial texture every max-pool window is an exact 4-way tie, whose
    # winner then depends on 1-ulp summation-order differences between tiles
    rng = np.random.default_rng(0)
    tex0 = [torch.from_numpy(((S.smooth_noise(rng, 3, 4096 >> i, 4096 >> i, cells=64) - 0.5) * 60).astype(np.float32))
            for i in range(4)]
    grads = {}
    eng.arena.g.zero_()
    eng.load_texture(tex0)
    grads = {}
    for sparse in (True, False):
        eng.sparse_tiles = sparse
        eng.set_view(view)
        assert [lv.index for lv in eng.view if lv.active] == [0, 1, 2, 3]
        eng.arena.g.zero_()
        lt = eng.loss_tensors()
        eng.forward_backward()
        grads[sparse] = (eng.losses(lt), eng.arena.g.clone())
    for k in grads[True][0]:
        np.testing.assert_allclose(grads[True][0][k], grads[False][0][k], rtol=1e-5)
    g_s, g_d = grads[True][1], grads[False][1]
    assert torch.isfinite(g_d).all() and float(g_d.abs().max()) > 0
    err = (g_s - g_d).abs()
    mx = float(g_d.abs().max())
    assert float((err > 1e-3 * g_d.abs() + 2e-4 * mx).float().mean()) < 1e-3 and float(err.max()) <= 2e-2 * mx
    # (b) a second pass doubles the accumulated gradient
    eng.forward_backward()
    err2 = (eng.arena.g - 2 * g_d).abs()
    assert float((err2 > 2e-3 * g_d.abs() + 4e-4 * mx).float().mean()) < 1e-3
    # (c) coverage: scatter a gradient image of ones -> exactly the touched texels are non-zero
    cover = torch.zeros_like(eng.arena.g)
    cover_layers = eng.arena.views(cover)
    for lv in eng.view:
        b = eng._level_bufs(lv.H, lv.W)
        ones = type(b.grad["img"])(3, lv.H, lv.W).from_dense(torch.ones(3, lv.H, lv.W))
        ops.tex_sample_bwd(cover_layers, lv.grid, ones, None)
    untouched = cover == 0
    assert 0.5 < float(untouched.float().mean()) < 1.0          # one view covers a small part of the texture
    assert float(g_d[untouched].abs().max()) == 0.0
    # (d) fused update (regulariser off for this check): zeroed gradient, clamped texture, untouched texels unchanged
    before = eng.arena.p.clone()
    eng.reg_coef = [0.0] * len(eng.reg_coef)
    eng.arena.g.copy_(g_d)
    eng.optimizer_step()
    assert float(eng.arena.g.abs().max()) == 0.0
    assert torch.equal(eng.arena.p[untouched], before[untouched])
    assert float(eng.arena.p.max()) <= O.CLAMP_HI and float(eng.arena.p.min()) >= O.CLAMP_LO
    assert torch.isfinite(eng.arena.p).all() and float((eng.arena.p - before).abs().max()) > 0.5
