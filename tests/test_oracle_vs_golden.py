"""Pins the oracle (oracle/stylemesh_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_goldens.py in the build container). CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import stylemesh_oracle as O
from conftest import batch_from_golden, load_golden
from golden_cases import (FLAGSETS, LOSS_WEIGHTS, MULTIVIEW_SEEDS, SMALL_LEVEL_HW, SMALL_ROOM, SMALL_VIEW_HW,
                          STYLE_HW, STYLE_SEED, STYLE_WEIGHTS, TEX, VGG_SEED)
from stylemesh_amd.data import synthetic as S
from stylemesh_amd.data import view_contract as VC

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


# ---------------------------------------------------------------- G1 texture
def test_g1_probe_values():
    d = load_golden("g1_texture")
    # SURVEY.md 8 a3 known answers: 3x4 arange texture, grid (-1,-1),(1,1),(0,0),(1.2,-3) -> 0, 11, 5.5, 3
    close(d["probe_out"].reshape(-1), [0, 11, 5.5, 3])
    tex = torch.arange(12.).view(1, 3, 4)
    grid = torch.tensor([[[[-1., -1.], [1., 1.], [0., 0.], [1.2, -3.]]]])
    close(O.grid_sample_border_explicit(tex, grid), d["probe_out"])


def test_g1_flat_and_hier_forward_backward():
    d = load_golden("g1_texture")
    grid, up = T(d["grid"]), T(d["upstream"])
    layers = [T(d[f"layer{i}"]).clone() for i in range(4)]
    clamped = [l.clamp(O.CLAMP_LO, O.CLAMP_HI) for l in layers]
    close(clamped[0], d["flat_data_after"], 0, 0)  # normalize() mutated the parameter
    close(O.grid_sample_border_explicit(clamped[0], grid), d["flat_out"], 1e-5, 1e-4)
    close(O.sample_texture(clamped[:1], grid), d["flat_out"], 1e-6, 1e-5)
    close(O.grid_sample_border_backward_explicit(clamped[0].shape, grid, up), d["flat_grad"], 1e-5, 1e-5)
    close(sum(O.grid_sample_border_explicit(c, grid) for c in clamped), d["hier_out"], 1e-5, 2e-4)
    for i in range(4):
        close(O.grid_sample_border_backward_explicit(clamped[i].shape, grid, up), d[f"hier_grad{i}"], 1e-5, 1e-5)
    close(O.tex_regularizer(clamped, [8, 4, 2, 0]), d["reg"], 1e-6, 0)
    close(O.sample_texture(clamped, O.identity_grid(24, 40))[0], d["hier_get_image"], 1e-5, 1e-4)


# ---------------------------------------------------------------- G2 VGG
def test_g2_vgg_forward_and_input_grad():
    d = load_golden("g2_vgg")
    state = S.seeded_vgg_state(int(d["vgg_seed"]))
    x = T(d["x"]).clone().requires_grad_(True)
    keys = [k[4:] for k in d.files if k.startswith("out_")]
    out = O.vgg_forward(state, x, keys)
    for k in keys:
        close(out[k], d["out_" + k], 1e-4, 1e-3)
    out_e = O.vgg_forward(state, x.detach(), ["p1", "p4", "r51"], explicit_pool=True)
    close(out_e["p1"], d["out_p1"], 1e-4, 1e-3)
    close(out_e["p4"], d["out_p4"], 1e-4, 1e-3)
    sum((out[k] * T(d["up_" + k])).sum() for k in ['r11', 'r21', 'r31', 'r41', 'r51', 'r42']).backward()
    close(x.grad, d["grad_x"], 1e-4, 1e-3 * float(np.abs(d["grad_x"]).max()))


# ---------------------------------------------------------------- G3 Gram / MSE
def test_g3_gram_masked_mse():
    d = load_golden("g3_gram")
    f = T(d["f"]).clone().requires_grad_(True)
    mask, target = T(d["mask"]), T(d["target"])
    close(O.gram_matrix(f), d["gram_full"])
    mf = O.masked_features(f, mask)
    assert mf.shape[2] == int(d["masked_n"])
    g = O.gram_matrix(mf)
    close(g, d["gram_masked"])
    # the mask-multiply form the HIP kernels use is the same quantity
    fm = (f * mask).reshape(1, 8, -1)
    close(torch.bmm(fm, fm.transpose(1, 2)) / mask.sum(), d["gram_masked"], 1e-5, 1e-6)
    loss = F.mse_loss(target, g)
    close(loss, d["style_mse"])
    loss.backward()
    close(f.grad, d["grad_style"], 1e-5, 1e-7)
    f.grad = None
    tf = T(d["tgt_feat"])
    cl = F.mse_loss(O.masked_features(tf, mask), O.masked_features(f, mask))
    close(cl, d["content_mse"])
    close((mask * (tf - f) ** 2).sum() / (8 * mask.sum()), d["content_mse"], 1e-5, 1e-7)
    cl.backward()
    close(f.grad, d["grad_content"], 1e-5, 1e-7)
    empty = O.masked_features(f, torch.zeros_like(mask))
    assert list(empty.shape) == list(d["empty_shape"])
    close(O.gram_matrix(empty), d["empty_gram"], 0, 0)


# ---------------------------------------------------------------- G4 style pyramid
def test_g4_pyramid_shapes_and_targets():
    d = load_golden("g4_style")
    for k in d.files:
        if k.startswith("shapes_") and not k.startswith("shapes_fwd_"):
            h, w = map(int, k[len("shapes_"):].split("x"))
            assert O.image_pyramid_sizes(h, w, [0, 1, 2, 3, 4]) == [tuple(s) for s in d[k]]
    style = S.style_image(int(d["style_seed"]), 600, 520)[None]
    pyr = O.image_pyramid(style, [0, 1, 2, 3, 4])
    close(pyr[0][0, :, ::5, ::5], d["pyr0_sub"], 1e-5, 1e-4)
    close(pyr[1][0, :, ::5, ::5], d["pyr1_sub"], 1e-5, 1e-4)
    state = S.seeded_vgg_state(VGG_SEED)
    layers = ['r11', 'r21', 'r31', 'r41', 'r51']
    tg = O.style_targets(state, style, layers)
    for li, layer in enumerate(layers):
        for lvl in (0, 1, 2):
            g = tg[li][lvl][0]
            if f"target_{layer}_{lvl}" in d.files:
                ref = d[f"target_{layer}_{lvl}"]
                close(g, ref, 1e-4, 1e-5 * float(np.abs(ref).max()))
            else:
                ref = d[f"target_{layer}_{lvl}_sub"]
                close(g[::5, ::7], ref, 1e-4, 1e-5 * float(np.abs(ref).max()))
                np.testing.assert_allclose(float(g.double().sum()), float(d[f"target_{layer}_{lvl}_sum"]), rtol=1e-5)


# ---------------------------------------------------------------- G5 pipeline loss + texture gradient
def make_oracle(cfgd, init_layers=None, **kw):
    cfg = O.OracleConfig(hierarchical=cfgd["hier"], style_weights=STYLE_WEIGHTS, angle_threshold=cfgd["thr"],
                         style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"], use_angle_weight=cfgd["angle"],
                         use_depth_scaling=cfgd["depth"], loss_weights=dict(LOSS_WEIGHTS), learning_rate=1,
                         decay_gamma=0.1, decay_step_size=1, **kw)
    return O.OraclePipeline(S.seeded_vgg_state(VGG_SEED), S.style_image(STYLE_SEED, *STYLE_HW), cfg, (TEX, TEX),
                            init_layers=init_layers)


@pytest.mark.parametrize("name", list(FLAGSETS))
def test_g5_forward_with_loss(name):
    d = load_golden("g5_" + name)
    cfgd = FLAGSETS[name]
    batch = batch_from_golden(d)
    init = [T(d[f"init{i}"]) for i in range(4)]
    pipe = make_oracle(cfgd, init)
    n_steps = 3 if cfgd["gram"] == "average" else 1
    for s in range(n_steps):
        tag = f"_s{s}" if n_steps > 1 else ""
        rec = {}
        losses, grads = pipe.grads(batch, rec)
        for lt in ("content", "style", "tex_reg", "total"):
            np.testing.assert_allclose(float(losses[lt]), float(d[f"loss_{lt}{tag}"]), rtol=2e-5)
        for i, g in enumerate(grads):
            ref = d[f"grad{i}{tag}"]
            close(g, ref, 1e-4, 2e-5 * float(np.abs(ref).max()))
    for k, p in enumerate(rec["preds"]):
        close(p, d[f"pred{k}"], 1e-5, 1e-4)


def test_g5_synthetic_view_is_reproducible():
    """The fixture's inputs are what the committed generator produces from the seed (so the GPU box can
    rebuild them without the fixture)."""
    d = load_golden("g5_with_angle_and_depth")
    b = S.make_view(3, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                    min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))
    g = batch_from_golden(d)
    for i in (0, 3, 4, 5, 6, 7, 10, 11, 12):
        close(b[i].float(), g[i].float(), 1e-6, 1e-6)
    for u, v in zip(b[9], g[9]):
        close(u, v, 1e-6, 1e-6)


# ---------------------------------------------------------------- G6 Adam + StepLR
@pytest.mark.parametrize("init_name", ["zero", "seeded"])
def test_g6_adam_steps(init_name):
    d = load_golden("g6_adam_" + init_name)
    batch = batch_from_golden(load_golden("g5_with_angle_and_depth"))
    init = None
    if init_name == "seeded":
        g5 = load_golden("g5_with_angle_and_depth")
        init = [T(g5[f"init{i}"]) for i in range(4)]
    pipe = make_oracle(FLAGSETS["with_angle_and_depth"], init)
    for step in range(5):
        losses, grads = pipe.grads(batch)
        np.testing.assert_allclose(float(losses["total"]), float(d[f"loss_total_step{step}"]), rtol=5e-5)
        if step == 0:
            for i, g in enumerate(grads):
                ref = d[f"grad{i}_step0"]
                close(g, ref, 1e-4, 2e-5 * float(np.abs(ref).max()))
        pipe.apply_adam(grads)
        if step % 2 == 1:
            pipe.end_epoch()
        if step in (0, 1, 4):
            for i in range(4):
                p_ref = d[f"p{i}_after{step + 1}"]
                p = pipe.layers[i].detach()
                # Adam's first updates are ~ -lr*sign(g): texels whose gradient is at rounding-noise level may
                # flip by a whole step between two correct implementations (SURVEY.md 7.2 hazard), so compare
                # the bulk tightly and bound the outlier fraction.
                bad = (p - T(p_ref)).abs() > 2e-3
                assert bad.sum() <= max(3, (2e-3 if step == 0 else 5e-3) * bad.numel()), (step, i, bad.sum())
                for mine, key, rel in ((pipe.m[i], f"m{i}_after{step + 1}", 2e-5), (pipe.v[i], f"v{i}_after{step + 1}", 1e-6)):
                    ref = T(d[key])
                    # bit-level differences of the first update (lerp vs mul/add, addcdiv rounding) occasionally flip a
                    # ReLU / max-pool decision in a LATER step's forward pass; the moments then differ in a small
                    # neighbourhood by < 1 % of max: after the first step the tight bound must hold for all but a
                    # handful of elements (<= 0.5 %: one flipped window's receptive field), the loose one everywhere
                    err, mx = (mine - ref).abs(), float(ref.abs().max())
                    tight = err > 1e-3 * ref.abs() + rel * mx
                    loose = err > 1e-3 * ref.abs() + 1e-2 * mx
                    if step == 0:
                        assert tight.sum() == 0, (key, tight.sum())
                    elif step == 1:
                        assert loose.sum() == 0 and tight.sum() <= max(16, 5e-3 * tight.numel()), (key, tight.sum(), loose.sum())
                    else:   # five steps at lr 1 from a zero texture: a few flipped windows, amplified
                        assert loose.sum() <= max(16, 1e-3 * loose.numel()) and float(err.max()) <= 5e-2 * mx, \
                            (key, loose.sum(), float(err.max()) / mx)


def test_adam_explicit_matches_torch():
    torch.manual_seed(0)
    p = torch.randn(1000)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1.0)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        g = torch.randn(1000) * 10 ** torch.randint(-6, 3, (1000,)).float()
        pt.grad = g.clone()
        opt.step()
        p, m, v = O.adam_step_explicit(p, g, m, v, step, 1.0)
        close(p, pt.detach(), 1e-6, 1e-6)


# ---------------------------------------------------------------- G7 input contract
def test_g7_depth_levels_and_masks():
    d = load_golden("g7_contract")
    out = VC.calculate_depth_level(d["known_depth"], np.linspace(256, 960, 5)[:4], 0.25)
    # SURVEY.md 8 a19 known-answer table
    np.testing.assert_array_equal(out[1].reshape(-1), [0, 0, 0, 0, 0, 1, 1, 1, 2, 3, 3, 3])
    np.testing.assert_array_equal(out[2].reshape(-1), [0, 0, 0, 1, 1, 0, 1, 2, 3, 3, 3, 3])
    for a, k in zip(out, ("known_cont", "known_rounded", "known_other", "known_weight")):
        np.testing.assert_allclose(a, d[k], rtol=1e-6, atol=1e-7)
    out = VC.calculate_depth_level(d["depth"], np.linspace(256, 960, 5)[:4], 0.25)
    for a, k in zip(out, ("cont", "rounded", "other", "weight")):
        np.testing.assert_allclose(a, d[k], rtol=1e-6, atol=1e-7)
    out = VC.calculate_depth_level(d["depth"].astype(np.float32), [256., 432., 608., 784.], 0.2)
    for a, k in zip(out, ("mp_cont", "mp_rounded", "mp_other", "mp_weight")):
        np.testing.assert_allclose(a, d[k], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(VC.calculate_mask(d["uv"], d["depth"]), d["mask_scannet"])
    np.testing.assert_array_equal(VC.calculate_mask(d["uv"]), d["mask_matterport"])
    close(VC.pre(T(d["rgb01"])), d["rgb_pre"], 1e-6, 1e-5)
    close(VC.post(VC.pre(T(d["rgb01"]))), d["rgb01"], 1e-5, 1e-6)


# ---------------------------------------------------------------- G8 multi-view mean gradient + Adam
def test_g8_multiview_mean_gradient_step():
    d = load_golden("g8_multiview")
    g5 = load_golden("g5_with_angle_and_depth")
    init = [T(g5[f"init{i}"]) for i in range(4)]
    assert list(d["view_seeds"]) == MULTIVIEW_SEEDS
    all_grads = []
    for s in MULTIVIEW_SEEDS:
        pipe = make_oracle(FLAGSETS["with_angle_and_depth"], init)
        batch = S.make_view(s, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                            min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))
        losses, grads = pipe.grads(batch)
        np.testing.assert_allclose(float(losses["total"]), float(d[f"loss_total_view{s}"]), rtol=5e-5)
        all_grads.append(grads)
    mean = [torch.stack([g[i] for g in all_grads]).mean(0) for i in range(4)]
    for i in range(4):
        ref = d[f"mean_grad{i}"]
        close(mean[i], ref, 1e-4, 2e-5 * float(np.abs(ref).max()))
    pipe.apply_adam(mean)
    for i in range(4):
        bad = (pipe.layers[i].detach() - T(d[f"p{i}_after"])).abs() > 2e-3
        assert bad.sum() <= max(3, 2e-3 * bad.numel())


# ---------------------------------------------------------------- explicit conventions vs ATen
def test_explicit_resize_and_erode_match_aten():
    torch.manual_seed(1)
    x = torch.randn(1, 3, 17, 23)
    for hw in [(40, 56), (9, 11), (17, 23), (64, 88), (5, 80)]:
        close(O.resize_nearest_explicit(x, hw), F.interpolate(x, hw, mode="nearest"), 0, 0)
        close(O.resize_bilinear_explicit(x, hw), F.interpolate(x, hw, mode="bilinear"), 1e-5, 5e-6)
    # SURVEY.md 8 a9 probes: nearest 7->3 picks 0,2,4; 7->16 picks 0,0,0,1,1,2,2,3,3,3,4,4,5,5,6,6
    assert O.nearest_index(3, 7).tolist() == [0, 2, 4]
    assert O.nearest_index(16, 7).tolist() == [0, 0, 0, 1, 1, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 6]
    close(O.resize_bilinear_explicit(torch.arange(7.).view(1, 1, 1, 7), (1, 3)).reshape(-1), [2 / 3, 3.0, 16 / 3], 1e-6)
    m = (torch.rand(1, 1, 20, 30) > 0.2).float()
    k = torch.ones(1, 1, 3, 3)
    ref = m * ((F.conv2d(m, k, padding=1) / 9).clamp(0, 1) == 1)
    close(O.erode_explicit(m), ref, 0, 0)
    y = torch.randn(1, 4, 9, 13)
    close(O.maxpool2x2_explicit(y), F.max_pool2d(y, 2, 2), 0, 0)


def test_g9_reprojection_warp():
    """Evaluation-metric warp against the reference's data/utils.py reproject / unproject."""
    d = load_golden("g9_reproject")
    for n in range(3):
        t = lambda k: torch.from_numpy(d[f"{k}{n}"])
        color, mask = O.reproject_explicit(t("c2w_src"), t("c2w_tar"), t("K"), t("depth_src"), t("depth_tar"),
                                           t("color_tar"), t("depth_tar") > 0)
        ref_mask = t("out_mask")
        # threshold tests on fp32 coordinates (bounds, floor, |dz| > 0.1, mask > 0.99): allow a stray pixel
        assert int((mask != ref_mask).sum()) <= 2, n
        both = (mask & ref_mask)[None]
        np.testing.assert_allclose((color * both).numpy(), (t("out_color") * both).numpy(), rtol=1e-4, atol=2e-3)
        pts = O.unproject_explicit(t("c2w_src"), t("K"), t("depth_src"))
        np.testing.assert_allclose(pts.numpy(), d[f"unproject{n}"], rtol=1e-5, atol=1e-5)
