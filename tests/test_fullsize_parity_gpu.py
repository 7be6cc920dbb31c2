"""The HIP engine against the oracle at BASELINE.json's full sizes, on the same seeded, non-constant texture and the same
synthetic views (VERDICT r2 task 1):

* c3 - ScanNet with_angle_and_depth: 4096^2 hier-4 texture, UV levels 256x341 .. 784x1045, multi, angle 30
  (reference model/model.py:178-327, scripts/train/optimize_texture_scannet_with_angle_and_depth.sh);
* c2 - ScanNet only2D: 2048^2 hier-4 texture, one UV level 256x341, single (optimize_texture_scannet_only2D.sh);
* c5 - Matterport with_angle_and_depth: 4096^2, UV levels 256x320 .. 784x980, angle 40, min_pyramid_depth 0.2
  (scripts/train/optimize_texture_matterport_with_angle_and_depth.sh:11-15);
* with_angle - ScanNet with_angle: 4096^2 hier-4, ONE UV level, multi, angle weighting on, depth scaling off
  (scripts/train/optimize_texture_scannet_with_angle.sh:3-20);
* dip - ScanNet dip: 4096^2 ONE-layer texture, tex_reg 0, style 1e-3, single, ``gram_mode average`` (the 10-deep
  detached Gram history of content_and_style_losses.py:319-323), ``index_repeat 1`` = a NEW view every step
  (scripts/train/optimize_texture_scannet_dip.sh:3-20; data/abstract_dataset.py:498-512).

Two kinds of test, both in the default fp16x2-split arithmetic AND with v_mfma_f32_32x32x2_f32 everywhere
(STYLEMESH_CONV_MODE / GRAM_MODE = f32), on three view seeds per config and the full-size 1528 x 1200 style image:

1. ``test_one_step_matches_oracle_at_full_size``: ONE step's losses and texture gradient. Stated fp32 tolerances: losses
   rtol 2e-4; gradient ``|err| <= 1e-3 |ref| + 2e-4 max|ref|`` on >= 99.5 % of the TOUCHED texels (the texels some view
   pixel maps to; on the others the data term must be exactly zero and the oracle's gradient the regulariser's alone),
   everywhere ``<= MAX_ERR[config] max|ref|`` (twice the largest value measured over seeds and modes). The texels
   beyond the tight bound are max-pool argmax flips and ReLU-gate ties - since round 5 IDENTIFIED,
   not budgeted: the engine's argmax codes and gate states are read back and compared with the oracle's for the same
   rendered images, the receptive-field footprints of the differing decisions are projected through the UV grids onto
   the texture, and the test asserts that EVERY out-of-bound texel lies inside them and that outside them the gradient
   agrees to 1e-5 of its maximum (``identify_decision_differences``). The fp32-MFMA mode shows the same fractions, and
   ``test_split_arithmetic_adds_no_flips_over_all_cases`` asserts that the split arithmetic adds nothing to them
   (mean split2 fraction <= 1.5 x mean f32 fraction + 2e-4 over the cases).
2. ``test_k_steps_texture_values_match_oracle_at_full_size``: FIVE training steps (3 on one view, 2 on the next: a view
   change inside) of the engine against five of the oracle - reference model/model.py:178-327 + Adam :387-395 - and the
   texture VALUES compared after every step: in lock-step (every step from the oracle's state) by DESIGN.md section 2's
   rule, free-running against a measured control - the oracle's distance from ITSELF under another fp32 summation
   order (see the test's docstring).

Everything measured is printed and written to gpurun_out/fullsize_parity.json (copied to profiles/ by the builder).
The oracle runs on the host cores (a few seconds per step at these sizes)."""
import json
import os
import time

import numpy as np
import pytest
import torch

import stylemesh_oracle as O
from conftest import REPO
from gpu_util import require_gpu
from stylemesh_amd.data import synthetic as S

pytestmark = pytest.mark.gpu

LOSS_WEIGHTS = {"content": 7e1, "style": 1e-4, "tex_reg": 5e3}
STYLE_WEIGHTS = [1000., 1000., 10., 10., 1000.]
STYLE_HW = (1528, 1200)   # "The Scream" (styles/120styles/17.jpg) is 1200 x 1528 px: the bench's style image
ROOM = (12.0, 9.0, 3.0)
CASES = {
    "c3": dict(tex=4096, level_hw=S.SCANNET_LEVEL_HW, view_hw=S.SCANNET_VIEW_HW, mode="multi", thr=30.0, angle=True,
               depth=True, min_depth=0.25, seeds=(2, 6, 9), active=[0, 1, 2, 3]),
    "c2": dict(tex=2048, level_hw=[S.SCANNET_VIEW_HW], view_hw=S.SCANNET_VIEW_HW, mode="single", thr=3000.0, angle=False,
               depth=False, min_depth=0.25, seeds=(2, 6, 9), active=[0]),
    "c5": dict(tex=4096, level_hw=S.MATTERPORT_LEVEL_HW, view_hw=S.MATTERPORT_VIEW_HW, mode="multi", thr=40.0, angle=True,
               depth=True, min_depth=0.2, seeds=(2, 6, 9), active=None),
    "with_angle": dict(tex=4096, level_hw=[S.SCANNET_VIEW_HW], view_hw=S.SCANNET_VIEW_HW, mode="multi", thr=30.0,
                       angle=True, depth=False, min_depth=0.25, seeds=(2, 6, 9), active=[0]),
    "dip": dict(tex=4096, level_hw=[S.SCANNET_VIEW_HW], view_hw=S.SCANNET_VIEW_HW, mode="single", thr=3000.0, angle=False,
                depth=False, min_depth=0.25, seeds=(2, 6, 9), active=[0], n_layers=1, gram_mode="average", decay=15,
                loss_weights={"content": 7e1, "style": 1e-3, "tex_reg": 0.0}),
}
MODES = ("split2", "f32")
# one step, gradient: fraction of the touched texels beyond the tight bound, and max |err| / max |ref| per arithmetic mode
# (= 2 x the largest value measured over the three seeds, profiles/r03/fullsize_parity_final.json and profiles/r04/: the
# maximum is the size of the largest max-pool argmax FLIP of the case - which window flips differs between the modes, in
# both directions: c3 seed 9 split2 2.4e-2 / f32 0.8e-2, c5 seed 2 split2 0.6e-2 / f32 1.2e-2, c5 seed 9 0.15e-2 / 0.5e-2)
# (a flip's footprint is a fixed number of texels; the ONE-layer texture of the dip script has no coarse layers, a view
# touches 2 % of it instead of 7 - 20 %, and one flip weighs that much more: seed 6 shows 0.70 % / 0.72 % in split2 / f32,
# seed 9 0.25 % / 1.50 % - which windows flip moves with every change of a summation order: bound = 2 x the largest seen)
FLIP_FRAC_MAX = {"c3": 0.005, "c2": 0.005, "c5": 0.005, "with_angle": 0.01, "dip": 0.03}
MAX_ERR = {"c3": {"split2": 5e-2, "f32": 1.8e-2}, "c2": {"split2": 5e-3, "f32": 5e-3},
           "c5": {"split2": 1.3e-2, "f32": 2.5e-2}, "with_angle": {"split2": 8e-3, "f32": 4e-2},
           "dip": {"split2": 3.7e-2, "f32": 3.7e-2}}


def seeded_texture(tex, n_layers=4, amp=60.0):
    """Smooth, non-constant texture (a constant one makes every max-pool window an exact tie)."""
    rng = np.random.default_rng(0)
    return [torch.from_numpy(((S.smooth_noise(rng, 3, tex >> i, tex >> i, cells=64) - 0.5) * amp).astype(np.float32))
            for i in range(n_layers)]


_SESSION = {}     # what this session's tests measured (the summary test reads THIS, not a file of some earlier run)


@pytest.fixture(scope="session")
def parity_results():
    return _SESSION


def _record(name, entry):
    _SESSION[name] = entry
    out_dir = os.path.join(REPO, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "fullsize_parity.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[name] = entry
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


_CACHE = {}


def _shared(key, make):
    if key not in _CACHE:
        _CACHE[key] = make()
    return _CACHE[key]


def _vgg():
    return _shared("vgg", lambda: S.seeded_vgg_state(0))


def _style():
    return _shared("style", lambda: S.style_image(1, *STYLE_HW))


def _oracle_targets():
    """Gram targets of the full-size style image on the oracle side: once per test session (a few seconds)."""
    return _shared("oracle_targets", lambda: O.style_targets(_vgg(), _style()[None], list(O.OracleConfig().style_layers)))


def _view(c, seed):
    return S.make_view(seed, view_hw=c["view_hw"], level_hw=c["level_hw"], level_heights=[h for h, _ in c["level_hw"]],
                       min_pyramid_depth=c["min_depth"], room=S.BoxRoom(ROOM))


def _oracle(c, tex0):
    ocfg = O.OracleConfig(hierarchical=True, style_weights=STYLE_WEIGHTS, angle_threshold=c["thr"],
                          style_pyramid_mode=c["mode"], gram_mode=c.get("gram_mode", "current"),
                          use_angle_weight=c["angle"], use_depth_scaling=c["depth"],
                          loss_weights=dict(c.get("loss_weights", LOSS_WEIGHTS)), learning_rate=1.0,
                          decay_step_size=c.get("decay", 3))
    return O.OraclePipeline(_vgg(), _style(), ocfg, (c["tex"], c["tex"]), n_layers=c.get("n_layers", 4), init_layers=tex0,
                            targets=_oracle_targets())


def _engine(c, mode, tex0, monkeypatch):
    from stylemesh_amd.runtime import ops
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    monkeypatch.setattr(ops, "CONV_MODE", mode)
    monkeypatch.setattr(ops, "GRAM_MODE", mode)
    cfg = EngineConfig(tex_w=c["tex"], tex_h=c["tex"], hierarchical=True, n_layers=c.get("n_layers", 4),
                       style_weights=STYLE_WEIGHTS, angle_threshold=c["thr"], style_pyramid_mode=c["mode"],
                       gram_mode=c.get("gram_mode", "current"), use_angle_weight=c["angle"], use_depth_scaling=c["depth"],
                       loss_weights=dict(c.get("loss_weights", LOSS_WEIGHTS)), learning_rate=1.0,
                       decay_step_size=c.get("decay", 3))
    eng = StepEngine(cfg, _vgg())
    eng.load_texture(tex0)
    eng.set_style_image(_style())
    return eng


def _tex0(c):
    n = c.get("n_layers", 4)
    return _shared(("tex", c["tex"], n), lambda: seeded_texture(c["tex"], n))


def _coverage(eng):
    """bool per arena element: does some pixel of the current view map to this texel? (scatter of ones, no weights)"""
    from stylemesh_amd.runtime import ops
    cover = torch.zeros_like(eng.arena.g)
    for lv in eng.view:
        if lv.active:
            b = eng._level_bufs(lv.H, lv.W)
            ones = type(b.grad["img"])(3, lv.H, lv.W).from_dense(torch.ones(3, lv.H, lv.W))
            ops.tex_sample_bwd(eng.arena.views(cover), lv.grid, ones, None)
    return (cover != 0).cpu()



# ---------------------------------------------------------------------------------------------------------------------
# Identification of the texels beyond the tight bound. The only places where two correct fp32
# implementations of this step can DECIDE differently are (i) a 2x2 max-pool window whose two largest activations are closer
# than the forward rounding noise (its gradient is routed to another pixel: content_and_style_losses.py:51,54,59,64,69) and
# (ii) a ReLU whose pre-activation is zero to rounding (its gate passes or blocks the gradient). Both are read back here -
# the engine's argmax codes / activations against the oracle's max_pool2d indices / activations of the SAME rendered
# images - , their receptive-field footprints are projected through the UV grids onto the texture, and the test asserts
# that the out-of-bound texels lie inside those footprints.
# ---------------------------------------------------------------------------------------------------------------------
_BLOCK_CONVS = (2, 2, 4, 4, 4)          # convs per VGG block


def _radius(block, convs_in_block):
    """image-pixel radius of the receptive field of a gradient entering conv #convs_in_block (counted from the block's
    first conv, 1-based) of ``block`` (1-based) and flowing down to the image"""
    r = convs_in_block * 2 ** (block - 1)
    for j in range(1, block):
        r += _BLOCK_CONVS[j - 1] * 2 ** (j - 1)
    return r


def _dilate(mask, r):
    """bool [H, W] dilated by a (2 r + 1)^2 box"""
    import torch.nn.functional as F
    if r == 0 or not bool(mask.any()):
        return mask
    m = mask[None, None].float()
    return F.max_pool2d(m, 2 * r + 1, 1, r)[0, 0] > 0


def _upsample_to(mask, scale, H, W):
    """bool [h, w] at 1 / scale resolution -> [H, W] (each cell covers scale x scale pixels; floor sizes padded)"""
    up = mask.repeat_interleave(scale, 0).repeat_interleave(scale, 1)
    out = torch.zeros(H, W, dtype=torch.bool, device=mask.device)
    out[:min(H, up.shape[0]), :min(W, up.shape[1])] = up[:H, :W]
    return out


def _engine_codes(b, pool, pre, Ho, Wo):
    """argmax codes [C, Ho, Wo] of the engine's pool ``pool`` (0..3 = dy * 2 + dx of the first maximum, 4 = maximum <= 0):
    from the code image of the pooling epilogue when it exists, else from the stored pre-pool activations."""
    import torch.nn.functional as F
    from stylemesh_amd.runtime import hip
    pooled = b.act[pool]
    if pool in b.code and int(b.code[pool].abs().max()) != 0:
        C8 = pooled.C // 8
        w = b.code[pool].view(C8, pooled.plane)[:, :(Ho + 2) * pooled.Wp].view(C8, Ho + 2, pooled.Wp)[:, 1:Ho + 1, 1:Wo + 1]
        return torch.stack([(w >> (4 * c)) & 15 for c in range(8)], 1).reshape(C8 * 8, Ho, Wo)
    x = b.act[pre].to_dense()
    vals, idx = F.max_pool2d(x[None], 2, 2, return_indices=True)
    wfull = x.shape[2]
    code = ((idx // wfull) % 2) * 2 + (idx % wfull) % 2
    code[vals <= 0] = 4
    return code[0]


def oracle_decisions(oracle_preds, oracle_active, vgg_state, deepest):
    """Per active level: the oracle's pool argmax codes (uint8 [C, Ho, Wo], 0..3 = dy * 2 + dx, 4 = maximum <= 0) and the
    open / closed state of every ReLU gate (bool [C, h, w]) of ITS rendered image - computed once per case."""
    import torch.nn.functional as F
    from stylemesh_amd.runtime import vgg as V
    names = V.OUT_NAMES[:V.depth_of(deepest) + 1]
    out = []
    for i in oracle_active:
        acts = O.vgg_forward(vgg_state, oracle_preds[i], names)
        codes, gates = {}, {}
        for kind, src, name, _, _ in V.NODES[:V.depth_of(deepest) + 1]:
            if kind == "pool":
                vals, ind = F.max_pool2d(acts[src], 2, 2, return_indices=True)
                wfull = acts[src].shape[3]
                c = ((ind // wfull) % 2) * 2 + (ind % wfull) % 2
                c[vals <= 0] = 4
                codes[name] = c[0].to(torch.uint8)
            else:
                gates[name] = acts[name][0] > 0
        out.append({"codes": codes, "gates": gates})
        del acts
    return out


def identify_decision_differences(eng, decisions, oracle_active, deepest):
    """-> (texel footprint bool [arena.n] on the CPU, counts dict). See the block comment above."""
    from stylemesh_amd.runtime import ops
    from stylemesh_amd.runtime import vgg as V
    names = V.OUT_NAMES[:V.depth_of(deepest) + 1]
    cover = torch.zeros_like(eng.arena.g)
    counts = {"flipped_windows": 0, "gate_differences": 0, "footprint_pixels": 0}
    levels = [lv for lv in eng.view if lv.active]
    assert [lv.index for lv in levels] == list(oracle_active)
    for lv, dec in zip(levels, decisions):
        b = eng._level_bufs(lv.H, lv.W)
        img_mask = torch.zeros(lv.H, lv.W, dtype=torch.bool, device="cuda")
        block, idx = 1, 0
        for kind, src, out, _, _ in V.NODES[:V.depth_of(deepest) + 1]:
            if kind == "pool":
                code_o = dec["codes"][out].cuda()
                Ho, Wo = code_o.shape[1], code_o.shape[2]
                code_e = _engine_codes(b, out, src, Ho, Wo)
                relevant = (b.grad[out].to_dense() != 0).any(0)           # positions whose pooled gradient is in use
                diff = (code_e != code_o) & relevant[None]
                counts["flipped_windows"] += int(diff.sum())
                fm = diff.any(0)
                if bool(fm.any()):
                    img_mask |= _dilate(_upsample_to(fm, 2 ** block, lv.H, lv.W), _radius(block, _BLOCK_CONVS[block - 1]))
                block, idx = block + 1, 0
                continue
            idx += 1
            if out in V.PRE_POOL and out != deepest and V.POOL_OUTPUT[out] in names:
                continue            # (its gate lives in the pool's codes: 4 = closed; never stored by the pooling epilogue)
            relevant = (b.grad[out].to_dense() != 0).any(0)
            if not bool(relevant.any()):
                continue
            diff = ((b.act[out].to_dense() > 0) != dec["gates"][out].cuda()) & relevant[None]
            counts["gate_differences"] += int(diff.sum())
            gm = diff.any(0)
            if bool(gm.any()):
                img_mask |= _dilate(_upsample_to(gm, 2 ** (block - 1), lv.H, lv.W), _radius(block, idx))
        counts["footprint_pixels"] += int(img_mask.sum())
        if bool(img_mask.any()):
            ones = type(b.grad["img"])(3, lv.H, lv.W).from_dense(img_mask[None].float().expand(3, -1, -1))
            ops.tex_sample_bwd(eng.arena.views(cover), lv.grid, ones, None)
    return (cover != 0).cpu(), counts


@pytest.mark.parametrize("seed_index", [0, 1, 2])
@pytest.mark.parametrize("name", list(CASES))
def test_one_step_matches_oracle_at_full_size(name, seed_index, monkeypatch):
    require_gpu()
    c = CASES[name]
    seed = c["seeds"][seed_index]
    _one_step_case(name, c, seed, _view(c, seed), _tex0(c), monkeypatch, f"{name}_seed{seed}")


def _one_step_case(name, c, seed, view, tex0, monkeypatch, record_key, flip_frac_max=None, max_err=None, outside_tol=1e-5):
    """ONE step's losses and texture gradient from the texture ``tex0``: engine (both arithmetic modes) against the oracle,
    the out-of-bound texels identified (module docstring). Returns the recorded entry."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    flip_frac_max = FLIP_FRAC_MAX[name] if flip_frac_max is None else flip_frac_max
    max_err = MAX_ERR[name] if max_err is None else max_err

    pipe = _oracle(c, tex0)
    rec = {}
    t0 = time.time()
    ref_losses, ref_grads = pipe.grads(view, rec)
    oracle_s = time.time() - t0
    g_ref = torch.cat([g.reshape(-1) for g in ref_grads])
    mx = float(g_ref.abs().max())
    decisions = oracle_decisions(rec["preds"], rec["active"], _vgg(), "r51")   # (style layers r11 .. r51 in every case)
    oracle_active = rec["active"]
    del pipe

    entry = {"seed": seed, "texels": int(g_ref.numel()), "max_ref": mx, "oracle_seconds": round(oracle_s, 1),
             "style_image": f"{STYLE_HW[1]}x{STYLE_HW[0]}"}
    fracs = {}
    for mode in MODES:
        eng = _engine(c, mode, tex0, monkeypatch)
        eng.set_view(view)
        active = [lv.index for lv in eng.view if lv.active]
        if c["active"] is not None:
            assert active == c["active"]
        assert len(active) >= 1 and rec["active"] == active
        eng.arena.g.zero_()
        lt = eng.loss_tensors()
        eng.forward_backward()
        mine = eng.losses(lt)
        g_mine = torch.cat([(g + k * p).reshape(-1) for g, k, p in zip(eng.grads, eng.reg_coef, eng.layers)]).cpu()
        g_data = eng.arena.g.cpu()
        touched = _coverage(eng)
        for k in ("content", "style", "tex_reg", "total"):
            np.testing.assert_allclose(mine[k], float(ref_losses[k]), rtol=2e-4, err_msg=f"{name} {mode} loss {k}")
        reg = torch.cat([(k * p).reshape(-1) for k, p in zip(eng.reg_coef, tex0)])
        # texels no pixel of the view maps to: an exactly-zero data term here, the regulariser's gradient alone there
        assert (~touched).any() and float(g_data[~touched].abs().max()) == 0.0
        assert float((g_ref - reg)[~touched].abs().max()) <= 1e-6 * float(reg.abs().max()) + 1e-12 * mx
        frac_touched = float(touched.float().mean())
        assert 0.0 < frac_touched < 0.6
        err = (g_mine - g_ref).abs()
        bad = (err > 1e-3 * g_ref.abs() + 2e-4 * mx) & touched
        flip_frac = float(bad.sum()) / float(touched.sum())
        fracs[mode] = flip_frac
        # WHERE the out-of-bound texels are: inside the footprints of the windows / gates that decided differently
        with _Mode(mode):
            assert eng.deepest == "r51"
            footprint, counts = identify_decision_differences(eng, decisions, oracle_active, eng.deepest)
        outside = bad & ~footprint
        entry[mode] = {"touched_fraction": round(frac_touched, 5), "fraction_of_touched_texels_beyond_tight_bound": flip_frac,
                       "max_err_over_max_ref": float(err.max()) / mx, "active_levels": active,
                       "loss_rel_err": {k: abs(mine[k] - float(ref_losses[k])) / max(abs(float(ref_losses[k])), 1e-30)
                                        for k in ("content", "style", "tex_reg", "total")},
                       "flipped_pool_windows": counts["flipped_windows"], "relu_gate_differences": counts["gate_differences"],
                       "footprint_fraction_of_touched": float((footprint & touched).sum()) / float(touched.sum()),
                       "out_of_bound_texels": int(bad.sum()), "out_of_bound_texels_outside_footprints": int(outside.sum()),
                       "fraction_of_touched_outside_footprints_beyond_bound":
                           float(outside.sum()) / max(float((touched & ~footprint).sum()), 1.0),
                       "max_err_outside_footprints_over_max_ref": float(err[touched & ~footprint].max()) / mx
                           if bool((touched & ~footprint).any()) else 0.0}
        del eng, g_mine, g_data, err, bad, footprint, outside
        torch.cuda.empty_cache()
    print(f"\n[{record_key}] {json.dumps(entry)}")
    _record(record_key, entry)
    for mode in MODES:
        e = entry[mode]
        # (a) EVERY out-of-bound texel lies inside the footprint of an identified decision (a flipped pool window, a ReLU
        # gate that differs); (b) outside the footprints the gradient is the oracle's to 1e-5 of its maximum - twenty times
        # tighter than the tight bound (measured over the thirty cases of round 5: <= 1.4e-6, profiles/r05/fullsize_parity.json)
        assert e["out_of_bound_texels_outside_footprints"] == 0, (name, mode, e)
        assert e["max_err_outside_footprints_over_max_ref"] <= outside_tol, (name, mode, e)
        assert e["out_of_bound_texels"] == 0 or e["flipped_pool_windows"] + e["relu_gate_differences"] > 0, (name, mode, e)
        assert e["fraction_of_touched_texels_beyond_tight_bound"] <= flip_frac_max, \
            f"{name} {mode}: {fracs[mode]:.5f} of the touched texels beyond 1e-3|ref| + 2e-4 max|ref|"
        assert e["max_err_over_max_ref"] <= max_err[mode], \
            f"{name} {mode}: max err {e['max_err_over_max_ref']:.3e} of max|ref|"
    return entry


# ---------------------------------------------------------------------------------------------------------------------
# LATE in the schedule. Every case above starts from the seeded smooth texture. The reference's schedule
# is 7 epochs x 20 repeats x 273 views with two x0.1 learning-rate decays (scripts/train/
# optimize_texture_scannet_with_angle_and_depth.sh:11-15; model/model.py:387-401): the texture the LAST steps see has been
# through thousands of lr-1 / 0.1 / 0.01 Adam updates, its rendered images carry the style's high frequencies, and the
# gradient tensors' dynamic range is whatever training made it - the regime in which the fp16x2 split's tensor-wide scale
# (elements > 2^18 below the maximum lose low bits) would show, if it showed anywhere. The late state is generated HERE, on
# the GPU box, by the engine itself (zero texture, the script's schedule shape over LATE_VIEWS views); then the one-step
# oracle comparison above runs FROM that texture in both arithmetic modes, with the same identification of every
# out-of-bound texel, and the operand census of the late step (stylemesh_amd/diagnostics.py) is recorded beside it.
# ---------------------------------------------------------------------------------------------------------------------
LATE_VIEWS = (0, 2, 6, 7, 9, 11, 12, 14)        # seeds whose views populate every UV level (bench.py's list)
LATE_EPOCHS, LATE_REPEAT = 7, 20                # epochs 0-2 at lr 1, 3-5 at 0.1, 6 at 0.01 (decay_step_size 3, gamma 0.1)


def _late_texture(c):
    """The texture (CPU layers) after the script-shaped schedule over LATE_VIEWS, from zero, default arithmetic."""
    def make():
        from stylemesh_amd.runtime import ops
        from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
        assert ops.CONV_MODE == "split2"
        cfg = EngineConfig(tex_w=c["tex"], tex_h=c["tex"], hierarchical=True, n_layers=4, style_weights=STYLE_WEIGHTS,
                           angle_threshold=c["thr"], style_pyramid_mode=c["mode"], use_angle_weight=c["angle"],
                           use_depth_scaling=c["depth"], loss_weights=dict(LOSS_WEIGHTS), learning_rate=1.0, decay_step_size=3)
        eng = StepEngine(cfg, _vgg())
        eng.set_style_image(_style())
        views = [_view(c, s_) for s_ in LATE_VIEWS]
        t0 = time.time()
        for epoch in range(LATE_EPOCHS):
            for v in views:
                for _ in range(LATE_REPEAT):
                    eng.training_step(v)
            eng.end_epoch()
        torch.cuda.synchronize()
        info = {"steps": eng.step_count, "epochs": LATE_EPOCHS, "views": len(views), "final_lr": eng.lr,
                "seconds": round(time.time() - t0, 1), "losses_last_step": eng.losses()}
        tex = [l.detach().clone().cpu() for l in eng.layers]
        del eng
        torch.cuda.empty_cache()
        return tex, info
    return _shared(("late", c["tex"]), make)


@pytest.mark.parametrize("seed", [LATE_VIEWS[2], 16], ids=["trained_view", "unseen_view"])
def test_one_step_matches_oracle_on_a_late_texture(seed, monkeypatch):
    """c3 from the texture 1120 steps and two learning-rate decays into training, on a view the training saw and on one it
    never saw: the same one-step comparison, the same identification; both arithmetic modes' flip fractions stay below 1 %
    and within 0.7 % of each other ON THIS STATE (the ratio is asserted over all cases, see below); and the census of the step's fp16x2 operands is recorded (share of elements
    > 2^18 below their tensor's bound)."""
    require_gpu()
    from stylemesh_amd.diagnostics import operand_census, summarize
    c = CASES["c3"]
    tex_late, info = _late_texture(c)
    assert info["steps"] == LATE_EPOCHS * LATE_REPEAT * len(LATE_VIEWS) and abs(info["final_lr"] - 1e-2) < 1e-12
    assert max(float(t.abs().max()) for t in tex_late) > 1.0          # a trained texture, not the zero it started from
    assert seed in LATE_VIEWS or seed == 16
    view = _view(c, seed)
    # (outside the identified footprints: measured 7.7e-6 of max|ref| in fp16x2 against 1.3e-6 in fp32-MFMA on this state -
    # late in training 3 - 7 % of the non-zero elements of the style-loss derivative matrices lie more than 2^18 below their
    # tensor's bound, profiles/r06/split2_dynamic_range.json; the early-training cases show <= 1.4e-6 in both modes. Bound:
    # 5e-5 = a quarter of the tight bound's absolute term)
    entry = _one_step_case("c3", c, seed, view, tex_late, monkeypatch, f"late_c3_seed{seed}", flip_frac_max=0.05,
                           max_err={"split2": 0.25, "f32": 0.25}, outside_tol=5e-5)
    # Flips are discrete events with large footprints: on ONE late state the two arithmetic modes' fractions scatter both ways
    # from run to run (six comparisons on three boxes, profiles/r06/README.md: fp16x2 0.02 / 0.05 / 0.06 / 0.09 / 0.16 / 0.19 %
    # against fp32-MFMA 0.02 / 0.03 / 0.30 / 0.03 / 0.23 / 0.05 %). The RATIO claim (fp16x2 <= 1.5 x fp32-MFMA + 2e-4) is made
    # where it is statistically meaningful - over all one-step cases of the session, these two included
    # (test_split_arithmetic_adds_no_flips_over_all_cases); here: both small, and no gap of the size a scale problem would open.
    key = "fraction_of_touched_texels_beyond_tight_bound"
    # (observed maxima over the round's runs: 0.30 % in a mode, 0.14 % between the modes - the bounds leave 3 - 5 x)
    assert entry["split2"][key] <= 1e-2 and entry["f32"][key] <= 1e-2, (entry["split2"][key], entry["f32"][key])
    assert entry["split2"][key] <= entry["f32"][key] + 7e-3, (entry["split2"][key], entry["f32"][key])
    # the operand census of this very step (dense tiles: every stored position is this step's)
    eng = _engine(c, "split2", tex_late, monkeypatch)
    eng.sparse_tiles = False
    eng.set_view(view)
    eng.arena.g.zero_()
    eng.forward_backward()
    census = operand_census(eng)
    summary = dict(summarize(census), late_state=info)
    print(f"\n[late c3 census] {json.dumps(summary)}")
    _record(f"late_c3_census_seed{seed}", {"summary": summary,
                               "share_beyond_2^18": {k: v["share_beyond_2^k"]["18"] for k, v in census.items()},
                               "median_log2_bound_over_x": {k: v["median_log2_bound_over_x"] for k, v in census.items()}})
    assert len(census) >= 30 and all(e["bound"] >= e["true_max"] for e in census.values())


def test_split_arithmetic_adds_no_flips_over_all_cases(parity_results):
    """Flips are discrete events (a case can show 0 in one mode and 0.1 % in the other): the comparison between the
    fp16x2-split and the fp32-MFMA arithmetic is made over ALL the one-step cases THIS session has run (the session
    fixture the test above records into; no file of an earlier run is read)."""
    cases = [v for k, v in parity_results.items() if "_seed" in k and all(m in v for m in MODES)]
    if len(cases) < 6:
        pytest.skip("needs the one-step cases of this session (run the whole file)")
    key = "fraction_of_touched_texels_beyond_tight_bound"
    mean = {m: float(np.mean([v[m][key] for v in cases])) for m in MODES}
    worst = {m: float(np.max([v[m]["max_err_over_max_ref"] for v in cases])) for m in MODES}
    print(f"\n[flip fractions over {len(cases)} cases] mean {mean}, worst max-err {worst}")
    _record("summary_one_step", {"cases": len(cases), "mean_flip_fraction": mean, "worst_max_err_over_max_ref": worst})
    assert mean["split2"] <= 1.5 * mean["f32"] + 2e-4, mean


K_STEPS, SWITCH_AT = 5, 3


class _Mode:
    """Run a block under one arithmetic mode (the mode is a module global read at launch time)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from stylemesh_amd.runtime import ops
        self.saved = ops.CONV_MODE, ops.GRAM_MODE
        ops.CONV_MODE = ops.GRAM_MODE = self.mode

    def __exit__(self, *exc):
        from stylemesh_amd.runtime import ops
        ops.CONV_MODE, ops.GRAM_MODE = self.saved


def _tex_stats(err):
    return {"beyond_1e-5": float((err > 1e-5).float().mean()), "beyond_2e-3": float((err > 2e-3).float().mean()),
            "beyond_2e-2": float((err > 2e-2).float().mean()), "beyond_0.3": float((err > 0.3).float().mean()),
            "max": float(err.max())}


def _flat(tensors):
    return torch.cat([t.detach().reshape(-1) for t in tensors])


# lock-step (the engine starts every step from the oracle's texture and Adam moments): DESIGN.md section 2's rule
LOCK_TIGHT_FRAC, LOCK_LOOSE_FRAC = 0.01, 0.001   # (measured: <= 0.31 % / <= 0.032 %)


# name -> (steps, the step's view = seeds[view_of_step(k)]): five steps with ONE view change for the index_repeat 20 / 100
# scripts; dip runs index_repeat 1 - a new view EVERY step - for 12 steps, so that the 10-deep Gram history fills and wraps
K_SCHEDULES = {"c3": (K_STEPS, lambda k: 0 if k < SWITCH_AT else 1), "c2": (K_STEPS, lambda k: 0 if k < SWITCH_AT else 1),
               "c5": (K_STEPS, lambda k: 0 if k < SWITCH_AT else 1), "with_angle": (K_STEPS, lambda k: 0 if k < SWITCH_AT else 1),
               "dip": (12, lambda k: k)}
DIP_SEEDS = (2, 6, 9, 0, 7, 11, 12, 14, 16, 18, 22, 23)


@pytest.mark.parametrize("name", ["c3", "c2", "c5", "with_angle", "dip"])
def test_k_steps_texture_values_match_oracle_at_full_size(name, monkeypatch):
    """Texture VALUES over five training steps with a view change inside (north_star: 'outputs match the reference
    PyTorch path's texture values'; reference model/model.py:178-327 + Adam :387-395), two ways, both modes:

    * LOCK-STEP: before every step the engine is given the oracle's texture and Adam moments, so that each step's own
      contribution is compared: |err| <= 2e-3 on >= 99 %, <= 2e-2 on >= 99.9 % of the texels (step 1: 1e-5 except
      sign-flip texels - the first update is -lr sign(g)).
    * FREE-RUNNING: five engine steps against five oracle steps. Adam at lr 1 normalises every texel's update to O(1)
      whatever its gradient's magnitude, so texels whose gradient is at the level of the fp32 summation-order noise take
      a different path after a few steps in ANY two fp32 implementations. The test measures exactly that as its
      CONTROL - the same oracle run with oneDNN switched off (ATen's im2col + GEMM convolutions instead: the reference
      op for op, another summation order) - and requires the engine's distance from the oracle to stay within 2 x the
      oracle's distance from itself (+ 0.5 % of the texels)."""
    require_gpu()
    c = CASES[name]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    n_steps, view_of = K_SCHEDULES[name]
    seeds = DIP_SEEDS if name == "dip" else c["seeds"]
    views = {}
    for k in range(n_steps):
        if view_of(k) not in views:
            views[view_of(k)] = _view(c, seeds[view_of(k)])
    tex0 = _tex0(c)
    schedule = [views[view_of(k)] for k in range(n_steps)]

    pipe, ctrl = _oracle(c, tex0), _oracle(c, tex0)
    free, lock = {}, {}
    for mode in MODES:
        free[mode] = _engine(c, mode, tex0, monkeypatch)
        lock[mode] = _engine(c, mode, tex0, monkeypatch)
    entry = {"steps": n_steps, "view_change_before_step": "every" if name == "dip" else SWITCH_AT + 1,
             "texels": int(free["split2"].arena.n),
             "control": [], "free": {m: [] for m in MODES}, "lock_step": {m: [] for m in MODES}}
    clamp = lambda t: t.clamp(O.CLAMP_LO, O.CLAMP_HI)
    t_oracle = 0.0
    for k, batch in enumerate(schedule):
        # lock-step engines start from the oracle's state before this step
        state = [_flat(pipe.layers), _flat(pipe.m), _flat(pipe.v)]
        for mode in MODES:
            e = lock[mode]
            for dst, src in zip((e.arena.p, e.arena.m, e.arena.v), state):
                dst.copy_(src)
            e.step_count = pipe.step_count
            e.sumsq.zero_()
            from stylemesh_amd.runtime import ops
            ops.clamp_sumsq(e.arena.p, e.arena.seg_end, e.sumsq)
        t0 = time.time()
        ref_loss = pipe.training_step(batch)
        t_oracle += time.time() - t0
        with torch.backends.mkldnn.flags(enabled=False):
            ctrl.training_step(batch)
        ref = clamp(_flat(pipe.layers))
        entry["control"].append(_tex_stats((clamp(_flat(ctrl.layers)) - ref).abs()))
        for mode in MODES:
            with _Mode(mode):
                lt = free[mode].training_step(batch)
                mine = free[mode].losses(lt)
                ll = lock[mode].training_step(batch)
                lock_loss = lock[mode].losses(ll)
            np.testing.assert_allclose(lock_loss["total"], ref_loss["total"], rtol=2e-4, err_msg=f"{name} {mode} step {k + 1}")
            st = _tex_stats((free[mode].arena.p.cpu() - ref).abs())
            st["loss_total_rel_err"] = abs(mine["total"] - ref_loss["total"]) / abs(ref_loss["total"])
            entry["free"][mode].append(st)
            entry["lock_step"][mode].append(_tex_stats((lock[mode].arena.p.cpu() - ref).abs()))
    entry["oracle_seconds"] = round(t_oracle, 1)
    for mode in MODES:
        assert float(free[mode].arena.g.abs().max()) == 0.0   # the fused update leaves a zeroed gradient
    del free, lock, pipe, ctrl
    torch.cuda.empty_cache()
    print(f"\n[{name} k-step] {json.dumps(entry)}")
    _record(f"{name}_ksteps", entry)
    for mode in MODES:
        for k in range(n_steps):
            m = entry["lock_step"][mode][k]
            what = f"{name} {mode} lock-step, step {k + 1}: {m}"
            if k == 0:
                assert m["beyond_1e-5"] <= 2e-4, what
            assert m["beyond_2e-3"] <= LOCK_TIGHT_FRAC and m["beyond_2e-2"] <= LOCK_LOOSE_FRAC, what
            f, ctl = entry["free"][mode][k], entry["control"][k]
            what = f"{name} {mode} free-running, step {k + 1}: {f} vs the oracle's own {ctl}"
            for key in ("beyond_2e-3", "beyond_2e-2", "beyond_0.3"):
                assert f[key] <= 2.0 * ctl[key] + 5e-3, what
    # the split arithmetic adds nothing: after the last step its fractions are those of the fp32-MFMA mode
    a, b = entry["free"]["split2"][-1], entry["free"]["f32"][-1]
    assert a["beyond_2e-3"] <= 1.5 * b["beyond_2e-3"] + 5e-3 and a["beyond_2e-2"] <= 1.5 * b["beyond_2e-2"] + 5e-3, (a, b)
