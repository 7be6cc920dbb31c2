"""ONE step of the HIP engine against ONE step of the oracle at BASELINE.json's full sizes, on the same seeded,
non-constant texture and the same synthetic view:

* c3 - ScanNet with_angle_and_depth: 4096^2 hier-4 texture, UV levels 256x341 .. 784x1045, multi, angle 30
  (reference model/model.py:178-327, scripts/train/optimize_texture_scannet_with_angle_and_depth.sh);
* c2 - ScanNet only2D: 2048^2 hier-4 texture, one UV level 256x341, single (optimize_texture_scannet_only2D.sh);
* c5 - Matterport with_angle_and_depth: 4096^2, UV levels 256x320 .. 784x980, angle 40, min_pyramid_depth 0.2
  (scripts/train/optimize_texture_matterport_with_angle_and_depth.sh:11-15).

Stated fp32 tolerances: losses rtol 2e-4; texture gradient by the ``grad_close`` rule of test_engine_gpu.py
(|err| <= 1e-3 |ref| + 2e-4 max|ref| on >= 97 % of the TOUCHED texels - counted over the texels some view pixel maps
to; on the others the data term must be exactly zero here and the oracle's gradient the regulariser's alone - and
<= 2e-2 max|ref| everywhere). The measured fraction of
texels beyond the tight bound (max-pool argmax flips) is printed and written to
gpurun_out/fullsize_parity.json (copied to profiles/ by the builder).

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
STYLE_HW = (764, 600)   # half of "The Scream" (1528 x 1200): the style image only sets constants (Gram targets)
ROOM = (12.0, 9.0, 3.0)
CASES = {
    "c3": dict(tex=4096, level_hw=S.SCANNET_LEVEL_HW, view_hw=S.SCANNET_VIEW_HW, mode="multi", thr=30.0, angle=True,
               depth=True, min_depth=0.25, seed=2, active=[0, 1, 2, 3]),
    "c2": dict(tex=2048, level_hw=[S.SCANNET_VIEW_HW], view_hw=S.SCANNET_VIEW_HW, mode="single", thr=3000.0, angle=False,
               depth=False, min_depth=0.25, seed=2, active=[0]),
    "c5": dict(tex=4096, level_hw=S.MATTERPORT_LEVEL_HW, view_hw=S.MATTERPORT_VIEW_HW, mode="multi", thr=40.0, angle=True,
               depth=True, min_depth=0.2, seed=2, active=None),
}


def seeded_texture(tex, n_layers=4, amp=60.0):
    """Smooth, non-constant texture (a constant one makes every max-pool window an exact tie)."""
    rng = np.random.default_rng(0)
    return [torch.from_numpy(((S.smooth_noise(rng, 3, tex >> i, tex >> i, cells=64) - 0.5) * amp).astype(np.float32))
            for i in range(n_layers)]


def _record(name, entry):
    out_dir = os.path.join(REPO, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "fullsize_parity.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[name] = entry
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.parametrize("name", list(CASES))
def test_one_step_matches_oracle_at_full_size(name):
    require_gpu()
    from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
    c = CASES[name]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    vgg = S.seeded_vgg_state(0)
    style = S.style_image(1, *STYLE_HW)
    view = S.make_view(c["seed"], view_hw=c["view_hw"], level_hw=c["level_hw"], level_heights=[h for h, _ in c["level_hw"]],
                       min_pyramid_depth=c["min_depth"], room=S.BoxRoom(ROOM))
    tex0 = seeded_texture(c["tex"])

    cfg = EngineConfig(tex_w=c["tex"], tex_h=c["tex"], hierarchical=True, n_layers=4, style_weights=STYLE_WEIGHTS,
                       angle_threshold=c["thr"], style_pyramid_mode=c["mode"], use_angle_weight=c["angle"],
                       use_depth_scaling=c["depth"], loss_weights=dict(LOSS_WEIGHTS), learning_rate=1.0, decay_step_size=3)
    eng = StepEngine(cfg, vgg)
    eng.load_texture(tex0)
    eng.set_style_image(style)
    eng.set_view(view)
    active = [lv.index for lv in eng.view if lv.active]
    if c["active"] is not None:
        assert active == c["active"]
    assert len(active) >= 1
    eng.arena.g.zero_()
    lt = eng.loss_tensors()
    eng.forward_backward()
    mine = eng.losses(lt)
    g_mine = torch.cat([(g + k * p).reshape(-1) for g, k, p in zip(eng.grads, eng.reg_coef, eng.layers)]).cpu()
    g_data = eng.arena.g.cpu()
    # coverage: scatter gradient images of ones (no pixel weights) -> exactly the texels some pixel maps to are non-zero
    from stylemesh_amd.runtime import ops
    cover = torch.zeros_like(eng.arena.g)
    for lv in eng.view:
        if lv.active:
            b = eng._level_bufs(lv.H, lv.W)
            ones = type(b.grad["img"])(3, lv.H, lv.W).from_dense(torch.ones(3, lv.H, lv.W))
            ops.tex_sample_bwd(eng.arena.views(cover), lv.grid, ones, None)
    touched = (cover != 0).cpu()

    ocfg = O.OracleConfig(hierarchical=True, style_weights=STYLE_WEIGHTS, angle_threshold=c["thr"],
                          style_pyramid_mode=c["mode"], use_angle_weight=c["angle"], use_depth_scaling=c["depth"],
                          loss_weights=dict(LOSS_WEIGHTS), learning_rate=1.0, decay_step_size=3)
    pipe = O.OraclePipeline(vgg, style, ocfg, (c["tex"], c["tex"]), init_layers=tex0)
    rec = {}
    t0 = time.time()
    ref_losses, ref_grads = pipe.grads(view, rec)
    oracle_s = time.time() - t0
    assert rec["active"] == active
    for k in ("content", "style", "tex_reg", "total"):
        np.testing.assert_allclose(mine[k], float(ref_losses[k]), rtol=2e-4, err_msg=f"{name} loss {k}")

    g_ref = torch.cat([g.reshape(-1) for g in ref_grads])
    reg = torch.cat([(k * p).reshape(-1) for k, p in zip(eng.reg_coef, tex0)])
    # texels no pixel of the view maps to: an exactly-zero data term here, the regulariser's gradient alone there
    assert (~touched).any() and float(g_data[~touched].abs().max()) == 0.0
    assert float((g_ref - reg)[~touched].abs().max()) <= 1e-6 * float(reg.abs().max()) + 1e-12
    frac_touched = float(touched.float().mean())
    assert 0.0 < frac_touched < 0.6
    mx = float(g_ref.abs().max())
    err = (g_mine - g_ref).abs()
    bad = (err > 1e-3 * g_ref.abs() + 2e-4 * mx) & touched
    flip_frac = float(bad.sum()) / float(touched.sum())
    entry = {"texels": int(g_ref.numel()), "touched_fraction": round(frac_touched, 5),
             "fraction_of_touched_texels_beyond_tight_bound": flip_frac, "max_err_over_max_ref": float(err.max()) / mx,
             "max_ref": mx, "active_levels": active, "oracle_seconds": round(oracle_s, 1),
             "loss_rel_err": {k: abs(mine[k] - float(ref_losses[k])) / max(abs(float(ref_losses[k])), 1e-30)
                              for k in ("content", "style", "tex_reg", "total")}}
    print(f"\n[{name}] {json.dumps(entry)}")
    _record(name, entry)
    assert flip_frac <= 0.03, f"{name}: {flip_frac:.4f} of the touched texels beyond 1e-3|ref| + 2e-4 max|ref|"
    assert float(err.max()) <= 2e-2 * mx, f"{name}: max err {float(err.max()):.3e} vs max|ref| {mx:.3e}"
