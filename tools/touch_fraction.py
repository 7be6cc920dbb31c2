"""Fraction of the gradient arena a bench view can touch, per chunk size (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stylemesh_amd.data import synthetic as S
from stylemesh_amd.runtime.engine import EngineConfig, StepEngine

wl = bench.WORKLOADS["c3"]
cfg = EngineConfig(tex_w=wl["tex"], tex_h=wl["tex"], hierarchical=True, n_layers=4, style_weights=bench.STYLE_WEIGHTS,
                   angle_threshold=wl["thr"], style_pyramid_mode=wl["mode"], use_angle_weight=wl["angle"],
                   use_depth_scaling=wl["depth"], loss_weights=dict(bench.LOSS_WEIGHTS), learning_rate=1.0, decay_step_size=3)
eng = StepEngine(cfg, S.seeded_vgg_state(0), device="cuda")
eng.set_style_image(S.style_image(1, *bench.STYLE_HW))
views = bench.make_views(wl, [0, 2, 6, 7])
union = {}
for i, v in enumerate(views):
    eng.set_view(bench.to_device(v, torch.device("cuda")))
    for cl in (4, 5, 6):
        f = eng.touch_flags(cl)
        union[cl] = f if cl not in union else torch.maximum(union[cl], f)
        print(f"view {i} chunk 2^{cl}: touched {float(f.float().mean()):.3f}   union so far {float(union[cl].float().mean()):.3f}")
    # exact element-level fraction: run one step and count non-zero gradient elements before the update
    eng.arena.g.zero_(); eng.forward_backward()
    print(f"   exact non-zero gradient fraction {float((eng.arena.g != 0).float().mean()):.3f}")
