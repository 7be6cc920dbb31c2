import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from stylemesh_amd.data import synthetic as S
from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
wl = B.WORKLOADS["c3"]
cfg = EngineConfig(tex_w=4096, tex_h=4096, style_weights=B.STYLE_WEIGHTS, angle_threshold=30.0, style_pyramid_mode="multi", loss_weights=dict(B.LOSS_WEIGHTS))
eng = StepEngine(cfg, S.seeded_vgg_state(0)); eng.set_style_image(S.style_image(1, *B.STYLE_HW))
views = [B.to_device(v, "cuda") for v in B.make_views(wl, [0, 2, 6])]
for sparse in (True, False):
    eng.sparse_tiles = sparse
    for v in views: eng.set_view(v); eng.training_step(v)
    torch.cuda.synchronize()
    ts = []
    for v in views * 2:
        t0 = time.perf_counter(); eng.set_view(v); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("sparse", sparse, "set_view ms:", [round(1e3 * t, 2) for t in ts])
