"""GPU time of a view change's preparation, per kernel family (GPU box): set_view of c3 / dip-sized views under torch's
profiler-free HIP events around ScatterPlan.build (the sort) and the whole prepare. Usage: time_setview.py [c3|c2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylemesh_amd.data import synthetic as S
from stylemesh_amd.runtime import ops
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
level_hw = S.SCANNET_LEVEL_HW if wl == "c3" else [S.SCANNET_VIEW_HW]
tex, n_layers = (4096, 4) if wl == "c3" else (4096, 1)
shapes = [(3, tex >> i, tex >> i) for i in range(n_layers)]
n = sum(c * h * w for c, h, w in shapes)
g = torch.zeros(n, device="cuda")
layers, off = [], 0
for c, h, w in shapes:
    layers.append(g[off:off + c * h * w].view(c, h, w)); off += c * h * w
plan = ops.ScatterPlan(layers, g)
view = S.make_view(3, view_hw=S.SCANNET_VIEW_HW, level_hw=level_hw, level_heights=[h for h, _ in level_hw], min_pyramid_depth=0.25,
                   room=S.BoxRoom((12.0, 9.0, 3.0)))
grids = [u[0].cuda().contiguous() for u in view[9]]
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan.build(grids, [None] * len(grids))
    e1.record(); torch.cuda.synchronize()
    print(f"{wl}: scatter plan (entries + sort + crossing runs) of {plan.n_entries / 1e6:.1f} M entries: {e0.elapsed_time(e1) * 1e3:.0f} us")
