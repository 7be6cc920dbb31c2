"""Would ONE Gram-forward block for both masks of a style layer (instead of a block per mask) save reads? Stages of 64 positions
that are live in the passed mask, the failed mask, both, either - three c3 bench views, all levels.   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from stylemesh_amd.runtime.engine import StepEngine
wl = B.WORKLOADS["c3"]
eng = StepEngine(B.engine_config(wl), B.S.seeded_vgg_state(0))
eng.set_style_image(B.S.style_image(1, *B.STYLE_HW))
tot = {}
for seed in (0, 2, 6):
    eng.set_view(B.to_device(B.make_views(wl, [seed])[0], "cuda"))
    for lv in eng.view:
        if not lv.active: continue
        for layer, m in lv.masks.items():
            p = m.planes[1:3]          # passed, failed
            n = p.shape[1] // 64 * 64
            c = (p[:, :n].reshape(2, -1, 64) != 0).any(-1)
            a = tot.setdefault(layer, [0, 0, 0, 0])
            a[0] += int(c[0].sum()); a[1] += int(c[1].sum()); a[2] += int((c[0] & c[1]).sum()); a[3] += int((c[0] | c[1]).sum())
for k, (a, b, both, any_) in tot.items():
    print(k, "stages live: passed", a, "failed", b, "both", both, "either", any_, " reads now / fused = %.2f" % ((a + b) / max(any_, 1)))
