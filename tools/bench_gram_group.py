"""Grouped Gram forward / backward (fp16x2) over the c3 style-layer shapes of all four UV levels: whole group, per tile
class and per layer (run on the GPU box). Usage: bench_gram_group.py [mask coverage = 0.8]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap

LAYERS = [(64, 1), (128, 2), (256, 4), (512, 8), (512, 16)]
LEVELS = [(256, 341), (432, 576), (608, 811), (784, 1045)]
cover = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
ops.GRAM_MODE = "split2"
probs, keep = {}, []
for li, (H0, W0) in enumerate(LEVELS):
    for la, (C, div) in enumerate(LAYERS):
        H, W = H0 // div, W0 // div
        f = FMap(C, H, W).from_dense(torch.relu(torch.randn(C, H, W, device="cuda")))
        yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
        m_all = ((yy >= (1 - cover) * H / 2) & (yy < H - (1 - cover) * H / 2)).float()
        passed = (xx < 0.45 * W).float()
        m0 = FMap(1, H, W).from_dense((m_all * passed)[None]); m1 = FMap(1, H, W).from_dense((m_all * (1 - passed))[None])
        S0 = torch.zeros(C, C, device="cuda"); S1 = torch.zeros(C, C, device="cuda")
        af = ops.new_amax("cuda", float(f.planes.abs().max()))
        keep.append((f, m0, m1, S0, S1, af))
        probs[(li, la)] = ops.gram_problem(f, m0, m1, S0, S1, af)

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

def fwd(keys):
    arr = ops.gram_problem_array([probs[k] for k in keys])
    return timed(lambda: ops.gram_masked_grouped(arr))

allk = list(probs)
mb = lambda keys: sum(keep[li * 5 + la][0].C * keep[li * 5 + la][0].H * keep[li * 5 + la][0].W * 4 for li, la in keys) / 1e6
print(f"mask coverage {cover}")
print(f"fwd all 20 problems: {fwd(allk):7.1f} us   ({mb(allk):.0f} MB of feature maps)")
for la, (C, div) in enumerate(LAYERS):
    ks = [k for k in allk if k[1] == la]
    print(f"fwd layer {la} (C={C:3d}, 4 levels): {fwd(ks):7.1f} us   ({mb(ks):.0f} MB)")
ks = [k for k in allk if k[1] >= 1]
print(f"fwd 128-channel tile class (layers 1-4): {fwd(ks):7.1f} us")
for li in range(4):
    ks = [k for k in allk if k[0] == li]
    print(f"fwd level {li} (5 layers): {fwd(ks):7.1f} us")
