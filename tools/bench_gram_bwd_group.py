"""Grouped Gram backward (fp16x2) over the c3 style-layer shapes of all four UV levels (run on the GPU box)."""
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
        D0 = torch.randn(C, C, device="cuda"); D0 = D0 + D0.T; D1 = torch.randn(C, C, device="cuda"); D1 = D1 + D1.T
        af = ops.new_amax("cuda", float(f.planes.abs().max()))
        ad = ops.new_amax("cuda", float(torch.maximum(D0.abs().max(), D1.abs().max())))
        df = FMap(C, H, W)
        ws = torch.empty(ops.gram_backward_ws_bytes(C), dtype=torch.uint8, device="cuda")
        keep.append((f, m0, m1, D0, D1, af, ad, df, ws))
        probs[(li, la)] = ops.gram_bwd_problem(f, m0, m1, D0, D1, df, ws, af, ad, relu_gate=(la == 4))

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

def bwd(keys):
    arr = ops.struct_array(hip.GramBwdProblem, [probs[k] for k in keys])
    return timed(lambda: ops.gram_backward_grouped(arr))
allk = list(probs)
print(f"mask coverage {cover}")
print(f"bwd all 20 problems: {bwd(allk):7.1f} us")
for la, (C, div) in enumerate(LAYERS):
    print(f"bwd layer {la} (C={C:3d}, 4 levels): {bwd([k for k in allk if k[1] == la]):7.1f} us")
print(f"bwd 128-row class (layers 1-4): {bwd([k for k in allk if k[1] >= 1]):7.1f} us")
