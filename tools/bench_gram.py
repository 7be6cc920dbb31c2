"""f32-MFMA vs bf16x3-split Gram forward / backward on the c3 style-layer shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap

LAYERS = [(64, 1), (128, 2), (256, 4), (512, 8), (512, 16)]
LEVELS = [(256, 341), (432, 576), (608, 811), (784, 1045)]
sel = [int(a) for a in sys.argv[1:]] or [0, 3]
tot = {m: [0.0, 0.0] for m in ("f32", "split", "split2")}
group, keep = [], []
for li in sel:
    H0, W0 = LEVELS[li]
    for C, div in LAYERS:
        H, W = H0 // div, W0 // div
        torch.manual_seed(C + H)
        f = FMap(C, H, W).from_dense(torch.relu(torch.randn(C, H, W, device="cuda")))
        # masks like a UV level: a blob covering ~40 % of the image, split into passed / failed halves
        yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
        m_all = ((yy > 0.2 * H) & (yy < 0.85 * H) & (xx > 0.1 * W) & (xx < 0.7 * W)).float()
        passed = (xx < 0.45 * W).float()
        m0 = FMap(1, H, W).from_dense((m_all * passed)[None]); m1 = FMap(1, H, W).from_dense((m_all * (1 - passed))[None])
        na = max(1, hip.lib.sm_gram_workspace_slabs(C, H, W))   # enough for either mode
        S0 = torch.zeros(na, C, C, device="cuda"); S1 = torch.zeros(na, C, C, device="cuda")
        D0 = torch.randn(C, C, device="cuda"); D0 = D0 + D0.T; D1 = torch.randn(C, C, device="cuda"); D1 = D1 + D1.T
        df = FMap(C, H, W)
        line = f"{H0}x{W0} C={C:3d} {H:4d}x{W:4d}"
        res = {}
        af = ops.new_amax("cuda", float(f.planes.abs().max()))
        ad = ops.new_amax("cuda", float(torch.maximum(D0.abs().max(), D1.abs().max())))
        for mode in ("f32", "split", "split2"):
            ops.GRAM_MODE = mode
            for which, fn in (("fwd", lambda: ops.gram_masked(f, m0, m1, S0, S1, amax_feat=af)),
                              ("bwd", lambda: ops.gram_backward(f, m0, m1, D0, D1, df, relu_gate=False, amax_feat=af, amax_d=ad))):
                for _ in range(2): fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 5
                res[(mode, which)] = us
                tot[mode][0 if which == "fwd" else 1] += us
            line += f" | {mode}: fwd {res[(mode,'fwd')]:7.1f} us  bwd {res[(mode,'bwd')]:7.1f} us"
        print(line, flush=True)
        group.append(ops.gram_problem(f, m0, m1, S0, S1, af)); keep.append((f, m0, m1, S0, S1, af))
for m, (a, b) in tot.items():
    print(f"{m}: fwd {a/1e3:.3f} ms  bwd {b/1e3:.3f} ms")
arr = ops.gram_problem_array(group)
for _ in range(2): ops.gram_masked_grouped(arr)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.gram_masked_grouped(arr)
e1.record(); torch.cuda.synchronize()
print(f"split2 grouped fwd over the {len(group)} problems above: {e0.elapsed_time(e1) / 5:.3f} ms")
