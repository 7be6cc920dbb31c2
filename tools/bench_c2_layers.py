"""The twelve fp16x2 conv layer shapes of the single-level c2 step (256 x 341 input), forward, conv + tail second pass,
for forced tail split counts (SM_CONV_FORCE_SPLITS): which decomposition does a small grid want? (GPU box)
Usage: bench_c2_layers.py [H W]   (run once per SM_CONV_FORCE_SPLITS value; 0 / unset = the library's own choice)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 341)
LAYERS = [("conv1_2", 64, 64, 1), ("conv2_1", 64, 128, 2), ("conv2_2", 128, 128, 2), ("conv3_1", 128, 256, 4),
          ("conv3_2", 256, 256, 4), ("conv4_1", 256, 512, 8), ("conv4_2", 512, 512, 8), ("conv5_1", 512, 512, 16)]
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tag = os.environ.get("SM_CONV_FORCE_SPLITS", "auto")
tot = 0.0
for name, cin, cout, div in LAYERS:
    h, w = H // div, W // div
    x = F.relu(torch.randn(cin, h, w, device="cuda"))
    wgt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    wf = ops.pack_conv_fwd(wgt)
    w2 = ops.pack_conv_split2(wf)
    b = torch.zeros(cout, device="cuda")
    xin, out = FMap(cin, h, w).from_dense(x), FMap(cout, h, w)
    am_in, am_out = ops.new_amax("cuda", float(x.abs().max())), ops.new_amax("cuda")
    t = timed(lambda: ops.conv3x3_grouped([(xin, out, None)], wf, b, hip.EPI_BIAS_RELU, None, 1.0, None, w2, am_in, am_out))
    gf = 2.0 * 9 * cin * cout * h * w / 1e9
    tot += t
    print(f"S={tag:>4} {name} {cin:3d}->{cout:3d} {h:3d}x{w:3d}: {t:6.1f} us  {gf / t * 1e3:6.1f} TFLOP/s")
print(f"S={tag:>4} sum {tot:.1f} us")
