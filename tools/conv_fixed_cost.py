"""Fixed vs per-chunk cost of a small-grid conv launch (GPU box): the conv4_x geometry (512 output channels, 32 x 42) with
C_in = 16 k input channels, K-split forced off (SM_CONV_FORCE_SPLITS=1 in the environment) or left to the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
cout, h, w = 512, 32, 42
z = torch.zeros(1024, device="cuda")
print(f"reference: back-to-back sm_zero_floats of 4 KB: {timed(lambda: ops.zero_floats(z)):.1f} us per launch")
for cin in (16, 32, 64, 128, 256, 512):
    x = F.relu(torch.randn(cin, h, w, device="cuda"))
    wgt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    wf = ops.pack_conv_fwd(wgt); w2 = ops.pack_conv_split2(wf); b = torch.zeros(cout, device="cuda")
    xin, out = FMap(cin, h, w).from_dense(x), FMap(cout, h, w)
    am_in, am_out = ops.new_amax("cuda", float(x.abs().max())), ops.new_amax("cuda")
    t = timed(lambda: ops.conv3x3_grouped([(xin, out, None)], wf, b, hip.EPI_BIAS_RELU, None, 1.0, None, w2, am_in, am_out))
    print(f"C_in {cin:3d} ({cin // 16:2d} chunks): {t:6.1f} us")
