"""One conv shape, one mode, a few launches (for rocprofv3 --kernel-trace --stats). Usage: one_conv.py mode cin cout H W"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
mode, cin, cout, H, W = sys.argv[1], *[int(a) for a in sys.argv[2:6]]
torch.manual_seed(0)
xd = F.relu(torch.randn(cin, H, W, device="cuda") * 3)
wgt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
b = torch.randn(cout, device="cuda") * 0.3
x = FMap(cin, H, W).from_dense(xd)
w = ops.pack_conv_fwd(wgt)
w3, w2 = ops.pack_conv_split(w), ops.pack_conv_split2(w)
amax_in = ops.new_amax("cuda", float(xd.abs().max()))
amax_out = ops.new_amax("cuda")
out = FMap(cout, H, W)
ops.CONV_MODE = mode
kw = dict(wt3=w3, wt2=w2, amax_in=amax_in, amax_out=amax_out) if mode == "split2" else dict(wt3=w3)
for _ in range(10):
    ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU, **kw)
torch.cuda.synchronize()
