"""Does the split conv's rate depend on the operand DATA (power management) or on its instruction stream?
Same launch on: zeros, a constant, ReLU'd normal data (the micro-benchmark input), dense normal data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
for cin, cout, H, W in [(256, 256, 196, 261), (512, 512, 98, 130), (64, 64, 784, 1045)]:
    wgt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    for wname, wg in (("w=randn", wgt), ("w=0", wgt * 0)):
        w = ops.pack_conv_fwd(wg)
        w2 = ops.pack_conv_split2(w) if wname == "w=randn" else (ops.pack_conv_split2(ops.pack_conv_fwd(wgt))[0] * 0, 1.0)
        b = torch.zeros(cout, device="cuda")
        for name, xd in (("zeros", torch.zeros(cin, H, W, device="cuda")), ("const 1", torch.ones(cin, H, W, device="cuda")),
                         ("relu(randn)", F.relu(torch.randn(cin, H, W, device="cuda") * 3)), ("randn", torch.randn(cin, H, W, device="cuda") * 3)):
            x = FMap(cin, H, W).from_dense(xd)
            out = FMap(cout, H, W)
            amax_in, amax_out = ops.new_amax("cuda", max(1e-3, float(xd.abs().max()))), ops.new_amax("cuda")
            kw = dict(wt2=w2, amax_in=amax_in, amax_out=amax_out)
            for _ in range(3): ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            print(f"{cin}->{cout} {H}x{W} {wname:8s} x={name:12s} {us:7.1f} us {2.0*9*cin*cout*H*W/us/1e6:6.1f} TF/s fp32-eq", flush=True)
