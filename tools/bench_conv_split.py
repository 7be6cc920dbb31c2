"""f32-MFMA conv vs bf16x3-split conv on the VGG layer shapes: accuracy against an fp64 convolution and speed
(run on the GPU box). Usage: bench_conv_split.py [level ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap

LAYERS = [(64, 64, 1), (128, 64, 2), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4), (256, 512, 8), (512, 512, 8), (512, 512, 16)]
LEVELS = [(256, 341), (432, 576), (608, 811), (784, 1045)]
sel = [int(a) for a in sys.argv[1:]] or [0, 3]
tot = {"f32": [0.0, 0.0], "split": [0.0, 0.0], "split2": [0.0, 0.0]}
for li in sel:
    H0, W0 = LEVELS[li]
    for cin, cout, div in LAYERS:
        H, W = H0 // div, W0 // div
        torch.manual_seed(cin + cout + H)
        xd = F.relu(torch.randn(cin, H, W, device="cuda") * 3)
        wgt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(cout, device="cuda") * 0.3
        x = FMap(cin, H, W).from_dense(xd)
        w = ops.pack_conv_fwd(wgt)
        w3 = ops.pack_conv_split(w)
        w2 = ops.pack_conv_split2(w)
        amax_in = ops.new_amax("cuda", float(xd.abs().max()))
        amax_out = ops.new_amax("cuda")
        out = FMap(cout, H, W)
        ref = F.relu(F.conv2d(xd[None].double(), wgt.double(), b.double(), padding=1))[0]
        line = f"{H0}x{W0} {cin:3d}->{cout:3d} {H:4d}x{W:4d}"
        for mode in ("f32", "split", "split2"):
            ops.CONV_MODE = mode
            kw = dict(wt3=w3, wt2=w2, amax_in=amax_in, amax_out=amax_out) if mode == "split2" else dict(wt3=w3)
            for _ in range(2): ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU, **kw)
            err = (out.to_dense().double() - ref).abs()
            torch.cuda.synchronize()
            n = 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            fl = 2.0 * 9 * cin * cout * H * W
            tot[mode][0] += us; tot[mode][1] += fl
            line += f" | {mode}: {us:7.1f} us {fl/us/1e6:6.1f} TF  max {float(err.max()/ref.abs().max()):.1e} rms {float((err**2).mean().sqrt()/(ref**2).mean().sqrt()):.1e}"
        print(line, flush=True)
for mode, (t, f) in tot.items():
    print(f"{mode}: {t/1e3:.2f} ms total, {f/t/1e6:.1f} TF/s")
