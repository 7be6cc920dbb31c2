"""Micro-benchmark of sm_conv3x3 on the VGG layer shapes of the four ScanNet UV levels (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap

LAYERS = [(3, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4), (256, 512, 8), (512, 512, 8), (512, 512, 16)]
LEVELS = [(256, 341), (432, 576), (608, 811), (784, 1045)]
which = sys.argv[1:] 
tot_t = tot_f = 0.0
for (H0, W0) in LEVELS:
    for cin, cout, div in LAYERS:
        H, W = H0 // div, W0 // div
        x = FMap(max(cin, 4), H, W); x.planes.normal_()
        w = ops.pack_conv_fwd(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05)
        b = torch.randn(cout, device="cuda")
        out = FMap(cout, H, W)
        for _ in range(2): ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU)
        torch.cuda.synchronize()
        n = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        fl = 2.0 * 9 * cin * cout * H * W
        mult = {(64,64,1):1,(128,128,2):1,(256,256,4):3,(512,512,8):3}.get((cin,cout,div),1)
        tot_t += us * mult; tot_f += fl * mult
        Wp = hip.row_stride(W); nt = (H * Wp + 127) // 128 * max(1, cout // 128)
        print(f"{H0}x{W0} {cin:3d}->{cout:3d} {H:4d}x{W:4d} blocks~{nt:5d}  {us:8.1f} us  {fl/us/1e6:6.1f} TF/s")
print(f"VGG fwd (13 convs x 4 levels): {tot_t/1e3:.2f} ms, {tot_f/tot_t/1e6:.1f} TF/s")
