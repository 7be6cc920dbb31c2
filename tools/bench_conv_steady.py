"""Steady-state ceiling of the conv kernel: large-K layers on grids that are exact multiples of the CU count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
for cin, cout, H, W in [(512, 512, 128, 763), (256, 256, 128, 763), (64, 64, 256, 763), (512, 512, 64, 763), (128, 128, 256, 763)]:
    x = FMap(cin, H, W); x.planes.normal_()
    w = ops.pack_conv_fwd(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05)
    b = torch.randn(cout, device="cuda"); out = FMap(cout, H, W)
    Wp = hip.row_stride(W); tiles = (H * Wp + (127 if cout >= 128 else 255)) // (128 if cout >= 128 else 256) * max(1, cout // 128)
    for _ in range(2): ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 3
    print(f"{cin}->{cout} {H}x{W} tiles {tiles} ({tiles/256:.2f} rounds)  {us:9.1f} us  {2.0*9*cin*cout*H*W/us/1e6:6.1f} TF/s")
