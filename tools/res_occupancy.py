"""Resident-input kernel: launch time against the number of tiles (256 CUs) - how many blocks does a CU hold at a time?
conv1_2 forward + pool over one level of W = 512 and H = 64 k (= 256 k tiles of 4 x 32 positions).   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
from bench_resident_lib import rows_list, timed

ops.CONV_MODE = "split2"
wgt = torch.randn(64, 64, 3, 3) * (2.0 / (9 * 64)) ** 0.5
b = (torch.randn(64) * 0.3).cuda()
w = ops.pack_conv_fwd(wgt).cuda()
w2 = ops.pack_conv_split2(w)
for k in (1, 2, 3, 4, 6, 9, 12):
    H, W = 64 * k, 512
    x = FMap(64, H, W).from_dense(F.relu(torch.randn(64, H, W, device="cuda")))
    amax_in = ops.new_amax("cuda", float(x.planes.abs().max()))
    out, pooled = FMap(64, H, W), FMap(64, H // 2, W // 2)
    codes = torch.zeros(8 * pooled.plane, dtype=torch.int32, device="cuda")
    for name, lst, quads in (("ring", rows_list([(H, W)], 2, 8), False), ("resident", rows_list([(H, W)], 4, 4), True)):
        t = timed(lambda: ops.conv3x3_grouped([(x, out, None, None, pooled, codes)], w, b, hip.EPI_BIAS_RELU | hip.EPI_POOL, lst,
                                              1.0, None, w2, amax_in, ops.new_amax("cuda"), quads=quads), n=50)
        print(f"{256 * k:5d} quads ({H} x {W}) {name:9s}: {t:7.1f} us")
