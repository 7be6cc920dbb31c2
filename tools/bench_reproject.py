"""Throughput of sm_reproject against its HBM traffic (run on the GPU box): per pair 5 + 3 + 1 planes read, 4 written."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stylemesh_amd.data import synthetic as S
from stylemesh_amd.eval import reprojection as R

for hw in ((240, 320), (480, 640), (968, 1296)):
    room = S.BoxRoom()
    K, c2w = S.camera_matrices((2.5, 2.0, 1.4), 0.7, 0.0, hw)
    _, c2w2 = S.camera_matrices((2.7, 2.1, 1.4), 0.8, 0.02, hw)
    d1 = torch.from_numpy(room.render((2.5, 2.0, 1.4), 0.7, 0.0, hw)[2]).cuda()
    d2 = torch.from_numpy(room.render((2.7, 2.1, 1.4), 0.8, 0.02, hw)[2]).cuda()
    a, b = torch.rand(3, *hw, device="cuda"), torch.rand(3, *hw, device="cuda")
    K, c2w, c2w2 = (torch.from_numpy(x).cuda() for x in (K, c2w, c2w2))
    acc = R.ReprojectionError()
    for _ in range(3): acc.update(a, c2w, d1, b, c2w2, d2, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): acc.update(a, c2w, d1, b, c2w2, d2, K)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    byt = hw[0] * hw[1] * 4 * (1 + 1 + 3 + 1 + 3 + 3) + hw[0] * hw[1]
    print(f"{hw[0]}x{hw[1]}: {us:7.1f} us per pair (incl. host-side pose maths + launch), {byt/us/1e3:.1f} GB/s algorithmic, mse {acc.compute():.5f}")
