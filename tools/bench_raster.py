"""Throughput of sm_raster_maps on a finely subdivided box room (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stylemesh_amd.data import synthetic as S
from stylemesh_amd import render as R

room = S.BoxRoom()
for subdiv in (16, 256):
    mesh = R.box_room_mesh(room, subdiv=subdiv)
    for hw in ((480, 640), (784, 1045)):
        K, c2w = S.camera_matrices((2.5, 2.0, 1.4), 0.7, 0.05, hw)
        intr = np.array([K[0, 0], K[1, 1], K[0, 2] + 0.5, K[1, 2] + 0.5], dtype=np.float32)
        for _ in range(3): R.render_maps(mesh, c2w, intr, hw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): R.render_maps(mesh, c2w, intr, hw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        F = mesh.faces.shape[0]
        print(f"{F:7d} triangles, {hw[0]}x{hw[1]}: {us:8.1f} us per frame  ({F/us:.1f} Mtri/s, {hw[0]*hw[1]/us:.1f} Mpix/s)")
