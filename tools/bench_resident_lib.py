"""Helpers of bench_resident.py / res_occupancy.py: dense row-block lists, event timing."""
import numpy as np
import torch
from stylemesh_amd.runtime import hip

PAD = 0xFFFFFF


def rows_list(hws, rows, group):
    """Dense list: per level, blocks of `rows` consecutive image rows x 32 columns, one entry per row (row-major inside a
    block), each level padded to a multiple of `group` entries."""
    parts = []
    for g, (H, W) in enumerate(hws):
        Wp = hip.row_stride(W)
        Y = np.arange((H + rows - 1) // rows)[:, None, None] * rows
        X = np.arange(0, W, 32)[None, :, None]
        I = np.arange(rows)[None, None, :]
        y = Y + I + 0 * X
        q = (y + 1) * Wp + X + 1
        e = np.where(y < H, (g << 24) | q, (g << 24) | PAD).reshape(-1)
        pad = (-len(e)) % group
        parts.append(np.concatenate([e, np.full(pad, (g << 24) | PAD)]))
    return torch.tensor(np.concatenate(parts).astype(np.int32), device="cuda")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def interior(f):
    return f.to_dense()
