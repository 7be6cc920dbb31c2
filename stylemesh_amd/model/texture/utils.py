"""UV <-> sampling-grid conversions (reference model/texture/utils.py:6-60). Pure tensor reshapes: the UV map of
a view is an (H,W,3) image (u, v, mip-LOD) in [0,1]; ``grid_sample`` wants (H,W,2) in [-1,1]."""
import torch


def to_grid_range(x):
    """[0,1] -> [-1,1]"""
    return (x * 2.0) - 1


def from_grid_range(x):
    """[-1,1] -> [0,1]"""
    return (x + 1) / 2.0


def cut_b_channel(x):
    return x[:2]


def add_b_channel(x):
    return torch.cat((x, torch.full_like(x[0], -1).unsqueeze(0)), dim=0)


def chw_to_hwc(x):
    return x.permute(1, 2, 0) if len(x.shape) == 3 else x.permute(0, 2, 3, 1)


def hwc_to_chw(x):
    return x.permute(2, 0, 1) if len(x.shape) == 3 else x.permute(0, 3, 1, 2)


def to_grid_format(x):
    return chw_to_hwc(cut_b_channel(x))


def from_grid_format(x):
    return add_b_channel(hwc_to_chw(x))


def to_grid(x):
    """UV map as CHW tensor image -> grid valid for grid_sample"""
    return to_grid_format(to_grid_range(x))


def from_grid(x):
    return from_grid_range(from_grid_format(x))
