"""Padded-planar feature maps in HBM (layout contract: include/stylemesh_hip.h).

``[C][plane]`` fp32; a plane holds ``H+2`` rows of ``Wp = round_up(W+1, 4)`` floats; pixel ``(y, x)`` sits at
``q = (y+1)*Wp + (x+1)``. The border (row 0, row H+1, column 0, columns > W) is zero and stays zero, which
turns the 3x3 convolution's im2col into pure offsets in ``q``. Each buffer carries ``SM_FMAP_GUARD`` floats of
zeroed slack on both sides so kernels may read a tile's halo without bounds checks.
"""
from __future__ import annotations

import torch

from . import hip


class FMap:
    __slots__ = ("C", "H", "W", "Wp", "plane", "buf", "ptr")

    def __init__(self, C: int, H: int, W: int, device="cuda"):
        self.C, self.H, self.W = C, H, W
        self.Wp = hip.row_stride(W)
        self.plane = hip.plane(H, W)
        self.buf = torch.zeros(2 * hip.SM_FMAP_GUARD + C * self.plane, dtype=torch.float32, device=device)
        self.ptr = self.buf.data_ptr() + 4 * hip.SM_FMAP_GUARD

    @property
    def planes(self) -> torch.Tensor:
        """View [C][plane] of the payload (shares memory)."""
        g = hip.SM_FMAP_GUARD
        return self.buf[g:g + self.C * self.plane].view(self.C, self.plane)

    def channel_ptr(self, c: int) -> int:
        return self.ptr + 4 * c * self.plane

    def to_dense(self, channels=None) -> torch.Tensor:
        """Interior as a dense [C,H,W] tensor (copy; tests / export only)."""
        C = self.C if channels is None else channels
        v = self.planes[:C, :(self.H + 2) * self.Wp].view(C, self.H + 2, self.Wp)
        return v[:, 1:self.H + 1, 1:self.W + 1].contiguous()

    def from_dense(self, x: torch.Tensor):
        """Fill the interior from a dense [C',H,W] tensor, C' <= C (tests / one-time setup only)."""
        c = x.shape[0]
        v = self.planes[:c, :(self.H + 2) * self.Wp].view(c, self.H + 2, self.Wp)
        v[:, 1:self.H + 1, 1:self.W + 1] = x.to(self.buf.device, torch.float32)
        return self

    def border_is_zero(self) -> bool:
        v = self.planes[:, :(self.H + 2) * self.Wp].view(self.C, self.H + 2, self.Wp)
        return bool((v[:, 0] == 0).all() and (v[:, self.H + 1] == 0).all() and (v[:, :, 0] == 0).all()
                    and (v[:, :, self.W + 1:] == 0).all())

    def zero_(self):
        self.buf.zero_()
        return self
