"""Event representation on the device: the producer of the uint8 event tensors the backbone consumes.

Mirror of the reference ``StackedHistogram`` (data/utils/representations.py:36-117): same constructor arguments, same
``construct(x, y, pol, time) -> uint8 (2*bins, H, W)``, ``get_shape`` / dtype helpers — computed by one HIP scatter
kernel + one clamp/narrow pass (rvt_stacked_histogram, rvt_amd/csrc/events.hpp) instead of ``put_(accumulate=True)``.
Integer work: bit-identical to the reference, including its accumulator wrap-around for hot pixels.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib as L


class StackedHistogram:
    def __init__(self, bins: int, height: int, width: int, count_cutoff: Optional[int] = None, fastmode: bool = True):
        assert bins >= 1 and height >= 1 and width >= 1
        self.bins, self.height, self.width = bins, height, width
        if count_cutoff is None:                              # representations.py:52-57
            self.count_cutoff = 255
        else:
            assert count_cutoff >= 1
            self.count_cutoff = min(count_cutoff, 255)
        self.fastmode = fastmode
        self.channels = 2
        self._scratch = None

    @staticmethod
    def get_numpy_dtype() -> np.dtype:
        return np.dtype('uint8')

    @staticmethod
    def get_torch_dtype() -> torch.dtype:
        return torch.uint8

    @property
    def dtype(self) -> torch.dtype:
        return torch.uint8

    def get_shape(self) -> Tuple[int, int, int]:
        return 2 * self.bins, self.height, self.width

    def construct(self, x: torch.Tensor, y: torch.Tensor, pol: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        dev = x.device
        assert y.device == pol.device == time.device == dev
        for t in (x, y, pol, time):
            assert not torch.is_floating_point(t) and not torch.is_complex(t)      # representations.py:78-81
        assert x.numel() == y.numel() == pol.numel() == time.numel()
        cells = 2 * self.bins * self.height * self.width
        if self._scratch is None or self._scratch.device != dev:
            self._scratch = torch.empty(cells, dtype=torch.int32, device=dev)
        out = torch.empty(self.get_shape(), dtype=torch.uint8, device=dev)
        x, y, pol, time = (t.to(torch.int64).contiguous() for t in (x, y, pol, time))
        L.call('rvt_stacked_histogram', L.ptr(x), L.ptr(y), L.ptr(pol), L.ptr(time), x.numel(), self.bins, self.height,
               self.width, self.count_cutoff, int(self.fastmode), L.ptr(self._scratch), L.ptr(out), L.stream_of(out))
        return out
