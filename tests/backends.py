"""Test backends: 'hip' = the real gfx950 library on cuda:0 (tests marked gpu);
'emu' = the CPU SIMT-emulator build of the SAME kernel sources (tests/emu), for GPU-less CI."""
import ctypes
import os
import subprocess

import pytest
import torch

from rvt_amd import _lib

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu')
EMU_SO = os.path.join(EMU_DIR, 'librvt_emu.so')
CSRC = os.path.join(os.path.dirname(EMU_DIR), '..', 'rvt_amd', 'csrc')


def _emu_stale() -> bool:
    if not os.path.exists(EMU_SO):
        return True
    t = os.path.getmtime(EMU_SO)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(EMU_DIR, f) for f in ('hip_emu.h', 'emu.cpp')]
    return any(os.path.getmtime(s) > t for s in srcs if os.path.isfile(s))


_emu_handle = None


def emu_library():
    global _emu_handle
    if _emu_handle is None:
        if _emu_stale():
            subprocess.run(['bash', os.path.join(EMU_DIR, 'build_emu.sh')], check=True, capture_output=True)
        _emu_handle = _lib._bind(ctypes.CDLL(EMU_SO))
    return _emu_handle


BACKENDS = [pytest.param('emu', id='emu'), pytest.param('hip', id='hip', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
    """Yields the torch device to allocate test tensors on."""
    if request.param == 'emu':
        _lib._install_test_library(emu_library())
        yield torch.device('cpu')
        _lib._install_test_library(None)
    else:
        _lib._install_test_library(None)
        yield torch.device('cuda', 0)
