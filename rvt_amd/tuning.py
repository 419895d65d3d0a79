"""The launch-geometry / routing record of librvt_hip.so (include/rvt_hip.h: RvtTuning, rvt_set_tuning).

Round 4: this record replaces ~25 RVT_* environment variables that used to be read once into function-local statics
in csrc/ and in stage.py.  Nothing in the package reads the environment any more:

  * the DEFAULTS are the production route — what bench.py times and a deployment runs;
  * the unit tests install TEST_GEOMETRY (tiny persistent grids, every kernel family forced on at test-size problems)
    with `tuning.use(**TEST_GEOMETRY)` and therefore SAY which route they exercise; the production-route parity tests
    (tests/test_production_route.py) run on the defaults;
  * A/B measurements (profiles/*.py) flip single fields with `tuning.override(...)`.

The record lives inside the loaded library (process-wide).  This module keeps the values the caller asked for and
pushes them into whichever library is active (the gfx950 build, or the emulator build the CPU tests install).
"""
from __future__ import annotations

import contextlib
import ctypes
from typing import Dict, Iterator

_GEOMETRY = ('gemm_resident', 'gemm_xcd_panels', 'wgrad_bn', 'wgrad_blocks', 'wgrad_slice_tokens', 'ppgemm', 'ppgemm_min_m',
             'ppgemm_all', 'ppgemm_grid', 'ppgemm_tn_items', 'one_per_cu_grid', 'stem', 'stem_depth', 'mlp_tm', 'mlp_chain',
             'mlp_chain_wgrad', 'chain_resident', 'attn_block_resident', 'dgrad_ln')
_LATE = ('lstm_scan_v2', 'route_stage_driver', 'route_mlp_store_pre', 'route_mlp_bwd_both', 'mlp_stream', 'ln_linear', 'conv_wgrad_tn', 'attn_staged', 'lstm_scan3', 'lstm_scan3_rb256', 'lstm_scan3_rb128', 'route_stage_driver_train', 'route_attn_preln', 'conv_fwd_pp')          # fields appended after the routes (struct order = _GEOMETRY + _ROUTES + _LATE)
_ROUTES = ('route_fused_mlp', 'route_mlp_bwd_fused', 'route_attn_block', 'route_lstm_scan', 'route_lstm_scan_wgrad',
           'route_conv_dgrad4', 'route_wgrad_stream')
FIELDS = _GEOMETRY + _ROUTES + _LATE


class RvtTuning(ctypes.Structure):
    """Mirror of `struct RvtTuning` (include/rvt_hip.h) — field order is part of the C ABI."""
    _fields_ = [('struct_bytes', ctypes.c_int)] + [(f, ctypes.c_int) for f in FIELDS]


# what the tests install: small grids so that test-size problems walk several tiles / K slices per workgroup, and every
# kernel family reachable at test sizes (production thresholds: >= 4096 rows for ppgemm, >= 8192 tokens per K slice, the
# ConvLSTM scan only where the weights stay on chip)
TEST_GEOMETRY: Dict[str, int] = dict(gemm_resident=8, wgrad_slice_tokens=128, route_lstm_scan=1, one_per_cu_grid=3,
                                     ppgemm_min_m=256, ppgemm_grid=8, ppgemm_tn_items=6, ppgemm_all=1)

_overrides: Dict[str, int] = {}          # what the caller asked for on top of the library defaults
_pushed_to = None                        # the library object that currently holds `_overrides`
_cache: Dict[str, int] = {}


def _defaults(lib) -> RvtTuning:
    t = RvtTuning()
    lib.rvt_tuning_defaults(ctypes.byref(t))
    return t


def push(lib) -> None:
    """Install the current overrides into `lib` (called by rvt_amd._lib whenever the active library changes)."""
    global _pushed_to, _cache
    t = _defaults(lib)
    for k, v in _overrides.items():
        setattr(t, k, int(v))
    if lib.rvt_set_tuning(ctypes.byref(t)) != 0:
        raise RuntimeError(f'rvt_set_tuning failed: {lib.rvt_last_error().decode()}')
    _pushed_to = lib
    _cache = {f: int(getattr(t, f)) for f in FIELDS}


def _active_lib():
    from . import _lib
    return _lib._lib if _lib._lib is not None else None


def use(**fields: int) -> None:
    """Replace the overrides (fields not named go back to the production defaults) and install them."""
    for k in fields:
        if k not in FIELDS:
            raise KeyError(f'unknown tuning field {k!r}; fields: {FIELDS}')
    _overrides.clear()
    _overrides.update({k: int(v) for k, v in fields.items()})
    lib = _active_lib()
    if lib is not None:
        push(lib)
    else:
        _cache.clear()


def production() -> None:
    """The production route: library defaults, no overrides."""
    use()


@contextlib.contextmanager
def override(**fields: int) -> Iterator[None]:
    """Temporarily change some fields on top of the current ones."""
    saved = dict(_overrides)
    try:
        use(**{**saved, **fields})
        yield
    finally:
        use(**saved)


def current() -> Dict[str, int]:
    """The record the active library holds (after the overrides)."""
    from . import _lib
    lib = _lib.get_lib()
    if _pushed_to is not lib or not _cache:
        push(lib)
    return dict(_cache)


def get(field: str) -> int:
    from . import _lib
    lib = _lib.get_lib()
    if _pushed_to is not lib or not _cache:
        push(lib)
    return _cache[field]


def overrides() -> Dict[str, int]:
    return dict(_overrides)
