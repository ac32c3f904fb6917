"""The C-ABI library loads without a GPU and exports every symbol include/bfgx.h declares; the ctypes
binding covers exactly that set.  No compute entry point is called here."""
import os
import re

import numpy as np
import pytest

from baryonification_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(REPO, 'include', 'bfgx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(bfgx_[A-Za-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported_and_bound():
    names = header_functions()
    assert len(names) >= 15
    L = _lib.load()
    for n in names:
        assert hasattr(L, n), "libbfgx.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names, "ctypes binding and header disagree"


def test_abi_version_and_device_count_callable_without_gpu():
    L = _lib.load()
    assert L.bfgx_abi_version() == 4 == _lib.ABI_VERSION
    assert L.bfgx_device_count() >= 0


def test_struct_sizes_match_header_layout():
    import ctypes as C
    assert C.sizeof(_lib.bfgx_cosmo) == 64
    assert C.sizeof(_lib.bfgx_massdef) == 16
    assert C.sizeof(_lib.bfgx_table) == 4 + 4 * 7 + 8 * 7 + 8 + 4 + 4 + 8       # ndim, n[7], axis[7], values, rdelta, log, eps_model
    assert C.sizeof(_lib.bfgx_catalog) == 8 + 4 * 8 + 4 * 8 + 2 * 8       # n, M z ra dec, extra[4], ln1pz lnM
    assert C.sizeof(_lib.bfgx_opts) == 24 + 8                           # 6 x int32, catalog_token


def test_compute_fails_loudly_without_gpu():
    """no CPU fallback: with no device the plan cannot be created"""
    L = _lib.load()
    if L.bfgx_device_count() > 0:
        pytest.skip("GPU present")
    import ctypes as C
    from baryonification_amd import engine, synthetic as syn
    z, M, r = np.geomspace(0.1, 0.2, 3), np.geomspace(1e13, 1e14, 3), np.geomspace(1e-2, 10, 8)
    model, keep = engine.model_from_tables([np.log(1 + z), np.log(M), np.log(r)], np.zeros((3, 3, 8)), syn.COSMO, 10.0)
    with pytest.raises(_lib.BfgxError, match="no HIP device"):
        engine.ShellPlan(model, keep, 16, 10)


def test_invalid_arguments_map_to_python_exceptions():
    from baryonification_amd import engine, synthetic as syn
    z, M, r = np.geomspace(0.1, 0.2, 3), np.geomspace(1e13, 1e14, 3), np.geomspace(1e-2, 10, 8)
    axes = [np.log(1 + z), np.log(M), np.log(r)]
    bad_axes = [axes[0][::-1].copy(), axes[1], axes[2]]
    model, keep = engine.model_from_tables(bad_axes, np.zeros((3, 3, 8)), syn.COSMO, 10.0)
    with pytest.raises(ValueError, match="ascending"):
        engine.ShellPlan(model, keep, 16, 10)
    model, keep = engine.model_from_tables(axes, np.zeros((3, 3, 8)), syn.COSMO, 10.0)
    with pytest.raises(ValueError, match="nside"):
        engine.ShellPlan(model, keep, 0, 10)
    with pytest.raises(ValueError):
        _lib.make_table(axes, np.zeros((3, 3, 7)))


def test_power_spectrum_work_size_is_host_arithmetic():
    """rows of the half spectrum are padded to whole 128-byte lines (8 complex values): pure host functions, no GPU needed"""
    from baryonification_amd import engine
    for n in (8, 16, 64, 128, 512, 1024):
        pitch = engine.fft_pitch(n)
        assert pitch % 8 == 0 and n // 2 + 1 <= pitch < n // 2 + 1 + 8
        assert engine.power_spectrum_work_doubles(n) == 2 * n * n * pitch
    assert engine.fft_pitch(512) == 264


def test_null_arguments_are_refused_before_any_device_call():
    import ctypes as C
    L = _lib.load()
    n = C.c_int64(0)
    assert L.bfgx_grid_baryonify_device(None, None, None, None, None, C.byref(n)) == _lib.ERR_INVALID
    assert b'NULL' in L.bfgx_last_error()
    assert L.bfgx_power_spectrum_device(0, None, 64, None, 100.0, 10, None, None, None, None) == _lib.ERR_INVALID


def test_round5_entries_refuse_bad_arguments_before_any_device_call():
    """the entries added in round 5 (one-call snapshot -> map, per-halo callable bridge, fused band entry) check their arguments first"""
    import ctypes as C
    L = _lib.load()
    n = C.c_int64(0)
    h = C.c_void_p()
    assert L.bfgx_shell_pairs_begin(None, None, 64, 0, 0, C.byref(h), None) == _lib.ERR_INVALID and not h.value
    assert L.bfgx_shell_pairs_radii(None, None) == _lib.ERR_INVALID
    assert L.bfgx_shell_pairs_apply(None, None, None, None, 1, None) == _lib.ERR_INVALID
    L.bfgx_shell_pairs_end(None)                                                  # (a no-op)
    assert L.bfgx_snapshot_displace_deposit_device(None, None, 0, None, None, None, None, 8, None, None, C.byref(n)) == _lib.ERR_INVALID
    st = _lib.bfgx_stats()
    assert L.bfgx_baryonify_snapshot_records_map(None, None, 3, 100.0, 0.0, 0, None, 32, 8, 16, 24, 0, 8, None, None, None, C.byref(st)) == _lib.ERR_INVALID
    assert L.bfgx_offsets_regrid_bands_device(None, None, 0, 1, None, 0, 0, 1, None, None, None, None) == _lib.ERR_INVALID
    assert b'NULL' in L.bfgx_last_error()
