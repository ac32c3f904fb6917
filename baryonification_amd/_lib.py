"""
ctypes binding of libbfgx.so (C ABI declared in include/bfgx.h).

The shared library is the product: if it is missing this module raises ImportError loudly --
there is NO Python/CPU fallback for the hot path.  Build it with `python -c "import
__graft_entry__ as g; g.build()"` or `make -C baryonification_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

BFGX_MAX_EXTRA = 4
BFGX_MAX_DIM = 3 + BFGX_MAX_EXTRA
KERNEL_KINDS = ('prep', 'offsets', 'regrid', 'paint', 'sum', 'count', 'bin', 'wide')

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 4        # include/bfgx.h BFGX_ABI_VERSION: 4 since BFGX_MAX_EXTRA = 4 (bfgx_table / bfgx_catalog hold 7 axes / 4 property columns)
LIB_PATH = os.environ.get('BFGX_LIB', os.path.join(_HERE, 'csrc', 'libbfgx.so'))   # BFGX_LIB: ablation builds only

OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_MASS, ERR_ASSERT = 0, -1, -2, -3, -4, -5, -6
# precision of the displacement path (include/bfgx.h BFGX_ACC_*)
ACC_AUTO, ACC_F32, ACC_F64, ACC_PARITY = -1, 0, 1, 3


def acc_mode(acc_f64):
    """runner.acc_f64 -> BFGX_ACC_*: None / 'auto' = the plan chooses from the table, False = fp32 pair math, True = fp64 throughout,
    'parity' / 3 = the parity-grade mode"""
    if acc_f64 is None or acc_f64 == 'auto' or (acc_f64 is not True and acc_f64 is not False and acc_f64 == -1):
        return ACC_AUTO
    if acc_f64 == 'parity' or (acc_f64 is not True and acc_f64 == 3):
        return ACC_PARITY
    return ACC_F64 if bool(acc_f64) else ACC_F32

c_double_p = C.POINTER(C.c_double)


class bfgx_cosmo(C.Structure):
    _fields_ = [('Omega_m', C.c_double), ('Omega_b', C.c_double), ('h', C.c_double), ('sigma8', C.c_double),
                ('n_s', C.c_double), ('w0', C.c_double), ('T_CMB', C.c_double), ('Neff', C.c_double)]


class bfgx_massdef(C.Structure):
    _fields_ = [('Delta', C.c_double), ('rho_type', C.c_int32), ('_pad', C.c_int32)]


class bfgx_table(C.Structure):
    _fields_ = [('ndim', C.c_int32), ('n', C.c_int32 * BFGX_MAX_DIM), ('axis', C.c_void_p * BFGX_MAX_DIM),
                ('values', C.c_void_p), ('rdelta_sampling', C.c_int32), ('log_values', C.c_int32),
                ('eps_model', C.c_double)]


class bfgx_catalog(C.Structure):
    _fields_ = [('n', C.c_int64), ('M', C.c_void_p), ('z', C.c_void_p), ('ra', C.c_void_p), ('dec', C.c_void_p),
                ('extra', C.c_void_p * BFGX_MAX_EXTRA), ('ln1pz', C.c_void_p), ('lnM', C.c_void_p)]


class bfgx_model(C.Structure):
    _fields_ = [('table', bfgx_table), ('cosmo_runner', bfgx_cosmo), ('massdef_runner', bfgx_massdef),
                ('cosmo_model', bfgx_cosmo), ('massdef_model', bfgx_massdef), ('eps_runner', C.c_double)]


class bfgx_opts(C.Structure):
    _fields_ = [('device', C.c_int32), ('acc_offsets_f64', C.c_int32), ('acc_paint_f64', C.c_int32),
                ('check_mass', C.c_int32), ('algo', C.c_int32), ('_pad', C.c_int32), ('catalog_token', C.c_uint64)]


class bfgx_grid(C.Structure):
    _fields_ = [('ndim', C.c_int32), ('npix', C.c_int32), ('bins', C.c_void_p), ('redshift', C.c_double)]


class bfgx_grid_catalog(C.Structure):
    _fields_ = [('n', C.c_int64), ('M', C.c_void_p), ('x', C.c_void_p), ('y', C.c_void_p), ('z', C.c_void_p),
                ('lnM', C.c_void_p), ('rmat', C.c_void_p), ('extra', C.c_void_p * BFGX_MAX_EXTRA)]


class bfgx_snapshot(C.Structure):
    _fields_ = [('ndim', C.c_int32), ('_pad', C.c_int32), ('n', C.c_int64), ('x', C.c_void_p), ('y', C.c_void_p), ('z', C.c_void_p),
                ('L', C.c_double), ('redshift', C.c_double)]


class bfgx_stats(C.Structure):
    _fields_ = [('n_pairs', C.c_int64), ('sum_in', C.c_double), ('sum_out', C.c_double),
                ('ms_h2d', C.c_double), ('ms_kernels', C.c_double), ('ms_d2h', C.c_double)]


# every symbol include/bfgx.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    'bfgx_abi_version': (C.c_int, []),
    'bfgx_last_error': (C.c_char_p, []),
    'bfgx_device_count': (C.c_int, []),
    'bfgx_cosmo_E2': (C.c_int, [_P(bfgx_cosmo), C.c_int64, C.c_void_p, C.c_void_p]),
    'bfgx_cosmo_radius': (C.c_int, [_P(bfgx_cosmo), _P(bfgx_massdef), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_cosmo_angular_diameter_distance': (C.c_int, [_P(bfgx_cosmo), C.c_int64, C.c_void_p, C.c_void_p]),
    'bfgx_cosmo_da_spline': (C.c_int, [_P(bfgx_cosmo), C.c_void_p, C.c_void_p]),
    'bfgx_cosmo_da_eval': (C.c_int, [_P(bfgx_cosmo), C.c_int64, C.c_void_p, C.c_void_p]),
    'bfgx_baryonify_shell': (C.c_int, [_P(bfgx_catalog), _P(bfgx_model), C.c_int64, C.c_void_p, C.c_void_p,
                                       _P(bfgx_opts), _P(bfgx_stats)]),
    'bfgx_paint_shell': (C.c_int, [_P(bfgx_catalog), _P(bfgx_model), C.c_int64, C.c_void_p,
                                   _P(bfgx_opts), _P(bfgx_stats)]),
    'bfgx_baryonify_shell_multi': (C.c_int, [_P(bfgx_catalog), _P(bfgx_model), C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                             _P(bfgx_opts), _P(bfgx_stats)]),
    'bfgx_paint_shell_multi': (C.c_int, [_P(bfgx_catalog), _P(bfgx_model), C.c_int64, C.c_void_p, C.c_int32, C.c_void_p,
                                         _P(bfgx_opts), _P(bfgx_stats)]),
    'bfgx_cache_clear': (None, []),
    'bfgx_debug_alloc_count': (C.c_longlong, []),
    'bfgx_debug_catalog_uploads': (C.c_longlong, []),
    'bfgx_debug_grid_pipe_fallbacks': (C.c_longlong, []),
    'bfgx_debug_host_spans': (None, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    'bfgx_host_alloc': (C.c_int, [C.c_size_t, _P(C.c_void_p)]),
    'bfgx_host_free': (None, [C.c_void_p]),
    'bfgx_plan_create': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int64, _P(bfgx_model), _P(C.c_void_p)]),
    'bfgx_plan_destroy': (None, [C.c_void_p]),
    'bfgx_offsets_device': (C.c_int, [C.c_void_p, _P(bfgx_catalog), C.c_void_p, C.c_int]),
    'bfgx_regrid_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'bfgx_plan_bands': (C.c_int, [C.c_void_p, _P(C.c_int32), C.c_void_p]),
    'bfgx_baryonify_device': (C.c_int, [C.c_void_p, _P(bfgx_catalog), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'bfgx_route_count_device': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'bfgx_route_fill_device': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    'bfgx_route_pack_device': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    'bfgx_bands_max_offset2_device': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'bfgx_max_offset2_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'bfgx_plan_tile_shape': (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32)]),
    'bfgx_disc_rings_device': (C.c_int, [C.c_void_p, _P(bfgx_catalog), C.c_void_p]),
    'bfgx_plan_set_route_margin': (C.c_int, [C.c_void_p, C.c_int32]),
    'bfgx_offsets_bands_device': (C.c_int, [C.c_void_p, _P(bfgx_catalog), C.c_int32, C.c_int32, C.c_void_p, C.c_int]),
    'bfgx_paint_bands_device': (C.c_int, [C.c_void_p, _P(bfgx_catalog), C.c_int32, C.c_int32, C.c_void_p, C.c_int]),
    'bfgx_plan_reach_rings': (C.c_int, [C.c_void_p, C.c_double, _P(C.c_int32)]),
    'bfgx_plan_set_band_reach': (C.c_int, [C.c_void_p, C.c_int32]),
    'bfgx_plan_band_apron': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _P(C.c_int64), _P(C.c_int64)]),
    'bfgx_regrid_bands_device': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                           C.c_void_p]),
    'bfgx_plan_far_fetch': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, _P(C.c_int64)]),
    'bfgx_plan_far_apply_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'bfgx_paint_device': (C.c_int, [C.c_void_p, _P(bfgx_catalog), C.c_void_p, C.c_int]),
    'bfgx_plan_set_algo': (C.c_int, [C.c_void_p, C.c_int]),
    'bfgx_plan_status': (C.c_int, [C.c_void_p]),
    'bfgx_route_step_device': (C.c_int, [C.c_void_p, C.POINTER(bfgx_catalog), C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_plan_set_catalog_blocks': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    'bfgx_offsets_regrid_bands_device': (C.c_int, [C.c_void_p, C.POINTER(bfgx_catalog), C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.c_int32, C.c_int32,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_plan_precision': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    'bfgx_plan_regrid_stats': (C.c_int, [C.c_void_p, _P(C.c_int64), _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    'bfgx_plan_timing_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'bfgx_plan_timing_read': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_project_profile': (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p]),
    'bfgx_enclosed_mass_2d': (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.c_void_p]),
    'bfgx_enclosed_mass_from_sigma': (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'bfgx_enclosed_mass_3d': (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'bfgx_displacement_rows': (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_pressure_profile': (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_void_p]),
    'bfgx_fftlog_kgrid': (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_void_p]),
    'bfgx_fftlog_transform': (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                        C.c_void_p, C.c_void_p]),
    'bfgx_fftlog_convolve': (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double,
                                       C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_void_p]),
    'bfgx_count_pairs_device': (C.c_int, [C.c_void_p, _P(bfgx_catalog), C.c_int, C.c_void_p, _P(C.c_int64)]),
    # regular-grid path
    'bfgx_baryonify_grid': (C.c_int, [_P(bfgx_grid_catalog), _P(bfgx_model), _P(bfgx_grid), C.c_void_p, C.c_void_p,
                                      _P(bfgx_opts), _P(bfgx_stats)]),
    'bfgx_paint_grid': (C.c_int, [_P(bfgx_grid_catalog), _P(bfgx_model), _P(bfgx_grid), C.c_void_p, _P(bfgx_opts), _P(bfgx_stats)]),
    'bfgx_regrid_pixels': (C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_deposit_particles': (C.c_int, [C.c_int, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p]),
    'bfgx_deposit_particles_records': (C.c_int, [C.c_int, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_void_p, C.c_void_p]),
    'bfgx_power_spectrum': (C.c_int, [C.c_int, C.c_int32, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_baryonify_snapshot': (C.c_int, [_P(bfgx_grid_catalog), _P(bfgx_model), _P(bfgx_snapshot), C.c_void_p, C.c_void_p, C.c_void_p,
                                          _P(bfgx_opts), _P(bfgx_stats)]),
    'bfgx_baryonify_snapshot_records': (C.c_int, [_P(bfgx_grid_catalog), _P(bfgx_model), C.c_int32, C.c_double, C.c_double, C.c_int64, C.c_void_p,
                                                  C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(bfgx_opts), _P(bfgx_stats)]),
    'bfgx_baryonify_snapshot_device': (C.c_int, [C.c_int, C.c_void_p, _P(bfgx_grid_catalog), _P(bfgx_model), _P(bfgx_snapshot), C.c_void_p,
                                                 C.c_void_p, C.c_void_p, _P(C.c_int64)]),
    'bfgx_snapshot_plan_create': (C.c_int, [C.c_int, C.c_void_p, _P(bfgx_model), C.c_int32, C.c_double, C.c_double, C.c_int64, _P(C.c_void_p)]),
    'bfgx_snapshot_plan_destroy': (None, [C.c_void_p]),
    'bfgx_snapshot_displace_device': (C.c_int, [C.c_void_p, _P(bfgx_grid_catalog), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, _P(C.c_int64)]),
    'bfgx_shell_pairs_begin': (C.c_int, [_P(bfgx_catalog), _P(bfgx_model), C.c_int64, C.c_int32, C.c_int32, _P(C.c_void_p), C.c_void_p]),
    'bfgx_shell_pairs_radii': (C.c_int, [C.c_void_p, C.c_void_p]),
    'bfgx_shell_pairs_apply': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, _P(bfgx_stats)]),
    'bfgx_shell_pairs_end': (None, [C.c_void_p]),
    'bfgx_baryonify_snapshot_records_map': (C.c_int, [_P(bfgx_grid_catalog), _P(bfgx_model), C.c_int32, C.c_double, C.c_double, C.c_int64, C.c_void_p,
                                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                                      _P(bfgx_opts), _P(bfgx_stats)]),
    'bfgx_snapshot_displace_deposit_device': (C.c_int, [C.c_void_p, _P(bfgx_grid_catalog), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_int32, C.c_void_p, C.c_void_p, _P(C.c_int64)]),
    'bfgx_grid_plan_create': (C.c_int, [C.c_int, C.c_void_p, _P(bfgx_grid), C.c_int64, _P(bfgx_model), _P(C.c_void_p)]),
    'bfgx_grid_plan_destroy': (None, [C.c_void_p]),
    'bfgx_grid_offsets_device': (C.c_int, [C.c_void_p, _P(bfgx_grid_catalog), C.c_void_p, _P(C.c_int64)]),
    'bfgx_grid_paint_device': (C.c_int, [C.c_void_p, _P(bfgx_grid_catalog), C.c_void_p, _P(C.c_int64)]),
    'bfgx_grid_regrid_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_grid_baryonify_device': (C.c_int, [C.c_void_p, _P(bfgx_grid_catalog), C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_int64)]),
    'bfgx_grid_deposit_baryonify_device': (C.c_int, [C.c_void_p, _P(bfgx_grid_catalog), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_int64)]),
    'bfgx_grid_plan_set_slab': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    'bfgx_grid_regrid_slab_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_deposit_particles_slab_device': (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'bfgx_fft_slab_planes_device': (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'bfgx_fft_slab_axis0_pk_device': (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_double, C.c_int32,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_grid_plan_timing_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'bfgx_grid_plan_timing_read': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'bfgx_deposit_particles_device': (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int32, C.c_void_p, C.c_void_p]),
    'bfgx_route_particles_count_device': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'bfgx_route_particles_fill_device': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                                   C.c_void_p, C.c_void_p]),
    'bfgx_power_spectrum_work_doubles': (C.c_int64, [C.c_int32]),
    'bfgx_fft_pitch': (C.c_int32, [C.c_int32]),
    'bfgx_power_spectrum_device': (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
}

_lib = None


def _preload_hip_runtime():
    """One process must use ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as
    the system one); if libbfgx pulled in /opt/rocm's copy first, a later `import torch` would mix the two stacks
    and find no GPU.  So when torch is installed (not necessarily imported) its runtime is loaded first and the
    dynamic linker resolves libbfgx's dependency to it."""
    import importlib.util
    import sys
    if 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load libbfgx.so once; raise ImportError (never fall back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "baryonification_amd: %s is missing. The HIP extension IS the product (no CPU fallback). "
                "Build it with `make -C %s` (hipcc --offload-arch=gfx950)." % (LIB_PATH, os.path.dirname(LIB_PATH)))
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)        # AttributeError here = ABI drift between header and library
            fn.restype = res
            fn.argtypes = args
        if L.bfgx_abi_version() != ABI_VERSION:
            raise ImportError("libbfgx.so ABI version %d != %d (a stale build? run `make -C baryonification_amd/csrc`)" % (L.bfgx_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


class BfgxError(RuntimeError):
    pass


def check(rc):
    """Map a bfgx_status to the exception the reference would raise on the same condition."""
    if rc == OK:
        return
    msg = load().bfgx_last_error().decode('utf-8', 'replace')
    if rc == ERR_MASS or rc == ERR_ASSERT:
        raise AssertionError(msg)                 # HealpixRunner.py:346, Map2DRunner.py:516 / :605
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NO_DEVICE:
        raise BfgxError("bfgx: " + msg)
    raise BfgxError("bfgx (status %d): %s" % (rc, msg))


def f8(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def make_cosmo(d, T_CMB=-1.0, Neff=-1.0):
    """bfgx_cosmo from the 6-key cosmology dict of io.py:79-85 (extra keys T_CMB / Neff honoured)."""
    return bfgx_cosmo(float(d['Omega_m']), float(d['Omega_b']), float(d['h']),
                      float(d.get('sigma8', np.nan) if d.get('sigma8') is not None else np.nan),
                      float(d.get('n_s', np.nan) if d.get('n_s') is not None else np.nan),
                      float(d.get('w0', -1.0)),
                      float(d.get('T_CMB', T_CMB)), float(d.get('Neff', Neff)))


def make_massdef(Delta=200.0, rho_type='critical'):
    if rho_type not in ('critical', 'matter'):
        raise NotImplementedError("mass definition rho_type %r" % (rho_type,))
    if isinstance(Delta, str):
        raise NotImplementedError("mass definition Delta=%r (numeric overdensities only)" % (Delta,))
    return bfgx_massdef(float(Delta), 0 if rho_type == 'critical' else 1, 0)


def make_table(axes, values, rdelta_sampling=False, log_values=False, eps_model=20.0):
    """Returns (bfgx_table, keepalive) -- keepalive holds the numpy buffers the struct points into."""
    axes = [f8(a) for a in axes]
    values = f8(values)
    if len(axes) > BFGX_MAX_DIM:
        raise NotImplementedError("a table with %d property axes (other_params): libbfgx reads out at most BFGX_MAX_EXTRA = %d of them "
                                  "(3 + %d table axes; the reference's ParamTabulatedProfile / BaryonificationClass accept any number, "
                                  "Tabulate.py:524-561, BaryonCorrection.py:205-221)" % (len(axes) - 3, BFGX_MAX_EXTRA, BFGX_MAX_EXTRA))
    if values.shape != tuple(a.size for a in axes):
        raise ValueError("table values shape %r does not match axes %r" % (values.shape, [a.size for a in axes]))
    t = bfgx_table()
    t.ndim = len(axes)
    for i, a in enumerate(axes):
        t.n[i] = a.size
        t.axis[i] = a.ctypes.data
    t.values = values.ctypes.data
    t.rdelta_sampling = int(bool(rdelta_sampling))
    t.log_values = int(bool(log_values))
    t.eps_model = float(eps_model)
    return t, (axes, values)


def table_coords(M, z):
    """The halo's (z, M) table coordinates exactly as the reference forms them -- np.log(1/a) with a = 1/(1+z)
    (HealpixRunner.py:295, BaryonCorrection.py:364, Tabulate.py:279) and np.log(M) (:369, :283) -- evaluated by the
    caller's numpy.  README.md:78-80 builds tables whose edges ARE the catalog's min/max, so the last bit of these logs
    decides whether an edge halo is inside the table; they travel through the ABI (bfgx_catalog.ln1pz / lnM)."""
    M, z = f8(M), f8(z)
    with np.errstate(all='ignore'):
        a = 1.0 / (1.0 + z)
        return f8(np.log(1.0 / a)), f8(np.log(M))


# ---- page-locked output arrays -----------------------------------------------------------------------------------
# The map a runner returns is a plain numpy array whose memory is page-locked (hipHostMalloc), so that the device-to-host
# copy of the result runs at the full PCIe rate.  Page-locking is slow (tens of ms for a 100 MB map), so buffers are pooled:
# when the array (and every view of it) is garbage-collected its buffer returns to the pool and the next process() call of
# the same size gets it back.  At most _PINNED_POOL_BYTES are kept; bfgx_cache_clear() style cleanup: pinned_pool_clear().
_PINNED_POOL_BYTES = 4 << 30
_pinned_pool = {}          # nbytes -> [ptr, ...]
_pinned_pooled = 0


def _pinned_release(ptr, nbytes):
    global _pinned_pooled
    try:
        if _pinned_pooled + nbytes <= _PINNED_POOL_BYTES:
            _pinned_pool.setdefault(nbytes, []).append(ptr)
            _pinned_pooled += nbytes
        else:
            load().bfgx_host_free(C.c_void_p(ptr))
    except Exception:        # interpreter shutdown
        pass


def pinned_pool_clear():
    global _pinned_pooled
    for nbytes, ptrs in _pinned_pool.items():
        for ptr in ptrs:
            load().bfgx_host_free(C.c_void_p(ptr))
    _pinned_pool.clear()
    _pinned_pooled = 0


def pinned_empty(n, dtype=np.float64):
    """np.empty(n, dtype) in page-locked memory (falls back to ordinary memory if page-locking fails)"""
    global _pinned_pooled
    import weakref
    dtype = np.dtype(dtype)
    nbytes = max(int(n) * dtype.itemsize, 1)
    ptrs = _pinned_pool.get(nbytes)
    if ptrs:
        ptr = ptrs.pop()
        _pinned_pooled -= nbytes
    else:
        out = C.c_void_p(0)
        if load().bfgx_host_alloc(nbytes, C.byref(out)) != OK or not out.value:
            return np.empty(int(n), dtype=dtype)
        ptr = out.value
    raw = (C.c_char * nbytes).from_address(ptr)
    weakref.finalize(raw, _pinned_release, ptr, nbytes)       # runs when the last numpy view of `raw` is gone
    return np.frombuffer(raw, dtype=dtype, count=int(n))


def make_catalog_host(M, z, ra, dec, extra=(), coords=None):
    cols = [f8(M), f8(z), f8(ra), f8(dec)] + [f8(e) for e in extra]
    c = bfgx_catalog()
    c.n = cols[0].size
    c.M, c.z, c.ra, c.dec = (cols[i].ctypes.data for i in range(4))
    for k, e in enumerate(cols[4:]):
        c.extra[k] = e.ctypes.data
    lnz, lnM = coords if coords is not None else table_coords(cols[0], cols[1])
    c.ln1pz, c.lnM = lnz.ctypes.data, lnM.ctypes.data
    cols += [lnz, lnM]
    return c, cols


def make_catalog_dev(n, M_ptr, z_ptr, ra_ptr, dec_ptr, extra_ptrs=(), ln1pz_ptr=0, lnM_ptr=0):
    """ln1pz_ptr / lnM_ptr: device copies of table_coords(M, z) (optional; 0 = the device derives them, which may
    classify a halo that sits exactly on a table edge differently from numpy)"""
    c = bfgx_catalog()
    c.n = int(n)
    c.M, c.z, c.ra, c.dec = int(M_ptr), int(z_ptr), int(ra_ptr), int(dec_ptr)
    for k, e in enumerate(extra_ptrs):
        c.extra[k] = int(e)
    c.ln1pz, c.lnM = (int(ln1pz_ptr) or None), (int(lnM_ptr) or None)
    return c


def make_grid(bins, ndim, redshift):
    """Returns (bfgx_grid, keepalive)"""
    b = f8(bins)
    g = bfgx_grid(int(ndim), int(b.size), b.ctypes.data, float(redshift))
    return g, b


def make_grid_catalog_host(M, x, y, z=None, lnM=None, rmat=None, extra=()):
    cols = {'M': f8(M), 'x': f8(x), 'y': f8(y)}
    if z is not None:
        cols['z'] = f8(z)
    if lnM is not None:
        cols['lnM'] = f8(lnM)
    if rmat is not None:
        cols['rmat'] = f8(rmat).reshape(-1, 4)
    c = bfgx_grid_catalog()
    c.n = cols['M'].size
    for k, v in cols.items():
        setattr(c, k, v.ctypes.data)
    ex = [f8(e) for e in extra]
    for k, e in enumerate(ex):
        c.extra[k] = e.ctypes.data
    return c, (cols, ex)


def make_grid_catalog_dev(n, M_ptr, x_ptr, y_ptr, z_ptr=0, lnM_ptr=0, rmat_ptr=0, extra_ptrs=()):
    c = bfgx_grid_catalog()
    c.n = int(n)
    c.M, c.x, c.y = int(M_ptr), int(x_ptr), int(y_ptr)
    c.z, c.lnM, c.rmat = (int(z_ptr) or None), (int(lnM_ptr) or None), (int(rmat_ptr) or None)
    for k, e in enumerate(extra_ptrs):
        c.extra[k] = int(e)
    return c
