"""
Host-array front-ends of the table-builder kernels (C ABI in include/bfgx.h).
numpy in, numpy out; they run once per model.  The radial grids are formed here with numpy exactly as
the reference forms them (np.geomspace), the heavy loops run on the GPU.

  los_grid / r_int_2d       the grids of Schneider19.py:225-233 and BaryonCorrection.py:639-641
  project_profile           SchneiderProfiles._projected_realspace      Schneider19.py:245-252
  enclosed_mass_2d          Baryonification2D.get_masses                BaryonCorrection.py:585-665
  displacement_rows         setup_interpolator per-mass loop            BaryonCorrection.py:226-301
  pressure_profile          Pressure._real                              Thermodynamic.py:240-271
"""
import numpy as np

from . import _lib

N_INT_2D = 50_000


def los_grid(r, padding_lo=0.1, padding_hi=10.0, n_per_decade=10, proj_cutoff=None):
    int_min = padding_lo * np.min(r)
    int_max = padding_hi * np.max(r)
    int_N = n_per_decade * np.int32(np.log10(int_max / int_min))     # fixed before the cutoff replaces int_max
    if proj_cutoff is not None:
        int_max = proj_cutoff
    return np.geomspace(int_min, int_max, int_N)


def r_int_2d(r):
    r_min = np.min([np.min(r), 1e-6])
    r_max = np.max([np.max(r), 1000])
    return np.geomspace(r_min / 1.5, r_max * 1.5, N_INT_2D)


def r_int_3d(r):
    """radial grid of Baryonification3D.get_masses (BaryonCorrection.py:525-527)"""
    r_min = np.min([np.min(r), 1e-6])
    r_max = np.max([np.max(r), 1000])
    return np.geomspace(r_min / 1.2, r_max * 1.2, N_INT_2D)


def _rows(a):
    a = _lib.f8(np.atleast_2d(a))
    return a, a.shape[0]


def project_profile(l, rho, r, scale=1.0, device=0):
    l, r = _lib.f8(l), _lib.f8(r)
    rho, nrows = _rows(rho)
    assert rho.shape[1] == l.size
    out = np.empty((nrows, r.size))
    _lib.check(_lib.load().bfgx_project_profile(device, nrows, l.size, l.ctypes.data, rho.ctypes.data, r.size,
                                                r.ctypes.data, float(scale), out.ctypes.data))
    return out


def enclosed_mass_2d(l, rho, a, r, device=0):
    l, r = _lib.f8(l), _lib.f8(r)
    rho, nrows = _rows(rho)
    r_int = _lib.f8(r_int_2d(r))
    out = np.empty((nrows, r.size))
    _lib.check(_lib.load().bfgx_enclosed_mass_2d(device, nrows, l.size, l.ctypes.data, rho.ctypes.data, float(a),
                                                 r_int.size, r_int.ctypes.data, r.size, r.ctypes.data, out.ctypes.data))
    return out


def enclosed_mass_from_sigma(r_int, Sigma, r, device=0):
    r_int, r = _lib.f8(r_int), _lib.f8(r)
    Sigma, nrows = _rows(Sigma)
    out = np.empty((nrows, r.size))
    _lib.check(_lib.load().bfgx_enclosed_mass_from_sigma(device, nrows, r_int.size, r_int.ctypes.data, Sigma.ctypes.data,
                                                         r.size, r.ctypes.data, out.ctypes.data))
    return out


def enclosed_mass_3d(r_int, rho, r, device=0):
    """3-D enclosed mass from the density sampled on r_int (Baryonification3D.get_masses, BaryonCorrection.py:528-546)"""
    r_int, r = _lib.f8(r_int), _lib.f8(r)
    rho, nrows = _rows(rho)
    out = np.empty((nrows, r.size))
    _lib.check(_lib.load().bfgx_enclosed_mass_3d(device, nrows, r_int.size, r_int.ctypes.data, rho.ctypes.data,
                                                 r.size, r.ctypes.data, out.ctypes.data))
    return out


def displacement_rows(r, M_dmo, M_dmb, device=0):
    r = _lib.f8(r)
    M_dmo, nrows = _rows(M_dmo)
    M_dmb, _ = _rows(M_dmb)
    out = np.empty((nrows, r.size))
    status = np.empty(nrows, dtype=np.int32)
    _lib.check(_lib.load().bfgx_displacement_rows(device, nrows, r.size, r.ctypes.data, M_dmo.ctypes.data,
                                                  M_dmb.ctypes.data, out.ctypes.data, status.ctypes.data))
    return out, status


def pressure_profile(rho_total, rho_gas, r_use, cutoff=np.inf, device=0):
    r500 = _lib.f8(np.geomspace(1e-6, 1000, 500))
    rho_total, nrows = _rows(rho_total)
    rho_gas, _ = _rows(rho_gas)
    r_use = _lib.f8(r_use)
    out = np.empty((nrows, r_use.size))
    _lib.check(_lib.load().bfgx_pressure_profile(device, nrows, r500.ctypes.data, rho_total.ctypes.data,
                                                 rho_gas.ctypes.data, r_use.size, r_use.ctypes.data, float(cutoff),
                                                 out.ctypes.data))
    return out
