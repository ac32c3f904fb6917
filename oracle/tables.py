"""
oracle/tables.py -- CPU restatement (numpy + scipy, float64) of the reference's table builders
(TEST INFRASTRUCTURE ONLY; never imported by the product):

  project_realspace      SchneiderProfiles._projected_realspace   BaryonForge/Profiles/Schneider19.py:195-265
  los_grid               the line-of-sight grid of the same       Schneider19.py:225-233
  enclosed_mass_2d       Baryonification2D.get_masses             BaryonForge/Profiles/BaryonCorrection.py:585-665
  displacement_rows      BaryonificationClass.setup_interpolator  BaryonCorrection.py:226-301 (per-mass loop body)
  pressure_profile       Pressure._real                           BaryonForge/Profiles/Thermodynamic.py:240-271

Unlike the reference these take the 3-D densities already SAMPLED on the relevant radial grid (rho[row, l]),
because that is the boundary of the GPU kernels (K4-K6): the profile physics stays on the host.
scipy (PchipInterpolator) is a dependency of the reference itself and is used here exactly as it is there.
"""
import warnings

import numpy as np
from scipy import interpolate

from .oracle import GNEWT, MPC_TO_METER, SOLAR_MASS

N_INT_2D = 50_000                 # BaryonCorrection.py:641
PRESSURE_AT_INFINITY = 1e-200     # Thermodynamic.py:38
G_MPC = GNEWT / MPC_TO_METER ** 3 * SOLAR_MASS          # Thermodynamic.py:11


def los_grid(r, padding_lo=0.1, padding_hi=10.0, n_per_decade=10, proj_cutoff=None):
    """Schneider19.py:225-233 (note: the node count is fixed BEFORE int_max is replaced by proj_cutoff)"""
    int_min = padding_lo * np.min(r)
    int_max = padding_hi * np.max(r)
    int_N = n_per_decade * np.int32(np.log10(int_max / int_min))
    if proj_cutoff is not None:
        int_max = proj_cutoff
    return np.geomspace(int_min, int_max, int_N)


def project_realspace(l, rho, r):
    """Schneider19.py:245-252: Sigma[i, j] = 2 trapz(interp(sqrt(l^2 + r_j^2), l, rho_i), l)"""
    rho = np.atleast_2d(rho)
    out = np.zeros([rho.shape[0], r.size])
    for i in range(rho.shape[0]):
        for j in range(r.size):
            out[i, j] = 2 * np.trapz(np.interp(np.sqrt(l ** 2 + r[j] ** 2), l, rho[i]), l)
    return out


def r_int_2d(r):
    """BaryonCorrection.py:639-641"""
    r_min = np.min([np.min(r), 1e-6])
    r_max = np.max([np.max(r), 1000])
    return np.geomspace(r_min / 1.5, r_max * 1.5, N_INT_2D)


def enclosed_mass_from_sigma(r_int, Sigma, r):
    """BaryonCorrection.py:645-661; Sigma already includes the factor a (:646)"""
    dlnr = np.log(r_int[1] / r_int[0])
    Sigma = np.where(Sigma < 0, 0, Sigma)
    M_enc = np.cumsum(2 * np.pi * r_int ** 2 * Sigma * dlnr, axis=-1)
    lnr = np.log(r)
    M_f = np.zeros([M_enc.shape[0], r.size])
    with np.errstate(divide='ignore', invalid='ignore'):
        for i in range(M_enc.shape[0]):
            Mask = (Sigma[i] > 0) & (np.isfinite(M_enc[i]))
            M_f[i] = np.exp(interpolate.PchipInterpolator(np.log(r_int)[Mask], np.log(M_enc[i])[Mask], extrapolate=False)(lnr))
    return M_f


def r_int_3d(r):
    """BaryonCorrection.py:525-527"""
    r_min = np.min([np.min(r), 1e-6])
    r_max = np.max([np.max(r), 1000])
    return np.geomspace(r_min / 1.2, r_max * 1.2, N_INT_2D)


def enclosed_mass_3d(r_int, rho, r):
    """Baryonification3D.get_masses, BaryonCorrection.py:528-546, from the density sampled on r_int"""
    dlnr = np.log(r_int[1] / r_int[0])
    rho = np.where(rho < 0, 0, rho)
    M_enc = np.cumsum(4 * np.pi * r_int ** 3 * rho * dlnr, axis=-1)
    lnr = np.log(r)
    M_f = np.zeros([M_enc.shape[0], r.size])
    with np.errstate(divide='ignore', invalid='ignore'):
        for i in range(M_enc.shape[0]):
            Mask = (rho[i] > 0) & (np.isfinite(M_enc[i]))
            M_f[i] = np.exp(interpolate.PchipInterpolator(np.log(r_int)[Mask], np.log(M_enc[i])[Mask], extrapolate=False)(lnr))
    return M_f


def enclosed_mass_2d(l, rho, a, r):
    """get_masses for a profile whose `projected` is `_projected_realspace` (l = los_grid(r_int_2d(r), ...))"""
    r_int = r_int_2d(r)
    Sigma = project_realspace(l, rho, r_int) * a
    return enclosed_mass_from_sigma(r_int, Sigma, r)


def displacement_rows(r, M_DMO, M_DMB, quiet=True):
    """BaryonCorrection.py:226-301 for every mass row; returns (offsets [rows, N_R], status [rows])
    status: 0 ok, 1 'nearly constant' (iterate > 30), 2 '< 5 usable points', both -> d = 0 (:290-297)"""
    out = np.zeros_like(M_DMO)
    status = np.zeros(M_DMO.shape[0], dtype=np.int32)
    with np.errstate(divide='ignore', invalid='ignore'), warnings.catch_warnings():
        if quiet:
            warnings.simplefilter('ignore')
        for i in range(M_DMO.shape[0]):
            ln_DMB, ln_DMO = np.log(M_DMB[i]), np.log(M_DMO[i])
            min_diff = -np.inf
            diff_mask = np.ones_like(ln_DMB).astype(bool)
            iterate = 0
            while (min_diff < 1e-5) & (diff_mask.sum() > 5):
                new_mask = ((np.diff(ln_DMB[diff_mask], prepend=0) > 1e-5) &
                            ((np.abs(ln_DMB - ln_DMO)[diff_mask] > 1e-6) | np.isnan(ln_DMO)[diff_mask]) &
                            np.isfinite(ln_DMB)[diff_mask])
                diff_mask[diff_mask] = new_mask
                diff_mask[0] = True
                iterate += 1
                if iterate > 30:
                    diff_mask = np.zeros_like(diff_mask).astype(bool)
                    status[i] = 1
                    break
                if diff_mask.sum() < 5:
                    status[i] = 2
                    break
                min_diff = np.min(np.diff(ln_DMB[diff_mask], prepend=0)[1:])
            if diff_mask.sum() > 5:
                fini_mask = ((np.diff(ln_DMO, prepend=0) > 1e-5) &
                             ((np.abs(ln_DMB - ln_DMO) > 1e-6) | np.isnan(ln_DMB)) & np.isfinite(ln_DMO))
                interp_DMB = interpolate.PchipInterpolator(ln_DMB[diff_mask], np.log(r)[diff_mask], extrapolate=False)
                interp_DMO = interpolate.PchipInterpolator(np.log(r)[fini_mask], ln_DMO[fini_mask], extrapolate=False)
                offset = np.exp(interp_DMB(interp_DMO(np.log(r)))) - r
                out[i] = np.where(np.isfinite(offset), offset, 0)
            else:
                if status[i] == 0:
                    status[i] = 2
                out[i] = 0.0
    return out, status


def pressure_profile(rho_total, rho_gas, r_use, cutoff=np.inf):
    """Thermodynamic.py:240-271 with r_integral = geomspace(1e-6, 1000, 500); returns cgs pressure"""
    r_integral = np.geomspace(1e-6, 1000, 500)
    rho_total, rho_gas = np.atleast_2d(rho_total), np.atleast_2d(rho_gas)
    dlnr = np.log(r_integral[1]) - np.log(r_integral[0])
    M_total = 4 * np.pi * np.cumsum(r_integral ** 3 * rho_total * dlnr, axis=-1)
    dP_dr = -G_MPC * M_total * rho_gas / r_integral ** 2
    prof = -np.cumsum((dP_dr * r_integral)[:, ::-1] * dlnr, axis=-1)[:, ::-1]
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        prof = interpolate.PchipInterpolator(np.log(r_integral), np.log(prof + PRESSURE_AT_INFINITY), axis=1, extrapolate=False)
        prof = np.exp(prof(np.log(r_use))) - PRESSURE_AT_INFINITY
        prof = np.where(np.isfinite(prof), prof, 0)
        prof = prof * (SOLAR_MASS * 1e3) / (MPC_TO_METER * 1e2)
        arg = (r_use[None, :] - cutoff)
        arg = np.where(arg > 30, np.inf, arg)
        kfac = 1 / (1 + np.exp(2 * arg))
    return prof * kfac
