"""
oracle.py -- numpy/ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (baryonification_amd/) never does.

It restates, in float64 on the CPU, what the reference does around its per-halo loop:

  * HealpixRunner.py:268-280  -- ccl.Cosmology(...) background + CubicSpline D_a(z)
  * HealpixRunner.py:293-297  -- a_j, R_j = mass_def.get_radius(cosmo, M_j, a_j), D_j = D_a(z_j)
  * BaryonCorrection.py:370   -- R = mass_def.get_radius(self.cosmo, M, a) / a
  * HealpixRunner.py:291-349  -- BaryonifyShell.process (C: bfgo_baryonify_offsets + bfgo_regrid)
  * HealpixRunner.py:418-447  -- PaintProfilesShell.process (C: bfgo_paint)

The background cosmology stands in for pyccl==2.8.0 (setup.py:25), which is not installable
here: flat wCDM + photons + massless neutrinos with pyccl-2.x defaults (T_CMB = 2.725 K,
Neff = 3.046) and CCL's physical constants.  Parity with CCL itself is UNPINNED.
"""
import ctypes as C
import os
import subprocess

import numpy as np
from scipy import interpolate

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# ----------------------------------------------------------------------------- constants
# pyccl.physical_constants (CCL 2.x; CODATA 2014 / IAU 2015)
CLIGHT = 299792458.0
GNEWT = 6.67408e-11
SOLAR_MASS = 1.9884754153381438e30
MPC_TO_METER = 3.085677581491367399198952281e22
STBOLTZ = 5.670367e-8
KBOLTZ = 1.38064852e-23
EV_IN_J = 1.6021766208e-19
T_CMB_DEFAULT = 2.725
NEFF_DEFAULT = 3.046
RHO_CRITICAL = ((3 * 100 * 100) / (8 * np.pi * GNEWT)) * (1000 * 1000 * MPC_TO_METER / SOLAR_MASS)


class Background(object):
    """Flat wCDM background; the subset of ccl.Cosmology the hot path touches."""

    def __init__(self, Omega_m, Omega_b, h, sigma8=None, n_s=None, w0=-1.0,
                 T_CMB=T_CMB_DEFAULT, Neff=NEFF_DEFAULT):
        self.Omega_m, self.Omega_b, self.h, self.w0 = float(Omega_m), float(Omega_b), float(h), float(w0)
        self.sigma8, self.n_s = sigma8, n_s
        rho_crit_si = 3.0 * (self.h * 1e5 / MPC_TO_METER) ** 2 / (8 * np.pi * GNEWT)   # kg / m^3
        self.Omega_g = 4 * STBOLTZ / CLIGHT ** 3 * T_CMB ** 4 / rho_crit_si
        T_nu = T_CMB * (4.0 / 11.0) ** (1.0 / 3.0)
        self.Omega_nu_rel = Neff * 7.0 / 8.0 * 4 * STBOLTZ / CLIGHT ** 3 * T_nu ** 4 / rho_crit_si
        self.Omega_r = self.Omega_g + self.Omega_nu_rel
        self.Omega_l = 1.0 - self.Omega_m - self.Omega_r

    @classmethod
    def from_dict(cls, d, **kw):
        return cls(d['Omega_m'], d['Omega_b'], d['h'], d.get('sigma8'), d.get('n_s'), d.get('w0', -1.0), **kw)

    def E2(self, a):
        a = np.asarray(a, dtype=np.float64)
        return self.Omega_m / a ** 3 + self.Omega_l * a ** (-3.0 * (1.0 + self.w0)) + self.Omega_r / a ** 4

    def rho_crit(self, a):
        """physical critical density, Msun / Mpc^3"""
        return RHO_CRITICAL * self.h ** 2 * self.E2(a)

    def get_radius(self, M, a, Delta=200.0, rho_type='critical'):
        """ccl MassDef(Delta, rho_type).get_radius: physical Mpc (rho_x(a, 'critical' | 'matter'), not comoving)"""
        a = np.asarray(a, dtype=np.float64)
        rho = self.rho_crit(a) if rho_type == 'critical' else RHO_CRITICAL * self.h ** 2 * self.Omega_m / a ** 3
        return (np.asarray(M, dtype=np.float64) / (4.18879020479 * Delta * rho)) ** (1.0 / 3.0)

    def comoving_distance_z(self, z):
        """chi(z) in Mpc by Gauss-Legendre panels on [0, z] (sorted ascending input)."""
        z = np.atleast_1d(np.asarray(z, dtype=np.float64))
        xg, wg = np.polynomial.legendre.leggauss(16)
        edges = np.concatenate([[0.0], z])
        out = np.zeros(z.size)
        acc = 0.0
        for i in range(z.size):
            lo, hi = edges[i], edges[i + 1]
            if hi > lo:
                zz = 0.5 * (hi - lo) * xg + 0.5 * (hi + lo)
                acc += 0.5 * (hi - lo) * np.sum(wg / np.sqrt(self.E2(1.0 / (1.0 + zz))))
            out[i] = acc
        return out * (CLIGHT / 1e5 / self.h)

    def angular_diameter_distance_z(self, z):
        z = np.atleast_1d(np.asarray(z, dtype=np.float64))
        return self.comoving_distance_z(z) / (1.0 + z)

    def Da_spline(self):
        """HealpixRunner.py:279-280: CubicSpline(linspace(0,30,1000), D_A) (not-a-knot)"""
        z_t = np.linspace(0, 30, 1000)
        return interpolate.CubicSpline(z_t, self.angular_diameter_distance_z(z_t))


# ----------------------------------------------------------------------------- C library

def build(force=False):
    so = os.path.join(_HERE, 'libbfgoracle.so')
    src = os.path.join(_HERE, 'bfg_oracle.c')
    if force or (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, '-B' if force else '-s'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        i64, dbl, vp = C.c_int64, C.c_double, C.c_void_p
        L.bfgo_pix2vec.argtypes = [i64, i64, vp, vp]
        L.bfgo_ang2vec_lonlat.argtypes = [i64, vp, vp, vp]
        L.bfgo_vec2ang_lonlat.argtypes = [i64, vp, vp, vp]
        L.bfgo_query_disc.argtypes = [i64, vp, dbl, vp, i64]
        L.bfgo_query_disc.restype = i64
        L.bfgo_get_interp_weights_lonlat.argtypes = [i64, i64, vp, vp, vp, vp]
        L.bfgo_rgi_eval.argtypes = [C.c_int, vp, vp, vp, vp]
        L.bfgo_rgi_eval.restype = dbl
        L.bfgo_baryonify_offsets.argtypes = [i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp,
                                             C.c_int, vp, vp, vp, C.c_int, dbl, dbl, vp, vp]
        L.bfgo_baryonify_offsets.restype = i64
        L.bfgo_regrid.argtypes = [i64, vp, vp, vp]
        L.bfgo_regrid_range.argtypes = [i64, i64, i64, vp, vp, vp]
        L.bfgo_paint.argtypes = [i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp,
                                 C.c_int, vp, vp, vp, dbl, vp, vp]
        L.bfgo_paint.restype = i64
        _LIB = L
    return _LIB


def _f8(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _ptr_array(arrs):
    """const double *const * from a list of contiguous f8 arrays (keeps them alive via closure)."""
    arr = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
    return arr


# ----------------------------------------------------------------------------- geometry probes

def pix2vec(nside, pix):
    pix = np.ascontiguousarray(pix, dtype=np.int64)
    out = np.empty((pix.size, 3))
    lib().bfgo_pix2vec(nside, pix.size, _ptr(pix), _ptr(out))
    return out


def ang2vec_lonlat(lon, lat):
    lon, lat = _f8(np.atleast_1d(lon)), _f8(np.atleast_1d(lat))
    out = np.empty((lon.size, 3))
    lib().bfgo_ang2vec_lonlat(lon.size, _ptr(lon), _ptr(lat), _ptr(out))
    return out


def vec2ang_lonlat(vec):
    vec = _f8(vec).reshape(-1, 3)
    lon, lat = np.empty(vec.shape[0]), np.empty(vec.shape[0])
    lib().bfgo_vec2ang_lonlat(vec.shape[0], _ptr(vec), _ptr(lon), _ptr(lat))
    return lon, lat


def query_disc(nside, vec, radius):
    vec = _f8(vec)
    cap = 4096
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().bfgo_query_disc(nside, _ptr(vec), float(radius), _ptr(out), cap)
        if n <= cap:
            return out[:n].copy()
        cap = int(n)


def get_interp_weights_lonlat(nside, lon, lat):
    lon, lat = _f8(np.atleast_1d(lon)), _f8(np.atleast_1d(lat))
    pix = np.empty((lon.size, 4), dtype=np.int64)
    wgt = np.empty((lon.size, 4))
    lib().bfgo_get_interp_weights_lonlat(nside, lon.size, _ptr(lon), _ptr(lat), _ptr(pix), _ptr(wgt))
    return pix, wgt


class Table(object):
    """A tabulated model: axes (ln(1+z), ln M, ln r[, params...]) + values, as the reference keeps
    them in raw_input_* (BaryonCorrection.py:309-313, Tabulate.py:231-235, 553-558)."""

    def __init__(self, axes, values, rdelta_sampling=False, eps_model=20.0, p_keys=()):
        self.axes = [_f8(a) for a in axes]
        self.values = _f8(values)
        assert self.values.shape == tuple(a.size for a in self.axes)
        self.rdelta_sampling = bool(rdelta_sampling)
        self.eps_model = float(eps_model)
        self.p_keys = list(p_keys)
        assert len(self.axes) == 3 + len(self.p_keys)

    def _cargs(self):
        n = np.array([a.size for a in self.axes], dtype=np.int32)
        ax = _ptr_array(self.axes)
        return len(self.axes), n, ax

    def eval(self, x):
        ndim, n, ax = self._cargs()
        x = _f8(x)
        return lib().bfgo_rgi_eval(ndim, _ptr(n), ax, _ptr(self.values), _ptr(x))


def halo_scalars(cat, bg_runner, bg_model=None, md_runner=(200.0, 'critical'), md_model=(200.0, 'critical')):
    """a_j, R_j (phys), D_j (phys), Rmod_j (comoving, model cosmology): HealpixRunner.py:293-297,
    BaryonCorrection.py:370.  md_* = (Delta, rho_type) of the runner's and the model's mass definitions."""
    bg_model = bg_model or bg_runner
    M, z = _f8(cat['M']), _f8(cat['z'])
    a = 1.0 / (1.0 + z)
    R = bg_runner.get_radius(M, a, *md_runner)
    D = bg_runner.Da_spline()(z)
    Rmod = bg_model.get_radius(M, a, *md_model) / a
    return a, R, D, Rmod


def table_coords(cat):
    """np.log(1/a), np.log(M): the halo's (z, M) table coordinates exactly as BaryonCorrection.py:364, :369 / Tabulate.py:279,
    :283 form them (a = 1/(1+z) from HealpixRunner.py:295), with numpy's own log so that halos on a table edge classify as
    in the reference."""
    M, z = _f8(cat['M']), _f8(cat['z'])
    a = 1.0 / (1.0 + z)
    with np.errstate(all='ignore'):
        return _f8(np.log(1.0 / a)), _f8(np.log(M))


def baryonify_offsets(nside, cat, table, eps_runner, bg_runner, bg_model=None, return_counts=False, md_runner=(200.0, 'critical'),
                      md_model=(200.0, 'critical')):
    a, R, D, Rmod = halo_scalars(cat, bg_runner, bg_model, md_runner, md_model)
    ra, dec, M = _f8(cat['ra']), _f8(cat['dec']), _f8(cat['M'])
    extra = [_f8(cat[k]) for k in table.p_keys]
    ndim, tn, tax = table._cargs()
    npix = 12 * nside * nside
    off = np.zeros((npix, 3))
    counts = np.zeros(ra.size, dtype=np.int64)
    lnz1, lnM = table_coords(cat)
    tot = lib().bfgo_baryonify_offsets(nside, ra.size, _ptr(ra), _ptr(dec), _ptr(M), _ptr(_f8(a)), _ptr(_f8(R)),
                                       _ptr(_f8(D)), _ptr(_f8(Rmod)), _ptr(lnz1), _ptr(lnM), len(extra), _ptr_array(extra),
                                       ndim, _ptr(tn), tax, _ptr(table.values), int(table.rdelta_sampling),
                                       float(eps_runner), table.eps_model, _ptr(off), _ptr(counts))
    assert tot == counts.sum()
    return (off, counts) if return_counts else off


def regrid(nside, orig_map, pix_offsets):
    orig_map, pix_offsets = _f8(orig_map), _f8(pix_offsets)
    new_map = np.zeros(orig_map.size)
    lib().bfgo_regrid(nside, _ptr(orig_map), _ptr(pix_offsets), _ptr(new_map))
    return new_map


def baryonify_shell(nside, orig_map, cat, table, eps_runner, bg_runner, bg_model=None, md_runner=(200.0, 'critical'),
                    md_model=(200.0, 'critical')):
    """BaryonifyShell.process(), HealpixRunner.py:240-349 (incl. the mass-conservation assert)."""
    off = baryonify_offsets(nside, cat, table, eps_runner, bg_runner, bg_model, md_runner=md_runner, md_model=md_model)
    new_map = regrid(nside, orig_map, off)
    new_sum, old_sum = np.sum(new_map), np.sum(orig_map)
    assert np.isclose(new_sum, old_sum), "ERROR in pixel regridding"
    return new_map


def paint_shell(nside, cat, log_table, eps_runner, bg_runner, return_counts=False, md_runner=(200.0, 'critical')):
    """PaintProfilesShell.process(), HealpixRunner.py:366-447, for a (Param)TabulatedProfile whose
    interpolator holds log(raw_input_2D)."""
    a, R, D, _ = halo_scalars(cat, bg_runner, None, md_runner)
    ra, dec, M = _f8(cat['ra']), _f8(cat['dec']), _f8(cat['M'])
    extra = [_f8(cat[k]) for k in log_table.p_keys]
    ndim, tn, tax = log_table._cargs()
    new_map = np.zeros(12 * nside * nside)
    counts = np.zeros(ra.size, dtype=np.int64)
    lnz1, lnM = table_coords(cat)
    lib().bfgo_paint(nside, ra.size, _ptr(ra), _ptr(dec), _ptr(M), _ptr(_f8(a)), _ptr(_f8(R)), _ptr(_f8(D)),
                     _ptr(lnz1), _ptr(lnM), len(extra), _ptr_array(extra), ndim, _ptr(tn), tax, _ptr(log_table.values),
                     float(eps_runner), _ptr(new_map), _ptr(counts))
    return (new_map, counts) if return_counts else new_map


def baryonify_shell_threads(nside, orig_map, cat, table, eps_runner, bg_runner, threads, bg_model=None):
    """The same scalar restatement spread over `threads` host threads (ctypes releases the GIL): halo slices with
    private pix_offsets buffers that are summed, then pixel ranges with private new_map buffers.  Only used to time a
    multi-core CPU baseline; returns (new_map, seconds_loop, seconds_regrid, pairs)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    a, R, D, Rmod = (_f8(v) for v in halo_scalars(cat, bg_runner, bg_model))
    ra, dec, M = _f8(cat['ra']), _f8(cat['dec']), _f8(cat['M'])
    extra = [_f8(cat[k]) for k in table.p_keys]
    ndim, tn, tax = table._cargs()
    npix = 12 * nside * nside
    n = ra.size
    cuts = np.linspace(0, n, threads + 1).astype(np.int64)
    lnz1, lnM = table_coords(cat)
    L = lib()

    def loop(i):
        lo, hi = int(cuts[i]), int(cuts[i + 1])
        off = np.zeros((npix, 3))
        ex = [e[lo:hi] for e in extra]
        tot = L.bfgo_baryonify_offsets(nside, hi - lo, _ptr(ra[lo:hi]), _ptr(dec[lo:hi]), _ptr(M[lo:hi]), _ptr(a[lo:hi]), _ptr(R[lo:hi]),
                                       _ptr(D[lo:hi]), _ptr(Rmod[lo:hi]), _ptr(lnz1[lo:hi]), _ptr(lnM[lo:hi]), len(ex), _ptr_array(ex), ndim, _ptr(tn), tax,
                                       _ptr(table.values), int(table.rdelta_sampling), float(eps_runner), table.eps_model,
                                       _ptr(off), None)
        return off, tot

    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex_:
        parts = list(ex_.map(loop, range(threads)))
    off = parts[0][0]
    for o_, _ in parts[1:]:
        off += o_
    pairs = sum(t for _, t in parts)
    t1 = time.time()
    orig_map = _f8(orig_map)
    pcuts = np.linspace(0, npix, threads + 1).astype(np.int64)

    def rg(i):
        nm = np.zeros(npix)
        L.bfgo_regrid_range(nside, int(pcuts[i]), int(pcuts[i + 1]), _ptr(orig_map), _ptr(off), _ptr(nm))
        return nm

    with ThreadPoolExecutor(threads) as ex_:
        maps = list(ex_.map(rg, range(threads)))
    new_map = maps[0]
    for m_ in maps[1:]:
        new_map += m_
    t2 = time.time()
    return new_map, t1 - t0, t2 - t1, pairs
