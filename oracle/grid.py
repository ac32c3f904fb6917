"""
grid.py -- numpy/ctypes front-end of the CPU oracle for the regular-grid path (TEST INFRASTRUCTURE ONLY;
same import rules as oracle.py).

Restates, in float64 on the CPU:

  * Map2DRunner.py:14-163   -- regrid_pixels_2D / regrid_pixels_3D            (C: bfgo_regrid_pixels)
  * Map2DRunner.py:431-607  -- BaryonifyGrid.process                           (C: bfgo_grid_loop mode 0 + regrid)
  * Map2DRunner.py:676-817  -- PaintProfilesGrid.process                       (C: bfgo_grid_loop mode 1)
  * Map2DRunner.py:283-318  -- DefaultRunnerGrid.build_Rmat (2D shear matrix)
  * io.py:622-670           -- ParticleSnapshot.make_map (np.histogramdd)      (C: bfgo_histogramdd)
  * examples/10_Reproduce_Schneider_deltaPk.ipynb cells 12, 15 -- the FFT P(k) summary (numpy)

The grid runners build `ccl.Cosmology` WITHOUT w0 (Map2DRunner.py:456-459), i.e. w0 = -1 whatever the catalog's
cosmology dict says; `grid_background` reproduces that.
"""
import ctypes as C

import numpy as np

from . import oracle as O

_f8, _ptr, _ptr_array = O._f8, O._ptr, O._ptr_array
_READY = False


def lib():
    global _READY
    L = O.lib()
    if not _READY:
        i64, dbl, vp, ci = C.c_int64, C.c_double, C.c_void_p, C.c_int
        L.bfgo_regrid_pixels.argtypes = [ci, i64, i64, vp, vp, vp]
        L.bfgo_grid_loop.argtypes = [ci, ci, i64, vp, i64, vp, vp, vp, vp, dbl, vp, vp, vp, ci, vp,
                                     ci, vp, vp, vp, ci, dbl, dbl, vp, vp]
        L.bfgo_grid_loop.restype = i64
        L.bfgo_histogramdd.argtypes = [ci, i64, vp, vp, i64, vp, vp]
        _READY = True
    return L


def grid_background(cosmo):
    """Map2DRunner.py:456-459: Omega_c, Omega_b, h, sigma8, n_s only -> w0 = -1"""
    return O.Background(cosmo['Omega_m'], cosmo['Omega_b'], cosmo['h'], cosmo.get('sigma8'), cosmo.get('n_s'), -1.0)


def regrid_pixels(grid, pix_positions, pix_values):
    """regrid_pixels_2D / regrid_pixels_3D: accumulates into `grid` in place (and returns it)"""
    assert grid.dtype == np.float64 and grid.flags.c_contiguous
    pos, val = _f8(pix_positions), _f8(pix_values)
    ndim = grid.ndim
    assert pos.shape == (val.size, ndim)
    lib().bfgo_regrid_pixels(ndim, grid.shape[0], val.size, _ptr(pos), _ptr(val), _ptr(grid))
    return grid


def build_Rmat(A, q):
    """DefaultRunnerGrid.build_Rmat for 2-vectors (Map2DRunner.py:283-318), vectorised over halos.
    A [n, 2] (normalised inside, as the reference does), q [n] -> [n, 2, 2]"""
    A = np.asarray(A, dtype=np.float64)
    A = A / np.linalg.norm(A, axis=1)[:, None]
    q = np.asarray(q, dtype=np.float64)
    beta = np.arccos(A[:, 0])
    eta = -np.log(q)
    etasq = eta * eta
    with np.errstate(divide='ignore', invalid='ignore'):
        eta2g = np.where(eta > 1e-4, np.tanh(0.5 * eta) / eta, 0.5 + etasq * ((-1 / 24) + etasq * (1 / 240)))
    g = eta2g * eta * np.exp(2j * beta)
    g1, g2 = g.real, g.imag
    det = np.sqrt(1 - np.abs(g) ** 2)
    R = np.empty((q.size, 2, 2))
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1] = 1 + g1, g2, g2, 1 - g1
    return R / det[:, None, None]


def _loop(mode, shape, bins, cat, redshift, table, eps_runner, bg_runner, bg_model, rmat):
    ndim = len(shape)
    npix = shape[0]
    bins = _f8(bins)
    a = 1.0 / (1.0 + redshift)
    M = _f8(cat['M'])
    # float32 logarithm of the float32 catalog mass, as the reference's read-out takes it (see bfg_oracle.c)
    lnM = _f8(cat['lnM']) if 'lnM' in cat else np.log(M.astype(np.float32)).astype(np.float64)
    hx, hy, hz = _f8(cat['x']), _f8(cat['y']), _f8(cat['z'])
    R = _f8(bg_runner.get_radius(M, a))
    Rmod = _f8((bg_model or bg_runner).get_radius(M, a) / a)
    extra = [_f8(cat[k]) for k in table.p_keys]
    tdim, tn, tax = table._cargs()
    out = np.zeros((npix ** ndim, ndim) if mode == 0 else npix ** ndim)
    rm = None if rmat is None else _f8(rmat).reshape(-1, 4)
    fail = C.c_int(0)
    pairs = lib().bfgo_grid_loop(mode, ndim, npix, _ptr(bins), M.size, _ptr(hx), _ptr(hy), _ptr(hz), _ptr(lnM), a,
                                 _ptr(R), _ptr(Rmod), None if rm is None else _ptr(rm), len(extra), _ptr_array(extra),
                                 tdim, _ptr(tn), tax, _ptr(table.values), int(table.rdelta_sampling), float(eps_runner),
                                 table.eps_model, _ptr(out), C.byref(fail))
    assert not fail.value, "Halo offsets are larger than res"
    return out, pairs


def baryonify_grid_offsets(shape, bins, cat, redshift, table, eps_runner, bg_runner, bg_model=None, rmat=None,
                           return_pairs=False):
    out, pairs = _loop(0, shape, bins, cat, redshift, table, eps_runner, bg_runner, bg_model, rmat)
    return (out, pairs) if return_pairs else out


def regrid_offsets(orig_map, pix_offsets):
    """Map2DRunner.py:577-599: non-finite offsets -> 0, add the pixel's own (x, y[, z]) index, regrid"""
    shape = orig_map.shape
    N = shape[0]
    x = np.arange(N)
    off = np.where(np.isfinite(pix_offsets), pix_offsets, 0)
    grids = np.meshgrid(*([x] * len(shape)), indexing='xy')
    for c, g in enumerate(grids):
        off[:, c] += g.flatten()
    new_map = np.zeros(shape)
    regrid_pixels(new_map, off, orig_map.flatten())
    return new_map


def baryonify_grid(orig_map, bins, cat, redshift, table, eps_runner, bg_runner, bg_model=None, rmat=None):
    """BaryonifyGrid.process() incl. the mass-conservation assert (Map2DRunner.py:601-605)"""
    orig_map = _f8(orig_map)
    off = baryonify_grid_offsets(orig_map.shape, bins, cat, redshift, table, eps_runner, bg_runner, bg_model, rmat)
    new_map = regrid_offsets(orig_map, off)
    assert np.isclose(np.sum(new_map), np.sum(orig_map)), "ERROR in pixel regridding"
    return new_map


def paint_grid(shape, bins, cat, redshift, log_table, eps_runner, bg_runner, rmat=None, return_pairs=False):
    """PaintProfilesGrid.process() for a (Param)TabulatedProfile; `log_table` holds log(raw_input_2D) for 2D maps
    and log(raw_input_3D) for 3D maps (Map2DRunner.py:750, :776)"""
    out, pairs = _loop(1, shape, bins, cat, redshift, log_table, eps_runner, bg_runner, None, rmat)
    out = out.reshape(shape)
    return (out, pairs) if return_pairs else out


def make_map(coords, mass, L, N_grid):
    """ParticleSnapshot.make_map (io.py:622-670): histogramdd on linspace(0, L, N_grid + 1) edges"""
    edges = np.linspace(0, L, N_grid + 1)
    cs = [_f8(c) for c in coords]
    w = _f8(mass)
    out = np.zeros((N_grid,) * len(cs))
    lib().bfgo_histogramdd(len(cs), w.size, _ptr_array(cs), _ptr(w), edges.size, _ptr(edges), _ptr(out))
    return out


def power_spectrum(Map, Lbox, Nk=180):
    """examples/10_Reproduce_Schneider_deltaPk.ipynb cells 12 + 15: |FFT|^2 averaged in Nk linear k-bins between the
    fundamental and the Nyquist frequency.  Returns (k_cen, Pk, k_count)."""
    Ngrd = Map.shape[0]
    kbins = np.linspace(2 * np.pi / Lbox, 2 * np.pi / Lbox * Ngrd / 2, Nk + 1)
    klin = np.fft.fftfreq(Ngrd, 1 / (2 * np.pi / (Lbox)) / Ngrd)
    k = np.sqrt(klin[:, None, None] ** 2 + klin[None, None, :] ** 2 + klin[None, :, None] ** 2).flatten()
    kinds = np.floor((k - kbins[0]) / (kbins[1] - kbins[0])).astype(int)
    kmsk = (kinds >= 0) & (kinds < Nk)
    k_c = np.bincount(kinds[kmsk], minlength=Nk)
    with np.errstate(invalid='ignore', divide='ignore'):
        k_cen = np.bincount(kinds[kmsk], minlength=Nk, weights=k[kmsk]) / k_c
        F = np.fft.fftn(Map)
        P = (np.conjugate(F) * F).real.flatten()
        Pk = np.bincount(kinds[kmsk], minlength=Nk, weights=P[kmsk]) / k_c
    return k_cen, Pk, k_c


def baryonify_snapshot(coords, L, cat, redshift, table, eps_runner, bg_runner, bg_model=None, return_pairs=False):
    """BaryonifySnapshot.process() (SnapshotRunner.py:199-262): coords = [x, y(, z)] particle columns in [0, L];
    returns the displaced, re-wrapped columns.  Brute force over halo x particle (C: bfgo_snapshot_offsets)."""
    L_ = lib()
    if not hasattr(L_, '_snap_ready'):
        i64, dbl, vp, ci = C.c_int64, C.c_double, C.c_void_p, C.c_int
        L_.bfgo_snapshot_offsets.argtypes = [ci, i64, vp, vp, vp, dbl, i64, vp, vp, vp, vp, dbl, vp, vp, ci, vp, vp, vp, ci, dbl, dbl, vp]
        L_.bfgo_snapshot_offsets.restype = i64
        L_._snap_ready = True
    ndim = len(coords)
    cs = [_f8(c) for c in coords]
    a = 1.0 / (1.0 + redshift)
    M = _f8(cat['M'])
    lnM = np.log(M.astype(np.float32)).astype(np.float64)       # float32 log of the float32 catalog mass (see _loop)
    hx, hy, hz = _f8(cat['x']), _f8(cat['y']), _f8(cat['z'])
    R = _f8(bg_runner.get_radius(M, a))
    Rmod = _f8((bg_model or bg_runner).get_radius(M, a) / a)
    tdim, tn, tax = table._cargs()
    off = np.zeros((cs[0].size, ndim))
    pairs = L_.bfgo_snapshot_offsets(ndim, cs[0].size, _ptr(cs[0]), _ptr(cs[1]), _ptr(cs[2]) if ndim == 3 else None, float(L), M.size,
                                     _ptr(hx), _ptr(hy), _ptr(hz), _ptr(lnM), a, _ptr(R), _ptr(Rmod), tdim, _ptr(tn), tax,
                                     _ptr(table.values), int(table.rdelta_sampling), float(eps_runner), table.eps_model, _ptr(off))
    out = []
    for c in range(ndim):                                       # :254-262
        v = cs[c] + off[:, c]
        v = np.where(v > L, v - L, v)
        v = np.where(v < 0, v + L, v)
        out.append(v)
    return (out, pairs) if return_pairs else out
