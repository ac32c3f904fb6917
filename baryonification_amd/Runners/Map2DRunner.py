"""
Regular-grid runners: drop-in for BaryonForge/Runners/Map2DRunner.py (`regrid_pixels_2D` :14-83, `regrid_pixels_3D`
:86-163, `DefaultRunnerGrid` :168-375, `BaryonifyGrid` :379-607, `PaintProfilesGrid` :610-817).  Same constructor
(positional order, attribute names), same `process()` return value (float64 array of the map's shape) and the same
exceptions.  `process()` does no per-halo work in Python except the 2x2 shear matrices of `use_ellipticity = True`:
catalog, map and the model's raw table go through the C ABI (include/bfgx.h, "regular-grid path") to the HIP kernels
in csrc/bfgx_grid.hpp.  There is no CPU fallback.

Not built: `PaintProfilesAnisGrid` (:820-942).
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..utils.cosmology import MassDef
from ..utils.Tabulate import ParamTabulatedProfile
from ._model import build_model

__all__ = ['DefaultRunnerGrid', 'BaryonifyGrid', 'PaintProfilesGrid', 'regrid_pixels_2D', 'regrid_pixels_3D']


def _regrid(grid, pix_positions, pix_values, ndim, device=0):
    if not (isinstance(grid, np.ndarray) and grid.dtype == np.float64 and grid.flags.c_contiguous and grid.ndim == ndim):
        raise ValueError("grid must be a C-contiguous float64 array with %d equal axes" % ndim)
    if len(set(grid.shape)) != 1:
        raise ValueError("grid must be square / cubic")
    pos = _lib.f8(pix_positions)
    val = _lib.f8(pix_values)
    if pos.shape != (val.size, ndim):
        raise ValueError("pix_positions must have shape (N, %d)" % ndim)
    rc = _lib.load().bfgx_regrid_pixels(int(device), ndim, grid.shape[0], val.size, pos.ctypes.data, val.ctypes.data,
                                        grid.ctypes.data)
    _lib.check(rc)


def regrid_pixels_2D(grid, pix_positions, pix_values):
    """Adds `pix_values`, spread over the unit squares at `pix_positions` (x, y), to the periodic square `grid`
    IN PLACE (grid[i, j]: i from y, j from x), as Map2DRunner.py:14-83."""
    _regrid(grid, pix_positions, pix_values, 2)


def regrid_pixels_3D(grid, pix_positions, pix_values):
    """3-D version (grid[i, j, k]: i from y, j from x, k from z), as Map2DRunner.py:86-163."""
    _regrid(grid, pix_positions, pix_values, 3)


class DefaultRunnerGrid(object):

    def __init__(self, HaloNDCatalog, GriddedMap, epsilon_max, model, use_ellipticity=False,
                 mass_def=MassDef(200, 'critical'), verbose=True):
        self.HaloNDCatalog = HaloNDCatalog
        self.GriddedMap = GriddedMap
        self.cosmo = HaloNDCatalog.cosmology
        self.model = model
        self.epsilon_max = epsilon_max
        self.mass_def = mass_def
        self.verbose = verbose
        self.use_ellipticity = use_ellipticity
        self.device = 0                # engine knobs (not in the reference): plain attributes, picklable
        self.last_stats = None
        if use_ellipticity:            # Map2DRunner.py:271-277
            names = HaloNDCatalog.cat.dtype.names
            assert 'q_ell' in names, "The 'q_ell' column is missing, but you set use_ellipticity = True"
            if not GriddedMap.is2D:
                assert 'c_ell' in names, "The 'c_ell' column is missing, but you set use_ellipticity = True"
            assert 'A_ell' in names, "The 'A_ell' column is missing, but you set use_ellipticity = True"

    def build_Rmat(self, A, q):
        """Shear matrix of a halo with orientation vector A (normalised IN PLACE) and axis ratio q: the reduced shear g of galsim's
        Shear(q, beta), |g| = tanh(eta / 2) with eta = -ln q, as [[1 + g1, g2], [g2, 1 - g1]] / sqrt(1 - |g|^2) (Map2DRunner.py:283-337).
        The operations and their order are the reference's: the matrices enter the parity tests bit for bit."""
        A /= np.linalg.norm(A)
        ndim = len(A)
        if ndim == 1:
            raise ValueError("Can't rotate a 1-dimensional vector")
        if ndim != 2:
            raise NotImplementedError("This method has not yet been verified. Use 2D ellipticity method instead")
        beta = np.arccos(np.dot(A, np.array([1., 0.])))                 # position angle against the first axis
        eta = -np.log(q)
        if eta > 1e-4:
            ratio = np.tanh(0.5 * eta) / eta                             # |g| / eta
        else:                                                            # its series about eta = 0
            e2 = eta * eta
            ratio = 0.5 + e2 * ((-1 / 24) + e2 * (1 / 240))
        g = ratio * eta * np.exp(2j * beta)
        return np.array([[1 + g.real, g.imag], [g.imag, 1 - g.real]]) / np.sqrt(1 - np.abs(g) ** 2)

    def coord_array(self, *args):
        """(N, len(args)) array of the flattened arguments, one per column (Map2DRunner.py:340-358)"""
        return np.stack([np.ravel(a) for a in args], axis=1)

    def pick_indices(self, center, width, Npix):
        """the 2 width indices around `center` on a periodic axis of Npix cells, wrapped once either way (Map2DRunner.py:361-391)"""
        idx = np.arange(center - width, center + width)
        idx = idx + Npix * (idx < 0)
        return idx - Npix * (idx >= Npix)

    # -- shared plumbing -----------------------------------------------------------------------
    def _check_keys(self, keys):
        if len(keys) > 0:                                             # Map2DRunner.py:471-474, :703-706
            txt = (f"You asked to use {keys} properties in Baryonification. You must pass a ParamTabulatedProfile"
                   f"as the model. You have passed {type(self.model)} instead")
            ok = isinstance(self.model, ParamTabulatedProfile) or type(self.model).__name__ == 'ParamTabulatedProfile'
            assert ok, txt

    def _runner_cosmo(self):
        """ccl.Cosmology(Omega_c, Omega_b, h, sigma8, n_s) of Map2DRunner.py:456-459: w0 is NOT passed on"""
        d = dict(self.cosmo)
        d['w0'] = -1.0
        return d

    def _rmats(self, what):
        """per-halo shear matrices, evaluated exactly as the runner does (float32 catalog columns, :490-493, :528)"""
        cat = self.HaloNDCatalog.cat
        out = np.zeros((cat.size, 2, 2))
        for j in range(cat.size):
            q_j = cat['q_ell'][j]
            A_j = cat['A_ell'][j]
            A_j = A_j / np.sqrt(np.sum(A_j ** 2))
            assert q_j > 0, "The axis ratio in halo %d is %s" % (j, what)
            out[j] = self.build_Rmat(A_j, q_j)
        return out

    def _catalog(self, keys, rmat):
        cat = self.HaloNDCatalog.cat
        is2D = self.GriddedMap.is2D
        # the read-out takes np.log of the float32 catalog mass, i.e. a float32 logarithm (BaryonCorrection.py:369,
        # Tabulate.py:283); evaluated here with the caller's numpy so that its last bit is the reference's
        with np.errstate(invalid='ignore', divide='ignore'):      # invalid masses are skipped by the kernels
            lnM = np.log(np.asarray(cat['M'], dtype=np.float32)).astype(np.float64)
        return _lib.make_grid_catalog_host(cat['M'], cat['x'], cat['y'], None if is2D else cat['z'], lnM, rmat,
                                           [cat[k] for k in keys])

    def _grid(self):
        G = self.GriddedMap
        if len(G.bins) != G.Npix:
            raise ValueError("GriddedMap.bins must hold one pixel-centre coordinate per pixel (%d != %d)" % (len(G.bins), G.Npix))
        return _lib.make_grid(G.bins, 2 if G.is2D else 3, self.HaloNDCatalog.redshift)


class BaryonifyGrid(DefaultRunnerGrid):
    """Displaces the mass of a periodic 2D / 3D grid around every halo (the grid must hold MASS, not density:
    pixels equal to 0 are empty, Map2DRunner.py:387-388)."""

    def process(self):
        keys = vars(self.model).get('p_keys', [])
        self._check_keys(keys)
        rmat = None
        if self.use_ellipticity:
            if not self.GriddedMap.is2D:
                raise NotImplementedError("Currently not able to ellipticities with 3D maps.")    # :559
            rmat = self._rmats("not positive")
        model, p_keys, keep = build_model(self, 'displacement', self._runner_cosmo())
        cat, cols = self._catalog(p_keys, rmat)
        grid, gkeep = self._grid()
        orig_map = _lib.f8(self.GriddedMap.map)
        new_map = _lib.pinned_empty(orig_map.size).reshape(orig_map.shape)     # page-locked and pooled: the copy back runs at the PCIe rate
        opts = _lib.bfgx_opts(int(self.device), 1, 1, 1, 1, 0)
        stats = _lib.bfgx_stats()
        rc = _lib.load().bfgx_baryonify_grid(C.byref(cat), C.byref(model), C.byref(grid), orig_map.ctypes.data,
                                             new_map.ctypes.data, C.byref(opts), C.byref(stats))
        _lib.check(rc)
        self.last_stats = {k: getattr(stats, k) for k, _ in stats._fields_}
        del keep, cols, gkeep
        return new_map


class PaintProfilesGrid(DefaultRunnerGrid):
    """Paints a tabulated profile around every halo into an empty grid: the projected profile on 2D maps, the real
    (3-D) profile on 3D maps (Map2DRunner.py:750, :776)."""

    def process(self):
        keys = vars(self.model).get('p_keys', []) if self.model is not None else []
        self._check_keys(keys)
        assert self.model is not None, "You must provide a model"
        rmat = None
        if self.use_ellipticity:
            if not self.GriddedMap.is2D:
                raise ValueError("use_ellipticity is not implemented for 3D maps")                # :784
            rmat = self._rmats("zero")
        model, p_keys, keep = build_model(self, 'projected' if self.GriddedMap.is2D else 'real', self._runner_cosmo())
        cat, cols = self._catalog(p_keys, rmat)
        grid, gkeep = self._grid()
        new_map = _lib.pinned_empty(int(np.prod(self.GriddedMap.map.shape))).reshape(self.GriddedMap.map.shape)
        opts = _lib.bfgx_opts(int(self.device), 1, 1, 0, 1, 0)
        stats = _lib.bfgx_stats()
        rc = _lib.load().bfgx_paint_grid(C.byref(cat), C.byref(model), C.byref(grid), new_map.ctypes.data, C.byref(opts),
                                         C.byref(stats))
        _lib.check(rc)
        self.last_stats = {k: getattr(stats, k) for k, _ in stats._fields_}
        del keep, cols, gkeep
        return new_map
