"""
Hot-path input containers, interface-compatible with BaryonForge/utils/io.py:9-140 (HaloLightConeCatalog),
:290-375 (LightconeShell), :143-287 (HaloNDCatalog), :378-497 (GriddedMap) and :500-670 (ParticleSnapshot):
same constructor arguments, same attributes (`cat`, `cosmo`, `map`, `NSIDE`, `bins`, `res`, `Npix`, `is2D`, ...),
same ValueError on a cosmology dict with missing keys, same pole clipping, same (float32) catalog dtypes.
"""
import warnings

import numpy as np

__all__ = ['HaloLightConeCatalog', 'LightconeShell', 'HaloNDCatalog', 'GriddedMap', 'ParticleSnapshot']

_COSMO_KEYS = ('Omega_m', 'sigma8', 'h', 'Omega_b', 'n_s', 'w0')


def _check_cosmo(cosmo):
    if cosmo is None or not all(k in cosmo.keys() for k in _COSMO_KEYS):
        raise ValueError("Not all cosmology parameters provided. I need Omega_m, sigma8, h, sigma8, Omega_b, n_s, w0")
    return cosmo


def npix2nside(npix):
    nside = int(round(np.sqrt(npix / 12.0)))
    if nside < 1 or 12 * nside * nside != npix:
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")
    return nside


class HaloLightConeCatalog(object):
    """Structured float64 array `cat` with fields M, z, ra, dec (+ any extra per-halo columns)."""

    def __init__(self, ra, dec, M, z, cosmo, **arrays):
        t = np.float64
        dtype = [('M', t), ('z', t), ('ra', t), ('dec', t)] + [(name, t) for name in arrays]
        ra = np.atleast_1d(ra)
        dec = np.atleast_1d(np.asarray(dec))
        cat = np.zeros(len(ra), dtype)
        if np.any(np.abs(dec) == 90):            # io.py:65-68
            dec = dec.astype(t)
            warnings.warn("Some halos found with declination exactly at the poles. Offsetting these by 4e-5 arcsec")
            dec = np.clip(dec, -90 + 1e-8, 90 - 1e-8)
        cat['ra'], cat['dec'], cat['z'], cat['M'] = ra, dec, z, M
        for name, arr in arrays.items():
            cat[name] = arr
        self.cat = cat
        self.cosmo = _check_cosmo(cosmo)

    @property
    def data(self):
        return self.cat

    @property
    def cosmology(self):
        return self.cosmo

    def __getitem__(self, key):
        other = {k: self.cat[k][key] for k in self.cat.dtype.names if k not in ('ra', 'dec', 'M', 'z')}
        return HaloLightConeCatalog(ra=self.cat['ra'][key], dec=self.cat['dec'][key], M=self.cat['M'][key],
                                    z=self.cat['z'][key], cosmo=self.cosmo, **other)

    def __len__(self):
        return self.cat.size

    def __str__(self):
        return (f"HaloLightConeCatalog with {self.cat.size} Halos at {self.cat['z'].min()} < z < {self.cat['z'].max()}.\n"
                f"Minimum log10(Mass) = {np.log10(self.cat['M'].min())}\n"
                f"Maximum log10(Mass) = {np.log10(self.cat['M'].max())}\n"
                f"Cosmology set to {self.cosmo}.")


class LightconeShell(object):
    """HEALPix RING-ordered map + cosmology dict."""

    def __init__(self, map=None, path=None, cosmo=None):
        if (path is None) and (map is None):
            raise ValueError("Need to provide either path to map, or provide map values in healpix ring configuration")
        elif isinstance(path, str):
            try:
                import healpy as hp
            except ImportError as e:       # FITS I/O is outside the hot path
                raise ImportError("reading a map from `path` needs healpy; pass the array via `map=`") from e
            self.map = hp.read_map(path)
        elif isinstance(map, np.ndarray):
            self.map = map
        else:
            raise ValueError("`map` must be a numpy array")
        self.NSIDE = npix2nside(self.map.size)
        self.cosmo = _check_cosmo(cosmo)

    @property
    def data(self):
        return self.map

    @property
    def cosmology(self):
        return self.cosmo


class HaloNDCatalog(object):
    """Halos of a 2D / 3D periodic box at one redshift.  As in the reference the structured array `cat` holds
    big-endian float32 columns M, x, y, z (+ extra columns; 2-D inputs keep their trailing shape): positions and
    masses are rounded to float32 on the way in (io.py:205-219)."""

    def __init__(self, x, y, M, redshift, cosmo, z=None, **arrays):
        dtype = [('M', '>f'), ('x', '>f'), ('y', '>f'), ('z', '>f')]
        dtype = dtype + [(name, '>f', np.shape(arr)[1:] if np.ndim(arr) > 1 else '') for name, arr in arrays.items()]
        N = 1 if not isinstance(x, (list, np.ndarray, tuple)) else len(x)
        cat = np.zeros(N, dtype)
        cat['x'] = x
        cat['y'] = y
        cat['z'] = 0 if z is None else z
        cat['M'] = M
        for name, arr in arrays.items():
            cat[name] = arr
        self.cat = cat
        self.redshift = redshift
        self.cosmo = _check_cosmo(cosmo)

    @property
    def data(self):
        return self.cat

    @property
    def cosmology(self):
        return self.cosmo

    def __getitem__(self, key):
        other = {k: self.cat[k][key] for k in self.cat.dtype.names if k not in ('x', 'y', 'z', 'M')}
        return HaloNDCatalog(x=self.cat['x'][key], y=self.cat['y'][key], z=self.cat['z'][key], M=self.cat['M'][key],
                             redshift=self.redshift, cosmo=self.cosmo, **other)

    def __len__(self):
        return self.cat.size

    def __str__(self):
        return (f"HaloNDCatalog with {self.cat.size} Halos at z = {self.redshift}.\n"
                f"Minimum log10(Mass) = {np.log10(self.cat['M'].min())}\n"
                f"Maximum log10(Mass) = {np.log10(self.cat['M'].max())}\n"
                f"Cosmology set to {self.cosmo}.")

    def __repr__(self):
        return f"HaloNDCatalog(cat = {self.cat!r}, \nredshift = {self.redshift!r}, \ncosmo = {self.cosmo})"


class GriddedMap(object):
    """Square (2D) or cubic (3D) periodic map, `bins` = pixel-centre coordinates along one axis.  `grid` and `inds`
    (np.meshgrid of the bins, np.arange over the map; conveniences in the reference, io.py:476-486) are built on
    access, so that a 512^3 map does not carry 4 GB of them."""

    def __init__(self, map=None, redshift=None, bins=None, cosmo=None):
        self.map = map
        self.redshift = redshift
        self.Npix = self.map.shape[0]
        self.res = bins[1] - bins[0]
        self.bins = bins
        self.is2D = True if len(self.map.shape) == 2 else False
        if self.is2D:
            assert self.map.shape[0] == self.map.shape[1]
        else:
            assert (self.map.shape[0] == self.map.shape[1]) & (self.map.shape[1] == self.map.shape[2])
        self.cosmo = _check_cosmo(cosmo)

    @property
    def grid(self):
        b = self.bins
        return np.meshgrid(b, b, indexing='xy') if self.is2D else np.meshgrid(b, b, b, indexing='xy')

    @property
    def inds(self):
        shape = (len(self.bins),) * (2 if self.is2D else 3)
        return np.arange(int(np.prod(shape))).reshape(shape)

    @property
    def data(self):
        return self.map

    @property
    def cosmology(self):
        return self.cosmo

    def __str__(self):
        return (f"GriddedMap of {'2D' if self.is2D else '3D'} shape {self.map.shape}, res = {self.res}, z = {self.redshift}.\n"
                f"Cosmology set to {self.cosmo}.")


class ParticleSnapshot(object):
    """Particles of a periodic box; `make_map(N_grid)` histograms their masses on the GPU (io.py:622-670)."""

    def __init__(self, x=None, y=None, z=None, M=None, L=None, redshift=None, cosmo=None):
        dtype = [('M', np.float64), ('x', np.float64), ('y', np.float64), ('z', np.float64)]
        cat = np.zeros(len(x), dtype)
        cat['x'] = x
        cat['y'] = y
        cat['z'] = 0 if z is None else z
        cat['M'] = M
        self.L = L
        self.cat = cat
        self.redshift = redshift
        self.is2D = True if z is None else False
        self.cosmo = _check_cosmo(cosmo)
        self.device = 0

    @property
    def data(self):
        return self.cat

    @property
    def cosmology(self):
        return self.cosmo

    def make_map(self, N_grid):
        import ctypes as C
        from .. import _lib
        edges = np.linspace(0, self.L, N_grid + 1)
        ndim = 2 if self.is2D else 3
        rec = self.cat
        fields = rec.dtype.fields or {}
        need = ('x', 'y', 'M') if self.is2D else ('x', 'y', 'z', 'M')
        if (rec.ndim == 1 and rec.flags.c_contiguous and rec.dtype.itemsize % 8 == 0 and rec.dtype.itemsize >= 16 and
                all(k in fields and fields[k][0] == np.float64 and fields[k][1] % 8 == 0 for k in need)):
            # the records as they are: no strided gather of the columns on the host (the library also makes the reference's NaN check of the
            # masses, io.py:636, and raises the same AssertionError)
            out = _lib.pinned_empty(N_grid ** ndim).reshape((N_grid,) * ndim)
            rc = _lib.load().bfgx_deposit_particles_records(int(self.device), ndim, rec.size, rec.ctypes.data if rec.size else None, rec.dtype.itemsize,
                                                            fields['x'][1], fields['y'][1], 0 if self.is2D else fields['z'][1], fields['M'][1],
                                                            int(N_grid), edges.ctypes.data, out.ctypes.data)
            _lib.check(rc)
            return out
        assert np.isnan(self.cat['M']).sum() == 0, "If you want to make a map, provide a value for the particle mass"      # io.py:636
        x, y, m = _lib.f8(self.cat['x']), _lib.f8(self.cat['y']), _lib.f8(self.cat['M'])
        z = None if self.is2D else _lib.f8(self.cat['z'])
        out = np.empty((N_grid,) * ndim)
        rc = _lib.load().bfgx_deposit_particles(int(self.device), ndim, x.size, x.ctypes.data, y.ctypes.data,
                                                None if z is None else z.ctypes.data, m.ctypes.data, int(N_grid),
                                                edges.ctypes.data, out.ctypes.data)
        _lib.check(rc)
        return out
