"""
Hot-path input containers, interface-compatible with BaryonForge/utils/io.py:9-140 (HaloLightConeCatalog)
and :290-375 (LightconeShell): same constructor arguments, same attributes (`cat`, `cosmo`, `map`,
`NSIDE`), same ValueError on a cosmology dict with missing keys, same pole clipping.
"""
import warnings

import numpy as np

__all__ = ['HaloLightConeCatalog', 'LightconeShell']

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
