"""
Pixel window functions and the convolved profile: drop-in for BaryonForge/utils/Pixel.py (`ConvolvedProfile` :10-224,
`GridPixelApprox` :229-366, `HealPixel` :369-470, `NoPix` :473-582).  Same class names, constructor arguments, attributes
(`Profile`, `Pixel`, `fft_par`, `isHarmonic`, `size`, `NSIDE`) and method signatures.

`ConvolvedProfile.real / .projected` evaluate the wrapped profile on the FFTLog grid on the host (as the reference does) and
hand the rows to the GPU (C ABI `bfgx_fftlog_convolve`: FFTLog forward, x window, FFTLog back, PCHIP in ln r).  The
reference's transform is `pyccl.pyutils._fftlog_transform`; pyccl is not available, so the transform implemented is the
published FFTLog algorithm (Hamilton 2000) -- see include/bfgx.h; parity with CCL's C code is unpinned.
"""
import ctypes as C

import numpy as np

from .. import _lib

__all__ = ['ConvolvedProfile', 'GridPixelApprox', 'HealPixel', 'NoPix']

# pyccl 2.8.0 HaloProfile.precision_fftlog defaults (profiles without the attribute get these)
FFTLOG_DEFAULTS = {'padding_lo_fftlog': 0.1, 'padding_lo_extra': 0.1, 'padding_hi_fftlog': 10.0, 'padding_hi_extra': 10.0,
                   'large_padding_2D': False, 'n_per_decade': 100, 'extrapol': 'linx_liny',
                   'plaw_fourier': -1.5, 'plaw_projected': -1.0}


def fftlog_transform(rs, frs, dim, mu, power_law_index, device=0):
    """ks, fks = pyccl.pyutils._fftlog_transform(rs, frs, dim, mu, power_law_index), on the GPU"""
    rs = _lib.f8(rs)
    f = _lib.f8(np.atleast_2d(frs))
    if rs.ndim != 1 or f.shape[1] != rs.size:
        raise ValueError("rs should be a 1D array and frs should have len(rs) columns")
    k = np.empty_like(rs)
    out = np.empty_like(f)
    _lib.check(_lib.load().bfgx_fftlog_transform(int(device), f.shape[0], rs.size, rs.ctypes.data, f.ctypes.data, int(dim), float(mu),
                                                 float(power_law_index), k.ctypes.data, out.ctypes.data))
    return k, (out[0] if np.ndim(frs) == 1 else out)


def _kgrid(r, dim, mu, plaw):
    r = _lib.f8(r)
    k = np.empty_like(r)
    _lib.check(_lib.load().bfgx_fftlog_kgrid(r.size, r.ctypes.data, int(dim), float(mu), float(plaw), k.ctypes.data))
    return k


def _convolve(r_fft, prof, dim, plaw_fwd, plaw_back, window, r_eval, r_scale, device=0):
    r_fft, prof2, window, r_eval = _lib.f8(r_fft), _lib.f8(np.atleast_2d(prof)), _lib.f8(window), _lib.f8(r_eval)
    out = np.empty((prof2.shape[0], r_eval.size))
    _lib.check(_lib.load().bfgx_fftlog_convolve(int(device), prof2.shape[0], r_fft.size, r_fft.ctypes.data, prof2.ctypes.data, int(dim), 0.0,
                                                float(plaw_fwd), float(plaw_back), window.ctypes.data, r_eval.size, r_eval.ctypes.data,
                                                float(r_scale), out.ctypes.data))
    return out


class ConvolvedProfile(object):
    """A profile convolved with an (isotropic) pixel window function (Pixel.py:10-224)."""

    def __init__(self, Profile, Pixel):
        self.Profile = Profile
        self.Pixel = Pixel
        self.fft_par = getattr(Profile, 'precision_fftlog', None) or dict(FFTLOG_DEFAULTS)
        self.isHarmonic = Pixel.isHarmonic

    def __getattr__(self, name):
        # delegate everything else to the wrapped profile (Pixel.py:76-97)
        if name in ('Profile', 'Pixel', 'fft_par', 'isHarmonic', '__setstate__', '__getstate__'):
            raise AttributeError(name)
        return getattr(self.Profile, name)

    def __getstate__(self):
        return self.__dict__.copy()

    def __setstate__(self, state):
        self.__dict__.update(state)

    def _fft_grid(self, r):
        """Pixel.py:134-139 / :196-201"""
        r_min = np.min([np.min(r) * self.fft_par['padding_lo_fftlog'], 1e-8])
        r_max = np.max([np.max(r) * self.fft_par['padding_hi_fftlog'], 1e3])
        n = self.fft_par['n_per_decade'] * np.int32(np.log10(r_max / r_min))
        return np.geomspace(r_min, r_max, n)

    def real(self, cosmo, r, M, a):
        r_in = np.atleast_1d(np.asarray(r, dtype=np.float64))
        r_fft = self._fft_grid(r_in)
        prof = np.atleast_2d(self.Profile.real(cosmo, r_fft, M, a))
        plaw = self.fft_par['plaw_fourier']
        window = self.Pixel.real(_kgrid(r_fft, 3, 0, plaw))                                   # :146-147
        r_eval = np.clip(r_in, self.Pixel.size / 5, None)                                     # :153
        out = _convolve(r_fft, prof, 3, plaw, plaw + 1, window, r_eval, 1.0)
        return _mirror(out, r, M)

    def projected(self, cosmo, r, M, a):
        r_in = np.atleast_1d(np.asarray(r, dtype=np.float64))
        D_A = 1.0
        if self.isHarmonic:
            assert a < 1, f"You cannot set a = 1, z = 0 when computing harmonic sky projections"
            D_A = _comoving_angular_distance(cosmo, a)                                        # :191
        r_fft = self._fft_grid(r_in)
        prof = np.atleast_2d(self.Profile.projected(cosmo, r_fft, M, a))
        r_t = r_fft / D_A if self.isHarmonic else r_fft                                       # :205
        plaw = self.fft_par['plaw_fourier'] + 1
        window = self.Pixel.projected(_kgrid(r_t, 2, 0, plaw))                                # :208-209
        r_eval = np.clip(r_in, self.Pixel.size / 5 * D_A, None)                               # :217-219
        out = _convolve(r_t, prof, 2, plaw, plaw, window, r_eval, D_A)
        return _mirror(out, r, M)


def _mirror(out, r, M):
    """the wrapped profile's rank convention: scalar r / scalar M drop their axis"""
    if np.ndim(r) == 0:
        out = np.squeeze(out, axis=-1)
    if np.ndim(M) == 0:
        out = np.squeeze(out, axis=0)
    return out


def _comoving_angular_distance(cosmo, a):
    """ccl.comoving_angular_distance(cosmo, a) [Mpc] for the flat cosmologies this package supports"""
    from .cosmology import Cosmology, cosmo_to_dict
    c = cosmo if isinstance(cosmo, Cosmology) else Cosmology.from_dict(cosmo_to_dict(cosmo))
    return float(c.angular_diameter_distance(float(a))) / float(a)


class GridPixelApprox(object):
    """The window of a square grid pixel approximated by a circular / spherical top hat of equal area / volume
    (Pixel.py:229-366)."""
    isHarmonic = False

    def __init__(self, size):
        self.size = size

    def beam(self, k, R):
        from scipy import special
        kr = k * (2 * R)                           # factor of 2: the reference passes the diameter (Pixel.py:322)
        with np.errstate(divide='ignore', invalid='ignore'):
            beam = np.where(kr > 0, 3 * special.spherical_jn(1, kr) / kr, 1)
        return beam

    def real(self, k):
        R = np.cbrt(self.size ** 3 / (4 / 3 * np.pi))
        return self.beam(k, R)

    def projected(self, k):
        R = np.sqrt(self.size ** 2 / np.pi)
        return self.beam(k, R)


class HealPixel(object):
    """HEALPix pixel window approximated by a Gaussian beam of FWHM = pixel size / sqrt(2) (Pixel.py:369-470)."""
    isHarmonic = True

    def __init__(self, NSIDE):
        self.NSIDE = NSIDE
        self.size = np.sqrt(4 * np.pi / (12 * NSIDE ** 2))         # hp.nside2resol(NSIDE) [rad]

    def real(self, k):
        return np.zeros_like(k)                    # no real-space form: the convolved profile comes out 0 (Pixel.py:519-521)

    def projected(self, k):
        sig = self.size / np.sqrt(8 * np.log(2)) / np.sqrt(2)
        return np.exp(-k * (1 + k) / 2 * sig ** 2)


class NoPix(object):
    """No pixel window at all (Pixel.py:473-582).  The reference's class lacks `isHarmonic` / `size`, without which
    ConvolvedProfile cannot be built around it; they are supplied here (physical space, zero size)."""
    isHarmonic = False
    size = 0.0

    def __init__(self):
        pass

    def real(self, k):
        return np.ones_like(k)

    def projected(self, k):
        return np.ones_like(k)
