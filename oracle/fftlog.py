"""
ORACLE (test infrastructure, never shipped): numpy restatement of the pixel-window convolution of
BaryonForge/utils/Pixel.py -- `ConvolvedProfile.real` (:106-157) and `.projected` (:160-224) -- and of the
third-party transform it calls, `pyccl.pyutils._fftlog_transform` (pyccl == 2.8.0, setup.py:25; the C file
ccl_fftlog.c, itself a C port of A. Hamilton's FFTLog, MNRAS 312 (2000) 257, Appendix B).

pyccl is not installable here and the reference holds no vector for this path, so **parity with CCL's
_fftlog_transform is unpinned**.  What is restated is the PUBLISHED algorithm (Hamilton 2000):

    fht:  b(k) = int a(r) (k r)^q J_mu(k r) k dr  on a log grid:  b = reverse( IFFT( FFT(a) u ) ),
          u_m = (k_c r_c)^(-2 pi i m / L) 2^z Gamma((mu+1+z)/2) / Gamma((mu+1-z)/2),  z = q + 2 pi i m / L,  L = N dln r,
          with the low-ringing choice of k_c r_c (FFTLog `krgood`) nearest to 1 and k_j r_(N-1-j) = k_c r_c
    3-D:  T3[f](k) = (2 pi)^-3  int d^3r f(r) j_0-kernel  = (2 pi k)^-3/2 k^-q  fht[ r^(3/2 - q) f ](mu = l + 1/2)
    2-D:  T2[f](k) = (2 pi)^-2  int d^2r f(r) J_mu(k r)   = k^-(1+q) / (2 pi) fht[ r^(1 - q) f ](mu)

with the bias exponent q = dim/2 + power_law_index (so that CCL's default power-law indices, -1.5 in 3-D and -1 in
2-D, give the unbiased transform).  In exact arithmetic the result does not depend on q.  The restatement is pinned
(tests/test_fftlog.py) against analytic Hankel pairs (Gaussian, power law x exponential) and against scipy.fft.fht,
an independent implementation of the same paper.
"""
import numpy as np
from scipy import interpolate, special

FFTLOG_DEFAULTS = {'padding_lo_fftlog': 0.1, 'padding_lo_extra': 0.1, 'padding_hi_fftlog': 10.0, 'padding_hi_extra': 10.0,
                   'large_padding_2D': False, 'n_per_decade': 100, 'extrapol': 'linx_liny',
                   'plaw_fourier': -1.5, 'plaw_projected': -1.0}     # pyccl HaloProfile.precision_fftlog defaults (2.8.0)


def _lngamma(x, y):
    """ln|Gamma(x + i y)|, arg Gamma(x + i y)"""
    g = special.loggamma(x + 1j * y)
    return g.real, g.imag


def good_lnkr(dln, mu, q, lnkr=0.0):
    """FFTLog's low-ringing value of ln(k_c r_c) closest to `lnkr` (Hamilton 2000 eq. B22; FFTLog `krgood`, ccl_fftlog.c goodkr)"""
    xp, xm = (mu + 1 + q) / 2, (mu + 1 - q) / 2
    y = np.pi / (2 * dln)
    _, argp = _lngamma(xp, y)
    _, argm = _lngamma(xm, y)
    arg = (np.log(2.0) - lnkr) / dln + (argp + argm) / np.pi
    return lnkr + (arg - np.round(arg)) * dln


def u_coefficients(N, dln, mu, q, lnkr):
    """u_m = (k_c r_c)^(-2 pi i m / L) U_mu(q + 2 pi i m / L), L = N dln, for m = 0 .. N/2 (Hamilton 2000 eq. B18; the
    coefficients of negative m are the conjugates)"""
    m = np.arange(N // 2 + 1)
    y = np.pi * m / (N * dln)
    xp, xm = (mu + 1 + q) / 2, (mu + 1 - q) / 2
    lnrp, phip = _lngamma(xp, y)
    lnrm, phim = _lngamma(xm, y)
    u = np.exp(q * np.log(2.0) + lnrp - lnrm) * np.exp(1j * (2 * y * (np.log(2.0) - lnkr) + phip + phim))
    if N % 2 == 0:
        u[-1] = u[-1].real                  # the Nyquist coefficient is made real
    return u


def fht(r, a, mu, q=0.0, lnkr=0.0, lowring=True):
    """Discrete Hankel transform on the log grid r (rows of `a`), FFTLog's `fhtq` form: returns k, b with
        b(k) = int a(r) (k r)^q J_mu(k r) k dr,     k_j r_(N-1-j) = k_c r_c = exp(lnkr)."""
    r = np.asarray(r, dtype=np.float64)
    a = np.atleast_2d(np.asarray(a, dtype=np.float64))
    N = r.size
    dln = np.log(r[-1] / r[0]) / (N - 1.0)
    if lowring:
        lnkr = good_lnkr(dln, mu, q, lnkr)
    u = u_coefficients(N, dln, mu, q, lnkr)
    b = np.fft.irfft(np.fft.rfft(a, axis=-1) * u, N, axis=-1)[:, ::-1]
    k = np.exp(lnkr) / r[::-1]
    return k, b


def fftlog_transform(rs, frs, dim, mu, power_law_index):
    """pyccl.pyutils._fftlog_transform(rs, frs, dim, mu, power_law_index) -> ks, fks   (see the module docstring)"""
    rs = np.asarray(rs, dtype=np.float64)
    f = np.atleast_2d(np.asarray(frs, dtype=np.float64))
    q = dim / 2.0 + power_law_index
    if dim == 3:
        k, b = fht(rs, f * rs ** (1.5 - q), mu + 0.5, q)
        out = b * (2 * np.pi * k) ** -1.5 * k ** -q
    elif dim == 2:
        k, b = fht(rs, f * rs ** (1.0 - q), mu, q)
        out = b * k ** -(1.0 + q) / (2 * np.pi)
    else:
        raise ValueError("dim must be 2 or 3")
    return k, (out[0] if np.ndim(frs) == 1 else out)


def _fft_grid(r, par):
    """Pixel.py:134-139 / :196-201"""
    r_min = np.min([np.min(r) * par['padding_lo_fftlog'], 1e-8])
    r_max = np.max([np.max(r) * par['padding_hi_fftlog'], 1e3])
    n = par['n_per_decade'] * np.int32(np.log10(r_max / r_min))
    return np.geomspace(r_min, r_max, n)


def convolved_real(profile_real, window_real, pixel_size, r, par=FFTLOG_DEFAULTS):
    """ConvolvedProfile.real, Pixel.py:106-157.  profile_real(r_fft) -> [N_M, n] (or [n]); window_real(k) -> [n]"""
    r = np.atleast_1d(np.asarray(r, dtype=np.float64))
    r_fft = _fft_grid(r, par)
    prof = profile_real(r_fft)
    k_out, Pk = fftlog_transform(r_fft, prof, 3, 0, par['plaw_fourier'])                       # :146
    r_out, prof = fftlog_transform(k_out, Pk * window_real(k_out), 3, 0, par['plaw_fourier'] + 1)   # :147
    r = np.clip(r, pixel_size / 5, None)                                                         # :153
    prof = interpolate.PchipInterpolator(np.log(r_out), prof, extrapolate=False, axis=-1)(np.log(r))
    return np.where(np.isnan(prof), 0, prof) * (2 * np.pi) ** 3                                  # :155


def convolved_projected(profile_projected, window_projected, pixel_size, r, is_harmonic, D_A=None, par=FFTLOG_DEFAULTS):
    """ConvolvedProfile.projected, Pixel.py:160-224 (D_A = ccl.comoving_angular_distance(cosmo, a) when harmonic)"""
    r = np.atleast_1d(np.asarray(r, dtype=np.float64))
    r_fft = _fft_grid(r, par)
    prof = profile_projected(r_fft)
    if is_harmonic:
        r_fft = r_fft / D_A                                                                      # :205
    k_out, Pk = fftlog_transform(r_fft, prof, 2, 0, par['plaw_fourier'] + 1)                    # :208
    r_out, prof = fftlog_transform(k_out, Pk * window_projected(k_out), 2, 0, par['plaw_fourier'] + 1)
    if is_harmonic:
        r_out = r_out * D_A
        r = np.clip(r, pixel_size / 5 * D_A, None)
    else:
        r = np.clip(r, pixel_size / 5, None)
    prof = interpolate.PchipInterpolator(np.log(r_out), prof, extrapolate=False, axis=-1)(np.log(r))
    return np.where(np.isnan(prof), 0, prof) * (2 * np.pi) ** 2
