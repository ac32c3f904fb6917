"""Pixel-window convolution: BaryonForge/utils/Pixel.py ConvolvedProfile / HealPixel / GridPixelApprox / NoPix.

The reference calls pyccl.pyutils._fftlog_transform, which is not available here (**parity with CCL unpinned**).  The oracle
(oracle/fftlog.py) restates the published FFTLog algorithm and is pinned here against (a) analytic Hankel pairs and (b)
scipy.fft.fht, an independent implementation of the same paper; the HIP path is then compared with the oracle (-m gpu).
The WRAPPER around the transform -- ConvolvedProfile.real / .projected, the windows -- is the reference's own code and is pinned by
tests/golden/pixel_s19.npz: the unmodified reference classes run under refshim with oracle/fftlog.py playing the CCL transform
(tests/golden/make_golden_pixel.py)."""
import os

import numpy as np
import pytest
from scipy import fft as sfft

from oracle import fftlog as F


def test_oracle_fht_equals_scipy_fht():
    r = np.geomspace(1e-4, 1e4, 801)          # odd and even lengths
    for n in (801, 800):
        rr = r[:n]
        dln = np.log(rr[1] / rr[0])
        a = np.exp(-rr ** 2 / 0.98) * rr
        for mu in (0.0, 0.5, 1.5):
            for q in (0.0, 0.3, -0.5):
                off = sfft.fhtoffset(dln, mu, bias=q)
                A = sfft.fht(a, dln, mu, offset=off, bias=q)            # scipy biases the arrays itself: A(k) = int a J k dr
                k, b = F.fht(rr, a * rr ** -q, mu, q)                   # FFTLog's fhtq form
                assert np.abs(k * rr[::-1] / np.exp(off) - 1).max() < 1e-12
                assert np.abs(A - b[0] * k ** -q).max() <= 1e-11 * np.abs(A).max()


@pytest.mark.parametrize('dim', [2, 3])
def test_oracle_transform_analytic_gaussian(dim):
    """f = exp(-r^2 / 2 s^2):  T3 = (2 pi)^-3 (2 pi s^2)^(3/2) exp(-k^2 s^2 / 2),  T2 = (2 pi)^-2 (2 pi s^2) exp(-k^2 s^2 / 2)"""
    r = np.geomspace(1e-5, 1e4, 900)
    s = 0.7
    f = np.exp(-r ** 2 / (2 * s ** 2))
    plaw = -1.5 if dim == 3 else -1.0           # CCL's defaults: the unbiased transform
    k, T = F.fftlog_transform(r, f, dim, 0, plaw)
    exact = (2 * np.pi * s ** 2) ** (dim / 2) * np.exp(-k ** 2 * s ** 2 / 2) / (2 * np.pi) ** dim
    m = (k > 1e-2) & (k < 6)
    assert np.abs(T - exact)[m].max() <= (2e-5 if dim == 3 else 5e-4) * exact.max()        # discretisation: 100 points per decade
    # and back: T[T[f]] = f / (2 pi)^dim
    r2, g = F.fftlog_transform(k, T, dim, 0, plaw)
    m2 = (r2 > 1e-2) & (r2 < 3)
    assert np.abs(r2 / r - 1).max() < 1e-12         # the same low-ringing k r both ways: the grid comes back
    assert np.abs(g * (2 * np.pi) ** dim - np.exp(-r2 ** 2 / (2 * s ** 2)))[m2].max() <= 1e-11      # discrete transforms invert exactly


def test_oracle_transform_analytic_yukawa():
    """3-D: f = exp(-m r) / r  ->  4 pi / (k^2 + m^2) / (2 pi)^3"""
    r = np.geomspace(1e-6, 1e5, 1100)
    mY = 0.8
    k, T = F.fftlog_transform(r, np.exp(-mY * r) / r, 3, 0, -1.5)
    exact = 4 * np.pi / (k ** 2 + mY ** 2) / (2 * np.pi) ** 3
    m = (k > 1e-2) & (k < 1e2)                 # (below 1e-2 the unbiased transform aliases the r^-1 cusp; a bias of -0.5 cures it)
    assert np.abs(T / exact - 1)[m].max() <= 1e-5
    k, T = F.fftlog_transform(r, np.exp(-mY * r) / r, 3, 0, -2.0)
    exact = 4 * np.pi / (k ** 2 + mY ** 2) / (2 * np.pi) ** 3
    m = (k > 1e-3) & (k < 1e2)
    assert np.abs(T / exact - 1)[m].max() <= 2e-5


def test_oracle_convolution_of_gaussians():
    """Gaussian profile (variance s^2) through a Gaussian window exp(-k^2 p^2 / 2): variances add.  This is the path of
    Pixel.py:146-155 / :208-222 with the reference's power-law indices; the reference clips r at pixel / 5 because the back
    transform rings below the pixel scale -- which is what this bias convention reproduces."""
    s, p = 0.5, 0.2
    r = np.geomspace(p / 5, 2.0, 60)
    for dim in (3, 2):
        if dim == 3:
            got = F.convolved_real(lambda x: np.exp(-x ** 2 / (2 * s ** 2)), lambda k: np.exp(-k ** 2 * p ** 2 / 2), p, r)
        else:
            got = F.convolved_projected(lambda x: np.exp(-x ** 2 / (2 * s ** 2)), lambda k: np.exp(-k ** 2 * p ** 2 / 2), p, r, False)
        st2 = s ** 2 + p ** 2
        exact = (s ** 2 / st2) ** (dim / 2) * np.exp(-r ** 2 / (2 * st2))
        assert np.abs(got - exact).max() <= 2e-3 * exact.max()


def test_pixel_windows_cpu():
    """GridPixelApprox / HealPixel / NoPix are plain numpy (Pixel.py:322-327, :349-366, :537-538)"""
    from baryonification_amd.utils.Pixel import GridPixelApprox, HealPixel, NoPix
    from scipy import special
    k = np.geomspace(1e-3, 1e3, 50)
    g = GridPixelApprox(0.3)
    R3, R2 = np.cbrt(0.3 ** 3 / (4 / 3 * np.pi)), np.sqrt(0.3 ** 2 / np.pi)
    assert np.allclose(g.real(k), 3 * special.spherical_jn(1, 2 * k * R3) / (2 * k * R3))
    assert np.allclose(g.projected(k), 3 * special.spherical_jn(1, 2 * k * R2) / (2 * k * R2))
    assert g.real(np.array([0.0]))[0] == 1 and not g.isHarmonic
    h = HealPixel(1024)
    assert h.isHarmonic and abs(np.degrees(h.size) * 60 - 3.435486411817406) < 1e-12       # hp.nside2resol(1024, arcmin=True)
    sig = h.size / np.sqrt(8 * np.log(2)) / np.sqrt(2)
    assert np.allclose(h.projected(k), np.exp(-k * (k + 1) / 2 * sig ** 2)) and np.all(h.real(k) == 0)
    assert np.all(NoPix().real(k) == 1) and np.all(NoPix().projected(k) == 1)


# ------------------------------------------------------------------------------------------ the reference's own wrapper (Pixel.py)
GOLDEN_PIXEL = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pixel_s19.npz')


def _ref_windows(name, size):
    """the window functions restated from Pixel.py:322-366 (top hat of equal volume / area, diameter passed to the beam), :537-538
    (Gaussian of FWHM = pixel / sqrt 2, k (k + 1)), :519-521 (no real-space HEALPix window), :576-582 (NoPix)"""
    from scipy import special

    def beam(k, R):
        kr = k * (2 * R)
        with np.errstate(invalid='ignore', divide='ignore'):
            return np.where(kr > 0, 3 * special.spherical_jn(1, kr) / kr, 1)
    if name.startswith('grid'):
        return (lambda k: beam(k, np.cbrt(size ** 3 / (4 / 3 * np.pi)))), (lambda k: beam(k, np.sqrt(size ** 2 / np.pi))), False
    if name.startswith('heal'):
        sig = size / np.sqrt(8 * np.log(2)) / np.sqrt(2)
        return (lambda k: np.zeros_like(k)), (lambda k: np.exp(-k * (1 + k) / 2 * sig ** 2)), True
    return (lambda k: np.ones_like(k)), (lambda k: np.ones_like(k)), False


def _pixel_cases():
    g = np.load(GOLDEN_PIXEL)
    return g, [str(c) for c in g['cases']]


def test_oracle_convolved_profile_equals_reference_pixel_py():
    """oracle.fftlog.convolved_real / convolved_projected == the reference's ConvolvedProfile fed the same profile rows"""
    g, cases = _pixel_cases()
    r = g['r']
    for key in cases:
        pname, xname, method = key.split('|')
        size = float(g['size|' + xname])
        w_real, w_proj, harmonic = _ref_windows(xname, size)
        rows, r_fft, exp = g[key + '|rows'], g[key + '|r_fft'], g[key + '|expected']

        def profile(x, rows=rows, r_fft=r_fft):
            assert x.shape == r_fft.shape and np.abs(x / r_fft - 1).max() < 1e-13          # the padding grid of Pixel.py:134-139
            return rows
        par = dict(F.FFTLOG_DEFAULTS, plaw_fourier=-2, padding_lo_fftlog=1e-2, padding_hi_fftlog=1e2, padding_lo_extra=1e-4,
                   padding_hi_extra=1e4)                                                     # Schneider19.py:124-128
        if method == 'real':
            got = F.convolved_real(profile, w_real, size, r, par)
        else:
            got = F.convolved_projected(profile, w_proj, size, r, harmonic, float(g['D_A_comoving']), par)
        assert got.shape == exp.shape == (3, r.size)
        assert np.abs(got - exp).max() <= 1e-12 * max(np.abs(exp).max(), 1e-300), key
    assert np.all(g['gas|heal256|real|expected'] == 0)                                       # Pixel.py:519-521


def test_product_windows_equal_reference_windows():
    from baryonification_amd.utils.Pixel import GridPixelApprox, HealPixel, NoPix
    g, _ = _pixel_cases()
    k = np.geomspace(1e-4, 1e5, 300)
    for name, px in (('grid0p5', GridPixelApprox(0.5)), ('grid0p08', GridPixelApprox(0.08)), ('heal256', HealPixel(256)),
                     ('heal2048', HealPixel(2048)), ('nopix', NoPix())):
        size = float(g['size|' + name])
        assert abs(px.size - size) <= 1e-15 * max(size, 1)
        w_real, w_proj, harmonic = _ref_windows(name, size)
        assert px.isHarmonic == harmonic
        assert np.abs(px.real(k) - w_real(k)).max() <= 1e-15 and np.abs(px.projected(k) - w_proj(k)).max() <= 1e-15


class _StoredProfile(object):
    """a profile object that returns the rows the reference's profile returned (golden), whatever the port of the profile does"""

    def __init__(self, g, key, par):
        self.g, self.key, self.precision_fftlog = g, key, par

    def _rows(self, r, scalar=False):
        r_fft = self.g[self.key + '|r_fft']
        assert r.shape == r_fft.shape and np.abs(r / r_fft - 1).max() < 1e-13
        return self.g[self.key + '|rows']

    def real(self, cosmo, r, M, a):
        return self._rows(r)

    def projected(self, cosmo, r, M, a):
        return self._rows(r)


@pytest.mark.gpu
def test_hip_convolved_profile_vs_reference_pixel_py(gpu):
    """the drop-in ConvolvedProfile (bfgx_fftlog_convolve on the GPU) == the reference's ConvolvedProfile on the same profile rows"""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    g, cases = _pixel_cases()
    r, M, a = g['r'], g['M'], float(g['a'])
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    par = dict(F.FFTLOG_DEFAULTS, plaw_fourier=-2, padding_lo_fftlog=1e-2, padding_hi_fftlog=1e2, padding_lo_extra=1e-4, padding_hi_extra=1e4)
    pixels = {'grid0p5': bfg.utils.GridPixelApprox(0.5), 'grid0p08': bfg.utils.GridPixelApprox(0.08), 'heal256': bfg.utils.HealPixel(256),
              'heal2048': bfg.utils.HealPixel(2048), 'nopix': bfg.utils.NoPix()}
    # the comoving angular distance the harmonic cases divide by: ours vs the refshim background the golden was made with
    from baryonification_amd.utils.Pixel import _comoving_angular_distance
    assert abs(_comoving_angular_distance(cosmo, a) / float(g['D_A_comoving']) - 1) < 1e-12
    for key in cases:
        pname, xname, method = key.split('|')
        conv = bfg.utils.ConvolvedProfile(_StoredProfile(g, key, par), pixels[xname])
        got = getattr(conv, method)(cosmo, r, M, a)
        exp = g[key + '|expected']
        assert got.shape == exp.shape
        # 1e-9 of the row's largest value: the two O(n^2) DFTs against numpy's FFT over 20+ decades of dynamic range
        for gr, er in zip(got, exp):
            assert np.abs(gr - er).max() <= 1e-9 * max(np.abs(er).max(), 1e-300), key
    # our port of the Gas profile through the same path: the whole f3 chain against the reference (port accuracy: 1e-6)
    keys, vals = [str(k) for k in g['par_keys']], g['par_vals']
    gas = bfg.Profiles.Gas(**dict(zip(keys, (float(v) for v in vals))))
    assert gas.precision_fftlog == par and bfg.Profiles.Pressure(gas=gas, darkmatterbaryon=gas).precision_fftlog == par      # Schneider19.py:124-128
    assert bfg.Profiles.Stars(**dict(zip(keys, (float(v) for v in vals)))).precision_fftlog['padding_hi_fftlog'] == 1e5          # :588
    rows = gas.projected(cosmo, g['gas|heal256|projected|r_fft'], M, a)
    ref_rows = g['gas|heal256|projected|rows']
    assert np.abs(rows / ref_rows - 1).max() <= 1e-6, np.abs(rows / ref_rows - 1).max()
    got = bfg.utils.ConvolvedProfile(gas, pixels['heal256']).projected(cosmo, r, M, a)
    exp = g['gas|heal256|projected|expected']
    assert np.abs(got - exp).max() <= 1e-6 * np.abs(exp).max()
    one = bfg.utils.ConvolvedProfile(gas, pixels['heal256']).projected(cosmo, r, 2e14, a)
    assert one.shape == g['gas|heal256|projected|scalarM'].shape and np.abs(one - g['gas|heal256|projected|scalarM']).max() <= 1e-6 * np.abs(one).max()


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('dim,plaw', [(3, -1.5), (3, -0.5), (2, -1.0), (2, -0.5)])
def test_hip_fftlog_transform_vs_oracle(gpu, dim, plaw):
    from baryonification_amd.utils.Pixel import fftlog_transform
    r = np.geomspace(1e-8, 3e3, 1100)
    rng = np.random.default_rng(3)
    rows = np.stack([np.exp(-(r / s) ** 1.3) / (1 + (r / 0.05) ** rng.uniform(0.5, 1.5)) for s in (0.3, 1.0, 7.0)])
    k, T = fftlog_transform(r, rows, dim, 0, plaw)
    ko, To = F.fftlog_transform(r, rows, dim, 0, plaw)
    assert np.abs(k / ko - 1).max() < 1e-12
    # pointwise over 20+ decades of dynamic range: compare where the transform is above the rounding floor of its row
    for a, b in zip(T, To):
        m = np.abs(b) > 1e-9 * np.abs(b).max()
        assert np.abs(a - b)[m].max() <= 1e-9 * np.abs(b).max()
    # single row, odd length
    k1, T1 = fftlog_transform(r[:1001], rows[0, :1001], dim, 0, plaw)
    _, T1o = F.fftlog_transform(r[:1001], rows[0, :1001], dim, 0, plaw)
    assert T1.shape == (1001,) and np.abs(T1 - T1o).max() <= 1e-9 * np.abs(T1o).max()


@pytest.mark.gpu
def test_hip_convolved_profile_vs_oracle(gpu):
    """ConvolvedProfile.real / .projected through the drop-in classes == the oracle's restatement of Pixel.py:106-224"""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    M = np.array([1e13, 1e14, 7e14])
    a = 1 / 1.25
    r = np.geomspace(1e-3, 3e2, 150)
    prof = bfg.Profiles.Gas(epsilon=4, theta_ej=4, theta_co=0.1, M_c=1e14, mu_beta=0.4, gamma=2.5, delta=7, eta=0.3, eta_delta=0.1,
                            tau=-1.5, tau_delta=0, A=0.055, M1=3e11, epsilon_h=0.015, a=0.3, n=2, p=0.3, q=0.707, cdelta=7,
                            proj_cutoff=50)
    # (with the reference's power-law indices the 3-D back transform is biased by q = +1, outside FFTLog's convergent strip: it
    # rings below the window scale -- the reason for the reference's clip at pixel / 5 -- so the comparison with the oracle, which
    # rings identically, is made where the result is meaningful and relative to the values there)
    for pixel in (bfg.utils.GridPixelApprox(0.5), bfg.utils.NoPix()):
        conv = bfg.utils.ConvolvedProfile(prof, pixel)
        m = r >= 0.1
        got = conv.real(cosmo, r, M, a)
        ora = F.convolved_real(lambda x: prof.real(cosmo, x, M, a), pixel.real, pixel.size, r, prof.precision_fftlog)
        assert got.shape == (3, r.size) and np.abs(got - ora)[:, m].max() <= 1e-8 * np.abs(ora[:, m]).max()
        got2 = conv.projected(cosmo, r, M, a)
        ora2 = F.convolved_projected(lambda x: prof.projected(cosmo, x, M, a), pixel.projected, pixel.size, r, False, None, prof.precision_fftlog)
        assert np.abs(got2 - ora2)[:, m].max() <= 1e-8 * np.abs(ora2[:, m]).max()
    # harmonic pixel: angles through D_A, clipping at pixel / 5 * D_A
    hpx = bfg.utils.HealPixel(256)
    conv = bfg.utils.ConvolvedProfile(prof, hpx)
    D_A = float(cosmo.angular_diameter_distance(a)) / a
    got = conv.projected(cosmo, r, M, a)
    ora = F.convolved_projected(lambda x: prof.projected(cosmo, x, M, a), hpx.projected, hpx.size, r, True, D_A, prof.precision_fftlog)
    assert np.abs(got - ora).max() <= 1e-8 * np.abs(ora).max()
    assert np.all(conv.real(cosmo, r, M, a) == 0)                      # no real-space HEALPix window (Pixel.py:519-521)
    # scalar-M call mirrors the rank, attribute access falls through to the wrapped profile
    assert conv.projected(cosmo, r, 1e14, a).shape == (r.size,) and conv.theta_ej == 4
    # with no window the convolved profile is the profile (above the clip)
    plain = prof.projected(cosmo, r, M, a)
    nop = bfg.utils.ConvolvedProfile(prof, bfg.utils.NoPix()).projected(cosmo, r, M, a)
    m = (r > 1e-2) & (r < 2)                  # (towards the truncation radius of the projection the steep profile aliases)
    assert np.abs(nop / plain - 1)[:, m].max() < 5e-2


@pytest.mark.gpu
def test_readme_quickstart_runs_end_to_end(gpu):
    """/root/reference README.md:60-94 with the package name swapped (N_samples_M accepted as the alias the README uses):
    Baryonification2D table -> BaryonifyShell, and Pressure convolved with HealPixel -> TabulatedProfile -> PaintProfilesShell"""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    nside = 64
    cat = syn.make_catalog(300, seed=11, z_lo=0.2, z_hi=0.3, logM_lo=13.0, logM_hi=14.8)
    cosmo_dict = dict(syn.COSMO)
    HealpixMap = syn.make_map(nside)
    Shell = bfg.utils.LightconeShell(map=HealpixMap, cosmo=cosmo_dict)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=cosmo_dict)
    cosmo = bfg.utils.Cosmology.from_dict(cosmo_dict)
    par = dict(epsilon=4, theta_ej=4, theta_co=0.1, M_c=1e14, mu_beta=0.4, gamma=2.5, delta=7, eta=0.3, eta_delta=0.1, tau=-1.5, tau_delta=0,
               A=0.055, M1=3e11, epsilon_h=0.015, a=0.3, n=2, p=0.3, q=0.707, cdelta=7, alpha_nt=0.2, nu_nt=0.5, gamma_nt=0.5)
    DMO = bfg.Profiles.DarkMatterOnly(proj_cutoff=100, **par)
    DMB = bfg.Profiles.DarkMatterBaryon(proj_cutoff=100, **par)
    model = bfg.Profiles.Baryonification2D(DMO, DMB, cosmo)
    model.setup_interpolator(z_min=Catalog.cat['z'].min(), z_max=Catalog.cat['z'].max(), N_samples_z=2,
                             M_min=Catalog.cat['M'].min(), M_max=Catalog.cat['M'].max(), N_samples_M=4,
                             R_min=1e-3, R_max=3e2, N_samples_R=100, verbose=False)
    PRESS = bfg.Profiles.Pressure(**dict(par, theta_ej=8, mu_theta_ej=0.1))
    Pixel = bfg.utils.HealPixel(NSIDE=nside)
    PRESS = bfg.utils.ConvolvedProfile(PRESS, Pixel)
    PRESS = bfg.utils.TabulatedProfile(PRESS, cosmo)
    PRESS.setup_interpolator(z_min=Catalog.cat['z'].min(), z_max=Catalog.cat['z'].max(), N_samples_z=2,
                             M_min=Catalog.cat['M'].min(), M_max=Catalog.cat['M'].max(), N_samples_Mass=4,
                             R_min=1e-3, R_max=3e2, N_samples_R=100, verbose=False)
    Runner = bfg.Runners.BaryonifyShell(Catalog, Shell, model=model, epsilon_max=20, verbose=False)
    new_map = Runner.process()
    assert new_map.shape == HealpixMap.shape and np.isclose(new_map.sum(), HealpixMap.sum()) and np.abs(new_map - HealpixMap).max() > 0
    Runner = bfg.Runners.PaintProfilesShell(Catalog, Shell, model=PRESS, epsilon_max=20, verbose=False)
    painted = Runner.process()
    assert painted.shape == HealpixMap.shape and np.isfinite(painted).all() and painted.min() >= 0 and painted.max() > 0
    # the pixel-convolved pressure map is smoother than the unconvolved one (which point-samples halos far smaller than these
    # NSIDE = 64 pixels, so the totals agree only roughly)
    P0 = bfg.utils.TabulatedProfile(bfg.Profiles.Pressure(**dict(par, theta_ej=8, mu_theta_ej=0.1)), cosmo)
    P0.setup_interpolator(z_min=Catalog.cat['z'].min(), z_max=Catalog.cat['z'].max(), N_samples_z=2,
                          M_min=Catalog.cat['M'].min(), M_max=Catalog.cat['M'].max(), N_samples_Mass=4,
                          R_min=1e-3, R_max=3e2, N_samples_R=100, verbose=False)
    plain = bfg.Runners.PaintProfilesShell(Catalog, Shell, model=P0, epsilon_max=20, verbose=False).process()
    assert 0.3 < painted.sum() / plain.sum() < 3 and painted.max() <= plain.max() * 1.001
