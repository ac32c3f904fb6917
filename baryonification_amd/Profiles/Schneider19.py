"""
One-halo Schneider+19 density profiles on the host (numpy/scipy): the integrands of the table-builder
kernels.  Interface-compatible with BaryonForge/Profiles/Schneider19.py for what the shell path needs --
same class names, same keyword parameters (`model_params`), `real(cosmo, r, M, a)`, `projected(...)`,
`set_parameter`, arithmetic between profiles -- so `Baryonification2D(DarkMatterOnly(**p), DarkMatterBaryon(**p),
cosmo).setup_interpolator(...)` works without BaryonForge/pyccl installed.  The numerical grids are part of
the definitions and are reproduced: DarkMatter :383-429, Stars :591-626, Gas :148-192 and
:687-742, CollisionlessMatter :959-1063, DarkMatterOnly :1128-1141, DarkMatterBaryon :1234-1267.

`projected()` runs the line-of-sight integral of `_projected_realspace` (:195-265) on the GPU
(tables.project_profile); there is no FFTLog path here.  TwoHalo needs the linear P(k), sigma(M) and the growth
factor from CCL: pass `xi_mm=callable(r, a)` and `bias=callable(M, a)` to include it, otherwise the one-halo
profiles are used alone.
"""
import operator
import warnings

import numpy as np
from scipy import interpolate

from .. import tables
from ..utils.cosmology import MassDef, RHO_CRITICAL, cosmo_to_dict

__all__ = ['model_params', 'SchneiderProfiles', 'DarkMatter', 'TwoHalo', 'Stars', 'Gas', 'CollisionlessMatter',
           'DarkMatterOnly', 'DarkMatterBaryon']

model_params = ['cdelta', 'epsilon', 'a', 'n', 'q', 'p', 'cutoff', 'proj_cutoff',
                'theta_ej', 'theta_co', 'M_c', 'gamma', 'delta',
                'mu_theta_ej', 'mu_theta_co', 'mu_beta', 'mu_gamma', 'mu_delta',
                'M_theta_ej', 'M_theta_co', 'M_gamma', 'M_delta',
                'nu_theta_ej', 'nu_theta_co', 'nu_M_c', 'nu_gamma', 'nu_delta',
                'zeta_theta_ej', 'zeta_theta_co', 'zeta_M_c', 'zeta_gamma', 'zeta_delta',
                'A', 'M1', 'eta', 'eta_delta', 'tau', 'tau_delta', 'epsilon_h',
                'alpha_nt', 'nu_nt', 'gamma_nt', 'mean_molecular_weight']

_R500 = np.geomspace(1e-6, 1000, 500)          # the normalisation grid of DarkMatter / Stars / Gas


def _set_parameter(obj, key, value):
    """recursive setattr through nested profiles (utils/Tabulate.py:12-65)"""
    for k in list(vars(obj)):
        v = getattr(obj, k)
        if k == key:
            setattr(obj, key, value)
        elif isinstance(v, SchneiderProfiles):
            _set_parameter(v, key, value)


def _kfac(r, cutoff):
    """exponential cutoff 1 / (1 + exp(2 (r - cutoff))), overflow-safe (e.g. :416-418)"""
    arg = r[None, :] - cutoff
    with np.errstate(over='ignore', invalid='ignore'):
        return 1.0 / (1.0 + np.exp(2.0 * np.where(arg > 30, np.inf, arg)))


def _squeeze(prof, r, M):
    if np.ndim(r) == 0:
        prof = np.squeeze(prof, axis=-1)
    if np.ndim(M) == 0:
        prof = np.squeeze(prof, axis=0)
    return prof


def _baryon_fraction(cosmo):
    d = cosmo_to_dict(cosmo)
    return d['Omega_b'] / d['Omega_m']


class SchneiderProfiles(object):
    model_param_names = model_params

    def __init__(self, mass_def=None, use_fftlog_projection=False, padding_lo_proj=0.1, padding_hi_proj=10,
                 n_per_decade_proj=10, xi_mm=None, bias=None, c_of_M=None, **kwargs):
        if use_fftlog_projection:
            raise NotImplementedError("FFTLog projection is not part of this build; the real-space projection is")
        for m in self.model_param_names:              # defaults as Schneider19.py:84-92
            if m in kwargs:
                setattr(self, m, kwargs[m])
            elif ('mu_' in m) or ('nu_' in m) or ('zeta_' in m):
                setattr(self, m, 0)
            elif 'M_' in m:
                setattr(self, m, 1e14)
            else:
                setattr(self, m, None)
        self.padding_lo_proj, self.padding_hi_proj, self.n_per_decade_proj = padding_lo_proj, padding_hi_proj, n_per_decade_proj
        self.mass_def = mass_def if mass_def is not None else MassDef(200, 'critical')
        self.xi_mm, self.bias = xi_mm, bias
        # c_of_M(cosmo, M, a) -> concentration: stands in for ccl's ConcentrationDiemer15 when cdelta is None (Schneider19.py:390-397: the reference's
        # default_config has cdelta = None).  A pyccl user passes  lambda cosmo, M, a: ccl.halos.ConcentrationDiemer15(mass_def=md)(cosmo, M, a)
        self.c_of_M = c_of_M
        self.cutoff = kwargs.get('cutoff', 1e3)
        self.proj_cutoff = kwargs.get('proj_cutoff', self.cutoff)
        self._projected = self._projected_realspace
        # ccl.halos.HaloProfile.precision_fftlog (pyccl 2.8.0 defaults) as SchneiderProfiles.__init__ overrides them
        # (Schneider19.py:124-128; Thermodynamic.py:87-91 sets the same): read by utils.Pixel.ConvolvedProfile (Pixel.py:72)
        self.precision_fftlog = {'padding_lo_fftlog': 0.1, 'padding_lo_extra': 0.1, 'padding_hi_fftlog': 10.0, 'padding_hi_extra': 10.0,
                                 'large_padding_2D': False, 'n_per_decade': 100, 'extrapol': 'linx_liny',
                                 'plaw_fourier': -1.5, 'plaw_projected': -1.0}
        self.update_precision_fftlog(plaw_fourier=-2)
        self.update_precision_fftlog(padding_lo_fftlog=1e-2, padding_hi_fftlog=1e2, padding_lo_extra=1e-4, padding_hi_extra=1e4)

    # -- protocol -------------------------------------------------------------------------------------
    @property
    def model_params(self):
        return {k: v for k, v in vars(self).items() if k in self.model_param_names}

    def set_parameter(self, key, value):
        _set_parameter(self, key, value)

    def update_precision_fftlog(self, **kwargs):
        """ccl.halos.HaloProfile.update_precision_fftlog"""
        for k, v in kwargs.items():
            if k not in self.precision_fftlog:
                raise KeyError("unknown FFTLog precision parameter %r" % (k,))
            self.precision_fftlog[k] = v

    def real(self, cosmo, r, M, a):
        return self._real(cosmo, r, M, a)

    def projected(self, cosmo, r, M, a):
        return self._projected(cosmo, r, M, a)

    def _R(self, cosmo, M, a):
        return np.atleast_1d(self.mass_def.get_radius(cosmo, np.atleast_1d(M), a)) / a      # comoving Mpc

    def _projected_realspace(self, cosmo, r, M, a):
        r_use = np.atleast_1d(r)
        l = tables.los_grid(r_use, self.padding_lo_proj, self.padding_hi_proj, self.n_per_decade_proj, self.proj_cutoff)
        rho = np.atleast_2d(self._real(cosmo, l, np.atleast_1d(M), a))
        proj = tables.project_profile(l, rho, r_use)                                        # GPU: :249-252
        if np.any(proj <= 0):
            warnings.warn("WARNING: Profile is zero/negative in some places."
                          "Likely a convolution artifact for objects smaller than the pixel scale")
        return _squeeze(proj, r, M)

    # -- arithmetic (utils/misc.py:7-127): the combination evaluates the ORIGINAL operands -------------
    def _combine(self, other, op, reflect=False):
        assert isinstance(other, (int, float, SchneiderProfiles)), \
            f"Object must be int/float/SchneiderProfile but is type '{type(other).__name__}'."
        out = self.__class__(**self.model_params, xi_mm=self.xi_mm, bias=self.bias, c_of_M=self.c_of_M, padding_lo_proj=self.padding_lo_proj,
                             padding_hi_proj=self.padding_hi_proj, n_per_decade_proj=self.n_per_decade_proj)
        me = self

        def _real(cosmo, r, M, a):
            A = me._real(cosmo, r, M, a)
            B = other._real(cosmo, r, M, a) if isinstance(other, SchneiderProfiles) else other
            return op(B, A) if reflect else op(A, B)

        out._real = _real
        out._operands = (me, other)
        return out

    def __add__(self, o): return self._combine(o, operator.add)
    def __sub__(self, o): return self._combine(o, operator.sub)
    def __mul__(self, o): return self._combine(o, operator.mul)
    def __truediv__(self, o): return self._combine(o, operator.truediv)
    def __pow__(self, o): return self._combine(o, operator.pow)
    def __radd__(self, o): return self._combine(o, operator.add, True)
    def __rsub__(self, o): return self._combine(o, operator.sub, True)
    def __rmul__(self, o): return self._combine(o, operator.mul, True)
    def __rtruediv__(self, o): return self._combine(o, operator.truediv, True)

    def __neg__(self):
        out = self._combine(0, operator.add)
        me = self
        out._real = lambda cosmo, r, M, a: -me._real(cosmo, r, M, a)
        return out

    def __str__(self):
        return self.__class__.__name__ + "(" + ", ".join(f"{m} = {getattr(self, m)}" for m in self.model_param_names) + ")"

    __repr__ = __str__

    # -- shared pieces --------------------------------------------------------------------------------
    def _concentration(self, cosmo, M, a):
        if self.cdelta is None:
            if self.c_of_M is None:
                raise NotImplementedError("cdelta=None needs the Diemer15 c(M) relation (CCL sigma(M)): pass cdelta, or c_of_M=callable(cosmo, M, a) "
                                          "(with pyccl: ccl.halos.ConcentrationDiemer15)")
            c = np.asarray(self.c_of_M(cosmo, np.atleast_1d(M), a), dtype=np.float64)
            return c * np.ones_like(M, dtype=np.float64)
        return self.cdelta * np.ones_like(M, dtype=np.float64)

    def _star_fraction(self, M, tau, eta):
        return 2 * self.A * ((M / self.M1) ** tau + (M / self.M1) ** eta) ** -1

    def _gas_params(self, M, z):
        c = 1 if self.cdelta is None else self.cdelta
        M_c = self.M_c * (1 + z) ** self.nu_M_c * c ** self.zeta_M_c
        x = (M / M_c) ** self.mu_beta
        beta = 3 * x / (1 + x)

        def scaled(name):
            return (getattr(self, name) * (M / getattr(self, 'M_' + name)) ** getattr(self, 'mu_' + name) *
                    (1 + z) ** getattr(self, 'nu_' + name) * c ** getattr(self, 'zeta_' + name))

        return [v[:, None] for v in (beta, scaled('theta_ej'), scaled('theta_co'), scaled('delta'), scaled('gamma'))]


def _dm_total_mass(par, cosmo, M, a, c_of_M=None):
    """trapz(4 pi r^2 rho_DM) on the 500-point grid with the DM cutoff lifted (e.g. :609-613)"""
    DM = DarkMatter(**par, c_of_M=c_of_M)
    DM.cutoff = 1e3
    rho = DM.real(cosmo, _R500, M, a)
    return np.atleast_1d(np.trapz(4 * np.pi * _R500 ** 2 * rho, _R500, axis=-1))[:, None]


class DarkMatter(SchneiderProfiles):
    """truncated NFW, normalised numerically to M within R (:383-429)"""

    def _shape(self, r, r_s, r_t):
        return 1 / (r / r_s * (1 + r / r_s) ** 2) * 1 / (1 + (r / r_t) ** 2) ** 2

    def _real(self, cosmo, r, M, a):
        r_use, M_use = np.atleast_1d(r), np.atleast_1d(M)
        R = self._R(cosmo, M_use, a)
        c = self._concentration(cosmo, M_use, a)
        r_s, r_t = (R / c)[:, None], (R * self.epsilon)[:, None]
        integrand = 4 * np.pi * _R500 ** 3 * self._shape(_R500, r_s, r_t)
        norm = np.array([interpolate.PchipInterpolator(np.log(_R500), f).antiderivative(nu=1)(np.log(Ri))
                         for f, Ri in zip(integrand, R)])
        rho_c = (M_use / norm)[:, None]
        return _squeeze(rho_c * self._shape(r_use, r_s, r_t) * _kfac(r_use, self.cutoff), r, M)


class Stars(SchneiderProfiles):
    """central galaxy: Gaussian-truncated r^-2 (:591-626)"""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.update_precision_fftlog(padding_lo_fftlog=1e-5, padding_hi_fftlog=1e5)      # Schneider19.py:581-588

    def _real(self, cosmo, r, M, a):
        r_use, M_use = np.atleast_1d(r), np.atleast_1d(M)
        R = self._R(cosmo, M_use, a)
        f_cga = self._star_fraction(M_use, self.tau + self.tau_delta, self.eta + self.eta_delta)[:, None]
        R_h = (self.epsilon_h * R)[:, None]
        M_tot = _dm_total_mass(self.model_params, cosmo, M_use, a, self.c_of_M)
        prof = f_cga * M_tot / (4 * np.pi ** (3 / 2) * R_h) * 1 / r_use ** 2 * np.exp(-(r_use / 2 / R_h) ** 2)
        return _squeeze(prof * _kfac(r_use, self.cutoff), r, M)


class Gas(SchneiderProfiles):
    """cored double power law (:687-742)"""

    def _shape(self, r, R_co, R_ej, beta, delta, gamma):
        return 1 / (1 + r / R_co) ** beta / (1 + (r / R_ej) ** gamma) ** ((delta - beta) / gamma)

    def _real(self, cosmo, r, M, a):
        r_use, M_use = np.atleast_1d(r), np.atleast_1d(M)
        R = self._R(cosmo, M_use, a)
        f_gas = (_baryon_fraction(cosmo) - self._star_fraction(M_use, self.tau, self.eta))[:, None]
        beta, theta_ej, theta_co, delta, gamma = self._gas_params(M_use, 1 / a - 1)
        R_co, R_ej = theta_co * R[:, None], theta_ej * R[:, None]
        norm = np.trapz(4 * np.pi * _R500 ** 2 * self._shape(_R500, R_co, R_ej, beta, delta, gamma), _R500, axis=-1)[:, None]
        M_tot = _dm_total_mass(self.model_params, cosmo, M_use, a, self.c_of_M)
        prof = self._shape(r_use, R_co, R_ej, beta, delta, gamma) * _kfac(r_use, self.cutoff)
        return _squeeze(prof * (f_gas * M_tot / norm), r, M)


class CollisionlessMatter(SchneiderProfiles):
    """dark matter + satellites after adiabatic relaxation against gas and stars (:933-1063)"""

    def __init__(self, gas=None, stars=None, darkmatter=None, max_iter=10, reltol=1e-2, r_min_int=1e-8, r_max_int=1e5,
                 r_steps=5000, **kwargs):
        self.Gas = gas if gas is not None else Gas(**kwargs)
        self.Stars = stars if stars is not None else Stars(**kwargs)
        self.DarkMatter = darkmatter if darkmatter is not None else DarkMatter(**kwargs)
        for p in (self.Gas, self.Stars, self.DarkMatter):
            p.set_parameter('cutoff', 1000)
        self.max_iter, self.reltol = max_iter, reltol
        self.r_min_int, self.r_max_int, self.r_steps = r_min_int, r_max_int, r_steps
        super().__init__(**kwargs)

    def _real(self, cosmo, r, M, a):
        r_use, M_use = np.atleast_1d(r), np.atleast_1d(M)
        if np.min(r) < self.r_min_int:
            warnings.warn(f"Decrease integral lower limit, r_min_int ({self.r_min_int}) < minimum radius ({np.min(r)})", UserWarning)
        if np.max(r) > self.r_max_int:
            warnings.warn(f"Increase integral upper limit, r_max_int ({self.r_max_int}) < maximum radius ({np.max(r)})", UserWarning)
        rg = np.geomspace(self.r_min_int, self.r_max_int, self.r_steps)
        lnr = np.log(rg)
        safe = (rg > 2 * rg.min()) & (rg < rg.max() / 2)
        f_star = self._star_fraction(M_use, self.tau, self.eta)[:, None]
        f_cga = self._star_fraction(M_use, self.tau + self.tau_delta, self.eta + self.eta_delta)[:, None]
        f_clm = 1 - _baryon_fraction(cosmo) + (f_star - f_cga)
        dlnr = lnr[1] - lnr[0]

        def enclosed(prof):
            return 4 * np.pi * np.cumsum(rg ** 3 * np.atleast_2d(prof.real(cosmo, rg, M_use, a)) * dlnr, axis=-1)

        M_i, M_cga, M_gas = enclosed(self.DarkMatter), enclosed(self.Stars), enclosed(self.Gas)
        out_lnM = np.empty_like(M_i)
        with np.errstate(over='ignore', divide='ignore', invalid='ignore'):
            for m in range(M_i.shape[0]):
                lnM_nfw = interpolate.PchipInterpolator(lnr, np.log(M_i[m]), extrapolate=True)
                lnM_cga = interpolate.PchipInterpolator(lnr, np.log(M_cga[m]), extrapolate=True)
                lnM_gas = interpolate.PchipInterpolator(lnr, np.log(M_gas[m]), extrapolate=True)
                zeta = np.ones_like(rg)
                for it in range(1, self.max_iter + 1):                     # fixed point of :1016-1044
                    ln_rf = np.log(rg * zeta)
                    M_f = f_clm[m] * M_i[m] + np.exp(lnM_cga(ln_rf)) + np.exp(lnM_gas(ln_rf))
                    zeta_new = self.a * ((M_i[m] / M_f) ** self.n - 1) + 1
                    rel = np.max(np.abs(zeta_new / zeta - 1)[safe])
                    zeta = zeta_new
                    if not (rel > self.reltol):
                        break
                    if it == self.max_iter:
                        warnings.warn("Profile of halo index %d did not converge after %d tries." % (m, it) +
                                      "Max_diff = %0.5f, Median_diff = %0.5f. Try increasing max_iter." % (rel, rel), UserWarning)
                out_lnM[m] = np.log(f_clm[m]) + lnM_nfw(np.log(rg / zeta))
            spline = interpolate.CubicSpline(lnr, out_lnM, axis=-1, extrapolate=False)
            lnq = np.log(r_use)
            prof = spline.derivative(nu=1)(lnq) * np.exp(spline(lnq)) / r_use / (4 * np.pi * r_use ** 2)
        prof = np.where(np.isnan(prof), 0, prof) * _kfac(r_use, self.cutoff)
        return _squeeze(prof, r, M)


class TwoHalo(SchneiderProfiles):
    """(1 + b(M) xi_mm(r)) rho_m (:432-521).  Needs user-supplied `xi_mm(r, a)` and `bias(M, a)` callables
    (the reference takes them from CCL's linear power spectrum, which is outside this build)."""

    def _real(self, cosmo, r, M, a):
        if self.xi_mm is None or self.bias is None:
            raise NotImplementedError("TwoHalo needs xi_mm(r, a) and bias(M, a) callables (CCL P(k)/sigma(M) are not available here)")
        r_use, M_use = np.atleast_1d(r), np.atleast_1d(M)
        d = cosmo_to_dict(cosmo)
        rho_m = RHO_CRITICAL * d['h'] ** 2 * d['Omega_m']                    # comoving matter density
        prof = (1 + np.atleast_1d(self.bias(M_use, a))[:, None] * np.atleast_1d(self.xi_mm(r_use, a))[None, :]) * rho_m
        return _squeeze(prof * _kfac(r_use, self.cutoff), r, M)


class _WithOptionalTwoHalo(SchneiderProfiles):
    def _two_halo(self, cosmo, r, M, a):
        if self.xi_mm is None or self.bias is None:
            return 0.0                                   # one-halo only
        return self.TwoHalo.real(cosmo, r, M, a)


class DarkMatterOnly(_WithOptionalTwoHalo):
    """DarkMatter + TwoHalo (:1128-1141)"""

    def __init__(self, darkmatter=None, twohalo=None, **kwargs):
        self.DarkMatter = darkmatter if darkmatter is not None else DarkMatter(**kwargs)
        self.TwoHalo = twohalo if twohalo is not None else TwoHalo(**kwargs)
        super().__init__(**kwargs)

    def _real(self, cosmo, r, M, a):
        return self.DarkMatter.real(cosmo, r, M, a) + self._two_halo(cosmo, r, M, a)


class DarkMatterBaryon(_WithOptionalTwoHalo):
    """(CLM + stars + gas) renormalised to the DM-only mass, + TwoHalo (:1234-1267)"""

    def __init__(self, gas=None, stars=None, collisionlessmatter=None, darkmatter=None, twohalo=None, **kwargs):
        self.Gas = gas if gas is not None else Gas(**kwargs)
        self.Stars = stars if stars is not None else Stars(**kwargs)
        self.DarkMatter = darkmatter if darkmatter is not None else DarkMatter(**kwargs)
        self.TwoHalo = twohalo if twohalo is not None else TwoHalo(**kwargs)
        # its own sub-profiles: CollisionlessMatter lifts their cutoff to 1000 (:945-947), ours keep the user's
        self.CollisionlessMatter = collisionlessmatter if collisionlessmatter is not None else CollisionlessMatter(**kwargs)
        super().__init__(**kwargs)

    def _baryonic(self, cosmo, r, M, a):
        return (self.CollisionlessMatter.real(cosmo, r, M, a) + self.Stars.real(cosmo, r, M, a) +
                self.Gas.real(cosmo, r, M, a))

    def _real(self, cosmo, r, M, a):
        rg = np.geomspace(1e-5, 100, 500)
        M_tot = np.trapz(4 * np.pi * rg ** 2 * self.DarkMatter.real(cosmo, rg, M, a), rg)
        M_dmb = np.trapz(4 * np.pi * rg ** 2 * self._baryonic(cosmo, rg, M, a), rg, axis=-1)
        factor = M_tot / M_dmb
        if np.ndim(factor) == 1:
            factor = factor[:, None]
        return self._baryonic(cosmo, r, M, a) * factor + self._two_halo(cosmo, r, M, a)
