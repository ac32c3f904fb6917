"""
install() puts stand-ins for `pyccl`, `healpy` and `numba` into sys.modules so that the UNMODIFIED
reference package can be imported from /root/reference inside the build container (none of the
three is installable offline: ordinary ModuleNotFoundError).

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py (and by the optional
reference-vs-oracle tests that skip when /root/reference is absent, e.g. on the GPU box).
The stand-ins are thin: geometry comes from refshim/healpy.py, background cosmology from
oracle.Background (flat wCDM + radiation with pyccl-2.x defaults).  Everything the reference
itself computes (runner loops, table read-out, RegularGridInterpolator use, regrid) runs as
shipped.  Parity of healpy/pyccl themselves remains UNPINNED.
"""
import os
import sys
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = '/root/reference'


def _make_pyccl():
    from oracle import oracle as O

    ccl = types.ModuleType('pyccl')
    ccl.__version__ = '2.8.0-refshim'

    pc = types.ModuleType('pyccl.physical_constants')
    for k in ('CLIGHT', 'GNEWT', 'SOLAR_MASS', 'MPC_TO_METER', 'STBOLTZ', 'KBOLTZ', 'EV_IN_J', 'RHO_CRITICAL'):
        setattr(pc, k, getattr(O, k))
    pc.T_CMB = O.T_CMB_DEFAULT
    ccl.physical_constants = pc

    class Cosmology(object):
        def __init__(self, Omega_c=None, Omega_b=None, h=None, n_s=None, sigma8=None, A_s=None,
                     Omega_k=0.0, Omega_g=None, Neff=O.NEFF_DEFAULT, m_nu=0.0, w0=-1.0, wa=0.0,
                     T_CMB=O.T_CMB_DEFAULT, matter_power_spectrum='halofit', transfer_function=None, **kw):
            assert wa == 0.0 and Omega_k == 0.0 and m_nu == 0.0, "refshim: flat wCDM only"
            self._bg = O.Background(Omega_c + Omega_b, Omega_b, h, sigma8, n_s, w0, T_CMB=T_CMB, Neff=Neff)
            p = types.SimpleNamespace(Omega_m=Omega_c + Omega_b, Omega_c=Omega_c, Omega_b=Omega_b, h=h,
                                      n_s=n_s, sigma8=sigma8, w0=w0, wa=wa)
            self.cosmo = types.SimpleNamespace(params=p)
            self._params = p
            self._pk_lin, self._pk_nl = {}, {}

        def __getitem__(self, k):
            return getattr(self._params, k)

        def compute_sigma(self):
            return None

    ccl.Cosmology = Cosmology
    core = types.ModuleType('pyccl.core')
    core.Cosmology = Cosmology
    ccl.core = core

    def angular_diameter_distance(cosmo, a1, a2=None):
        assert a2 is None
        a = np.atleast_1d(np.asarray(a1, dtype=np.float64))
        z = 1.0 / a - 1.0
        order = np.argsort(z)
        out = np.empty_like(z)
        out[order] = cosmo._bg.comoving_distance_z(z[order]) * a[order]
        return out if np.ndim(a1) else out[0]

    def comoving_angular_distance(cosmo, a):
        aa = np.atleast_1d(np.asarray(a, dtype=np.float64))
        z = 1.0 / aa - 1.0
        order = np.argsort(z)
        out = np.empty_like(z)
        out[order] = cosmo._bg.comoving_distance_z(z[order])
        return out if np.ndim(a) else out[0]

    def rho_x(cosmo, a, species, is_comoving=False):
        bg = cosmo._bg
        a = np.asarray(a, dtype=np.float64)
        com = a ** 3 if is_comoving else 1.0
        rc0 = O.RHO_CRITICAL * bg.h ** 2
        if species == 'critical':
            return rc0 * bg.E2(a) * com
        if species == 'matter':
            return rc0 * bg.Omega_m / a ** 3 * com
        raise NotImplementedError(species)

    def _unavailable(name):
        def f(*a, **k):
            raise NotImplementedError("refshim.pyccl has no %s (needs CCL's Boltzmann/sigma(M) machinery)" % name)
        return f

    ccl.angular_diameter_distance = angular_diameter_distance
    ccl.comoving_angular_distance = comoving_angular_distance
    ccl.rho_x = rho_x
    for nm in ('sigmaM', 'growth_factor', 'correlation_3d', 'correlation_3dRsd', 'linear_matter_power'):
        setattr(ccl, nm, _unavailable(nm))
    bgm = types.ModuleType('pyccl.background')
    bgm.h_over_h0 = lambda cosmo, a: np.sqrt(cosmo._bg.E2(a))
    ccl.background = bgm
    ccl.h_over_h0 = bgm.h_over_h0

    pyutils = types.ModuleType('pyccl.pyutils')
    # CCL's C FFTLog is absent: the published algorithm restated in oracle/fftlog.py stands in for it (parity with the CCL binary
    # stays unpinned), so that the reference's OWN wrapper around it (utils/Pixel.py ConvolvedProfile: padding grid, windows,
    # clip at pixel / 5, PCHIP read-back) runs unmodified -- tests/golden/make_golden_pixel.py
    from oracle import fftlog as _F
    pyutils._fftlog_transform = _F.fftlog_transform
    ccl.pyutils = pyutils

    # ---- halos
    halos = types.ModuleType('pyccl.halos')
    massdef = types.ModuleType('pyccl.halos.massdef')
    profiles = types.ModuleType('pyccl.halos.profiles')
    conc = types.ModuleType('pyccl.halos.concentration')

    class MassDef(object):
        def __init__(self, Delta, rho_type, c_m_relation=None):
            self.Delta, self.rho_type, self.c_m_relation = Delta, rho_type, c_m_relation

        def get_Delta(self, cosmo, a):
            assert self.Delta not in ('vir', 'fof'), "refshim: numeric Delta only"
            return self.Delta

        def get_radius(self, cosmo, M, a):
            Delta = self.get_Delta(cosmo, a)
            rho = rho_x(cosmo, a, self.rho_type, is_comoving=False)
            return (np.asarray(M, dtype=np.float64) / (4.18879020479 * Delta * rho)) ** (1.0 / 3.0)

        def get_mass(self, cosmo, R, a):
            Delta = self.get_Delta(cosmo, a)
            return 4.18879020479 * rho_x(cosmo, a, self.rho_type, is_comoving=False) * Delta * np.asarray(R) ** 3

        def __eq__(self, other):
            return isinstance(other, MassDef) and (self.Delta, self.rho_type) == (other.Delta, other.rho_type)

        def __hash__(self):
            return hash((self.Delta, self.rho_type))

    massdef.MassDef = MassDef
    halos.MassDef = MassDef

    class HaloProfile(object):
        def __init__(self, *, mass_def=None, concentration=None, is_number_counts=False):
            if mass_def is not None or not hasattr(self, 'mass_def'):
                self.mass_def = mass_def
            self.concentration = concentration
            self.precision_fftlog = {'padding_lo_fftlog': 0.1, 'padding_lo_extra': 0.1,
                                     'padding_hi_fftlog': 10., 'padding_hi_extra': 10.,
                                     'large_padding_2D': False, 'n_per_decade': 100,
                                     'extrapol': 'linx_liny', 'plaw_fourier': -1.5, 'plaw_projected': -1.}

        def update_precision_fftlog(self, **kwargs):
            self.precision_fftlog.update(kwargs)

        def real(self, cosmo, r, M, a, mass_def=None):
            return self._real(cosmo, r, M, a)

        def projected(self, cosmo, r_t, M, a, mass_def=None):
            if getattr(self, '_projected', None):
                return self._projected(cosmo, r_t, M, a)
            raise NotImplementedError("refshim.pyccl: FFTLog projection unavailable")

        def fourier(self, *a, **k):
            raise NotImplementedError("refshim.pyccl: FFTLog unavailable")

    profiles.HaloProfile = HaloProfile
    halos.HaloProfile = HaloProfile

    class ConcentrationConstant(object):
        def __init__(self, c=1, mass_def=None, *, mdef=None):
            self.c, self.mass_def = c, mass_def if mass_def is not None else mdef

        def get_concentration(self, cosmo, M, a):
            return self.c * np.ones_like(np.asarray(M, dtype=np.float64))

        __call__ = get_concentration

    class _NoConcentration(object):
        def __init__(self, *a, **k):
            pass

        def get_concentration(self, cosmo, M, a):
            raise NotImplementedError("refshim.pyccl: mass-concentration relations need sigma(M)")

        __call__ = get_concentration

    conc.ConcentrationConstant = ConcentrationConstant
    conc.ConcentrationDiemer15 = _NoConcentration
    conc.ConcentrationDuffy08 = _NoConcentration
    halos.ConcentrationConstant = ConcentrationConstant
    halos.massdef, halos.profiles, halos.concentration = massdef, profiles, conc
    halos.mass_translator = _unavailable('mass_translator')
    ccl.halos = halos

    mods = {'pyccl': ccl, 'pyccl.core': core, 'pyccl.physical_constants': pc, 'pyccl.background': bgm,
            'pyccl.pyutils': pyutils, 'pyccl.halos': halos, 'pyccl.halos.massdef': massdef,
            'pyccl.halos.profiles': profiles, 'pyccl.halos.concentration': conc}
    return mods


def _make_numba():
    nb = types.ModuleType('numba')

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    nb.njit = njit
    nb.jit = njit
    return nb


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'BaryonForge'))


def install():
    """Register the stand-ins and make `import BaryonForge` resolve to /root/reference."""
    repo = os.path.dirname(os.path.dirname(_HERE))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    import importlib.util
    for real in ('pyccl', 'healpy', 'numba'):
        if real in sys.modules:
            continue
        if real == 'pyccl':
            sys.modules.update(_make_pyccl())
        elif real == 'numba':
            sys.modules['numba'] = _make_numba()
        else:
            spec = importlib.util.spec_from_file_location('healpy', os.path.join(_HERE, 'healpy.py'))
            mod = importlib.util.module_from_spec(spec)
            sys.modules['healpy'] = mod
            spec.loader.exec_module(mod)
    if reference_available() and REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
    os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
    sys.dont_write_bytecode = True            # never write into the read-only reference tree
