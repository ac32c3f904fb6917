"""
Host-side stand-ins for the few pyccl objects the runner/model protocol carries around
(`ccl.Cosmology`, `ccl.halos.massdef.MassDef`), backed by the background functions of libbfgx
(bfgx_cosmo_* in include/bfgx.h).  They let the package be used where pyccl is not installed;
real pyccl objects are accepted everywhere too (duck-typed through `cosmo_to_dict` /
`massdef_to_tuple`).

Reference call sites mirrored: HealpixRunner.py:268-280, :296; BaryonCorrection.py:370.
"""
import ctypes as C

import numpy as np

from .. import _lib

__all__ = ['Cosmology', 'MassDef', 'cosmo_to_dict', 'massdef_to_tuple']

# pyccl.physical_constants (CCL 2.x) used on the host side
GNEWT, SOLAR_MASS, MPC_TO_METER = 6.67408e-11, 1.9884754153381438e30, 3.085677581491367399198952281e22
RHO_CRITICAL = ((3 * 100 * 100) / (8 * np.pi * GNEWT)) * (1000 * 1000 * MPC_TO_METER / SOLAR_MASS)   # h^2 Msun / Mpc^3


def cosmo_to_dict(cosmo):
    """6-key cosmology dict from: a dict, our Cosmology, or a pyccl Cosmology (cosmo['Omega_m'] API)."""
    if isinstance(cosmo, dict):
        d = dict(cosmo)
    elif isinstance(cosmo, Cosmology):
        d = dict(cosmo.params)
    else:  # pyccl.Cosmology supports item access for its parameters
        d = {}
        for k in ('Omega_b', 'h', 'sigma8', 'n_s', 'w0', 'T_CMB', 'Neff'):
            try:
                d[k] = float(cosmo[k])
            except Exception:
                pass
        try:
            d['Omega_m'] = float(cosmo['Omega_m'])
        except Exception:
            d['Omega_m'] = float(cosmo['Omega_c']) + float(cosmo['Omega_b'])
    for k in ('Omega_m', 'Omega_b', 'h'):
        if k not in d:
            raise ValueError("cosmology is missing %r" % k)
    return d


def massdef_to_tuple(md):
    """(Delta, rho_type) from our MassDef or a pyccl MassDef."""
    if md is None:
        return 200.0, 'critical'
    return md.Delta, md.rho_type


class Cosmology(object):
    """Flat wCDM background with pyccl-2.x radiation defaults.  Signature follows ccl.Cosmology for the
    keywords the reference passes (HealpixRunner.py:268-272)."""

    def __init__(self, Omega_c=None, Omega_b=None, h=None, n_s=None, sigma8=None, w0=-1.0,
                 Omega_m=None, T_CMB=None, Neff=None, matter_power_spectrum=None, **ignored):
        if Omega_m is None:
            Omega_m = Omega_c + Omega_b
        self.params = {'Omega_m': float(Omega_m), 'Omega_b': float(Omega_b), 'h': float(h),
                       'sigma8': sigma8, 'n_s': n_s, 'w0': float(w0)}
        if T_CMB is not None:
            self.params['T_CMB'] = float(T_CMB)
        if Neff is not None:
            self.params['Neff'] = float(Neff)

    @classmethod
    def from_dict(cls, d):
        d = cosmo_to_dict(d)
        return cls(Omega_m=d['Omega_m'], Omega_b=d['Omega_b'], h=d['h'], n_s=d.get('n_s'),
                   sigma8=d.get('sigma8'), w0=d.get('w0', -1.0), T_CMB=d.get('T_CMB'), Neff=d.get('Neff'))

    def __getitem__(self, k):
        if k == 'Omega_c':
            return self.params['Omega_m'] - self.params['Omega_b']
        return self.params[k]

    def compute_sigma(self):
        return None

    def _c(self):
        return _lib.make_cosmo(self.params)

    def E2(self, a):
        a = _lib.f8(np.atleast_1d(a))
        out = np.empty_like(a)
        _lib.check(_lib.load().bfgx_cosmo_E2(C.byref(self._c()), a.size, a.ctypes.data, out.ctypes.data))
        return out

    def angular_diameter_distance(self, a):
        """physical Mpc, like ccl.angular_diameter_distance(cosmo, a)"""
        aa = _lib.f8(np.atleast_1d(a))
        z = 1.0 / aa - 1.0
        out = np.empty_like(z)
        _lib.check(_lib.load().bfgx_cosmo_angular_diameter_distance(C.byref(self._c()), z.size, z.ctypes.data,
                                                                    out.ctypes.data))
        return out if np.ndim(a) else out[0]

    def Da_of_z(self, z):
        """the runner's CubicSpline D_a(z) (HealpixRunner.py:279-280, :297)"""
        zz = _lib.f8(np.atleast_1d(z))
        out = np.empty_like(zz)
        _lib.check(_lib.load().bfgx_cosmo_da_eval(C.byref(self._c()), zz.size, zz.ctypes.data, out.ctypes.data))
        return out if np.ndim(z) else out[0]


class MassDef(object):
    """ccl.halos.massdef.MassDef(Delta, rho_type) for numeric Delta."""

    def __init__(self, Delta=200, rho_type='critical', c_m_relation=None):
        self.Delta, self.rho_type = Delta, rho_type
        _lib.make_massdef(Delta, rho_type)   # validates

    def get_radius(self, cosmo, M, a):
        """physical Mpc"""
        M = _lib.f8(np.atleast_1d(M))
        aa = _lib.f8(np.broadcast_to(np.asarray(a, dtype=np.float64), M.shape))
        out = np.empty_like(M)
        c = _lib.make_cosmo(cosmo_to_dict(cosmo))
        md = _lib.make_massdef(self.Delta, self.rho_type)
        _lib.check(_lib.load().bfgx_cosmo_radius(C.byref(c), C.byref(md), M.size, M.ctypes.data, aa.ctypes.data,
                                                 out.ctypes.data))
        return out

    def __eq__(self, other):
        return (getattr(other, 'Delta', None), getattr(other, 'rho_type', None)) == (self.Delta, self.rho_type)

    def __hash__(self):
        return hash((self.Delta, self.rho_type))

    def __repr__(self):
        return "MassDef(%r, %r)" % (self.Delta, self.rho_type)
