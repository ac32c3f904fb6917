"""
Gas pressure from hydrostatic equilibrium: interface-compatible with `Pressure` of
BaryonForge/Profiles/Thermodynamic.py:42-278, plus the thin scalings built on it that the quickstart's tSZ flow uses:
`ElectronPressure` (:431-457), `NonThermalFrac` (:282-368) and `ThermalSZ` (:662-775).  `real()` evaluates the total and gas densities on the
hard-coded 500-point grid on the host and runs the two radial integrals + log-log PCHIP read-out of
Thermodynamic.py:240-271 on the GPU (tables.pressure_profile); `projected()` is the GPU line-of-sight kernel.
"""
import numpy as np

from .. import tables
from .Schneider19 import DarkMatterBaryon, Gas, SchneiderProfiles, _squeeze

__all__ = ['Pressure', 'ElectronPressure', 'NonThermalFrac', 'ThermalSZ']

# constants of Thermodynamic.py:10-38 (CCL's physical constants, CODATA 2014)
MPC_TO_METER = 3.085677581491367399198952281e22
SIGMA_T_CGS = 6.652458e-29 * 1e2 ** 2          # m^2 -> cm^2
M_E_CGS = 9.10938e-31 * 1e3                    # kg -> g
C_CGS = 2.99792458e8 * 1e2                     # m/s -> cm/s
Y_HELIUM = 0.24
PTH_TO_PE = (4 - 2 * Y_HELIUM) / (8 - 5 * Y_HELIUM)


class Pressure(SchneiderProfiles):

    def __init__(self, gas=None, darkmatterbaryon=None, **kwargs):
        self.Gas = gas if gas is not None else Gas(**kwargs)
        # the reference default is DarkMatterBaryon - TwoHalo (:163); our DarkMatterBaryon without the
        # xi_mm/bias callables already is the one-halo total
        self.DarkMatterBaryon = darkmatterbaryon if darkmatterbaryon is not None else DarkMatterBaryon(**kwargs)
        self.Gas.set_parameter('cutoff', 1000)
        self.DarkMatterBaryon.set_parameter('cutoff', 1000)
        super().__init__(**kwargs)

    def _real(self, cosmo, r, M, a):
        r_use, M_use = np.atleast_1d(np.asarray(r, dtype=np.float64)), np.atleast_1d(M)
        r500 = np.geomspace(1e-6, 1000, 500)
        rho_total = np.atleast_2d(self.DarkMatterBaryon.real(cosmo, r500, M_use, a))
        rho_gas = np.atleast_2d(self.Gas.real(cosmo, r500, M_use, a))
        prof = tables.pressure_profile(rho_total, rho_gas, r_use, cutoff=self.cutoff)      # cgs
        return _squeeze(prof, r, M)


class ElectronPressure(Pressure):
    """P_e = P_th (4 - 2Y) / (8 - 5Y), Y = 0.24 (Thermodynamic.py:453-457)"""

    def _real(self, cosmo, r, M, a):
        return PTH_TO_PE * super()._real(cosmo, r, M, a)


class NonThermalFrac(SchneiderProfiles):
    """Non-thermal pressure fraction f_nt = alpha f(z) (r / R200c)^gamma clipped to [0, 1] (Thermodynamic.py:338-368);
    used as `Pressure(...) * (1 - NonThermalFrac(...))`."""

    def __init__(self, alpha_nt=None, nu_nt=None, gamma_nt=None, **kwargs):
        super().__init__(alpha_nt=alpha_nt, nu_nt=nu_nt, gamma_nt=gamma_nt, **kwargs)
        self.alpha_nt, self.nu_nt, self.gamma_nt = alpha_nt, nu_nt, gamma_nt

    def _real(self, cosmo, r, M, a):
        r_use = np.atleast_1d(np.asarray(r, dtype=np.float64))
        z = 1 / a - 1
        R = self._R(cosmo, M, a)
        f_max = 6 ** -self.gamma_nt / self.alpha_nt
        f_z = np.min([(1 + z) ** self.nu_nt, (f_max - 1) * np.tanh(self.nu_nt * z) + 1])
        f_nt = np.clip(self.alpha_nt * f_z * (r_use / R[:, None]) ** self.gamma_nt, 0, 1)
        return _squeeze(f_nt, r, M)


class ThermalSZ(SchneiderProfiles):
    """Compton-y profile: the projected gas pressure (GPU line-of-sight kernel) times a (comoving -> physical path),
    Mpc -> cm, sigma_T / (m_e c^2) and the gas-to-electron pressure factor (Thermodynamic.py:736-754).  `real()`
    returns the reference's -99 sentinel (:757-767) so that the object can be tabulated."""

    def __init__(self, pressure=None, **kwargs):
        self.pressure = pressure if pressure is not None else Pressure(**kwargs)
        super().__init__(**kwargs)

    def Pgas_to_Pe(self, cosmo, r, M, a):
        return PTH_TO_PE

    def projected(self, cosmo, r, M, a):
        r_use, M_use = np.atleast_1d(r), np.atleast_1d(M)
        prof = np.atleast_2d(self.pressure.projected(cosmo, r_use, M_use, a))
        prof = prof * a * (MPC_TO_METER * 1e2)
        prof = prof * SIGMA_T_CGS / (M_E_CGS * C_CGS ** 2)
        prof = prof * self.Pgas_to_Pe(cosmo, r_use, M_use, a)
        return _squeeze(prof, r, M)

    def real(self, cosmo, r, M, a):
        return np.ones((np.atleast_1d(M).size, np.atleast_1d(r).size)) * -99

    def _real(self, *args):
        return np.nan

    def _projected(self, *args):
        return np.nan
