"""
Gas pressure from hydrostatic equilibrium: interface-compatible with `Pressure` of
BaryonForge/Profiles/Thermodynamic.py:42-278.  `real()` evaluates the total and gas densities on the
hard-coded 500-point grid on the host and runs the two radial integrals + log-log PCHIP read-out of
Thermodynamic.py:240-271 on the GPU (tables.pressure_profile); `projected()` is the GPU line-of-sight kernel.
"""
import numpy as np

from .. import tables
from .Schneider19 import DarkMatterBaryon, Gas, SchneiderProfiles, _squeeze

__all__ = ['Pressure']


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
