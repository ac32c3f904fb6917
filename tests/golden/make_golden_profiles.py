#!/usr/bin/env python
"""Golden values of the reference's one-halo Schneider19 profiles (DarkMatter, Stars, Gas, CollisionlessMatter,
Schneider19.py:335-1063), evaluated by the UNMODIFIED reference classes under the refshim stand-ins.
Run in the build container only.  Data only."""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from oracle.refshim import install  # noqa: E402

install.install()
import BaryonForge as bfg  # noqa: E402
import pyccl as ccl  # noqa: E402
from baryonification_amd import synthetic as syn  # noqa: E402
from make_golden_tables import PAR  # noqa: E402

d = syn.COSMO
cosmo = ccl.Cosmology(Omega_c=d['Omega_m'] - d['Omega_b'], Omega_b=d['Omega_b'], h=d['h'], sigma8=d['sigma8'],
                      n_s=d['n_s'], w0=d['w0'], matter_power_spectrum='linear')
r = np.geomspace(1e-4, 60, 50)
M = np.geomspace(1e12, 3e15, 5)
a = 1 / 1.25
out = {'r': r, 'M': M, 'a': a}
warnings.simplefilter('ignore')
for name in ('DarkMatter', 'Stars', 'Gas', 'CollisionlessMatter'):
    out[name] = getattr(bfg.Profiles, name)(**PAR).real(cosmo, r, M, a)
    out[name + '_scalarM'] = getattr(bfg.Profiles, name)(**PAR).real(cosmo, r, 2e14, a)
# a mass/redshift-dependent gas model exercises _get_gas_params (:148-192)
PAR2 = dict(PAR, mu_theta_ej=0.3, nu_theta_ej=0.5, zeta_theta_ej=0.2, mu_delta=0.1, nu_M_c=-0.3, zeta_M_c=0.4, M_delta=3e13)
out['Gas_scaled'] = bfg.Profiles.Gas(**PAR2).real(cosmo, r, M, a)
out['par2_keys'] = np.array(sorted(PAR2))
out['par2_vals'] = np.array([PAR2[k] for k in sorted(PAR2)])
# thermodynamic scalings built on Pressure (Thermodynamic.py:282-368, :431-457, :662-775), one-halo total matter as in
# make_golden_tables.py
r_t = np.geomspace(5e-3, 20, 16)
M_t = np.array([3e13, 4e14])
Pgas = bfg.Profiles.Gas(**PAR)
Ptot = bfg.Profiles.CollisionlessMatter(**PAR) + bfg.Profiles.Stars(**PAR) + bfg.Profiles.Gas(**PAR)
Pth = bfg.Profiles.Pressure(gas=Pgas, darkmatterbaryon=Ptot, **PAR)
fnt = bfg.Profiles.NonThermalFrac(**PAR)
out['thermo_r'], out['thermo_M'] = r_t, M_t
out['NonThermalFrac'] = fnt.real(cosmo, r_t, M_t, a)
out['NonThermalFrac_z0'] = fnt.real(cosmo, r_t, M_t, 1.0)
out['ElectronPressure'] = bfg.Profiles.ElectronPressure(gas=Pgas, darkmatterbaryon=Ptot, **PAR).real(cosmo, r_t, M_t, a)
out['ThermalSZ'] = bfg.Profiles.ThermalSZ(pressure=Pth * (1 - fnt), **PAR).projected(cosmo, r_t, M_t, a)
out['ThermalSZ_real'] = bfg.Profiles.ThermalSZ(pressure=Pth, **PAR).real(cosmo, r_t, M_t, a)
np.savez_compressed(os.path.join(HERE, 'profiles_s19.npz'), **out)
print({k: np.shape(v) for k, v in out.items()})
