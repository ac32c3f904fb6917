#!/usr/bin/env python
"""
Golden vectors for the table builders, produced by the UNMODIFIED reference classes
imported from /root/reference under the refshim stand-ins (see make_golden.py for the rationale):

    BaryonForge.Profiles.Baryonification2D.setup_interpolator / get_masses   (BaryonCorrection.py:136-321, 585-665)
    BaryonForge.Profiles.SchneiderProfiles._projected_realspace              (Schneider19.py:195-265)
    BaryonForge.Profiles.Pressure._real                                      (Thermodynamic.py:174-278)
    + the one-halo Schneider19 profiles DarkMatter, Stars, Gas, CollisionlessMatter that feed them.

Stored: the 3-D densities sampled on the grids the builders use (the inputs of the GPU kernels), and the
reference's enclosed masses, displacement table and pressure profile (the expected outputs).  Data only.
TwoHalo needs CCL's P(k)/sigma(M) and is left out: DMO = DarkMatter, DMB = CollisionlessMatter + Stars + Gas.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle.refshim import install  # noqa: E402

install.install()
import BaryonForge as bfg  # noqa: E402
import pyccl as ccl  # noqa: E402
from baryonification_amd import synthetic as syn  # noqa: E402
from oracle import tables as OT  # noqa: E402

# examples/default_config.npy (decoded in SURVEY.md section 5), cdelta = 7, proj_cutoff = 50
PAR = dict(epsilon=4.0, theta_ej=4.0, theta_co=0.1, M_c=1e14, mu_beta=0.1, gamma=2.5, delta=7.0, eta=0.3, eta_delta=0.1,
           tau=-1.5, tau_delta=0.0, A=0.055, M1=3e11, epsilon_h=0.015, a=0.3, n=2.0, p=0.3, q=0.707, cdelta=7.0,
           alpha_nt=0.2, nu_nt=0.5, gamma_nt=0.5, cutoff=1000.0, proj_cutoff=50.0, mu_theta_ej=0.1, mu_theta_co=0.0,
           M_theta_ej=5e13, M_theta_co=5e13)


def main():
    d = syn.COSMO
    cosmo = ccl.Cosmology(Omega_c=d['Omega_m'] - d['Omega_b'], Omega_b=d['Omega_b'], h=d['h'], sigma8=d['sigma8'],
                          n_s=d['n_s'], w0=d['w0'], matter_power_spectrum='linear')
    DMO = bfg.Profiles.DarkMatter(**PAR)
    DMB = bfg.Profiles.CollisionlessMatter(**PAR) + bfg.Profiles.Stars(**PAR) + bfg.Profiles.Gas(**PAR)
    model = bfg.Profiles.Baryonification2D(DMO, DMB, cosmo, epsilon_max=20)

    z_range = np.array([0.2, 0.3])
    M_range = np.geomspace(1e13, 1e15, 4)
    r = np.geomspace(1e-3, 3e2, 60)
    t0 = time.time()
    model.setup_interpolator(z_min=0.2, z_max=0.3, N_samples_z=2, z_linear_sampling=True, M_min=1e13, M_max=1e15,
                             N_samples_Mass=4, R_min=1e-3, R_max=3e2, N_samples_R=60, verbose=False)
    print("reference setup_interpolator: %.1f s" % (time.time() - t0))
    d_ref = model.raw_input_d.copy()

    # inputs of the kernels: densities on the line-of-sight grid the reference builds inside get_masses
    r_int = OT.r_int_2d(r)
    l = OT.los_grid(r_int, DMO.padding_lo_proj, DMO.padding_hi_proj, DMO.n_per_decade_proj, DMO.proj_cutoff)
    rho_dmo = np.stack([DMO.real(cosmo, l, M_range, 1 / (1 + z)) for z in z_range])
    rho_dmb = np.stack([DMB.real(cosmo, l, M_range, 1 / (1 + z)) for z in z_range])
    t0 = time.time()
    M_dmo = np.stack([model.get_masses(DMO, r, M_range, 1 / (1 + z)) for z in z_range])
    M_dmb = np.stack([model.get_masses(DMB, r, M_range, 1 / (1 + z)) for z in z_range])
    print("reference get_masses x4: %.1f s" % (time.time() - t0))
    # a coarse check of the projection itself (reference method on a short radius list)
    r_chk = np.geomspace(2e-3, 40, 24)
    DMOp = bfg.Profiles.DarkMatter(**PAR)
    sig_ref = DMOp.projected(cosmo, r_chk, M_range, 1 / 1.2)
    l_chk = OT.los_grid(r_chk, DMOp.padding_lo_proj, DMOp.padding_hi_proj, DMOp.n_per_decade_proj, DMOp.proj_cutoff)
    rho_chk = DMOp.real(cosmo, l_chk, M_range, 1 / 1.2)

    # oracle pins
    for zi, z in enumerate(z_range):
        a = 1 / (1 + z)
        m1 = OT.enclosed_mass_2d(l, rho_dmo[zi], a, r)
        m2 = OT.enclosed_mass_2d(l, rho_dmb[zi], a, r)
        dd, st = OT.displacement_rows(r, M_dmo[zi], M_dmb[zi])
        print("z=%.2f  oracle/ref: M_DMO %.2e  M_DMB %.2e  d %.2e (abs, max|d|=%.3e)  status %s" % (
            z, np.nanmax(np.abs(m1 / M_dmo[zi] - 1)), np.nanmax(np.abs(m2 / M_dmb[zi] - 1)),
            np.abs(dd - d_ref[zi]).max(), np.abs(d_ref[zi]).max(), st))
    print("projection oracle/ref: %.2e" % np.abs(OT.project_realspace(l_chk, rho_chk, r_chk) / sig_ref - 1).max())

    # Pressure (Thermodynamic.py:174-278) with an explicit one-halo total-matter profile
    Pgas = bfg.Profiles.Gas(**PAR)
    Ptot = bfg.Profiles.CollisionlessMatter(**PAR) + bfg.Profiles.Stars(**PAR) + bfg.Profiles.Gas(**PAR)
    P = bfg.Profiles.Pressure(gas=Pgas, darkmatterbaryon=Ptot, **PAR)
    r_p = np.geomspace(5e-3, 30, 30)
    a_p = 1 / 1.25
    P_ref = P.real(cosmo, r_p, M_range, a_p)
    r500 = np.geomspace(1e-6, 1000, 500)
    rho_tot = P.DarkMatterBaryon.real(cosmo, r500, M_range, a_p)
    rho_gas = P.Gas.real(cosmo, r500, M_range, a_p)
    P_or = OT.pressure_profile(rho_tot, rho_gas, r_p, cutoff=P.cutoff)
    print("pressure oracle/ref: %.2e   (cutoff %s)" % (np.abs(P_or / P_ref - 1).max(), P.cutoff))

    np.savez_compressed(os.path.join(HERE, 'tables_s19.npz'), z_range=z_range, M_range=M_range, r=r, l=l,
                        rho_dmo=rho_dmo, rho_dmb=rho_dmb, M_dmo=M_dmo, M_dmb=M_dmb, d_ref=d_ref,
                        r_chk=r_chk, l_chk=l_chk, rho_chk=rho_chk, sig_ref=sig_ref,
                        r_p=r_p, a_p=a_p, rho_tot=rho_tot, rho_gas=rho_gas, P_ref=P_ref, P_cutoff=P.cutoff,
                        par_keys=np.array(sorted(PAR)), par_vals=np.array([PAR[k] for k in sorted(PAR)]))


if __name__ == '__main__':
    main()
