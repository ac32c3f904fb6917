#!/usr/bin/env python
"""Timing of the table builders at the reference notebooks' grid (N_M = 30, N_R = 2000 per
redshift sample; examples/04_Baryonify_Density_Shell.ipynb:222-226 reports 2.4-3.4 s per z-sample on a laptop)."""
import json
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, '.')
import baryonification_amd as bfg
from baryonification_amd import synthetic as syn, tables as T

warnings.simplefilter('ignore')
par = dict(syn.S19_PARAMS)
cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
M = np.geomspace(1e12, 1e16, 30)
r = np.geomspace(1e-3, 1e2, 2000)
a = 1 / 1.2
DMO, DMB = bfg.Profiles.DarkMatterOnly(**par), bfg.Profiles.DarkMatterBaryon(**par)
model = bfg.Profiles.Baryonification2D(DMO, DMB, cosmo)
model.get_masses(DMO, r, M[:2], a)                     # warm-up (library load, first launches)
r_int = T.r_int_2d(r)
l = T.los_grid(r_int, DMO.padding_lo_proj, DMO.padding_hi_proj, DMO.n_per_decade_proj, DMO.proj_cutoff)
t0 = time.perf_counter(); rho_o = DMO.real(cosmo, l, M, a); rho_b = DMB.real(cosmo, l, M, a); t_host = time.perf_counter() - t0
t0 = time.perf_counter(); Mo = T.enclosed_mass_2d(l, rho_o, a, r); Mb = T.enclosed_mass_2d(l, rho_b, a, r); t_mass = time.perf_counter() - t0
t0 = time.perf_counter(); d, st = T.displacement_rows(r, Mo, Mb); t_disp = time.perf_counter() - t0
t0 = time.perf_counter(); sig = T.project_profile(l, rho_o, r_int); t_proj = time.perf_counter() - t0
P = bfg.Profiles.Pressure(**par)
rp = np.geomspace(1e-3, 1e2, 2000)
P.real(cosmo, rp[:4], M[:2], a)
t0 = time.perf_counter(); pr = P.real(cosmo, rp, M, a); t_press = time.perf_counter() - t0
t0 = time.perf_counter(); pp = P.projected(cosmo, rp, M, a); t_pproj = time.perf_counter() - t0
print(json.dumps({
    "grid": "N_M=30, N_R=2000, one redshift sample, Schneider19 default_config parameters",
    "host_profiles_s (numpy, 2 x 30 rows on %d LOS nodes)" % l.size: t_host,
    "gpu_enclosed_mass_2d_x2_s (30 rows x 50000-pt projection + prefix sum + PCHIP; incl. H2D/D2H)": t_mass,
    "gpu_displacement_rows_s": t_disp,
    "gpu_projection_only_s (30 x 50000 radii x %d nodes)" % l.size: t_proj,
    "per_z_sample_total_s": t_host + t_mass + t_disp,
    "pressure_real_s (30 rows; host densities + GPU integrals)": t_press,
    "pressure_projected_s (30 x 2000, GPU line-of-sight of the GPU pressure profile)": t_pproj,
    "status_nonzero_rows": int((st != 0).sum()),
    "reference_published": "2.37-3.39 s per z-sample (displacement table), 5.37-6.91 s (tSZ table) -- laptop, FFTLog-convolved profiles",
}))
