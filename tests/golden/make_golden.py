#!/usr/bin/env python
"""
Generates tests/golden/*.npz by running the UNMODIFIED reference (imported from /root/reference)
on small synthetic inputs.  Run in the build container only:

    python tests/golden/make_golden.py

healpy / pyccl / numba cannot be installed offline, so oracle/refshim/ provides stand-ins for
them (geometry + background cosmology); everything BaryonForge itself computes -- the runner
loops, `_readout`, the RegularGridInterpolator use, the regrid -- runs as shipped:

    BaryonForge.Runners.BaryonifyShell.process        (HealpixRunner.py:240-349)
    BaryonForge.Runners.PaintProfilesShell.process    (HealpixRunner.py:366-447)
    BaryonForge.Profiles.Baryonification2D.displacement/_readout   (BaryonCorrection.py:324-431)
    BaryonForge.utils.TabulatedProfile.projected/_readout          (Tabulate.py:246-358)

Each fixture stores the inputs and the reference's output map (data only, no reference source).
The script also prints the oracle-vs-reference residual for every case (the oracle's pin).
"""
import os
import sys
import time

import numpy as np
from scipy import interpolate

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle.refshim import install  # noqa: E402

install.install()
import BaryonForge as bfg  # noqa: E402  (the reference, from /root/reference)
import pyccl as ccl  # noqa: E402  (refshim stand-in)

from baryonification_amd import synthetic as syn  # noqa: E402
from oracle import oracle as O  # noqa: E402

COSMO_B = dict(syn.COSMO, Omega_m=0.27, h=0.70)      # a second cosmology for the model side


def ccl_cosmo(d):
    return ccl.Cosmology(Omega_c=d['Omega_m'] - d['Omega_b'], Omega_b=d['Omega_b'], h=d['h'],
                         sigma8=d['sigma8'], n_s=d['n_s'], w0=d['w0'], matter_power_spectrum='linear')


def ref_displacement_model(z, M, r_axis, d_table, rdelta, eps_model, cosmo_dict):
    """reference Baryonification2D carrying a ready-made table (attributes as set at
    BaryonCorrection.py:306-316); DMO/DMB profile objects are not needed for read-out"""
    model = bfg.Profiles.Baryonification2D.__new__(bfg.Profiles.Baryonification2D)
    model.cosmo = ccl_cosmo(cosmo_dict)
    model.epsilon_max = eps_model
    model.mass_def = ccl.halos.massdef.MassDef(200, 'critical')
    model.p_keys = []
    grid = (np.log(1 + z), np.log(M), np.log(r_axis))
    model.raw_input_d = d_table
    model.raw_input_z_range, model.raw_input_M_range, model.raw_input_r_range = grid
    model.interp_d = interpolate.RegularGridInterpolator(grid, d_table, bounds_error=False, fill_value=np.nan)
    model.Rdelta_sampling = rdelta
    return model


def ref_tabulated_profile(z, M, r, P_table, cosmo_dict):
    """reference TabulatedProfile with the attributes set at Tabulate.py:229-238"""
    prof = bfg.utils.TabulatedProfile(model=None, cosmo=ccl_cosmo(cosmo_dict))
    grid = (np.log(1 + z), np.log(M), np.log(r))
    prof.raw_input_3D = P_table
    prof.raw_input_2D = P_table
    prof.raw_input_z_range, prof.raw_input_M_range, prof.raw_input_r_range = grid
    with np.errstate(divide='ignore'):
        prof.interp3D = interpolate.RegularGridInterpolator(grid, np.log(P_table), bounds_error=False)
        prof.interp2D = interpolate.RegularGridInterpolator(grid, np.log(P_table), bounds_error=False)
    return prof


def ref_param_profile(z, M, r, P_table, p_axes, cosmo_dict):
    """reference ParamTabulatedProfile with the attributes set at Tabulate.py:551-561"""
    prof = bfg.utils.ParamTabulatedProfile(model=None, cosmo=ccl_cosmo(cosmo_dict))
    prof.p_keys = list(p_axes.keys())
    grid = tuple([np.log(1 + z), np.log(M), np.log(r)] + [p_axes[k] for k in prof.p_keys])
    prof.raw_input_3D = P_table
    prof.raw_input_2D = P_table
    prof.raw_input_z_range, prof.raw_input_M_range, prof.raw_input_r_range = grid[:3]
    for k in prof.p_keys:
        setattr(prof, 'raw_input_%s_range' % k, p_axes[k])
    prof.interp3D = interpolate.RegularGridInterpolator(grid, np.log(P_table), bounds_error=False)
    prof.interp2D = interpolate.RegularGridInterpolator(grid, np.log(P_table), bounds_error=False)
    return prof


def run_param_paint(name, nside, cat, eps_runner, z, M, r, p_axes, P_table, cosmo=syn.COSMO):
    """PaintProfilesShell with per-halo property columns (HealpixRunner.py:407-425, Tabulate.py:569-621)"""
    extra = {k: cat[k] for k in p_axes}
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=cosmo, **extra)
    cat_used = {k: np.array(Catalog.cat[k]) for k in Catalog.cat.dtype.names}
    Shell = bfg.utils.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
    model = ref_param_profile(z, M, r, P_table, p_axes, cosmo)
    out = bfg.Runners.PaintProfilesShell(Catalog, Shell, eps_runner, model, verbose=False).process()
    otab = O.Table([np.log(1 + z), np.log(M), np.log(r)] + [p_axes[k] for k in p_axes], np.log(P_table), p_keys=list(p_axes))
    oout = O.paint_shell(nside, cat_used, otab, eps_runner, O.Background.from_dict(cosmo))
    print(f"{name:14s} paint     nside={nside:4d} N={cat['M'].size:5d} p_keys={list(p_axes)}  "
          f"max|oracle-ref|/max|ref| = {np.abs(oout - out).max() / np.abs(out).max():.3e}  nonzero px = {int((out != 0).sum())}")
    np.savez_compressed(os.path.join(HERE, name + '.npz'), kind='paint', nside=nside, eps_runner=eps_runner, eps_model=0.0,
                        rdelta=False, cat_M=cat_used['M'], cat_z=cat_used['z'], cat_ra=cat_used['ra'], cat_dec=cat_used['dec'],
                        tab_z=z, tab_M=M, tab_r=r, tab_values=P_table, map_in=np.zeros(0, dtype=np.uint8),
                        cosmo_runner=np.array([cosmo[k] for k in ('Omega_m', 'Omega_b', 'h', 'sigma8', 'n_s', 'w0')]),
                        cosmo_model=np.array([cosmo[k] for k in ('Omega_m', 'Omega_b', 'h', 'sigma8', 'n_s', 'w0')]),
                        p_keys=np.array(list(p_axes)), **{'p_axis_' + k: v for k, v in p_axes.items()},
                        **{'cat_' + k: cat_used[k] for k in p_axes}, expected=out)


def special_catalog(N, seed, z_lo, z_hi, logM_lo, logM_hi):
    cat = syn.make_catalog(N, seed=seed, z_lo=z_lo, z_hi=z_hi, logM_lo=logM_lo, logM_hi=logM_hi)
    # hand-placed halos: poles (exact -> clipped by the catalog class), near-pole, ra wrap-around
    cat['dec'][:6] = [90.0, -90.0, 89.7, -89.6, 0.0, 35.0]
    cat['ra'][:6] = [10.0, 200.0, 123.0, 300.0, 0.0, 359.999]
    cat['M'][:6] = 10.0 ** np.array([14.9, 14.7, 15.0, 14.5, 14.8, 14.6]) * (10 ** (logM_hi - 15.3))
    return cat


def run_case(name, kind, nside, cat, eps_runner, eps_model, table_axes, table, rdelta=False,
             cosmo_runner=syn.COSMO, cosmo_model=syn.COSMO, map_seed=syn.SEED_MAP, md_runner=(200.0, 'critical'),
             md_model=(200.0, 'critical')):
    z, M, r_axis = table_axes
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=cosmo_runner)
    cat_used = {k: np.array(Catalog.cat[k]) for k in ('M', 'z', 'ra', 'dec')}      # after pole clipping
    t0 = time.time()
    if kind == 'baryonify':
        hmap = syn.make_map(nside, seed=map_seed)
        Shell = bfg.utils.LightconeShell(map=hmap, cosmo=cosmo_runner)
        model = ref_displacement_model(z, M, r_axis, table, rdelta, eps_model, cosmo_model)
        model.mass_def = ccl.halos.massdef.MassDef(*md_model)
        out = bfg.Runners.BaryonifyShell(Catalog, Shell, eps_runner, model, mass_def=ccl.halos.massdef.MassDef(*md_runner),
                                         verbose=False).process()
        otab = O.Table([np.log(1 + z), np.log(M), np.log(r_axis)], table, rdelta, eps_model)
        oout = O.baryonify_shell(nside, hmap, cat_used, otab, eps_runner, O.Background.from_dict(cosmo_runner),
                                 O.Background.from_dict(cosmo_model), md_runner, md_model)
    else:
        hmap = np.zeros(12 * nside * nside)
        Shell = bfg.utils.LightconeShell(map=hmap, cosmo=cosmo_runner)
        model = ref_tabulated_profile(z, M, r_axis, table, cosmo_model)
        out = bfg.Runners.PaintProfilesShell(Catalog, Shell, eps_runner, model, mass_def=ccl.halos.massdef.MassDef(*md_runner),
                                             verbose=False).process()
        with np.errstate(divide='ignore'):
            otab = O.Table([np.log(1 + z), np.log(M), np.log(r_axis)], np.log(table))
        oout = O.paint_shell(nside, cat_used, otab, eps_runner, O.Background.from_dict(cosmo_runner), md_runner=md_runner)
    dt = time.time() - t0
    scale = np.abs(out).max()
    print(f"{name:14s} {kind:9s} nside={nside:4d} N={cat['M'].size:5d} ref+oracle {dt:6.1f}s  "
          f"max|oracle-ref|/max|ref| = {np.abs(oout - out).max() / scale:.3e}   changed px = "
          f"{int((out != hmap).sum())}")
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'), kind=kind, nside=nside, eps_runner=eps_runner, eps_model=eps_model,
        rdelta=rdelta, cat_M=cat_used['M'], cat_z=cat_used['z'], cat_ra=cat_used['ra'], cat_dec=cat_used['dec'],
        tab_z=z, tab_M=M, tab_r=r_axis, tab_values=table,
        map_in=hmap.astype(np.uint8) if kind == 'baryonify' else np.zeros(0, dtype=np.uint8),
        cosmo_runner=np.array([cosmo_runner[k] for k in ('Omega_m', 'Omega_b', 'h', 'sigma8', 'n_s', 'w0')]),
        cosmo_model=np.array([cosmo_model[k] for k in ('Omega_m', 'Omega_b', 'h', 'sigma8', 'n_s', 'w0')]),
        md_runner=np.array([md_runner[0], 0.0 if md_runner[1] == 'critical' else 1.0]),
        md_model=np.array([md_model[0], 0.0 if md_model[1] == 'critical' else 1.0]),
        expected=out)


def main_massdef():
    """runner mass definition 500c, model mass definition 200m (HealpixRunner.py:296, BaryonCorrection.py:370)"""
    pad = 1e-9
    cat = special_catalog(200, 31, 0.02, 0.08, 13.0, 15.2)
    ax = syn.table_grid(cat, Nz=5, NM=6, NR=100, pad=pad)
    run_case('massdef_baryonify', 'baryonify', 64, cat, 14.0, 9.0, ax, syn.displacement_table(*ax), md_runner=(500.0, 'critical'),
             md_model=(200.0, 'matter'))
    run_case('massdef_paint', 'paint', 64, cat, 12.0, 0.0, ax, syn.paint_table(*ax), md_runner=(180.0, 'matter'))


def main():
    pad = 1e-9   # table z/M edges = catalog min/max widened by 1e-9 (edge membership is ulp-fragile in the reference)
    # C1: BASELINE.json configs[0] -- 1e3 halos, NSIDE 128, closed-form displacement table 10 x 10 x 500
    cat = syn.make_catalog(1000)
    ax = syn.table_grid(cat, pad=pad)
    run_case('c1_baryonify', 'baryonify', 128, cat, 10.0, 10.0, ax, syn.displacement_table(*ax))

    # low-z, massive halos on a coarse map: discs of 10^2-10^3 pixels, polar caps, phi wrap-around,
    # runner epsilon != model epsilon (the r < eps*R mask bites)
    cat = special_catalog(300, 7, 0.01, 0.06, 13.0, 15.3)
    ax = syn.table_grid(cat, Nz=6, NM=7, NR=120, R_min=1e-3, R_max=3e2, pad=pad)
    run_case('lowz_baryonify', 'baryonify', 64, cat, 15.0, 12.0, ax, syn.displacement_table(*ax))

    # Rdelta-sampled table (BaryonCorrection.py:374-379) + model cosmology != runner cosmology
    cat = special_catalog(150, 11, 0.02, 0.08, 13.0, 15.0)
    z, M, _ = syn.table_grid(cat, Nz=5, NM=6, pad=pad)
    rd = np.geomspace(1e-3, 30, 90)                           # r / R_Delta axis
    Rc = syn._Rc(z, M, COSMO_B)[:, :, None]
    x = rd[None, None, :]
    d_rd = -0.05 * Rc * x * np.exp(-x) / (1 + x * x)
    run_case('rdelta_baryonify', 'baryonify', 64, cat, 10.0, 20.0, (z, M, rd), d_rd, rdelta=True,
             cosmo_model=COSMO_B, map_seed=5)

    # painting: same low-z catalog; profile set to 0 beyond 6 R_c (log -> -inf rows in the table)
    cat = special_catalog(300, 7, 0.01, 0.06, 13.0, 15.3)
    ax = syn.table_grid(cat, Nz=6, NM=7, NR=120, pad=pad)
    P = syn.paint_table(*ax)
    P[ax[2][None, None, :] / syn._Rc(ax[0], ax[1])[:, :, None] > 6.0] = 0.0
    run_case('lowz_paint', 'paint', 64, cat, 8.0, 0.0, ax, P)

    # painting at the C1 catalog: many halos with empty discs (no fallback in the paint runner)
    cat = syn.make_catalog(500)
    ax = syn.table_grid(cat, pad=pad)
    run_case('c1_paint', 'paint', 128, cat, 10.0, 0.0, ax, syn.paint_table(*ax))


def main_params():
    pad = 1e-9
    rng = np.random.default_rng(99)
    cat = special_catalog(250, 21, 0.01, 0.06, 13.0, 15.3)
    cat['cdelta'] = rng.uniform(3.0, 9.0, 250)
    cat['cdelta'][:3] = [3.0, 9.0, 9.5]                      # on both axis ends, and outside (-> NaN -> paints nothing)
    cat['fgas'] = rng.uniform(0.05, 0.15, 250)
    z, M, r = syn.table_grid(cat, Nz=4, NM=5, NR=70, pad=pad)
    base = syn.paint_table(z, M, r)
    c_ax = np.array([3.0, 5.0, 9.0])
    f_ax = np.linspace(0.05, 0.15, 4)
    P4 = base[..., None] * (1 + 0.1 * (c_ax - 5.0))[None, None, None, :]
    run_param_paint('param1_paint', 64, cat, 8.0, z, M, r, {'cdelta': c_ax}, P4)
    P5 = P4[..., None] * (f_ax / 0.1)[None, None, None, None, :] ** 1.5
    run_param_paint('param2_paint', 64, cat, 8.0, z, M, r, {'cdelta': c_ax, 'fgas': f_ax}, P5)
    # three and four property axes (Tabulate.py:524-561 takes any number): 32 / 64 corner rows per halo
    cat['theta'] = rng.uniform(2.0, 6.0, 250)
    cat['theta'][5] = 6.5                                    # outside the third axis
    cat['mu'] = rng.uniform(-0.2, 0.4, 250)
    cat['mu'][7:9] = [-0.2, 0.4]                             # on both ends of the fourth
    t_ax = np.array([2.0, 3.5, 6.0])
    m_ax = np.array([-0.2, 0.1, 0.4])
    P6 = P5[..., None] * (1 + 0.05 * (t_ax - 4.0))[None, None, None, None, None, :]
    run_param_paint('param3_paint', 64, cat, 8.0, z, M, r, {'cdelta': c_ax, 'fgas': f_ax, 'theta': t_ax}, P6)
    P7 = P6[..., None] * np.exp(0.3 * m_ax)[None, None, None, None, None, None, :]
    run_param_paint('param4_paint', 64, cat, 8.0, z, M, r, {'cdelta': c_ax, 'fgas': f_ax, 'theta': t_ax, 'mu': m_ax}, P7)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'params':
        main_params()
    elif len(sys.argv) > 1 and sys.argv[1] == 'massdef':
        main_massdef()
    else:
        main()
