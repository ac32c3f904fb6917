"""Table builders.  CPU part: the oracle restatement (oracle/tables.py) reproduces what the
UNMODIFIED reference classes produced (tests/golden/tables_s19.npz, see make_golden_tables.py).  GPU part: the HIP
kernels against those golden vectors and the oracle."""
import os

import numpy as np
import pytest

from oracle import tables as OT

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def G():
    f = np.load(os.path.join(HERE, 'golden', 'tables_s19.npz'))
    return {k: f[k] for k in f.files}


def test_oracle_projection_matches_reference(G):
    s = OT.project_realspace(G['l_chk'], G['rho_chk'], G['r_chk'])
    assert np.abs(s / G['sig_ref'] - 1).max() < 1e-13
    assert np.array_equal(OT.los_grid(G['r_chk'], proj_cutoff=50.0), G['l_chk'])


def test_oracle_enclosed_mass_and_displacement_match_reference(G):
    zi = 1
    a = 1 / (1 + G['z_range'][zi])
    m = OT.enclosed_mass_2d(G['l'], G['rho_dmo'][zi][:2], a, G['r'])          # 2 of 4 masses: keeps the CPU suite short
    assert np.nanmax(np.abs(m / G['M_dmo'][zi][:2] - 1)) < 1e-13
    for zi in range(2):
        d, st = OT.displacement_rows(G['r'], G['M_dmo'][zi], G['M_dmb'][zi])
        assert np.abs(d - G['d_ref'][zi]).max() <= 1e-13 * np.abs(G['d_ref'][zi]).max()
        assert np.all(st == 0)


def test_oracle_pressure_matches_reference(G):
    P = OT.pressure_profile(G['rho_tot'], G['rho_gas'], G['r_p'], cutoff=float(G['P_cutoff']))
    assert np.abs(P / G['P_ref'] - 1).max() < 1e-13


def test_oracle_degenerate_mass_profiles_give_zero_displacement():
    r = np.geomspace(1e-3, 1e2, 40)
    M = np.ones((1, 40)) * 1e14                                   # constant enclosed mass: nothing usable
    d, st = OT.displacement_rows(r, M, M * 1.1)
    assert np.all(d == 0) and st[0] in (1, 2)


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_projection(gpu, G):
    from baryonification_amd import tables as T
    s = T.project_profile(G['l_chk'], G['rho_chk'], G['r_chk'])
    assert np.abs(s / G['sig_ref'] - 1).max() < 1e-12
    assert np.array_equal(T.los_grid(G['r_chk'], proj_cutoff=50.0), G['l_chk'])


@pytest.mark.gpu
def test_hip_enclosed_mass(gpu, G):
    from baryonification_amd import tables as T
    for zi in range(2):
        a = 1 / (1 + G['z_range'][zi])
        for rho, ref in ((G['rho_dmo'][zi], G['M_dmo'][zi]), (G['rho_dmb'][zi], G['M_dmb'][zi])):
            m = T.enclosed_mass_2d(G['l'], rho, a, G['r'])
            assert np.array_equal(np.isnan(m), np.isnan(ref))
            assert np.nanmax(np.abs(m / ref - 1)) < 1e-10      # parallel prefix sum vs np.cumsum: rounding only


@pytest.mark.gpu
def test_hip_displacement_rows(gpu, G):
    from baryonification_amd import tables as T
    for zi in range(2):
        d, st = T.displacement_rows(G['r'], G['M_dmo'][zi], G['M_dmb'][zi])
        assert np.all(st == 0)
        # d = M_DMB^-1(M_DMO(r)) - r: where the mass profile flattens the inverse amplifies libm ulps, so the
        # bound is on the scale of the table, with a tight bound on the typical element
        err = np.abs(d - G['d_ref'][zi])
        assert err.max() <= 2e-9 * np.abs(G['d_ref'][zi]).max()
        assert np.median(err[G['d_ref'][zi] != 0] / np.abs(G['d_ref'][zi][G['d_ref'][zi] != 0])) < 1e-11
    # end to end: densities -> enclosed masses -> displacement, all on the GPU
    zi = 0
    a = 1 / (1 + G['z_range'][zi])
    d, st = T.displacement_rows(G['r'], T.enclosed_mass_2d(G['l'], G['rho_dmo'][zi], a, G['r']),
                                T.enclosed_mass_2d(G['l'], G['rho_dmb'][zi], a, G['r']))
    assert np.abs(d - G['d_ref'][zi]).max() <= 1e-8 * np.abs(G['d_ref'][zi]).max()
    # degenerate rows behave like the reference (d = 0 + status)
    r = np.geomspace(1e-3, 1e2, 40)
    M = np.ones((2, 40)) * 1e14
    d, st = T.displacement_rows(r, M, M * 1.1)
    do, so = OT.displacement_rows(r, M, M * 1.1)
    assert np.all(d == 0) and np.array_equal(st, so)


@pytest.mark.gpu
def test_hip_displacement_rows_random_profiles_vs_oracle(gpu):
    """noisy, partly non-monotone mass profiles exercise the iterative mask"""
    from baryonification_amd import tables as T
    rng = np.random.default_rng(3)
    r = np.geomspace(1e-3, 3e2, 200)
    base = 1e14 * (r / 1.0) ** 1.5 / (1 + (r / 2.0) ** 1.5)
    M_dmo = np.stack([base * (1 + 0.0 * rng.random(r.size)) for _ in range(6)])
    M_dmb = np.stack([base * (1 + 0.05 * np.tanh((np.log(r) + k * 0.3)) + 2e-6 * rng.standard_normal(r.size)) for k in range(6)])
    M_dmb[2, 50:60] = M_dmb[2, 49]                       # flat stretch
    M_dmb[3, 100] = np.nan
    d, st = T.displacement_rows(r, M_dmo, M_dmb)
    do, so = OT.displacement_rows(r, M_dmo, M_dmb)
    assert np.array_equal(st, so)
    assert np.abs(d - do).max() <= 1e-9 * max(np.abs(do).max(), 1e-30)


@pytest.mark.gpu
def test_hip_pressure(gpu, G):
    from baryonification_amd import tables as T
    P = T.pressure_profile(G['rho_tot'], G['rho_gas'], G['r_p'], cutoff=float(G['P_cutoff']))
    assert np.abs(P / G['P_ref'] - 1).max() < 1e-10


# ------------------------------------------------------------------------------------------------- 3-D builder
@pytest.fixture(scope='module')
def G3():
    f = np.load(os.path.join(HERE, 'golden', 'tables3d_s19.npz'))
    return {k: f[k] for k in f.files}


def _analytic_rho(r, M):
    """the closed-form density of tests/golden/make_golden_tables3d.py (hole, negative patch, far cut-off)"""
    M = np.atleast_1d(M)[:, None]
    rs = 0.3 * (M / 1e14) ** (1.0 / 3.0)
    x = r[None, :] / rs
    rho = M / (4 * np.pi * rs ** 3) / (x * (1 + x) ** 2) / (1 + (r[None, :] / 30.0) ** 2) ** 2
    rho = np.where((r[None, :] > 2.0) & (r[None, :] < 2.2), 0.0, rho)
    rho = np.where((r[None, :] > 8.0) & (r[None, :] < 8.3), -1e-3 * rho, rho)
    return np.where(r[None, :] > 900.0, 0.0, rho)


def test_oracle_enclosed_mass_3d_matches_reference(G3):
    r_int = OT.r_int_3d(G3['an_r'])
    assert r_int.size == 50_000 and np.isclose(r_int[0], 1e-6 / 1.2) and np.isclose(r_int[-1], 1.2e3 * 1.2)
    m = OT.enclosed_mass_3d(r_int, _analytic_rho(r_int, G3['an_M']), G3['an_r'])
    assert np.array_equal(np.isnan(m), np.isnan(G3['an_Menc']))
    assert np.nanmax(np.abs(m / G3['an_Menc'] - 1)) < 1e-13
    for zi in range(2):           # the displacement step is shared with the 2-D builder
        d, st = OT.displacement_rows(G3['r'], G3['M_dmo'][zi], G3['M_dmb'][zi])
        assert np.abs(d - G3['d_ref'][zi]).max() <= 1e-13 * np.abs(G3['d_ref'][zi]).max() and np.all(st == 0)


@pytest.mark.gpu
def test_hip_enclosed_mass_3d(gpu, G3):
    from baryonification_amd import tables as T
    r_int = T.r_int_3d(G3['an_r'])
    assert np.array_equal(r_int, OT.r_int_3d(G3['an_r']))
    m = T.enclosed_mass_3d(r_int, _analytic_rho(r_int, G3['an_M']), G3['an_r'])
    assert np.array_equal(np.isnan(m), np.isnan(G3['an_Menc']))
    assert np.nanmax(np.abs(m / G3['an_Menc'] - 1)) < 1e-10          # parallel prefix sum vs np.cumsum, r*r*r vs r**3


@pytest.mark.gpu
@pytest.mark.filterwarnings('ignore')
def test_baryonification3d_end_to_end(gpu, G3):
    """our Schneider19 port + GPU builders reproduce the reference's Baryonification3D table (notebook-10 settings)"""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    par = dict(zip([str(k) for k in G3['par_keys']], G3['par_vals'].tolist()))
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    DMO = bfg.Profiles.DarkMatter(**par)
    DMB = bfg.Profiles.CollisionlessMatter(**par) + bfg.Profiles.Stars(**par) + bfg.Profiles.Gas(**par)
    model = bfg.Profiles.Baryonification3D(DMO, DMB, cosmo, epsilon_max=20)
    m = model.get_masses(DMO, G3['r'], G3['M_range'], 1.0)
    assert np.nanmax(np.abs(m / G3['M_dmo'][0] - 1)) < 1e-9
    assert model.get_masses(DMO, G3['r'], 1e14, 1.0).shape == (80,)
    model.setup_interpolator(z_min=0, z_max=0.01, N_samples_z=2, z_linear_sampling=True, M_min=1e13, M_max=1e15,
                             N_samples_Mass=3, R_min=1e-4, R_max=300, N_samples_R=80, verbose=False)
    err = np.abs(model.raw_input_d - G3['d_ref'])
    assert err.max() <= 2e-4 * np.abs(G3['d_ref']).max()             # CollisionlessMatter port differs at ~1e-6
    nz = G3['d_ref'] != 0
    assert np.median(err[nz] / np.abs(G3['d_ref'][nz])) < 1e-5
