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
