"""Host-side Schneider19 one-halo profiles (baryonification_amd/Profiles/Schneider19.py) against values produced by
the UNMODIFIED reference classes (tests/golden/profiles_s19.npz, make_golden_profiles.py), and the end-to-end
table construction (profiles -> GPU kernels -> displacement / pressure tables)."""
import os
import warnings

import numpy as np
import pytest

import baryonification_amd as bfg
from baryonification_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))


def _par(G, keys='par_keys', vals='par_vals'):
    return {str(k): float(v) for k, v in zip(G[keys], G[vals])}


@pytest.fixture(scope='module')
def P():
    f = np.load(os.path.join(HERE, 'golden', 'profiles_s19.npz'))
    return {k: f[k] for k in f.files}


@pytest.fixture(scope='module')
def T():
    f = np.load(os.path.join(HERE, 'golden', 'tables_s19.npz'))
    return {k: f[k] for k in f.files}


@pytest.mark.filterwarnings('ignore')
@pytest.mark.parametrize('name,tol', [('DarkMatter', 1e-13), ('Stars', 1e-11), ('Gas', 1e-13), ('CollisionlessMatter', 1e-5)])
def test_profiles_match_reference(P, T, name, tol):
    par = _par(T)
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    prof = getattr(bfg.Profiles, name)(**par)
    got = prof.real(cosmo, P['r'], P['M'], float(P['a']))
    ref = P[name]
    assert got.shape == ref.shape
    nz = ref != 0
    assert np.abs(got[nz] / ref[nz] - 1).max() < tol and np.all(got[~nz] == 0)
    got1, ref1 = prof.real(cosmo, P['r'], 2e14, float(P['a'])), P[name + '_scalarM']
    assert got1.shape == P['r'].shape and np.abs(got1[ref1 != 0] / ref1[ref1 != 0] - 1).max() < tol and np.all(got1[ref1 == 0] == 0)
    assert np.ndim(prof.real(cosmo, 0.1, 2e14, float(P['a']))) == 0


@pytest.mark.filterwarnings('ignore')
def test_gas_mass_redshift_concentration_scalings(P):
    par = _par(P, 'par2_keys', 'par2_vals')
    got = bfg.Profiles.Gas(**par).real(bfg.utils.Cosmology.from_dict(syn.COSMO), P['r'], P['M'], float(P['a']))
    assert np.abs(got / P['Gas_scaled'] - 1).max() < 1e-13


def test_profile_protocol(T):
    par = _par(T)
    dm = bfg.Profiles.DarkMatter(**par)
    assert dm.model_params['epsilon'] == par['epsilon'] and dm.mu_beta == par['mu_beta'] and dm.M_gamma == 1e14
    assert bfg.Profiles.Gas().nu_delta == 0 and bfg.Profiles.Gas().theta_ej is None       # defaults, Schneider19.py:84-92
    clm = bfg.Profiles.CollisionlessMatter(**dict(par, cutoff=50.0))
    assert clm.cutoff == 50.0 and clm.Gas.cutoff == 1000 and clm.DarkMatter.cutoff == 1000     # :945-947
    dmb = bfg.Profiles.DarkMatterBaryon(**par)
    dmb.set_parameter('theta_ej', 9.0)
    assert dmb.Gas.theta_ej == 9.0 and dmb.CollisionlessMatter.Gas.theta_ej == 9.0
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    r = np.geomspace(1e-2, 5, 7)
    s = (dm + 2 * dm - dm / 2.0).real(cosmo, r, 1e14, 0.8)
    assert np.allclose(s, 2.5 * dm.real(cosmo, r, 1e14, 0.8), rtol=1e-14)
    with pytest.raises(NotImplementedError):
        bfg.Profiles.TwoHalo(**par).real(cosmo, r, 1e14, 0.8)
    with pytest.raises(NotImplementedError):
        bfg.Profiles.DarkMatter(**dict(par, cdelta=None)).real(cosmo, r, 1e14, 0.8)
    # cdelta = None (the reference's default_config: Diemer15, Schneider19.py:390-397) with a user's c(M): a constant c_of_M == cdelta, and a
    # mass-dependent one reaches the sub-profiles and the arithmetic of profiles too
    calls = []

    def c7(cosmo_, M_, a_):
        calls.append((np.shape(M_), a_))
        return par['cdelta'] * np.ones_like(M_)
    dmc = bfg.Profiles.DarkMatter(**dict(par, cdelta=None), c_of_M=c7)
    assert np.allclose(dmc.real(cosmo, r, np.array([1e13, 1e14]), 0.8), dm.real(cosmo, r, np.array([1e13, 1e14]), 0.8), rtol=1e-14) and calls
    cm = lambda cosmo_, M_, a_: 5.0 * (np.asarray(M_) / 1e14) ** -0.1
    dmm = bfg.Profiles.DarkMatter(**dict(par, cdelta=None), c_of_M=cm)
    ref5 = bfg.Profiles.DarkMatter(**dict(par, cdelta=5.0)).real(cosmo, r, 1e14, 0.8)
    assert np.allclose(dmm.real(cosmo, r, 1e14, 0.8), ref5, rtol=1e-13)
    assert not np.allclose(dmm.real(cosmo, r, 1e13, 0.8), bfg.Profiles.DarkMatter(**dict(par, cdelta=5.0)).real(cosmo, r, 1e13, 0.8), rtol=1e-3)
    assert np.allclose((dmm + dmm).real(cosmo, r, 1e14, 0.8), 2 * ref5, rtol=1e-13)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        full = bfg.Profiles.DarkMatterBaryon(**dict(par, cdelta=None), c_of_M=cm).real(cosmo, r, 1e14, 0.8)
        want = bfg.Profiles.DarkMatterBaryon(**dict(par, cdelta=5.0)).real(cosmo, r, 1e14, 0.8)
    # (cdelta = None also sets c = 1 in the gas parameters' zeta powers, Schneider19.py:175: equal here because the zetas default to 0)
    assert np.allclose(full, want, rtol=1e-12)


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.filterwarnings('ignore')
def test_setup_interpolator_end_to_end(gpu, T):
    """our profiles + GPU table builders reproduce the reference's displacement table"""
    par = _par(T)
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    DMO = bfg.Profiles.DarkMatter(**par)
    DMB = bfg.Profiles.CollisionlessMatter(**par) + bfg.Profiles.Stars(**par) + bfg.Profiles.Gas(**par)
    model = bfg.Profiles.Baryonification2D(DMO, DMB, cosmo, epsilon_max=20)
    model.setup_interpolator(z_min=0.2, z_max=0.3, N_samples_z=2, z_linear_sampling=True, M_min=1e13, M_max=1e15,
                             N_samples_M=4, R_min=1e-3, R_max=3e2, N_samples_R=60, verbose=False)
    assert model.raw_input_d.shape == (2, 4, 60) and model.p_keys == []
    assert np.allclose(model.raw_input_r_range, np.log(T['r'])) and np.allclose(model.raw_input_z_range, np.log(1 + T['z_range']))
    err = np.abs(model.raw_input_d - T['d_ref'])
    assert err.max() <= 2e-4 * np.abs(T['d_ref']).max()          # CollisionlessMatter differs at ~1e-6 (spline derivative)
    assert np.median(err[T['d_ref'] != 0] / np.abs(T['d_ref'][T['d_ref'] != 0])) < 1e-5
    d = model.displacement(np.geomspace(0.01, 5, 5), 1e14, 1 / 1.25)
    assert np.all(np.isfinite(d))
    m = model.get_masses(DMO, T['r'], T['M_range'], 1 / 1.2)
    assert np.nanmax(np.abs(m / T['M_dmo'][0] - 1)) < 1e-9
    assert model.get_masses(DMO, T['r'], 1e14, 1 / 1.2).shape == (60,)


@pytest.mark.gpu
@pytest.mark.filterwarnings('ignore')
def test_quickstart_flow_with_own_profiles(gpu, T):
    """README quickstart: profiles -> Baryonification2D table -> BaryonifyShell, and Pressure -> TabulatedProfile ->
    PaintProfilesShell, all inside this package"""
    par = _par(T)
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    cat = syn.make_catalog(400, z_lo=0.1, z_hi=0.2, logM_lo=13.0, logM_hi=15.0)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    Shell = bfg.utils.LightconeShell(map=syn.make_map(128), cosmo=syn.COSMO)
    model = bfg.Profiles.Baryonification2D(bfg.Profiles.DarkMatterOnly(**par), bfg.Profiles.DarkMatterBaryon(**par), cosmo, epsilon_max=10)
    model.setup_interpolator(z_min=cat['z'].min() * 0.999, z_max=cat['z'].max() * 1.001, N_samples_z=2,
                             M_min=cat['M'].min() * 0.999, M_max=cat['M'].max() * 1.001, N_samples_M=3,
                             R_min=1e-3, R_max=3e2, N_samples_R=80, verbose=False)
    new_map = bfg.Runners.BaryonifyShell(Catalog, Shell, 10, model, verbose=False).process()
    assert np.isclose(new_map.sum(), Shell.map.sum()) and np.abs(new_map - Shell.map).max() > 1e-3
    press = bfg.utils.TabulatedProfile(bfg.Profiles.Pressure(**par), cosmo)
    press.setup_interpolator(z_min=cat['z'].min() * 0.999, z_max=cat['z'].max() * 1.001, N_samples_z=2,
                             M_min=cat['M'].min() * 0.999, M_max=cat['M'].max() * 1.001, N_samples_M=3,
                             R_min=1e-3, R_max=50, N_samples_R=60, verbose=False)
    y = bfg.Runners.PaintProfilesShell(Catalog, bfg.utils.LightconeShell(map=np.zeros(12 * 128 ** 2), cosmo=syn.COSMO), 5, press,
                                       verbose=False).process()
    assert y.min() >= 0 and y.max() > 0 and np.isfinite(y).all()


# ------------------------------------------------------------------------------------------------- thermodynamic scalings
def test_nonthermal_fraction_matches_reference(P, T):
    """pure host formula (Thermodynamic.py:347-368): runs without a GPU"""
    par = _par(T)
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    f = bfg.Profiles.NonThermalFrac(**par)
    for key, a in (('NonThermalFrac', float(P['a'])), ('NonThermalFrac_z0', 1.0)):
        got = f.real(cosmo, P['thermo_r'], P['thermo_M'], a)
        assert got.shape == P[key].shape and np.abs(got - P[key]).max() < 1e-13
    assert np.all((P['NonThermalFrac'] >= 0) & (P['NonThermalFrac'] <= 1))
    sz = bfg.Profiles.ThermalSZ(pressure=f, **par)
    assert np.all(sz.real(cosmo, P['thermo_r'], P['thermo_M'], 0.8) == -99)                 # the reference's sentinel (:757-767)


@pytest.mark.gpu
@pytest.mark.filterwarnings('ignore')
def test_thermal_sz_and_electron_pressure_match_reference(gpu, P, T):
    """Pressure integrals + line-of-sight projection on the GPU, scalings on the host (quickstart tSZ flow, README.md:84-87
    without the pixel-window convolution)"""
    par = _par(T)
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    a = float(P['a'])
    gas = bfg.Profiles.Gas(**par)
    tot = bfg.Profiles.CollisionlessMatter(**par) + bfg.Profiles.Stars(**par) + bfg.Profiles.Gas(**par)
    pe = bfg.Profiles.ElectronPressure(gas=gas, darkmatterbaryon=tot, **par).real(cosmo, P['thermo_r'], P['thermo_M'], a)
    assert np.abs(pe / P['ElectronPressure'] - 1).max() < 1e-5                              # CollisionlessMatter port ~1e-6
    pth = bfg.Profiles.Pressure(gas=gas, darkmatterbaryon=tot, **par)
    y = bfg.Profiles.ThermalSZ(pressure=pth * (1 - bfg.Profiles.NonThermalFrac(**par)), **par).projected(cosmo, P['thermo_r'], P['thermo_M'], a)
    ref = P['ThermalSZ']
    nz = ref != 0
    assert y.shape == ref.shape and np.abs(y[nz] / ref[nz] - 1).max() < 1e-5 and np.all(y[~nz] == 0)
    assert np.ndim(bfg.Profiles.ThermalSZ(pressure=pth, **par).projected(cosmo, 0.5, 1e14, a)) == 0
