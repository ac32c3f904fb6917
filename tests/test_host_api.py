"""Host logic of the drop-in layer (no GPU): containers, runner constructor contract, model protocol,
pickling -- mirrored from BaryonForge/utils/io.py and Runners/HealpixRunner.py:149-165."""
import pickle
import warnings

import numpy as np
import pytest

import baryonification_amd as bfg
from baryonification_amd import synthetic as syn
from baryonification_amd.Runners._model import build_model
from helpers import load_golden, product_runner


def test_package_surface_matches_reference_names():
    for name in ('BaryonifyShell', 'PaintProfilesShell', 'DefaultRunner'):
        assert hasattr(bfg.Runners, name) and hasattr(bfg, name)
    for name in ('HaloLightConeCatalog', 'LightconeShell', 'TabulatedProfile', 'ParamTabulatedProfile',
                 'SplitJoinParallel', 'SimpleParallel'):
        assert hasattr(bfg.utils, name)
    assert hasattr(bfg.Profiles, 'Baryonification2D')


def test_catalog_fields_pole_clip_and_slicing():
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        cat = bfg.utils.HaloLightConeCatalog(ra=np.array([1., 2., 3.]), dec=np.array([90, -90, 10]), M=np.ones(3) * 1e14,
                                             z=np.array([.1, .2, .3]), cosmo=syn.COSMO, cdelta=np.array([4., 5., 6.]))
        assert any('poles' in str(x.message) for x in w)
    assert cat.cat.dtype.names == ('M', 'z', 'ra', 'dec', 'cdelta')
    assert cat.cat['dec'][0] == 90 - 1e-8 and cat.cat['dec'][1] == -90 + 1e-8       # io.py:65-68
    sub = cat[1:]
    assert sub.cat.size == 2 and sub.cat['cdelta'][0] == 5. and sub.cosmology is syn.COSMO
    with pytest.raises(ValueError):
        bfg.utils.HaloLightConeCatalog(ra=[1.], dec=[2.], M=[1e14], z=[.1], cosmo={'Omega_m': .3})


def test_shell_nside_and_errors():
    sh = bfg.utils.LightconeShell(map=np.zeros(12 * 16 * 16), cosmo=syn.COSMO)
    assert sh.NSIDE == 16 and sh.data is sh.map
    with pytest.raises(ValueError):
        bfg.utils.LightconeShell(cosmo=syn.COSMO)
    with pytest.raises(ValueError):
        bfg.utils.LightconeShell(map=np.zeros(100), cosmo=syn.COSMO)
    with pytest.raises(ValueError):
        bfg.utils.LightconeShell(map=np.zeros(12), cosmo={'h': .7})


def test_runner_contract():
    g = load_golden('lowz_baryonify')
    r = product_runner(g)
    for attr in ('HaloLightConeCatalog', 'LightconeShell', 'cosmo', 'model', 'mass_def', 'epsilon_max', 'use_ellipticity', 'verbose'):
        assert hasattr(r, attr)                                   # Parallelize.py:237-243 reads these
    assert r.cosmo is r.HaloLightConeCatalog.cosmology
    # SplitJoinParallel re-instantiates positionally (Parallelize.py:271)
    r2 = type(r)(r.HaloLightConeCatalog, r.LightconeShell, r.epsilon_max, r.model, r.use_ellipticity, r.mass_def, verbose=False)
    assert r2.epsilon_max == r.epsilon_max
    with pytest.raises(NotImplementedError):
        bfg.Runners.BaryonifyShell(r.HaloLightConeCatalog, r.LightconeShell, 10, r.model, use_ellipticity=True)
    pickle.loads(pickle.dumps(r))                                 # joblib/loky requirement


def test_model_protocol_errors():
    g = load_golden('lowz_baryonify')
    r = product_runner(g)
    r.model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO))
    with pytest.raises(NameError):                                # BaryonCorrection.py:425-426
        r.process()
    with pytest.raises(NameError):
        r.model.displacement(1.0, 1e14, 0.9)
    p = product_runner(load_golden('lowz_paint'))
    p.model = None
    with pytest.raises(AssertionError):                           # HealpixRunner.py:415
        p.process()
    # property keys demand a ParamTabulatedProfile (HealpixRunner.py:284-287)
    r = product_runner(g)
    r.model.p_keys = ['cdelta']
    with pytest.raises(AssertionError):
        r.process()


def test_build_model_struct_contents():
    g = load_golden('rdelta_baryonify')
    r = product_runner(g)
    m, p_keys, keep = build_model(r, 'displacement')
    assert p_keys == [] and m.table.ndim == 3 and m.table.rdelta_sampling == 1 and m.table.log_values == 0
    assert [m.table.n[i] for i in range(3)] == [g['tab_z'].size, g['tab_M'].size, g['tab_r'].size]
    assert m.eps_runner == g['eps_runner'] and m.table.eps_model == g['eps_model']
    assert m.cosmo_runner.Omega_m == g['cosmo_runner']['Omega_m'] and m.cosmo_model.Omega_m == g['cosmo_model']['Omega_m']
    assert np.allclose(keep[0][0], np.log(1 + g['tab_z']))


def test_host_readouts_match_oracle_rgi():
    """model.displacement / profile.projected (host convenience read-outs) follow the reference semantics"""
    from oracle import oracle as O
    g = load_golden('lowz_baryonify')
    r = product_runner(g)
    tab = O.Table([np.log(1 + g['tab_z']), np.log(g['tab_M']), np.log(g['tab_r'])], g['tab_values'])
    a, M = 1 / 1.03, 3e14
    rr = np.geomspace(5e-4, 50, 40)
    d = r.model.displacement(rr, M, a)
    R = O.Background.from_dict(g['cosmo_model']).get_radius(M, a) / a
    exp = np.array([tab.eval([np.log(1 / a), np.log(M), np.log(x)]) if x < g['eps_model'] * R else 0.0 for x in rr])
    assert np.array_equal(np.isnan(d), np.isnan(exp)) and np.isnan(d).any()
    assert np.nanmax(np.abs(d - exp)) < 1e-15
    assert np.ndim(r.model.displacement(1.0, M, a)) == 0


# ------------------------------------------------------------------------------------------ regular-grid path

def test_grid_containers_match_reference_layout():
    rng = np.random.default_rng(1)
    x, y, z, M = rng.uniform(0, 50, 5), rng.uniform(0, 50, 5), rng.uniform(0, 50, 5), 10 ** rng.uniform(12, 15, 5)
    A = rng.normal(size=(5, 2))
    cat = bfg.utils.HaloNDCatalog(x=x, y=y, M=M, redshift=0.2, cosmo=syn.COSMO, z=z, q_ell=np.full(5, 0.7), A_ell=A)
    assert cat.cat.dtype.names == ('M', 'x', 'y', 'z', 'q_ell', 'A_ell')
    assert cat.cat.dtype['M'] == np.dtype('>f4') and cat.cat['A_ell'].shape == (5, 2)      # io.py:205-206
    assert np.array_equal(cat.cat['M'], M.astype(np.float32)) and cat.redshift == 0.2
    assert cat.cosmology is syn.COSMO and cat[1:3].cat.size == 2 and cat[1:3].redshift == 0.2
    cat2 = bfg.utils.HaloNDCatalog(x=x, y=y, M=M, redshift=0.0, cosmo=syn.COSMO)
    assert np.all(cat2.cat['z'] == 0)
    with pytest.raises(ValueError):
        bfg.utils.HaloNDCatalog(x=x, y=y, M=M, redshift=0.0, cosmo={'Omega_m': 0.3})

    bins = (np.arange(16) + 0.5) * 2.0
    g2 = bfg.utils.GriddedMap(map=np.zeros((16, 16)), redshift=0.1, bins=bins, cosmo=syn.COSMO)
    assert g2.is2D and g2.Npix == 16 and g2.res == 2.0 and g2.inds.shape == (16, 16) and len(g2.grid) == 2
    g3 = bfg.utils.GriddedMap(map=np.zeros((16, 16, 16)), redshift=0.1, bins=bins, cosmo=syn.COSMO)
    assert not g3.is2D and g3.inds[1, 2, 3] == (1 * 16 + 2) * 16 + 3 and g3.data is g3.map
    with pytest.raises(AssertionError):
        bfg.utils.GriddedMap(map=np.zeros((16, 8)), redshift=0.1, bins=bins, cosmo=syn.COSMO)

    snap = bfg.utils.ParticleSnapshot(x=x, y=y, z=None, M=np.ones(5), L=50.0, redshift=0.0, cosmo=syn.COSMO)
    assert snap.is2D and snap.cat.dtype['x'] == np.float64 and np.all(snap.cat['z'] == 0)


def test_grid_runner_contract():
    rng = np.random.default_rng(2)
    bins = (np.arange(16) + 0.5) * 2.0
    cat = bfg.utils.HaloNDCatalog(x=rng.uniform(0, 32, 4), y=rng.uniform(0, 32, 4), M=np.full(4, 1e14), redshift=0.2, cosmo=syn.COSMO)
    gmap = bfg.utils.GriddedMap(map=np.ones((16, 16)), redshift=0.2, bins=bins, cosmo=syn.COSMO)
    r = bfg.Runners.BaryonifyGrid(cat, gmap, 5.0, None)
    for name in ('HaloNDCatalog', 'GriddedMap', 'cosmo', 'model', 'epsilon_max', 'mass_def', 'verbose', 'use_ellipticity'):
        assert hasattr(r, name)
    assert r.cosmo is cat.cosmology and r._runner_cosmo()['w0'] == -1.0          # Map2DRunner.py:456-459 drops w0
    assert pickle.loads(pickle.dumps(r)).epsilon_max == 5.0
    assert np.array_equal(r.pick_indices(1, 3, 16), [14, 15, 0, 1, 2, 3])
    with pytest.raises(AssertionError, match='q_ell'):
        bfg.Runners.BaryonifyGrid(cat, gmap, 5.0, None, use_ellipticity=True)
    with pytest.raises(AssertionError, match='must provide a model'):
        bfg.Runners.PaintProfilesGrid(cat, gmap, 5.0, None).process()
    # build_Rmat: the shear matrix has unit determinant and reduces to the identity for q -> 1
    Rm = r.build_Rmat(np.array([0.6, 0.8]), 0.5)
    assert np.isclose(np.linalg.det(Rm), 1.0) and np.allclose(Rm, Rm.T)
    assert np.allclose(r.build_Rmat(np.array([1.0, 0.0]), 1.0 - 1e-9), np.eye(2), atol=1e-8)
    with pytest.raises(NotImplementedError):
        r.build_Rmat(np.array([1.0, 0.0, 0.0]), 0.5)
    # argument checks of the regrid functions happen before any device work
    with pytest.raises(ValueError):
        bfg.Runners.regrid_pixels_2D(np.zeros((8, 8), dtype=np.float32), np.zeros((3, 2)), np.zeros(3))
    with pytest.raises(ValueError):
        bfg.Runners.regrid_pixels_3D(np.zeros((8, 8, 8)), np.zeros((3, 2)), np.zeros(3))


@pytest.mark.gpu
def test_warm_process_call_allocates_nothing(gpu):
    """the one-shot calls re-use a cached plan and pooled device buffers: after the first call with a model, further calls
    (same model, same or smaller catalog) perform no hipMalloc; a different model builds its own plan; results are unchanged"""
    from baryonification_amd import _lib
    from helpers import load_golden, product_runner
    L = _lib.load()
    L.bfgx_cache_clear()
    g = load_golden('lowz_baryonify')
    r = product_runner(g, acc_f64=True)
    out1 = r.process()
    n1 = L.bfgx_debug_alloc_count()
    out2 = r.process()
    out3 = r.process()
    assert L.bfgx_debug_alloc_count() == n1                      # warm calls: no device allocation at all
    assert np.array_equal(out1, out2) or np.abs(out1 - out2).max() <= 1e-13 * np.abs(out1).max()      # LDS add order only
    assert np.abs(out3 - g['expected']).max() <= 1e-10 * np.abs(g['expected']).max()
    # the returned maps are independent arrays (page-locked, pooled on release), not views of one buffer
    assert out1.ctypes.data != out2.ctypes.data and out2.ctypes.data != out3.ctypes.data
    # another model -> another plan (allocations), and both stay cached
    gp = load_golden('lowz_paint')
    rp = product_runner(gp, acc_f64=True)
    p1 = rp.process()
    n2 = L.bfgx_debug_alloc_count()
    assert n2 > n1
    p2 = rp.process()
    out4 = r.process()
    assert L.bfgx_debug_alloc_count() == n2
    assert np.abs(p2 - gp['expected']).max() <= 1e-10 * np.abs(gp['expected']).max() and np.abs(p1 - p2).max() <= 1e-13 * np.abs(p1).max()
    assert np.abs(out4 - g['expected']).max() <= 1e-10 * np.abs(g['expected']).max()
    # a released map returns its page-locked buffer to the pool
    addr = out1.ctypes.data
    del out1
    import gc
    gc.collect()
    out5 = r.process()
    assert out5.ctypes.data == addr
    L.bfgx_cache_clear()
    _lib.pinned_pool_clear()


@pytest.mark.gpu
def test_in_place_catalog_edit_between_calls_is_seen(gpu):
    """the contiguous catalog columns are cached on the catalog object between process() calls; an in-place edit of ONE halo that a
    sampled fingerprint would miss -- and a swap of two halos that leaves every column sum unchanged -- must be seen, as the reference,
    which re-reads `cat` on every call, sees it"""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    from oracle import oracle as O
    nside, N = 64, 20_000
    cat = syn.make_catalog(N, seed=3, logM_lo=13.0, logM_hi=14.5)
    z, M, r = syn.table_grid(cat, pad=1e-3)
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=10.0)
    d = syn.displacement_table(z, M, r)
    model.set_table(z, M, r, d)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    hmap = syn.make_map(nside)
    runner = bfg.Runners.BaryonifyShell(Catalog, bfg.utils.LightconeShell(map=hmap, cosmo=syn.COSMO), 10.0, model, verbose=False)
    runner.acc_f64 = True
    bg = O.Background.from_dict(syn.COSMO)
    tab = O.Table([np.log(1 + z), np.log(M), np.log(r)], d, False, 10.0)

    def oracle():
        used = {k: np.array(Catalog.cat[k]) for k in ('M', 'z', 'ra', 'dec')}
        return O.baryonify_shell(nside, hmap, used, tab, 10.0, bg)

    out0 = runner.process()
    assert np.abs(out0 - oracle()).max() <= 1e-10 * hmap.max() and hasattr(Catalog, '_bfgx_columns')
    j = 7777                                                    # not a multiple of any sampling stride, not the last element
    Catalog.cat['M'][j] *= 1.05                                 # stays inside the table (pad)
    out1 = runner.process()
    ora1 = oracle()
    assert np.abs(out1 - ora1).max() <= 1e-10 * hmap.max() and np.abs(out1 - out0).max() > 0
    a, b = 123, 4567                                            # swap two halos' positions: every column sum is unchanged
    for k in ('ra', 'dec'):
        Catalog.cat[k][a], Catalog.cat[k][b] = Catalog.cat[k][b], Catalog.cat[k][a]
    out2 = runner.process()
    assert np.abs(out2 - oracle()).max() <= 1e-10 * hmap.max() and np.abs(out2 - out1).max() > 0
    assert np.array_equal(runner.process(), out2) or np.abs(runner.process() - out2).max() <= 1e-13 * hmap.max()      # unchanged catalog: cached


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['subpixel', 'polar', 'several_rings', 'many_rings'])
def test_host_entry_in_band_ranges_equals_one_pass(gpu, case, monkeypatch):
    """bfgx_baryonify_shell moves large maps in band ranges (a range is regridded and sent back while the next ones are still arriving).
    Its result against the one-pass route of the same entry (BFGX_NO_PIPELINE) and the oracle: sub-pixel displacements (one ring of
    apron), halos on the poles (deposits the gathering regrid lists and the host adds), displacements of a few rings (wider aprons) and
    of many rings (falls back to the one-pass route by itself)"""
    import baryonification_amd as bfg
    from baryonification_amd import _lib, synthetic as syn
    from oracle import oracle as O
    nside, N = 512, 30_000
    cat = syn.make_catalog(N, seed=11, logM_lo=13.0, logM_hi=15.0)
    if case == 'polar':
        cat['dec'][:400] = np.where(np.arange(400) % 2 == 0, 89.97, -89.95) + np.linspace(0, 0.02, 400)
    z, M, r = syn.table_grid(cat, pad=1e-3)
    scale = {'subpixel': 1.0, 'polar': 1.0, 'several_rings': 60.0, 'many_rings': 400.0}[case]
    d = scale * syn.displacement_table(z, M, r)
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=10.0)
    model.set_table(z, M, r, d)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    hmap = syn.make_map(nside)
    hmap[::23] = 0.0
    runner = bfg.Runners.BaryonifyShell(Catalog, bfg.utils.LightconeShell(map=hmap, cosmo=syn.COSMO), 10.0, model, verbose=False)
    runner.acc_f64 = True
    monkeypatch.setenv('BFGX_PIPE_CHUNKS', '5')               # (a 25 MB map would go in one piece)
    piped = runner.process()
    st = dict(runner.last_stats)
    monkeypatch.setenv('BFGX_NO_PIPELINE', '1')
    whole = runner.process()
    monkeypatch.delenv('BFGX_NO_PIPELINE')
    assert np.isfinite(piped).all() and np.isclose(piped.sum(), hmap.sum(), rtol=1e-12) and np.isclose(st['sum_out'], st['sum_in'], rtol=1e-12)
    assert np.abs(piped - hmap).max() > 0
    assert np.abs(piped - whole).max() <= 1e-12 * hmap.max()
    if case in ('subpixel', 'polar'):
        used = {k: np.array(Catalog.cat[k]) for k in ('M', 'z', 'ra', 'dec')}
        ora = O.baryonify_shell(nside, hmap, used, O.Table([np.log(1 + z), np.log(M), np.log(r)], d, False, 10.0), 10.0, O.Background.from_dict(syn.COSMO))
        assert np.abs(piped - ora).max() <= 1e-10 * hmap.max()
    # fp32 pair math through the same route
    runner.acc_f64 = False
    p32 = runner.process()
    assert np.abs(p32 - piped).max() <= 1e-5 * max(1.0, scale) * hmap.mean() and np.isclose(p32.sum(), hmap.sum(), rtol=1e-9)
    # the parity-grade mode (split fp32 pix_offsets: the band-range route hands the regrid both halves) and the default, where the plan
    # picks from the table: both within SURVEY 8(d)'s 1e-6 mean(map) of fp64 throughout whatever the displacement
    runner.acc_f64 = 'parity'
    ppar = runner.process()
    assert np.abs(ppar - piped).max() <= 1e-8 * hmap.mean() and np.isclose(ppar.sum(), hmap.sum(), rtol=1e-12)
    runner.acc_f64 = None
    pdef = runner.process()
    assert np.abs(pdef - piped).max() <= 1e-6 * hmap.mean() and np.isclose(pdef.sum(), hmap.sum(), rtol=1e-9)


@pytest.mark.gpu
def test_catalog_columns_stay_on_the_device_between_calls(gpu):
    """bfgx_opts.catalog_token: a process() call on an unchanged catalog copies no catalog column; an edited catalog is copied again"""
    import baryonification_amd as bfg
    from baryonification_amd import _lib, synthetic as syn
    nside, N = 64, 5000
    cat = syn.make_catalog(N, seed=5, logM_lo=13.0, logM_hi=14.5)
    z, M, r = syn.table_grid(cat, pad=1e-3)
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=10.0)
    model.set_table(z, M, r, syn.displacement_table(z, M, r))
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    runner = bfg.Runners.BaryonifyShell(Catalog, bfg.utils.LightconeShell(map=syn.make_map(nside), cosmo=syn.COSMO), 10.0, model, verbose=False)
    runner.acc_f64 = True
    lib = _lib.load()
    out0 = runner.process()
    n0 = lib.bfgx_debug_catalog_uploads()
    out1 = runner.process()
    close = lambda a, b: np.abs(a - b).max() <= 1e-11             # (the order of the LDS adds is not fixed)
    assert lib.bfgx_debug_catalog_uploads() == n0 and close(out0, out1)
    j = int(np.argmax(Catalog.cat['M']))
    Catalog.cat['M'][j] *= 0.8
    out2 = runner.process()
    assert lib.bfgx_debug_catalog_uploads() == n0 + 1
    assert np.abs(out2 - out1).max() > 1e-6
    # the raw entry without a token copies every time
    c, keep, _ = runner._catalog([])
    from baryonification_amd.Runners._model import build_model
    m, _, mkeep = build_model(runner, 'displacement')
    import ctypes as C
    o = _lib.bfgx_opts(0, 1, 1, 1, 1, 0)
    res = _lib.pinned_empty(12 * nside * nside)
    hmap = _lib.f8(runner.LightconeShell.map)
    for _ in range(2):
        _lib.check(lib.bfgx_baryonify_shell(C.byref(c), C.byref(m), nside, hmap.ctypes.data, res.ctypes.data, C.byref(o), None))
    assert lib.bfgx_debug_catalog_uploads() == n0 + 3 and close(res, out2)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [True, 'mixed'])
def test_paint_host_entry_in_band_ranges_equals_one_pass(gpu, mode, monkeypatch):
    """bfgx_paint_shell paints large fp64 maps band range by band range, copying each back while the next is painted: against the one-pass
    route of the same entry and the oracle, with halos on the poles (the generic kernel's tiles) in the catalog"""
    from oracle import oracle as O
    nside, N = 512, 20_000
    cat = syn.make_catalog(N, seed=12, logM_lo=13.0, logM_hi=15.0)
    cat['dec'][:300] = np.where(np.arange(300) % 2 == 0, 89.9, -89.8) + np.linspace(0, 0.05, 300)
    z, M, r = syn.table_grid(cat, pad=1e-3)
    P = syn.paint_table(z, M, r)
    prof = bfg.utils.TabulatedProfile(None, bfg.utils.Cosmology.from_dict(syn.COSMO))
    prof.set_table(z, M, r, P)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    runner = bfg.Runners.PaintProfilesShell(Catalog, bfg.utils.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=syn.COSMO), 10.0, prof, verbose=False)
    runner.acc_f64 = mode
    monkeypatch.setenv('BFGX_PIPE_CHUNKS', '6')
    piped = runner.process()
    monkeypatch.setenv('BFGX_NO_PIPELINE', '1')
    whole = runner.process()
    assert piped.max() > 0 and np.isfinite(piped).all()
    assert np.abs(piped - whole).max() <= 1e-12 * np.abs(whole).max()
    used = {k: np.array(Catalog.cat[k]) for k in ('M', 'z', 'ra', 'dec')}
    ora = O.paint_shell(nside, used, O.Table([np.log(1 + z), np.log(M), np.log(r)], np.log(P)), 10.0, O.Background.from_dict(syn.COSMO))
    assert np.abs(piped - ora).max() <= (1e-10 if mode is True else 5e-5) * np.abs(ora).max()


def test_more_than_four_property_axes_are_refused_loudly():
    """Tabulate.py:524-561 / BaryonCorrection.py:205-221 accept any number of `other_params`; libbfgx reads out up to four property
    axes (BFGX_MAX_EXTRA: 64 corner rows per halo; goldens param3_paint / param4_paint).  A fifth must raise, naming the limit -- never a
    silent truncation"""
    import pytest
    from baryonification_amd import _lib
    from baryonification_amd.Runners import _model as RM
    ax = [np.linspace(0.1, 0.2, 2), np.linspace(28, 34, 3), np.linspace(-3, 3, 4)] + [np.linspace(0, 1, 2) for _ in range(5)]
    with pytest.raises(NotImplementedError, match="BFGX_MAX_EXTRA = 4"):
        _lib.make_table(ax, np.zeros([a.size for a in ax]))
    _lib.make_table(ax[:7], np.zeros([a.size for a in ax[:7]]))          # four property axes are fine

    class FiveParams(object):                                            # a duck-typed ParamTabulatedProfile with five property axes
        p_keys = ['p0', 'p1', 'p2', 'p3', 'p4']
        epsilon_max, cosmo, mass_def = 10.0, None, None
        raw_input_z_range, raw_input_M_range, raw_input_r_range = ax[0], ax[1], ax[2]
        raw_input_p0_range, raw_input_p1_range, raw_input_p2_range, raw_input_p3_range, raw_input_p4_range = ax[3:8]
        raw_input_d = np.zeros([a.size for a in ax])

    class R(object):
        model = FiveParams()
    R.model.__dict__['p_keys'] = FiveParams.p_keys
    with pytest.raises(NotImplementedError, match="5 property axes"):
        RM.build_model(R(), 'displacement')


def test_catalog_fingerprint_without_xxhash(monkeypatch):
    """xxhash is optional: without it the fingerprint comes from hashlib.blake2b, and still sees a one-halo edit"""
    import sys
    from baryonification_amd.Runners import HealpixRunner as HR
    cat = np.zeros(1000, dtype=[('M', 'f8'), ('z', 'f8'), ('ra', 'f8'), ('dec', 'f8')])
    cat['M'] = np.arange(1000) + 1.0
    names = ['M', 'z', 'ra', 'dec']
    with_xx = HR._catalog_fingerprint(cat, names)
    monkeypatch.setitem(sys.modules, 'xxhash', None)               # `import xxhash` now raises ImportError
    f0 = HR._catalog_fingerprint(cat, names)
    assert f0 is not None and f0[:3] == (1000, str(cat.dtype), tuple(names)) and f0 == HR._catalog_fingerprint(cat, names)
    assert with_xx[:3] == f0[:3]
    cat['M'][777] *= 1.0000001
    assert HR._catalog_fingerprint(cat, names) != f0
    assert HR._catalog_fingerprint(cat[:0], names)[3] == 0


@pytest.mark.gpu
def test_round4_fault_sequence_small_arrays_are_never_page_locked_in_place(gpu, monkeypatch):
    """The call sequence in front of the two GPU memory faults of round 4 (gpurun_out/r04_t20.log, r04_t27.log), replayed in
    one process: (1) the streamed route of bfgx_baryonify_shell forced onto a small map (BFGX_PIPE_CHUNKS), which then page-locked the caller's
    map_in / map_out IN PLACE; (2) ParticleSnapshot.make_map on a 5-particle snapshot, whose records entry page-locked a 160-byte array;
    (3) the ordinary one-shot calls that faulted (bfgx_baryonify_shell on a 393 KB map, bfgx_baryonify_grid on a 2 MB map): plain pageable
    copies from heap arrays that share pages with what (1) / (2) had registered and unregistered.  Small arrays sit in the process heap
    beside other live objects; hipHostRegister works on whole pages, so two such registrations (or one and the runtime's own pinning of a
    pageable copy) overlap on the shared boundary pages and unregistering one takes the other's mapping with it.  The invariant that
    removes it: the one-shot entries page-lock a caller's array in place only when it owns its pages (>= 32 MiB: a mapping of its own)."""
    import ctypes as C
    import baryonification_amd as bfg
    from baryonification_amd import _lib, synthetic as syn
    import helpers as H
    lib = _lib.load()

    def spans():
        a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        lib.bfgx_debug_host_spans(C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value
    pinned0, staged0, _ = spans()
    # (1) the streamed shell route on a small map
    nside, N = 128, 4000
    cat = syn.make_catalog(N, seed=3, logM_lo=13.0, logM_hi=14.5)
    z, M, r = syn.table_grid(cat, pad=1e-3)
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=10.0)
    model.set_table(z, M, r, syn.displacement_table(z, M, r))
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    hmap = syn.make_map(nside)
    runner = bfg.Runners.BaryonifyShell(Catalog, bfg.utils.LightconeShell(map=hmap, cosmo=syn.COSMO), 10.0, model, verbose=False)
    runner.acc_f64 = True
    monkeypatch.setenv('BFGX_PIPE_CHUNKS', '4')
    piped = runner.process()
    monkeypatch.delenv('BFGX_PIPE_CHUNKS')
    pinned1, staged1, smallest1 = spans()
    assert pinned1 == pinned0, "a 1.5 MB map was page-locked in place"
    assert staged1 >= staged0 + 1                                  # (it went through a page-locked staging buffer of the library's)
    # (2) the records entry on a 5-particle snapshot
    cos = syn.COSMO
    Snap = bfg.utils.ParticleSnapshot(x=np.array([1.0, 2.0, 3.0, 4.0, 5.0]), y=np.array([1.0, 2.0, 3.0, 4.0, 5.0]), z=np.array([1.0, 2.0, 3.0, 4.0, 5.0]),
                                      M=np.ones(5), L=10.0, redshift=0.0, cosmo=cos)
    m5 = Snap.make_map(8)
    assert m5.sum() == 5.0
    # (3) the two ordinary calls that faulted: results must be right, and nothing small may have been page-locked in place on the way
    whole = runner.process()
    assert np.abs(whole - piped).max() <= 1e-12 * hmap.max() and np.isclose(whole.sum(), hmap.sum(), rtol=1e-12)
    c = H.load_grid_golden('grid3d_baryonify')
    out = H.grid_product_runner(c).process()
    assert np.abs(out - c['expected']).max() <= 1e-10 * np.abs(c['expected']).max()
    pinned2, _, smallest2 = spans()
    assert pinned2 == pinned0
    assert smallest2 < 0 or smallest2 >= (32 << 20)                # whatever this process page-locked in place before owned its pages


def test_which_route_a_callable_model_takes():
    """Runners/_model.py: a plain Python callable (no table, no setup_interpolator) is called per halo -- the reference's loop,
    HealpixRunner.py:321, :441 -- on catalogs of up to EXACT_MAX_HALOS halos or when model.bfgx_exact is set; table classes never are"""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    from baryonification_amd.Runners import _model as RM

    class Plain(object):
        def displacement(self, r, M, a):
            return 0 * r

        def projected(self, cosmo, r, M, a):
            return 0 * r

    cat = syn.make_catalog(50)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    Shell = bfg.utils.LightconeShell(map=np.ones(12 * 16 * 16), cosmo=syn.COSMO)
    m = Plain()
    r = bfg.Runners.BaryonifyShell(Catalog, Shell, 5.0, m, verbose=False)
    assert RM.plain_callable(m, 'displacement') and RM.plain_callable(m, 'projected') and not RM.plain_callable(m, 'real')
    assert RM.wants_exact(r, 'displacement') and RM.wants_exact(r, 'projected')
    m.bfgx_exact = False
    assert not RM.wants_exact(r, 'displacement')
    m.bfgx_exact = True
    old = RM.EXACT_MAX_HALOS
    try:
        RM.EXACT_MAX_HALOS = 10
        assert RM.wants_exact(r, 'displacement')                       # asked for: whatever the size
        del m.bfgx_exact
        assert not RM.wants_exact(r, 'displacement')                   # by size: 50 halos > 10
    finally:
        RM.EXACT_MAX_HALOS = old
    tab = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO))
    assert not RM.plain_callable(tab, 'displacement') and not RM.plain_callable(None, 'projected') and not RM.plain_callable(object(), 'displacement')
