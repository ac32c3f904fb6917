"""Parity tests proper: the HIP path, called through the drop-in runners (-> ctypes -> C ABI), against
(a) the reference's own outputs (golden fixtures) and (b) the CPU oracle on the same inputs.

Stated tolerances (north_star: "within a stated fp64->fp32 tolerance"):
  * geometry and table read-out are fp64 on the device; with fp64 accumulators the only differences
    are libm ulps and atomic summation order          ->  |d| <= 1e-10 * max|map|
  * default BaryonifyShell path: fp32 atomics for pix_offsets (values ~1e-5 of a unit vector), fp64 regrid
                                                        ->  |d| <= 1e-6 * mean(map) per pixel
  * PaintProfilesShell with fp32 accumulators         ->  |d| <= 1e-5 * max|map|
"""
import numpy as np
import pytest

from helpers import GOLDEN_CASES, load_golden, oracle_run, product_runner

pytestmark = pytest.mark.gpu


ALGOS = [1, 0]       # 1 = tile-owned LDS accumulation (default), 0 = per-halo global atomics
# tables with property axes (p_keys) exist on the tiled path only: those fixtures are not paired with algo 0
CASES = [(n, a) for a in ALGOS for n in GOLDEN_CASES if not (a == 0 and n.startswith('param'))]


def run(g, acc_f64=None, algo=1):
    assert not (algo == 0 and g['p_keys']), "tables with property axes (p_keys) run on the tiled path only"
    r = product_runner(g, acc_f64=acc_f64)
    r.algo = algo
    return r.process()


@pytest.mark.parametrize('name,algo', CASES)
def test_hip_vs_reference_golden_f64_accumulators(gpu, name, algo):
    g = load_golden(name)
    out = run(g, True, algo)
    exp = g['expected']
    assert out.dtype == np.float64 and out.shape == exp.shape
    assert np.abs(out - exp).max() <= 1e-10 * np.abs(exp).max()


@pytest.mark.parametrize('name,algo', CASES)
def test_hip_vs_reference_golden_f32_accumulators(gpu, name, algo):
    g = load_golden(name)
    out = run(g, False, algo)
    exp = g['expected']
    if g['kind'] == 'baryonify':
        assert np.abs(out - exp).max() <= 1e-6 * exp.mean()
        assert np.isclose(out.sum(), g['map_in'].sum())
    else:
        assert np.abs(out - exp).max() <= 1e-5 * np.abs(exp).max()


@pytest.mark.parametrize('name', [n for n in GOLDEN_CASES if n.endswith('baryonify')])
def test_hip_vs_reference_golden_parity_grade_and_default(gpu, name):
    """BFGX_ACC_PARITY (fp64 pair math with 1e-11 elementary functions, split fp32 pix_offsets, fp64 regrid geometry; on these small shells
    the barrier-per-tile form of the fast kernel and the lean regrid) and the DEFAULT (BFGX_ACC_AUTO: the plan picks from the table) against
    the reference's own output: SURVEY 8(d)'s 1e-6 mean(map) for the default whatever the table, 1e-8 for the parity-grade mode"""
    g = load_golden(name)
    exp = g['expected']
    out = run(g, 'parity', 1)
    assert out.dtype == np.float64 and np.abs(out - exp).max() <= 1e-8 * exp.mean()
    assert np.isclose(out.sum(), g['map_in'].sum(), rtol=1e-12)
    out = run(g, None, 1)
    assert np.abs(out - exp).max() <= 1e-6 * exp.mean() and np.isclose(out.sum(), g['map_in'].sum())


@pytest.mark.parametrize('name,algo', CASES)
def test_hip_vs_oracle(gpu, name, algo):
    g = load_golden(name)
    out = run(g, True, algo)
    ora = oracle_run(g)
    assert np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max()


def test_default_accumulators(gpu):
    """defaults: BaryonifyShell picks its precision from the table (fp32 pair math + fp32 pix_offsets while the table moves a pixel by less than
    0.1 pixel sides, the parity-grade mode beyond: 1e-6 mean(map) either way); PaintProfilesShell f32 pair math accumulated in f64 into the f64
    map (stated tolerance 5e-5 of the pixel's value); acc_f64 = True is fp64 throughout"""
    g = load_golden('lowz_baryonify')
    r = product_runner(g)
    out = r.process()
    assert np.abs(out - g['expected']).max() <= 1e-6 * g['expected'].mean()
    g = load_golden('lowz_paint')
    exp = g['expected']
    out = product_runner(g).process()
    assert out.dtype == np.float64 and np.all(np.abs(out - exp) <= 5e-5 * np.abs(exp) + 1e-12 * np.abs(exp).max())
    assert np.abs(out - exp).max() > 1e-10 * np.abs(exp).max()                       # (it IS the mixed mode)
    out = product_runner(g, acc_f64=True).process()
    assert np.abs(out - exp).max() <= 1e-10 * np.abs(exp).max()


def test_plan_picks_its_precision_from_the_table(gpu):
    """BFGX_ACC_AUTO: fp32 pair math while the table cannot move a pixel by more than 0.1 pixel sides of the plan's NSIDE, the parity-grade mode
    beyond; a table the fast kernel cannot take (a property axis) resolves the parity-grade mode to fp64 throughout; explicit requests stay"""
    import torch
    from baryonification_amd import _lib, engine, synthetic as syn
    cat = syn.make_catalog(1000)
    z, M, r = syn.table_grid(cat, pad=1e-3)
    axes = [np.log(1 + z), np.log(M), np.log(r)]
    d = syn.displacement_table(z, M, r)
    st = torch.cuda.current_stream().cuda_stream
    got = {}
    for nside, scale in ((1024, 1.0), (1024, 40.0), (64, 40.0), (4096, 8.0)):
        model, keep = engine.model_from_tables(axes, scale * d, syn.COSMO, 10.0, 10.0)
        plan = engine.ShellPlan(model, keep, nside, 1000, 0, st)
        got[(nside, scale)] = plan.precision(_lib.ACC_AUTO)
        assert plan.precision(_lib.ACC_F32)[0] == _lib.ACC_F32 and plan.precision(_lib.ACC_F64)[0] == _lib.ACC_F64
        assert plan.precision(_lib.ACC_PARITY)[0] == _lib.ACC_PARITY
        plan.close()
    assert got[(1024, 1.0)][0] == _lib.ACC_F32 and 0.01 < got[(1024, 1.0)][1] < 0.4, got
    assert got[(1024, 40.0)][0] == _lib.ACC_PARITY and np.isclose(got[(1024, 40.0)][1], 40 * got[(1024, 1.0)][1], rtol=1e-9), got
    assert got[(64, 40.0)][0] == _lib.ACC_F32 and np.isclose(got[(64, 40.0)][1] * 16, got[(1024, 40.0)][1], rtol=1e-9), got      # coarse pixels
    assert got[(4096, 8.0)][0] == _lib.ACC_PARITY and got[(4096, 8.0)][1] < got[(1024, 40.0)][1], got                           # fine pixels
    # a 4-axis table (model.p_keys): the generic tile kernel -- the parity-grade request runs as fp64 throughout there
    p = np.linspace(0.5, 1.5, 3)
    d4 = 40.0 * d[..., None] * p[None, None, None, :]
    model, keep = engine.model_from_tables(axes + [p], d4, syn.COSMO, 10.0, 10.0)
    plan = engine.ShellPlan(model, keep, 1024, 1000, 0, st)
    assert plan.precision(_lib.ACC_AUTO)[0] == _lib.ACC_F64 and plan.precision(_lib.ACC_PARITY)[0] == _lib.ACC_F64
    plan.close()


def test_empty_catalog(gpu):
    """no halos: baryonify returns the regridded (identity) map, paint returns zeros"""
    g = load_golden('lowz_baryonify')
    for k in g['cat']:
        g['cat'][k] = g['cat'][k][:0]
    out = product_runner(g, acc_f64=True).process()
    assert np.abs(out - g['map_in']).max() <= 1e-12 * g['map_in'].max()
    g = load_golden('lowz_paint')
    for k in g['cat']:
        g['cat'][k] = g['cat'][k][:0]
    assert np.all(product_runner(g).process() == 0)


def test_halos_outside_table_contribute_nothing(gpu):
    """RegularGridInterpolator fill_value = NaN -> offset/paint 0 (HealpixRunner.py:323, :442)"""
    g = load_golden('lowz_baryonify')
    g['cat']['M'] = g['cat']['M'] * 1e3          # all above the table's M range
    out = product_runner(g, acc_f64=True).process()
    ora = oracle_run(g)
    assert np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max()
    assert np.abs(out - g['map_in']).max() <= 1e-12 * g['map_in'].max()


@pytest.mark.parametrize('nside', [4, 16, 64, 256, 1024])
def test_tile_binning_is_complete(gpu, nside):
    """pair census through the tile path (halo -> tile entries, rows clipped to tiles) must equal the
    per-halo census exactly (integer equality): no pixel of any disc is lost or double counted"""
    import torch
    from baryonification_amd import _lib, engine, synthetic as syn
    rng = np.random.default_rng(nside)
    N = 4000
    cat = syn.make_catalog(N, seed=100 + nside, z_lo=0.005, z_hi=0.4, logM_lo=12.0, logM_hi=15.5)
    cat['dec'][:8] = [90 - 1e-8, -90 + 1e-8, 89.9, -89.9, 89.0, -88.5, 0.0, 0.0]
    cat['ra'][:8] = [0.0, 10.0, 359.99, 0.01, 180.0, 90.0, 0.0, 359.9999]
    cat['M'][:8] = 3e15
    cat['z'][:8] = [0.01, 0.01, 0.02, 0.02, 0.05, 0.01, 0.008, 0.008]
    z, M, r = syn.table_grid(cat, Nz=4, NM=4, NR=32, pad=1e-9)
    model, keep = engine.model_from_tables([np.log(1 + z), np.log(M), np.log(r)], syn.displacement_table(z, M, r),
                                           syn.COSMO, 10.0)
    dev = torch.device('cuda', 0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in cat.items()}
    plan = engine.ShellPlan(model, keep, nside, N, 0, torch.cuda.current_stream().cuda_stream)
    cd = _lib.make_catalog_dev(N, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr())
    for fb in (True, False):
        plan.set_algo(0)
        n0 = plan.count_pairs(cd, fallback4=fb)
        plan.set_algo(1)
        n1 = plan.count_pairs(cd, fallback4=fb)
        assert n0 == n1 and n0 > 0
    # and against the oracle's own census
    from oracle import oracle as O
    tab = O.Table([np.log(1 + z), np.log(M), np.log(r)], syn.displacement_table(z, M, r), False, 10.0)
    sub = {k: v[:400] for k, v in cat.items()}
    _, counts = O.baryonify_offsets(nside, sub, tab, 10.0, O.Background.from_dict(syn.COSMO), return_counts=True)
    t2 = {k: torch.from_numpy(v).to(dev) for k, v in sub.items()}
    cd2 = _lib.make_catalog_dev(400, t2['M'].data_ptr(), t2['z'].data_ptr(), t2['ra'].data_ptr(), t2['dec'].data_ptr())
    assert plan.count_pairs(cd2, fallback4=True) == int(counts.sum())
    plan.close()


def test_invalid_halos_are_ignored(gpu):
    """NaN / non-positive masses, |dec| > 90 or NaN positions touch no pixel (and must never fault the GPU)"""
    g = load_golden('lowz_baryonify')
    ref = run(g, True, 1)
    n = g['cat']['M'].size
    for k in g['cat']:
        g['cat'][k] = np.concatenate([g['cat'][k], g['cat'][k][:6]])
    g['cat']['M'][n:n + 2] = [np.nan, -1e14]
    g['cat']['dec'][n + 2] = 91.0
    g['cat']['ra'][n + 3] = np.nan
    g['cat']['z'][n + 4] = np.nan
    g['cat']['M'][n + 5] = 0.0
    for algo in ALGOS:
        out = run(g, True, algo)
        assert np.isfinite(out).all() and np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max()


def test_entry_list_regrowth(gpu, monkeypatch):
    """a deliberately tiny halo->tile entry list must be regrown (one-shot API) or reported (resident API)"""
    g = load_golden('lowz_baryonify')
    ref = run(g, True, 1)
    monkeypatch.setenv('BFGX_ENTRY_CAP', '64')
    out = run(g, True, 1)                                   # host API: overflow -> exact regrowth -> same answer
    assert np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max()
    import torch
    from baryonification_amd import _lib, engine
    axes = [np.log(1 + g['tab_z']), np.log(g['tab_M']), np.log(g['tab_r'])]
    model, keep = engine.model_from_tables(axes, g['tab_values'], g['cosmo_runner'], g['eps_runner'], g['eps_model'])
    dev = torch.device('cuda', 0)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in g['cat'].items()}
    plan = engine.ShellPlan(model, keep, g['nside'], t['M'].numel(), 0, torch.cuda.current_stream().cuda_stream)
    cd = _lib.make_catalog_dev(t['M'].numel(), t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr())
    off = torch.zeros(12 * g['nside'] ** 2 * 3, dtype=torch.float64, device=dev)
    plan.offsets(cd, off.data_ptr(), True)
    with pytest.raises(ValueError, match='overflowed'):
        plan.status()
    plan.close()


@pytest.mark.parametrize('seed,nside,zr,logM,eps', [
    (1, 1, (0.01, 0.05), (14.0, 15.5), 8.0),       # 12 pixels: every disc takes the <4-pixel fallback or covers whole rings
    (2, 2, (0.005, 0.02), (14.5, 15.8), 30.0),     # discs of tens of degrees, poles inside, radius > pi/2 for some
    (3, 8, (0.02, 0.2), (13.0, 15.5), 12.0),
    (4, 32, (0.05, 1.5), (12.0, 15.0), 10.0),      # wide redshift range: D_A spline, (1+z) scalings
    (5, 128, (0.3, 3.0), (11.5, 14.5), 20.0),      # high z: sub-pixel discs -> fallback for nearly all
    (6, 512, (0.1, 0.2), (13.5, 15.5), 6.0),       # several tiles per halo, >4-tile halos
])
def test_randomized_regimes_vs_oracle(gpu, seed, nside, zr, logM, eps):
    """random catalogs far from the benchmark's regime, both BaryonifyShell accumulator types and PaintProfilesShell,
    against the CPU oracle on the same inputs"""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    from oracle import oracle as O
    N = 600
    cat = syn.make_catalog(N, seed=1000 + seed, z_lo=zr[0], z_hi=zr[1], logM_lo=logM[0], logM_hi=logM[1])
    rng = np.random.default_rng(seed)
    cat['dec'][:4] = [90.0 - 1e-8, -90.0 + 1e-8, 89.99, 0.0]          # poles and the phi = 0 seam
    cat['ra'][:4] = [0.0, 123.0, 359.999, 1e-9]
    z, M, r = syn.table_grid(cat, Nz=5, NM=6, NR=96, R_min=1e-3, R_max=1e3, pad=1e-9)
    d = syn.displacement_table(z, M, r) * (1 + 0.3 * np.sin(3 * np.log(r))[None, None, :])     # not monotone in r
    P = syn.paint_table(z, M, r)
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    hmap = syn.make_map(nside, seed=seed)
    axes = [np.log(1 + z), np.log(M), np.log(r)]
    bg = O.Background.from_dict(syn.COSMO)
    used = {k: np.array(Catalog.cat[k]) for k in ('M', 'z', 'ra', 'dec')}
    # BaryonifyShell, model epsilon < runner epsilon (the r < eps R mask bites)
    model = bfg.Profiles.Baryonification2D(None, None, cosmo, epsilon_max=0.7 * eps)
    model.set_table(z, M, r, d)
    Shell = bfg.utils.LightconeShell(map=hmap, cosmo=syn.COSMO)
    ora = O.baryonify_shell(nside, hmap, used, O.Table(axes, d, False, 0.7 * eps), eps, bg)
    for acc64, tol in ((True, 1e-10 * np.abs(ora).max()), (False, 1e-6 * ora.mean())):
        runner = bfg.Runners.BaryonifyShell(Catalog, Shell, eps, model, verbose=False)
        runner.acc_f64 = acc64
        out = runner.process()
        assert np.abs(out - ora).max() <= tol, (acc64, np.abs(out - ora).max(), tol)
    # PaintProfilesShell
    prof = bfg.utils.TabulatedProfile(None, cosmo)
    prof.set_table(z, M, r, P)
    pshell = bfg.utils.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=syn.COSMO)
    pr = bfg.Runners.PaintProfilesShell(Catalog, pshell, eps, prof, verbose=False)
    with np.errstate(divide='ignore'):
        orap = O.paint_shell(nside, used, O.Table(axes, np.log(P)), eps, bg)
    pr.acc_f64 = True                                                 # fp64 throughout
    assert np.abs(pr.process() - orap).max() <= 1e-10 * np.abs(orap).max()
    pr.acc_f64 = None                                                 # default: fp32 pair math into the fp64 map
    out = pr.process()
    assert np.all(np.abs(out - orap) <= 5e-5 * np.abs(orap) + 1e-12 * np.abs(orap).max())


@pytest.mark.parametrize('name', ['lowz_baryonify', 'c1_baryonify', 'rdelta_baryonify', 'lowz_paint', 'param1_paint'])
@pytest.mark.parametrize('ndev', [1, 3])
def test_multi_device_c_entry_on_one_gpu(gpu, name, ndev):
    """bfgx_*_shell_multi (all GPUs of a node from one C call) with the ONE device of this box named `ndev` times: the halo
    shards, the peer-to-peer slice exchange (device-to-device copies here), the banded gathering regrid and the assembly of the
    disjoint slices are the code an 8-GPU node runs.  Through the drop-in SplitJoinParallel; result == single-device run."""
    import baryonification_amd as bfg
    g = load_golden(name)
    exp = g['expected']
    r = product_runner(g, acc_f64=True)
    sj = bfg.utils.SplitJoinParallel(r, njobs=ndev, devices=[0] * ndev)
    out = sj.process()
    assert out.dtype == np.float64 and out.shape == exp.shape
    assert np.abs(out - exp).max() <= 1e-10 * np.abs(exp).max()
    if g['kind'] == 'baryonify':
        assert np.isclose(sj.last_stats['sum_out'], sj.last_stats['sum_in']) and np.isclose(out.sum(), g['map_in'].sum())
        # default accumulators (f32 pix_offsets) through the same path
        r32 = product_runner(g)
        out32 = bfg.utils.SplitJoinParallel(r32, njobs=ndev, devices=[0] * ndev).process()
        assert np.abs(out32 - exp).max() <= 1e-6 * exp.mean()


@pytest.mark.parametrize('scale', [30.0, 400.0])
def test_multi_device_c_entry_large_displacements(gpu, scale):
    """the same one-call multi-device entry with the displacement table scaled up so that pixels move by a good fraction of a
    ring / by several rings: the devices pull 16 rings of apron, agree on the reach from the summed offsets and regrid with it;
    result == the single-device call == the CPU oracle on the scaled table"""
    import baryonification_amd as bfg
    from helpers import oracle_run
    g = dict(load_golden('lowz_baryonify'))
    g['tab_values'] = g['tab_values'] * scale
    ora = oracle_run(g)
    one = product_runner(g, acc_f64=True).process()
    assert np.abs(one - ora).max() <= 1e-10 * np.abs(ora).max()
    pix = np.sqrt(4 * np.pi / one.size)
    for ndev in (2, 3):
        sj = bfg.utils.SplitJoinParallel(product_runner(g, acc_f64=True), njobs=ndev, devices=[0] * ndev)
        out = sj.process()
        assert np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max()
        assert np.isclose(sj.last_stats['sum_out'], sj.last_stats['sum_in']) and np.isclose(out.sum(), g['map_in'].sum())
        # the default (the plan picks the precision from the table) holds SURVEY 8(d)'s tolerance on a table that moves pixels this far
        outd = bfg.utils.SplitJoinParallel(product_runner(g), njobs=ndev, devices=[0] * ndev).process()
        assert np.abs(outd - ora).max() <= 1e-6 * ora.mean()
    assert np.abs(one - g['map_in']).max() > 0 and pix > 0


def test_callable_models_per_halo_and_tabulated(gpu):
    """HealpixRunner.py:321, :441 call model.displacement(r, M, a) / model.projected(cosmo, r, M, a) on ANY object.  A plain-Python model
    (no raw_input_*, no setup_interpolator) goes through process() in one of two ways: called once per halo on that halo's own pixel
    separations -- the reference's loop, reproduced to 1e-10 (catalogs up to 50 000 halos, or model.bfgx_exact = True) --, or tabulated once
    on the catalog's (z, M) support (larger catalogs, or model.bfgx_exact = False): then the result equals the oracle fed the same tabulation
    and differs from the reference by the table's interpolation error; objects with neither method are still refused"""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    from baryonification_amd.Runners import _model as RM
    from oracle import oracle as O
    nside, N, eps = 128, 800, 8.0
    cat = syn.make_catalog(N, seed=77)
    cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
    bg = O.Background.from_dict(syn.COSMO)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    used = {k: np.array(Catalog.cat[k]) for k in ('M', 'z', 'ra', 'dec')}

    class PlainDisplacement(object):
        """closed form, vectorised in r; the model-side cut at epsilon_max R (BaryonCorrection.py:381-382) is its own business"""
        epsilon_max = 6.0
        calls = 0

        def displacement(self, r, M, a):
            type(self).calls += 1
            Rc = bg.get_radius(np.atleast_1d(M), a)[0] / a
            x = np.asarray(r) / Rc
            return np.where(x < self.epsilon_max, -0.05 * Rc * x * np.exp(-x) / (1 + x * x), 0.0)

    class PlainProfile(object):
        def projected(self, cosmo, r, M, a):
            Rc = bg.get_radius(np.atleast_1d(M), a) / a
            x = np.asarray(r)[None, :] / Rc[:, None]
            out = (np.atleast_1d(M)[:, None] / 1e14) * np.exp(-x) / (1 + x) ** 2
            return out if np.ndim(M) else out[0]

    hmap = syn.make_map(nside, seed=5)
    Shell = bfg.utils.LightconeShell(map=hmap, cosmo=syn.COSMO)
    model = PlainDisplacement()
    model.bfgx_exact = False                                                    # the tabulated bridge (the default above 50 000 halos)
    runner = bfg.Runners.BaryonifyShell(Catalog, Shell, eps, model, verbose=False)
    runner.acc_f64 = True
    out = runner.process()
    n_calls = PlainDisplacement.calls
    assert n_calls == RM.BRIDGE_N_Z * RM.BRIDGE_N_M and hasattr(model, '_bfgx_tabulated')
    holder = model._bfgx_tabulated[1]
    axes = [holder.raw_input_z_range, holder.raw_input_M_range, holder.raw_input_r_range]
    ora = O.baryonify_shell(nside, hmap, used, O.Table(axes, holder.raw_input_d, False, model.epsilon_max), eps, bg)
    assert np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max() and np.abs(out - hmap).max() > 0 and np.isclose(out.sum(), hmap.sum())
    runner.process()
    assert PlainDisplacement.calls == n_calls                                   # the table is built once
    # against the reference's OWN form of the loop -- model.displacement() called per halo (HealpixRunner.py:291-331), restated here
    # with the oracle's HEALPix primitives: the bridge differs from it by the interpolation error of the table, which is what the
    # warning states.  Measured on this profile (smooth but for its own cut at 6 R, which the table smears over one radial cell): 2.4e-3 of
    # the largest |new - old| at the default sampling; stated: 5e-3
    Da = bg.Da_spline()
    off = np.zeros((hmap.size, 3))
    for j in range(N):
        M_j, z_j = used['M'][j], used['z'][j]
        a_j = 1 / (1 + z_j)
        R_j, D_j = bg.get_radius(np.atleast_1d(M_j), a_j)[0], float(Da(z_j))
        vec_j = O.ang2vec_lonlat(used['ra'][j], used['dec'][j]).reshape(3)
        pix = O.query_disc(nside, vec_j, R_j * eps / D_j)
        if pix.size < 4:
            pix = O.get_interp_weights_lonlat(nside, np.atleast_1d(used['ra'][j]), np.atleast_1d(used['dec'][j]))[0].reshape(-1)
        vec = O.pix2vec(nside, pix)
        diff = (vec - vec_j) * D_j
        r_sep = np.sqrt((diff ** 2).sum(axis=1))
        with np.errstate(invalid='ignore', divide='ignore'):
            o = (model.displacement(r_sep / a_j, M_j, a_j) * a_j)[:, None] * (diff / r_sep[:, None])
        o = np.where(np.isfinite(o), o, 0)
        nw = vec * D_j + o
        off[pix] += nw / np.sqrt((nw ** 2).sum(axis=1))[:, None] - vec
    direct_map = O.regrid(nside, hmap, off)
    PlainDisplacement.calls = n_calls
    err = np.abs(out - direct_map).max() / np.abs(direct_map - hmap).max()
    print("callable bridge vs per-halo calls: %.2e of the largest change of the map" % err)
    assert err < 5e-3
    # the EXACT route (the default for a catalog of this size; model.bfgx_exact = True for any): the model is called once per halo on the
    # separations of that halo's own pixels, as the reference does -- no table, no interpolation error
    del model.bfgx_exact
    PlainDisplacement.calls = 0
    out_exact = runner.process()
    assert PlainDisplacement.calls == N and np.isclose(out_exact.sum(), hmap.sum())
    err_exact = np.abs(out_exact - direct_map).max() / np.abs(direct_map - hmap).max()
    print("per-halo route vs the reference's loop: %.2e of the largest change of the map" % err_exact)
    assert err_exact <= 1e-8 and np.abs(out_exact - direct_map).max() <= 1e-10 * np.abs(direct_map).max()       # (3e-10 of the change measured)
    assert runner.last_stats['n_pairs'] > 4 * N
    model.bfgx_exact = False
    PlainDisplacement.calls = n_calls
    # a finer table on request, and a changed parameter is seen (the cached table is keyed by the model's attributes)
    model.bfgx_table_grid = (12, 40, 1000)
    with pytest.warns(RuntimeWarning, match="tabulated once on 12 x 40 x 1000"):
        out_fine = runner.process()
    assert PlainDisplacement.calls == n_calls + 12 * 40
    assert np.abs(out_fine - direct_map).max() < np.abs(out - direct_map).max()
    model.epsilon_max = 5.0                                                     # an instance attribute now: the fingerprint changes
    runner.process()
    assert PlainDisplacement.calls == n_calls + 2 * 12 * 40
    del model.bfgx_table_grid
    model.epsilon_max = 6.0
    prof = PlainProfile()
    pshell = bfg.utils.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=syn.COSMO)
    pr = bfg.Runners.PaintProfilesShell(Catalog, pshell, eps, prof, verbose=False)
    pr.acc_f64 = True
    # exact route first (default at this size): the reference's loop (HealpixRunner.py:417-445) restated with the oracle's primitives
    painted_exact = pr.process()
    ref_paint = np.zeros(12 * nside * nside)
    for j in range(N):
        M_j, z_j = used['M'][j], used['z'][j]
        a_j = 1 / (1 + z_j)
        R_j, D_j = bg.get_radius(np.atleast_1d(M_j), a_j)[0], float(Da(z_j))
        vec_j = O.ang2vec_lonlat(used['ra'][j], used['dec'][j]).reshape(3)
        pix = O.query_disc(nside, vec_j, R_j * eps / D_j)
        r_sep = np.sqrt((((O.pix2vec(nside, pix) - vec_j) * D_j) ** 2).sum(axis=1))
        P = prof.projected(cosmo, r_sep / a_j, M_j, a_j)
        ref_paint[pix] += np.where(np.isfinite(P), P, 0)
    assert ref_paint.max() > 0 and np.abs(painted_exact - ref_paint).max() <= 1e-10 * ref_paint.max()
    prof.bfgx_exact = False
    painted = pr.process()
    hp_ = prof._bfgx_tabulated[1]
    with np.errstate(divide='ignore'):
        orap = O.paint_shell(nside, used, O.Table([hp_.raw_input_z_range, hp_.raw_input_M_range, hp_.raw_input_r_range], np.log(hp_.raw_input_2D)), eps, bg)
    assert painted.max() > 0 and np.abs(painted - orap).max() <= 1e-10 * np.abs(orap).max()
    # the tabulation itself is faithful: the table read-out against the callable at the halos' own (z, M) and a few radii
    j = np.argmax(cat['M'])
    a_j = 1 / (1 + cat['z'][j])
    rr = np.geomspace(0.02, 5.0, 7)
    direct = prof.projected(cosmo, rr, cat['M'][j], a_j)
    assert np.abs(hp_.projected(cosmo, rr, cat['M'][j], a_j) / direct - 1).max() < 2e-3
    with pytest.raises(TypeError):
        bfg.Runners.BaryonifyShell(Catalog, Shell, eps, object(), verbose=False).process()
    with pytest.raises(NameError):                                              # a table class that was never set up: as the reference
        bfg.Runners.BaryonifyShell(Catalog, Shell, eps, bfg.Profiles.Baryonification2D(None, None, cosmo), verbose=False).process()
