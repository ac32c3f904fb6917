"""Parity of the regular-grid HIP path (csrc/bfgx_grid.hpp, bfgx_fft.hpp), called through the drop-in runners and
functions (-> ctypes -> C ABI), against (a) the reference's own outputs (tests/golden/grid_*.npz) and (b) the CPU
oracle on the same inputs at sizes the oracle finishes in seconds.

Stated tolerances: everything on this path is fp64 (geometry, read-out, accumulation by fp64 atomics); differences
are libm ulps and atomic summation order                               ->  |d| <= 1e-10 * max|map|
regrid_pixels_* / make_map have no libm in them                        ->  1e-13 * max (summation order only)
P(k): fp64 radix-2 FFT vs numpy's pocketfft, bin sums in another order ->  1e-10 relative per bin
"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', H.GRID_RUNNER_CASES)
def test_grid_runner_vs_reference_golden(gpu, name):
    g = H.load_grid_golden(name)
    out = H.grid_product_runner(g).process()
    exp = g['expected']
    assert out.dtype == np.float64 and out.shape == exp.shape
    assert np.abs(out - exp).max() <= 1e-10 * np.abs(exp).max()
    if g['kind'] == 'baryonify':
        assert np.isclose(out.sum(), g['map_in'].sum())


@pytest.mark.parametrize('name', H.GRID_RUNNER_CASES)
def test_grid_runner_vs_oracle(gpu, name):
    g = H.load_grid_golden(name)
    out = H.grid_product_runner(g).process()
    ora = H.grid_oracle_run(g)
    assert np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max()


@pytest.mark.parametrize('name', ['grid2d_regrid', 'grid3d_regrid'])
def test_regrid_pixels_vs_reference_golden(gpu, name):
    import baryonification_amd as bfg
    g = H.load_grid_golden(name)
    grid = np.zeros((g['npix'],) * g['ndim'])
    (bfg.Runners.regrid_pixels_2D if g['ndim'] == 2 else bfg.Runners.regrid_pixels_3D)(grid, g['pos'], g['val'])
    assert np.abs(grid - g['expected']).max() <= 1e-13 * np.abs(g['expected']).max()
    # in place and additive, like the reference's njit function
    (bfg.Runners.regrid_pixels_2D if g['ndim'] == 2 else bfg.Runners.regrid_pixels_3D)(grid, g['pos'], g['val'])
    assert np.abs(grid - 2 * g['expected']).max() <= 1e-13 * np.abs(g['expected']).max()


@pytest.mark.parametrize('name', ['grid2d_make_map', 'grid3d_make_map'])
def test_make_map_vs_reference_golden(gpu, name):
    import baryonification_amd as bfg
    g = H.load_grid_golden(name)
    xyz = g['xyz']
    Snap = bfg.utils.ParticleSnapshot(x=xyz[:, 0], y=xyz[:, 1], z=xyz[:, 2] if g['ndim'] == 3 else None, M=g['mass'],
                                      L=float(g['L']), redshift=0.0, cosmo=H.load_grid_golden('grid2d_paint')['cosmo_runner'])
    out = Snap.make_map(int(g['N_grid']))
    assert out.shape == g['expected'].shape
    assert np.abs(out - g['expected']).max() <= 1e-13 * np.abs(g['expected']).max()
    assert np.array_equal(out != 0, g['expected'] != 0)            # bin membership is exact (edges, inclusive last edge)


@pytest.mark.parametrize('mode', ['tiled', 'atomic'])
@pytest.mark.parametrize('ndim,N,npart,kind', [(3, 80, 200_000, 'uniform'), (3, 33, 50_000, 'clustered'), (2, 100, 100_000, 'uniform'),
                                               (2, 300, 150_000, 'clustered'), (3, 128, 5, 'uniform'), (3, 64, 70_000, 'one-cell')])
def test_make_map_tiled_and_atomic_forms(gpu, monkeypatch, mode, ndim, N, npart, kind):
    """ParticleSnapshot.make_map through both device forms (sort into 8192-cell tiles + LDS-owned accumulation; one global
    atomic per particle) == the oracle's np.histogramdd restatement: grids that are not multiples of the tile shape,
    particles outside the box / exactly on edges (dropped / inclusive last edge), everything in one cell, a handful of particles"""
    import baryonification_amd as bfg
    from oracle import grid as G
    monkeypatch.setenv('BFGX_DEPOSIT', mode)
    rng = np.random.default_rng(N + npart)
    L = 250.0
    if kind == 'uniform':
        xyz = rng.uniform(-0.02 * L, 1.02 * L, (npart, ndim))              # ~8 % outside
    elif kind == 'clustered':
        c = rng.uniform(0, L, (40, ndim))
        xyz = c[rng.integers(0, 40, npart)] + rng.normal(0, 0.01 * L, (npart, ndim))
    else:
        xyz = np.full((npart, ndim), 0.4321 * L) + rng.uniform(0, 1e-3, (npart, ndim))
    k = min(4, npart)
    xyz[:k] = np.array([[0.0] * ndim, [L] * ndim, [L / N] * ndim, [L * (1 - 1e-16)] * ndim])[:k]
    mass = rng.uniform(0.5, 2.0, npart)
    Snap = bfg.utils.ParticleSnapshot(x=xyz[:, 0], y=xyz[:, 1], z=xyz[:, 2] if ndim == 3 else None, M=mass, L=L, redshift=0.0,
                                      cosmo=H.load_grid_golden('grid2d_paint')['cosmo_runner'])
    Map = Snap.make_map(N)
    ora = G.make_map([xyz[:, d] for d in range(ndim)], mass, L, N)
    assert Map.shape == (N,) * ndim and np.array_equal(Map != 0, ora != 0)
    assert np.abs(Map - ora).max() <= 1e-12 * ora.max()


@pytest.mark.parametrize('N,Nk', [(8, 3), (16, 5), (32, 12), (64, 180), (128, 60), (32, 300), (256, 180)])      # (Nk > 254: bins computed per mode, no table)
def test_power_spectrum_vs_numpy_restatement(gpu, N, Nk):
    from baryonification_amd.engine import power_spectrum
    from oracle import grid as G
    rng = np.random.default_rng(N)
    Map = rng.poisson(4.0, (N, N, N)).astype(np.float64)
    Map[3, 5, 7] += 500.0
    L = 305.0
    k_cen, Pk, k_c = power_spectrum(Map, L, Nk)
    ok_cen, oPk, ok_c = G.power_spectrum(Map, L, Nk)
    assert np.array_equal(k_c, ok_c)
    good = ok_c > 0
    assert np.array_equal(np.isnan(Pk), ~good)
    assert np.abs(Pk[good] / oPk[good] - 1).max() <= 1e-10
    assert np.abs(k_cen[good] / ok_cen[good] - 1).max() <= 1e-12


def _big_case(ndim, N, nh, seed):
    """synthetic grid case above the fixture sizes: many halos, overlapping cutouts, periodic wraps"""
    from baryonification_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    L = 2.0 * N
    bins = (np.arange(N) + 0.5) * (L / N)
    M = (10 ** rng.uniform(12.5, 15.0, nh)).astype(np.float32).astype(np.float64)
    pos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
    zr = 0.3
    z = np.linspace(zr - 0.05, zr + 0.05, 3)
    Mt = np.geomspace(10 ** 12.4, 10 ** 15.1, 8)
    r = np.geomspace(1e-3, 2e2, 200)
    cat = {'M': M, 'x': pos[:, 0], 'y': pos[:, 1], 'z': pos[:, 2]}
    hmap = rng.poisson(2.0, (N,) * ndim).astype(np.float64)
    return dict(ndim=ndim, N=N, L=L, bins=bins, cat=cat, redshift=zr, z=z, Mt=Mt, r=r, map=hmap,
                d=syn.displacement_table(z, Mt, r), P=syn.paint_table(z, Mt, r), cosmo=dict(syn.COSMO))


@pytest.mark.parametrize('ndim,N,nh', [(2, 512, 4000), (3, 80, 1500)])
def test_grid_baryonify_vs_oracle_larger(gpu, ndim, N, nh):
    import baryonification_amd as bfg
    from oracle import grid as G
    from oracle import oracle as O
    c = _big_case(ndim, N, nh, 100 + ndim)
    cat = c['cat']
    HCat = bfg.utils.HaloNDCatalog(x=cat['x'], y=cat['y'], z=cat['z'] if ndim == 3 else None, M=cat['M'], redshift=c['redshift'],
                                   cosmo=c['cosmo'])
    GMap = bfg.utils.GriddedMap(map=c['map'], redshift=c['redshift'], bins=c['bins'], cosmo=c['cosmo'])
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(c['cosmo']), epsilon_max=8.0)
    model.set_table(c['z'], c['Mt'], c['r'], c['d'])
    runner = bfg.Runners.BaryonifyGrid(HCat, GMap, 6.0, model, verbose=False)
    out = runner.process()
    tab = O.Table([np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])], c['d'], False, 8.0)
    off, pairs = G.baryonify_grid_offsets(c['map'].shape, c['bins'], cat, c['redshift'], tab, 6.0, G.grid_background(c['cosmo']),
                                          return_pairs=True)
    ora = G.regrid_offsets(c['map'], off)
    assert np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max()
    assert np.isclose(out.sum(), c['map'].sum(), rtol=1e-12)
    assert 0 < runner.last_stats['n_pairs'] <= pairs           # contributing pairs <= cutout pixels


def test_grid_paint_3d_vs_oracle_larger(gpu):
    import baryonification_amd as bfg
    from oracle import grid as G
    from oracle import oracle as O
    c = _big_case(3, 96, 3000, 7)
    cat = c['cat']
    HCat = bfg.utils.HaloNDCatalog(x=cat['x'], y=cat['y'], z=cat['z'], M=cat['M'], redshift=c['redshift'], cosmo=c['cosmo'])
    GMap = bfg.utils.GriddedMap(map=np.zeros((96,) * 3), redshift=c['redshift'], bins=c['bins'], cosmo=c['cosmo'])
    model = bfg.utils.TabulatedProfile(None, bfg.utils.Cosmology.from_dict(c['cosmo']))
    model.set_table(c['z'], c['Mt'], c['r'], c['P'], table_3D=2.0 * c['P'])        # 3D maps read raw_input_3D
    out = bfg.Runners.PaintProfilesGrid(HCat, GMap, 4.0, model, verbose=False).process()
    tab = O.Table([np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])], np.log(2.0 * c['P']))
    ora = G.paint_grid((96,) * 3, c['bins'], cat, c['redshift'], tab, 4.0, G.grid_background(c['cosmo']))
    assert np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max()


def test_grid_nan_poisoning_and_invalid_halos(gpu):
    """a halo outside the table's mass range poisons (-> zeroes) the offsets of every pixel of its ball, exactly as the
    reference's post-loop np.where(np.isfinite(...)) does; NaN / non-positive halos are ignored"""
    import baryonification_amd as bfg
    from oracle import grid as G
    from oracle import oracle as O
    c = _big_case(2, 128, 300, 9)
    cat = {k: v.copy() for k, v in c['cat'].items()}
    cat['M'][0] = 3e15                                         # above the table's last mass node
    HCat = bfg.utils.HaloNDCatalog(x=cat['x'], y=cat['y'], M=cat['M'], redshift=c['redshift'], cosmo=c['cosmo'])
    used = {k: np.array(HCat.cat[k], dtype=np.float64) for k in ('M', 'x', 'y', 'z')}
    GMap = bfg.utils.GriddedMap(map=c['map'], redshift=c['redshift'], bins=c['bins'], cosmo=c['cosmo'])
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(c['cosmo']), epsilon_max=8.0)
    model.set_table(c['z'], c['Mt'], c['r'], c['d'])
    out = bfg.Runners.BaryonifyGrid(HCat, GMap, 6.0, model, verbose=False).process()
    tab = O.Table([np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])], c['d'], False, 8.0)
    ora = G.baryonify_grid(c['map'], c['bins'], used, c['redshift'], tab, 6.0, G.grid_background(c['cosmo']))
    assert np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max()
    # invalid rows change nothing
    bad = {k: np.concatenate([v, [np.nan if k == 'M' else 10.0, -1.0 if k == 'M' else 20.0]]) for k, v in used.items()}
    HCat2 = bfg.utils.HaloNDCatalog(x=bad['x'], y=bad['y'], M=bad['M'], redshift=c['redshift'], cosmo=c['cosmo'])
    out2 = bfg.Runners.BaryonifyGrid(HCat2, GMap, 6.0, model, verbose=False).process()
    assert np.abs(out2 - out).max() <= 1e-12 * np.abs(out).max()


def test_grid_work_item_table_regrowth(gpu):
    """the work-item table starts too small (BFGX_GRID_ITEM_CAP) and is regrown inside the call: same result"""
    c = H.load_grid_golden('grid3d_baryonify')
    base = H.grid_product_runner(c).process()
    os.environ['BFGX_GRID_ITEM_CAP'] = '3'
    try:
        again = H.grid_product_runner(c).process()
    finally:
        del os.environ['BFGX_GRID_ITEM_CAP']
    assert np.abs(again - base).max() <= 1e-12 * np.abs(base).max()


def test_grid_plan_resident_path(gpu):
    """device-resident GridPlan (what bench.py --mode grid uses) == one-shot host API"""
    import torch
    import baryonification_amd as bfg
    from baryonification_amd import _lib, engine
    c = _big_case(3, 64, 800, 5)
    cat = c['cat']
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(c['cosmo']), epsilon_max=8.0)
    model.set_table(c['z'], c['Mt'], c['r'], c['d'])
    HCat = bfg.utils.HaloNDCatalog(x=cat['x'], y=cat['y'], z=cat['z'], M=cat['M'], redshift=c['redshift'], cosmo=c['cosmo'])
    GMap = bfg.utils.GriddedMap(map=c['map'], redshift=c['redshift'], bins=c['bins'], cosmo=c['cosmo'])
    host = bfg.Runners.BaryonifyGrid(HCat, GMap, 6.0, model, verbose=False).process()

    cos = dict(c['cosmo'], w0=-1.0)
    m, keep = engine.model_from_tables([np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])], c['d'], cos, 6.0, 8.0)
    dev = torch.device('cuda:0')
    t = {k: torch.tensor(cat[k], dtype=torch.float64, device=dev) for k in ('M', 'x', 'y', 'z')}
    lnM = torch.tensor(np.log(cat['M'].astype(np.float32)).astype(np.float64), device=dev)
    plan = engine.GridPlan(m, keep, c['bins'], 3, c['redshift'], cat['M'].size, 0, torch.cuda.current_stream().cuda_stream)
    dcat = _lib.make_grid_catalog_dev(cat['M'].size, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr(),
                                      lnM.data_ptr())
    off = torch.empty((64 ** 3, 3), dtype=torch.float64, device=dev)
    m_in = torch.tensor(c['map'], device=dev)
    m_out = torch.empty_like(m_in)
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    plan.timing_enable(True)
    npairs = plan.offsets(dcat, off.data_ptr())
    plan.regrid(m_in.data_ptr(), off.data_ptr(), m_out.data_ptr(), sums.data_ptr())
    torch.cuda.synchronize()
    tm = plan.timing_read()
    assert npairs > 0 and tm['offsets'][1] == 1 and tm['regrid'][1] == 1
    out = m_out.cpu().numpy()
    assert np.abs(out - host).max() <= 1e-12 * np.abs(host).max()
    s = sums.cpu().numpy()
    assert np.isclose(s[0], c['map'].sum()) and np.isclose(s[1], s[0], rtol=1e-12)
    plan.close()


@pytest.mark.parametrize('ndim,N,nh', [(3, 64, 800), (3, 64, 12), (2, 128, 300), (2, 250, 9), (3, 61, 500), (3, 32, 0)])
def test_grid_cell_owned_pass(gpu, ndim, N, nh):
    """bfgx_grid_baryonify_device (halos listed per block of cells, offsets summed in registers, no pix_offsets array)
    == bfgx_grid_offsets_device + bfgx_grid_regrid_device; grids that are not a multiple of the block size; list regrowth"""
    import torch
    from baryonification_amd import _lib, engine
    c = _big_case(ndim, N, max(nh, 1), 17)
    cat = {k: v[:nh] for k, v in c['cat'].items()}
    cos = dict(c['cosmo'], w0=-1.0)
    m, keep = engine.model_from_tables([np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])], c['d'], cos, 6.0, 8.0)
    dev = torch.device('cuda:0')
    t = {k: torch.tensor(cat[k], dtype=torch.float64, device=dev) for k in ('M', 'x', 'y', 'z')}
    lnM = torch.tensor(np.log(cat['M'].astype(np.float32)).astype(np.float64), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    plan = engine.GridPlan(m, keep, c['bins'], ndim, c['redshift'], max(nh, 1), 0, stream)
    dcat = _lib.make_grid_catalog_dev(nh, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr() if ndim == 3 else 0,
                                      lnM.data_ptr())
    off = torch.empty((N ** ndim, ndim), dtype=torch.float64, device=dev)
    m_in = torch.tensor(c['map'], device=dev)
    ref, out = torch.empty_like(m_in), torch.full_like(m_in, float('nan'))
    s_ref, s_out = (torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(2))
    n_ref = plan.offsets(dcat, off.data_ptr())
    plan.regrid(m_in.data_ptr(), off.data_ptr(), ref.data_ptr(), s_ref.data_ptr())
    n_out = plan.baryonify(dcat, m_in.data_ptr(), out.data_ptr(), s_out.data_ptr())
    torch.cuda.synchronize()
    a, b = ref.cpu().numpy(), out.cpu().numpy()
    assert n_out == n_ref and (n_ref > 0) == (nh > 0)
    assert np.isfinite(b).all() and np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    assert np.allclose(s_out.cpu().numpy(), s_ref.cpu().numpy(), rtol=1e-12, atol=0)
    # maps that start at an odd multiple of 8 bytes (a caller's view into a larger buffer)
    big_in, big_out = torch.empty(m_in.numel() + 1, dtype=torch.float64, device=dev), torch.empty(m_in.numel() + 1, dtype=torch.float64, device=dev)
    big_in[1:] = m_in.reshape(-1)
    big_out.fill_(float('nan'))
    assert big_in[1:].data_ptr() % 16 == 8
    assert plan.baryonify(dcat, big_in[1:].data_ptr(), big_out[1:].data_ptr()) == n_ref
    torch.cuda.synchronize()
    assert np.abs(a - big_out[1:].cpu().numpy().reshape(a.shape)).max() <= 1e-12 * np.abs(a).max()
    if nh:
        assert not np.array_equal(a, c['map'])
        os.environ['BFGX_GRID_ITEM_CAP'] = '5'                           # block lists start too small and are regrown
        try:
            small = engine.GridPlan(m, keep, c['bins'], ndim, c['redshift'], nh, 0, stream)
            out.fill_(float('nan'))
            assert small.baryonify(dcat, m_in.data_ptr(), out.data_ptr()) == n_ref
            torch.cuda.synchronize()
            assert np.abs(a - out.cpu().numpy()).max() <= 1e-12 * np.abs(a).max()
            small.close()
        finally:
            del os.environ['BFGX_GRID_ITEM_CAP']
    plan.close()


@pytest.mark.parametrize('ndim,N,nh,npart,mass', [(3, 64, 600, 1_200_000, False), (3, 61, 400, 1_100_000, True), (2, 250, 40, 1_050_000, False),
                                                    (3, 64, 600, 5000, False)])
def test_fused_deposit_baryonify_equals_the_two_calls(gpu, ndim, N, nh, npart, mass):
    """bfgx_grid_deposit_baryonify_device (make_map + BaryonifyGrid: the deposit's last kernel also stores the start value of map_out and
    the map's sum) == bfgx_deposit_particles_device followed by bfgx_grid_baryonify_device: map_in bit for bit, map_out to the order of
    the fp64 sums; the tile-owned deposit (>= 2^20 particles) and the atomic one (small inputs: the copy pass runs as before)"""
    import torch
    from baryonification_amd import _lib, engine
    c = _big_case(ndim, N, nh, 23)
    cat = c['cat']
    cos = dict(c['cosmo'], w0=-1.0)
    m, keep = engine.model_from_tables([np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])], c['d'], cos, 6.0, 8.0)
    dev = torch.device('cuda:0')
    t = {k: torch.tensor(cat[k], dtype=torch.float64, device=dev) for k in ('M', 'x', 'y', 'z')}
    lnM = torch.tensor(np.log(cat['M'].astype(np.float32)).astype(np.float64), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    plan = engine.GridPlan(m, keep, c['bins'], ndim, c['redshift'], nh, 0, stream)
    dcat = _lib.make_grid_catalog_dev(nh, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr() if ndim == 3 else 0,
                                      lnM.data_ptr())
    bins = np.asarray(c['bins'], dtype=np.float64)
    res = bins[1] - bins[0]
    edges = np.concatenate([bins - res / 2, [bins[-1] + res / 2]])
    rng = np.random.default_rng(4)
    pos = rng.uniform(edges[0] - 0.01 * res, edges[-1] + 0.01 * res, (3, npart))          # a few particles outside the box: dropped
    pos[:, :100] = edges[3]                                                                  # some exactly on a bin edge
    part = torch.tensor(pos, device=dev)
    pm = torch.tensor(rng.uniform(0.5, 2.0, npart), device=dev) if mass else None
    d_edges = torch.tensor(edges, device=dev)
    a_in, a_out, b_in, b_out = (torch.full((N ** ndim,), float('nan'), dtype=torch.float64, device=dev) for _ in range(4))
    s_a, s_b = (torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(2))
    engine.deposit_particles_device(part[0].data_ptr(), part[1].data_ptr(), part[2].data_ptr() if ndim == 3 else 0, pm.data_ptr() if mass else 0,
                                    npart, N, d_edges.data_ptr(), a_in.data_ptr(), ndim, 0, stream)
    n_a = plan.baryonify(dcat, a_in.data_ptr(), a_out.data_ptr(), s_a.data_ptr())
    n_b = plan.deposit_baryonify(dcat, npart, part[0].data_ptr(), part[1].data_ptr(), part[2].data_ptr() if ndim == 3 else 0,
                                 pm.data_ptr() if mass else 0, d_edges.data_ptr(), b_in.data_ptr(), b_out.data_ptr(), s_b.data_ptr())
    torch.cuda.synchronize()
    ai, ao, bi, bo = (v.cpu().numpy() for v in (a_in, a_out, b_in, b_out))
    assert n_a == n_b > 0 and np.isfinite(bo).all()
    # unit masses: every cell is an integer count, identical bit for bit; with masses the LDS adds of a tile arrive in any order
    assert (np.abs(ai - bi).max() <= 1e-13 * ai.max() and ai.sum() > 0.5 * npart) if mass else (np.array_equal(ai, bi) and 0.9 * npart < ai.sum() < npart)
    assert np.abs(ao - bo).max() <= 1e-12 * np.abs(ao).max() and not np.array_equal(bo, bi)
    sa, sb = s_a.cpu().numpy(), s_b.cpu().numpy()
    assert np.allclose(sa, sb, rtol=1e-12, atol=0) and np.isclose(sb[0], bi.sum(), rtol=1e-12) and np.isclose(sb[1], sb[0], rtol=1e-10)
    plan.close()


def test_grid_cell_owned_pass_refuses_what_it_cannot_do(gpu):
    """a plan that owns a slab, a plan holding a profile (log) table and a catalog larger than the plan fail loudly"""
    import torch
    from baryonification_amd import _lib, engine
    c = _big_case(3, 32, 20, 3)
    cat = c['cat']
    cos = dict(c['cosmo'], w0=-1.0)
    axes = [np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])]
    m, keep = engine.model_from_tables(axes, c['d'], cos, 6.0, 8.0)
    dev = torch.device('cuda:0')
    t = {k: torch.tensor(cat[k], dtype=torch.float64, device=dev) for k in ('M', 'x', 'y', 'z')}
    dcat = _lib.make_grid_catalog_dev(20, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr(), 0)
    m_in = torch.tensor(c['map'], device=dev)
    m_out = torch.empty_like(m_in)
    plan = engine.GridPlan(m, keep, c['bins'], 3, c['redshift'], 20, 0, torch.cuda.current_stream().cuda_stream)
    assert plan.baryonify(dcat, m_in.data_ptr(), m_out.data_ptr()) > 0
    plan.set_slab(8, 8)
    with pytest.raises(ValueError, match='slab'):
        plan.baryonify(dcat, m_in.data_ptr(), m_out.data_ptr())
    plan.close()
    small = engine.GridPlan(m, keep, c['bins'], 3, c['redshift'], 10, 0, torch.cuda.current_stream().cuda_stream)
    with pytest.raises(ValueError, match='max_halos'):
        small.baryonify(dcat, m_in.data_ptr(), m_out.data_ptr())
    small.close()
    mp, keep_p = engine.model_from_tables(axes, np.log(c['P']), cos, 6.0, 8.0, log_values=True)
    paint = engine.GridPlan(mp, keep_p, c['bins'], 3, c['redshift'], 20, 0, torch.cuda.current_stream().cuda_stream)
    with pytest.raises(ValueError, match='paint'):
        paint.baryonify(dcat, m_in.data_ptr(), m_out.data_ptr())
    paint.close()


def test_grid_one_shot_scatter_path(gpu):
    """BFGX_GRID_PATH=scatter: the one-shot API through the halo-owned kernels (pix_offsets array + regrid, what the slab and
    multi-GPU entry points run) == the default cell-owned pass == the committed golden"""
    for name in ('grid3d_baryonify', 'grid2d_baryonify'):
        c = H.load_grid_golden(name)
        base = H.grid_product_runner(c).process()
        os.environ['BFGX_GRID_PATH'] = 'scatter'
        try:
            again = H.grid_product_runner(c).process()
        finally:
            del os.environ['BFGX_GRID_PATH']
        assert np.abs(again - base).max() <= 1e-12 * np.abs(base).max()
        assert np.abs(again - c['expected']).max() <= 1e-10 * np.abs(c['expected']).max()


def test_grid_edge_cases(gpu):
    """empty catalog, a single halo whose cutout is clipped to half the box, an all-zero map"""
    import baryonification_amd as bfg
    from oracle import grid as G
    from oracle import oracle as O
    c = _big_case(3, 32, 4, 11)
    cos = c['cosmo']
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(cos), epsilon_max=50.0)
    model.set_table(c['z'], c['Mt'], c['r'], c['d'])
    prof = bfg.utils.TabulatedProfile(None, bfg.utils.Cosmology.from_dict(cos))
    prof.set_table(c['z'], c['Mt'], c['r'], c['P'])
    GMap = bfg.utils.GriddedMap(map=c['map'], redshift=c['redshift'], bins=c['bins'], cosmo=cos)
    none = bfg.utils.HaloNDCatalog(x=np.zeros(0), y=np.zeros(0), z=np.zeros(0), M=np.zeros(0), redshift=c['redshift'], cosmo=cos)
    out = bfg.Runners.BaryonifyGrid(none, GMap, 5.0, model, verbose=False).process()
    assert np.array_equal(out, c['map'])                                  # zero offsets: every pixel deposits into itself
    assert np.all(bfg.Runners.PaintProfilesGrid(none, GMap, 5.0, prof, verbose=False).process() == 0)
    # one cluster with epsilon_max = 40: R_q is clipped to max(bins)/2 and the cutout spans (almost) the whole box
    one = bfg.utils.HaloNDCatalog(x=np.array([31.0]), y=np.array([3.0]), z=np.array([60.0]), M=np.array([9e14]), redshift=c['redshift'], cosmo=cos)
    used = {k: np.array(one.cat[k], dtype=np.float64) for k in ('M', 'x', 'y', 'z')}
    r1 = bfg.Runners.BaryonifyGrid(one, GMap, 40.0, model, verbose=False)
    out = r1.process()
    tab = O.Table([np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])], c['d'], False, 50.0)
    ora = G.baryonify_grid(c['map'], c['bins'], used, c['redshift'], tab, 40.0, G.grid_background(cos))
    assert np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max() and r1.last_stats['n_pairs'] > 20000
    zero = bfg.utils.GriddedMap(map=np.zeros((32,) * 3), redshift=c['redshift'], bins=c['bins'], cosmo=cos)
    assert np.all(bfg.Runners.BaryonifyGrid(one, zero, 40.0, model, verbose=False).process() == 0)


# ------------------------------------------------------------------------------------------ BASELINE config 5 sizes
def test_config5_size_512_cubed(gpu):
    """512^3 (BASELINE config 5's grid): the 512-point LDS FFT lines and the 512^3 deposit / regrid meet the oracle.
    P(k) of a 512^3 map against the numpy restatement of the notebook cells (np.fft.fftn, np.bincount); make_map of 3e6
    particles against the oracle's histogramdd; regrid_pixels_3D of 2e6 displaced pixels against the oracle's 5^3-cell loops."""
    import baryonification_amd as bfg
    from baryonification_amd.engine import power_spectrum
    from oracle import grid as G
    N, L = 512, 1000.0
    rng = np.random.default_rng(512)
    # (a) deposit
    npart = 3_000_000
    xyz = rng.uniform(0, L, (npart, 3))
    xyz[:5] = [[0, 0, 0], [L, L, L], [L / 2, 0, L], [1e-12, L - 1e-12, 500.0], [L / N, 2 * L / N, 3 * L / N]]     # edges: inclusive last edge
    mass = rng.uniform(0.5, 2.0, npart)
    Snap = bfg.utils.ParticleSnapshot(x=xyz[:, 0], y=xyz[:, 1], z=xyz[:, 2], M=mass, L=L, redshift=0.0,
                                      cosmo=H.load_grid_golden('grid2d_paint')['cosmo_runner'])
    Map = Snap.make_map(N)
    ora = G.make_map([xyz[:, 0], xyz[:, 1], xyz[:, 2]], mass, L, N)
    assert Map.shape == (N, N, N) and np.abs(Map - ora).max() <= 1e-12 * ora.max() and np.array_equal(Map != 0, ora != 0)
    assert np.isclose(Map.sum(), mass.sum(), rtol=1e-12)
    del ora
    # (b) P(k): every 512-point line of the LDS FFT, all 180 bins
    k_cen, Pk, k_c = power_spectrum(Map, L, 180)
    ok_cen, oPk, ok_c = G.power_spectrum(Map, L, 180)
    assert np.array_equal(k_c, ok_c) and np.all(ok_c > 0)
    assert np.abs(Pk / oPk - 1).max() <= 1e-10 and np.abs(k_cen / ok_cen - 1).max() <= 1e-11      # (means over ~1e6 modes per bin)
    del Map
    # (c) regrid: 2e6 pixels of the 512^3 grid displaced by up to two cells, some across the periodic faces
    n = 2_000_000
    idx = rng.choice(N ** 3, n, replace=False)
    base = np.stack(np.unravel_index(idx, (N, N, N)), axis=1).astype(np.float64)
    pos = base + rng.normal(0, 0.7, (n, 3))
    pos[:1000] = base[:1000]                              # undisplaced pixels deposit into themselves
    val = rng.uniform(0.1, 3.0, n)
    grid = np.zeros((N, N, N))
    bfg.Runners.regrid_pixels_3D(grid, pos, val)
    ora = G.regrid_pixels(np.zeros((N, N, N)), pos, val)
    assert np.abs(grid - ora).max() <= 1e-12 * ora.max() and np.isclose(grid.sum(), val.sum(), rtol=1e-12)


def test_config5_halo_loop_512_cubed_cell_owned_vs_scatter_and_oracle(gpu):
    """BASELINE config 5 exactly as bench.py --mode grid3d builds it (512^3 cells, 1e5 halos, closed-form table): the cell-owned
    BaryonifyGrid pass the bench times (bfgx_grid_baryonify_device, no pix_offsets array) == the scatter form (offsets + regrid,
    oracle-pinned at 80^3) at 1e-12; the scatter form's pix_offsets == the oracle's FULL halo loop (Map2DRunner.py:539-575); the
    regrid of a 4M-cell sample of source cells == the oracle's 5^3-cell loops (Map2DRunner.py:86-163); the mass sums."""
    import torch
    from baryonification_amd import _lib, engine, synthetic as syn
    from oracle import grid as G
    from oracle import oracle as O
    dev = torch.device('cuda:0')
    N, nh, eps, zr = 512, 100_000, 5.0, 0.0
    L = 205.0 / syn.COSMO['h']
    bins = (np.arange(N) + 0.5) * (L / N)
    rng = np.random.default_rng(syn.SEED_CATALOG)
    M = syn.make_catalog(nh, seed=syn.SEED_CATALOG)['M'].astype(np.float32).astype(np.float64)
    pos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
    cat = {'M': M, 'x': pos[:, 0].copy(), 'y': pos[:, 1].copy(), 'z': pos[:, 2].copy()}
    z, Mt, r = np.array([0.0, 0.01]), np.geomspace(0.99e12, 1.01e15, 10), np.geomspace(1e-3, 3e2, 500)
    table = syn.displacement_table(z, Mt, r)
    axes = [np.log(1 + z), np.log(Mt), np.log(r)]
    model, keep = engine.model_from_tables(axes, table, dict(syn.COSMO, w0=-1.0), eps, eps)
    stream = torch.cuda.current_stream().cuda_stream
    plan = engine.GridPlan(model, keep, bins, 3, zr, nh, device=0, stream=stream)
    t = {k: torch.from_numpy(v).to(dev) for k, v in cat.items()}
    lnM = torch.from_numpy(np.log(M.astype(np.float32)).astype(np.float64)).to(dev)
    cat_dev = _lib.make_grid_catalog_dev(nh, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr(), lnM.data_ptr())
    torch.manual_seed(syn.SEED_MAP)
    d_map = torch.poisson(torch.full((N ** 3,), 8.0, dtype=torch.float64, device=dev))       # ~0.03 % empty cells
    d_off = torch.empty(N ** 3 * 3, dtype=torch.float64, device=dev)
    d_cell = torch.full((N ** 3,), float('nan'), dtype=torch.float64, device=dev)
    d_scat = torch.full((N ** 3,), float('nan'), dtype=torch.float64, device=dev)
    s_cell, s_scat = (torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(2))
    n_cell = plan.baryonify(cat_dev, d_map.data_ptr(), d_cell.data_ptr(), s_cell.data_ptr())
    n_scat = plan.offsets(cat_dev, d_off.data_ptr())
    plan.regrid(d_map.data_ptr(), d_off.data_ptr(), d_scat.data_ptr(), s_scat.data_ptr())
    torch.cuda.synchronize()
    total = float(d_map.sum().item())
    scale = float(d_scat.abs().max().item())
    # (1) the two forms of the halo loop + regrid
    assert n_cell == n_scat > 1e7
    assert torch.isfinite(d_cell).all().item() and (d_cell - d_scat).abs().max().item() <= 1e-12 * scale
    assert not torch.equal(d_cell, d_map)
    # (4) mass: HealpixRunner-style assert of Map2DRunner.py:601-605, from the kernels' own sums and from the maps
    for s in (s_cell.cpu().numpy(), s_scat.cpu().numpy()):
        assert np.isclose(s[0], total, rtol=1e-12) and np.isclose(s[1], s[0], rtol=1e-11)
    assert np.isclose(float(d_cell.sum().item()), total, rtol=1e-11)
    del d_cell
    # (2) pix_offsets against the oracle's full halo loop (1.5e7 contributing cutout cells)
    tab = O.Table(axes, table, False, eps)
    ora_off, ora_pairs = G.baryonify_grid_offsets((N, N, N), bins, cat, zr, tab, eps, G.grid_background(syn.COSMO), return_pairs=True)
    assert ora_pairs >= n_scat                  # (the oracle counts every pixel of the cutouts, the GPU the contributing ones)
    off = d_off.cpu().numpy().reshape(-1, 3)
    assert np.array_equal(np.isfinite(off), np.isfinite(ora_off))               # NaN-poisoned cells are the same cells
    fin = np.isfinite(ora_off)
    assert np.abs(off[fin] - ora_off[fin]).max() <= 1e-10 * np.abs(ora_off[fin]).max()
    assert np.count_nonzero(ora_off[fin]) > 1e7
    # (3) regrid of a 4M-cell sample of SOURCE cells (the first 16 planes): GPU regrid of the map zeroed elsewhere vs the oracle
    ns = 1 << 22
    masked = torch.zeros_like(d_map)
    masked[:ns] = d_map[:ns]
    plan.regrid(masked.data_ptr(), d_off.data_ptr(), d_scat.data_ptr(), s_scat.data_ptr())
    torch.cuda.synchronize()
    got = d_scat.cpu().numpy().reshape(N, N, N)
    p = np.arange(ns)
    i, j, k = p // (N * N), (p // N) % N, p % N
    own = np.stack([j, i, k], axis=1).astype(np.float64)         # meshgrid(indexing='xy'): column 0 carries the SECOND array axis
    chk = np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing='xy')
    assert [int(g.flatten()[1 * 9 + 2 * 3 + 0]) for g in chk] == [2, 1, 0]
    posn = np.where(np.isfinite(off[:ns]), off[:ns], 0) + own           # the same offsets the GPU regrid read (pinned in (2))
    ora = G.regrid_pixels(np.zeros((N, N, N)), posn, d_map[:ns].cpu().numpy())
    assert np.abs(got - ora).max() <= 1e-12 * ora.max() and np.isclose(got.sum(), ora.sum(), rtol=1e-12)
    plan.close()


# ------------------------------------------------------------------------------------------ slab decomposition (config 5 over N GPUs)
@pytest.mark.parametrize('N,W', [(64, 2), (64, 4), (128, 8)])
def test_slab_kernels_side_by_side_equal_the_full_grid(gpu, N, W, monkeypatch):
    """The per-rank pieces of utils/GridSlabs on ONE GPU, the W ranks played one after the other and the exchanges done with
    torch indexing: slab deposit, slab halo loop (cutouts clipped to the slab), slab regrid with aprons (added periodically to
    the neighbours), plane FFTs -> transpose -> axis-0 FFT + partial P(k) sums == the single-GPU pipeline on the full grid."""
    import torch
    from baryonification_amd import _lib, engine, synthetic as syn
    from baryonification_amd.utils.GridSlabs import HipBackend
    monkeypatch.setenv('BFGX_DEPOSIT', 'tiled')
    dev = torch.device('cuda:0')
    L, nh, eps, Nk = 120.0, 400, 5.0, 24
    rng = np.random.default_rng(N * W)
    bins = (np.arange(N) + 0.5) * (L / N)
    edges = torch.from_numpy(np.linspace(0, L, N + 1)).to(dev)
    M = (10 ** rng.uniform(13.0, 14.8, nh)).astype(np.float32).astype(np.float64)
    pos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
    pos[:4] = [[0.3, 60.0, 60.0], [L - 0.2, 10.0, 100.0], [L / W, 50.0, 50.0], [L / W - 0.01, 5.0, 5.0]]     # balls across slab faces / the box face
    z, Mt, r = np.array([0.0, 0.01]), np.geomspace(0.99e12, 1.01e15, 10), np.geomspace(1e-3, 3e2, 500)
    table = syn.displacement_table(z, Mt, r) * (180.0 * 64 / N)         # displacements of up to ~2 cells: aprons in use
    model, keep = engine.model_from_tables([np.log(1 + z), np.log(Mt), np.log(r)], table, dict(syn.COSMO, w0=-1.0), eps, eps)
    t = {k: torch.from_numpy(v.copy()).to(dev) for k, v in (('M', M), ('x', pos[:, 0]), ('y', pos[:, 1]), ('z', pos[:, 2]))}
    lnM = torch.from_numpy(np.log(M.astype(np.float32)).astype(np.float64)).to(dev)
    cat_dev = _lib.make_grid_catalog_dev(nh, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr(), lnM.data_ptr())
    npart = 300_000
    rows = torch.from_numpy(np.concatenate([rng.uniform(-0.01 * L, 1.01 * L, (3, npart)), rng.uniform(0.5, 2.0, (1, npart))], axis=0)).to(dev)
    rows[:3, 0] = 0.0
    rows[:3, 1] = L

    be = HipBackend(model, keep, bins, 0.0, nh, device=0)
    # ---- the full grid on one GPU
    full_map = be.deposit(rows, edges, 0, N)
    full_off = be.offsets(cat_dev, 0, N)
    pairs_full = be.pairs
    full_buf, full_sums, missed = be.regrid(full_map, full_off, 0, 0, N)
    assert int(missed.item()) == 0 and pairs_full > 0
    amax = float(torch.nan_to_num(full_off[..., 1], nan=0.0, posinf=0.0, neginf=0.0).abs().max().item())
    assert 0.5 < amax < 8.0, amax
    work = torch.empty(engine.power_spectrum_work_doubles(N), dtype=torch.float64, device=dev)
    sums_full = torch.zeros((2, Nk), dtype=torch.float64, device=dev)
    cnt_full = torch.zeros(Nk, dtype=torch.int64, device=dev)
    engine.power_spectrum_device(full_buf.data_ptr(), N, L, Nk, work.data_ptr(), sums_full[0].data_ptr(), sums_full[1].data_ptr(), cnt_full.data_ptr())
    torch.cuda.synchronize()

    # ---- W slabs, one after the other
    cnt = N // W
    apron = min(int(np.ceil(amax)) + 1, (N - cnt) // 2)
    new = torch.zeros_like(full_buf)
    pairs = 0
    s_in = s_out = 0.0
    for rk in range(W):
        lo = rk * cnt
        slab = be.deposit(rows, edges, lo, cnt)                           # particles of other planes are dropped
        assert (slab - full_map[lo:lo + cnt]).abs().max().item() <= 1e-13 * full_map.max().item()      # (order of the LDS additions)
        off = be.offsets(cat_dev, lo, cnt)
        pairs += be.pairs
        a_, b_ = torch.nan_to_num(off, nan=-7.0), torch.nan_to_num(full_off[lo:lo + cnt], nan=-7.0)
        assert (a_ - b_).abs().max().item() <= 1e-12 * b_.abs().max().item()                             # (order of the atomics)
        buf, sums, missed = be.regrid(slab, off, apron, lo, cnt)
        assert int(missed.item()) == 0
        planes = (torch.arange(lo - apron, lo + cnt + apron, device=dev) % N)
        new.index_add_(0, planes, buf)                                    # own planes + what exchange_aprons hands to the neighbours
        s_in += sums[0].item(); s_out += sums[1].item()
    assert pairs == pairs_full                                            # clipped cutouts: every contributing (halo, cell) pair exactly once
    assert np.isclose(s_in, full_sums[0].item(), rtol=1e-13) and np.isclose(s_out, full_sums[1].item(), rtol=1e-12)
    assert (new - full_buf).abs().max().item() <= 1e-12 * full_buf.abs().max().item()
    # an apron that is too thin is reported, not silently dropped
    if amax > 1.0:
        _, _, missed = be.regrid(full_map[:cnt].contiguous(), full_off[:cnt].contiguous(), 0, 0, cnt)
        assert int(missed.item()) == 1
    # P(k): plane passes per slab, transpose, axis-0 pass + partial sums per column block
    works = [be.fft_planes(full_buf[rk * cnt:(rk + 1) * cnt].contiguous()) for rk in range(W)]
    allw = torch.cat(works, dim=0)                                        # [N][N][nz] after the plane passes
    psum = torch.zeros((2, Nk), dtype=torch.float64, device=dev)
    pcnt = torch.zeros(Nk, dtype=torch.int64, device=dev)
    for rk in range(W):
        cols = allw[:, rk * cnt:(rk + 1) * cnt, :].contiguous()           # what transpose_planes_to_columns delivers to rank rk
        s2, c2 = be.fft_axis0_pk(cols, rk * cnt, L, Nk)
        psum += s2; pcnt += c2
    torch.cuda.synchronize()
    assert torch.equal(pcnt, cnt_full) and int(pcnt.sum().item()) > 0
    ok = cnt_full > 0
    assert ((psum - sums_full).abs()[:, ok] <= 1e-11 * sums_full.abs()[:, ok]).all().item()
    be.close()


@pytest.mark.parametrize('world,ncols', [(1, 3), (4, 4), (8, 3)])
def test_particle_routing_kernels(gpu, world, ncols):
    """bfgx_route_particles_count_device / _fill_device (slab decomposition over GPUs): owners follow np.histogramdd's bin rule on the
    first coordinate, particles outside the edges are dropped, every destination receives exactly its particles (any order)"""
    import torch
    from baryonification_amd import engine
    dev = torch.device('cuda:0')
    N, L, n = 64, 100.0, 700_000
    rng = np.random.default_rng(world)
    cols = rng.uniform(-0.02 * L, 1.02 * L, (ncols, n))
    cols[0, :5] = [0.0, L, L / 2, np.nextafter(L, 2 * L), -1e-12]           # first edge, last edge (inclusive), an inner edge, just outside
    edges = np.linspace(0, L, N + 1)
    d_cols = torch.from_numpy(cols).to(dev)
    d_edges = torch.from_numpy(edges).to(dev)
    owner = torch.empty(n, dtype=torch.uint8, device=dev)
    counts = torch.empty(world, dtype=torch.int32, device=dev)
    engine.route_particles_count_device(d_cols[0].data_ptr(), n, N, d_edges.data_ptr(), world, owner.data_ptr(), counts.data_ptr())
    b = np.searchsorted(edges, cols[0], side='right') - 1
    b[cols[0] == edges[-1]] = N - 1
    inside = (b >= 0) & (b < N)
    want_owner = np.where(inside, b // (N // world), 255)
    assert np.array_equal(owner.cpu().numpy(), want_owner)
    c = counts.cpu().numpy().astype(np.int64)
    assert np.array_equal(c, np.bincount(want_owner[inside], minlength=world)) and c.sum() < n
    total = int(c.sum())
    start = np.concatenate([[0], np.cumsum(c)[:-1]])
    packed = torch.full((ncols, total), float('nan'), dtype=torch.float64, device=dev)
    cursor = torch.empty(world, dtype=torch.int32, device=dev)
    engine.route_particles_fill_device([d_cols[i].data_ptr() for i in range(ncols)], n, owner.data_ptr(), world, start, total, cursor.data_ptr(),
                                       packed.data_ptr())
    got = packed.cpu().numpy()
    assert np.isfinite(got).all()
    for d in range(world):
        sel = np.nonzero(want_owner == d)[0]
        blk = got[:, start[d]:start[d] + c[d]]
        order_g, order_w = np.lexsort(blk[::-1]), np.lexsort(cols[:, sel][::-1])
        assert np.array_equal(blk[:, order_g], cols[:, sel][:, order_w])          # the same particles, column by column


def test_callable_models_on_the_grid_and_snapshot_runners(gpu):
    """Map2DRunner.py:534, :752, :781 and SnapshotRunner.py:228 call model.displacement / .projected / .real per halo on any object.
    Plain-Python models go through BaryonifyGrid, PaintProfilesGrid (2D: projected, 3D: real) and BaryonifySnapshot: each is tabulated
    once (two redshift slices around the box's own, the catalog's mass range) and the result equals the oracle fed that tabulation"""
    import baryonification_amd as bfg
    from baryonification_amd.Runners import _model as RM
    from oracle import grid as G
    from oracle import oracle as O
    c = _big_case(2, 256, 900, 41)
    cat, cos, zr = c['cat'], c['cosmo'], c['redshift']
    bg = G.grid_background(cos)

    class PlainDisplacement(object):
        epsilon_max = 7.0
        calls = 0

        def displacement(self, r, M, a):
            type(self).calls += 1
            Rc = bg.get_radius(np.atleast_1d(M), a)[0] / a
            x = np.asarray(r) / Rc
            return np.where(x < self.epsilon_max, -0.08 * Rc * x * np.exp(-x) / (1 + x * x), 0.0)

    class PlainProfile(object):
        def _f(self, r, M, a, s):
            Rc = bg.get_radius(np.atleast_1d(M), a) / a
            x = np.asarray(r)[None, :] / Rc[:, None]
            out = s * (np.atleast_1d(M)[:, None] / 1e14) * np.exp(-x) / (1 + x) ** 2
            return out if np.ndim(M) else out[0]

        def projected(self, cosmo, r, M, a):
            return self._f(r, M, a, 1.0)

        def real(self, cosmo, r, M, a):
            return self._f(r, M, a, 3.0)

    def table_of(holder, values, eps=None):
        axes = [holder.raw_input_z_range, holder.raw_input_M_range, holder.raw_input_r_range]
        if eps is not None:
            return O.Table(axes, values, False, eps)
        with np.errstate(divide='ignore'):
            return O.Table(axes, np.log(values))

    HCat2 = bfg.utils.HaloNDCatalog(x=cat['x'], y=cat['y'], z=None, M=cat['M'], redshift=zr, cosmo=cos)
    GMap2 = bfg.utils.GriddedMap(map=c['map'], redshift=zr, bins=c['bins'], cosmo=cos)
    model = PlainDisplacement()
    runner = bfg.Runners.BaryonifyGrid(HCat2, GMap2, 5.0, model, verbose=False)
    out = runner.process()
    assert PlainDisplacement.calls == 2 * RM.BRIDGE_N_M
    hold = model._bfgx_tabulated[1]
    assert hold.raw_input_z_range[0] < np.log(1 + zr) < hold.raw_input_z_range[-1]
    ora = G.regrid_offsets(c['map'], G.baryonify_grid_offsets(c['map'].shape, c['bins'], cat, zr, table_of(hold, hold.raw_input_d, 7.0), 5.0, bg))
    assert np.abs(out - c['map']).max() > 0 and np.abs(out - ora).max() <= 1e-10 * np.abs(ora).max()
    runner.process()
    assert PlainDisplacement.calls == 2 * RM.BRIDGE_N_M                               # tabulated once

    prof = PlainProfile()
    zero2 = bfg.utils.GriddedMap(map=np.zeros_like(c['map']), redshift=zr, bins=c['bins'], cosmo=cos)
    p2 = bfg.Runners.PaintProfilesGrid(HCat2, zero2, 4.0, prof, verbose=False).process()
    hp2 = prof._bfgx_tabulated[1]
    ora2 = G.paint_grid(c['map'].shape, c['bins'], cat, zr, table_of(hp2, hp2.raw_input_2D), 4.0, bg)
    assert p2.max() > 0 and np.abs(p2 - ora2).max() <= 1e-10 * np.abs(ora2).max()

    c3 = _big_case(3, 64, 400, 42)
    cat3 = c3['cat']
    HCat3 = bfg.utils.HaloNDCatalog(x=cat3['x'], y=cat3['y'], z=cat3['z'], M=cat3['M'], redshift=zr, cosmo=cos)
    zero3 = bfg.utils.GriddedMap(map=np.zeros((64,) * 3), redshift=zr, bins=c3['bins'], cosmo=cos)
    prof3 = PlainProfile()
    p3 = bfg.Runners.PaintProfilesGrid(HCat3, zero3, 4.0, prof3, verbose=False).process()
    hp3 = prof3._bfgx_tabulated[1]
    ora3 = G.paint_grid((64,) * 3, c3['bins'], cat3, zr, table_of(hp3, hp3.raw_input_3D), 4.0, bg)
    assert p3.max() > 0 and np.abs(p3 - ora3).max() <= 1e-10 * np.abs(ora3).max()
    # the 3D table holds real(), not projected(): three times larger in this model
    j = int(np.argmax(cat3['M']))
    rr = np.geomspace(0.05, 3.0, 5)
    assert np.abs(hp3.real(None, rr, cat3['M'][j], 1 / (1 + zr)) / prof3.real(None, rr, cat3['M'][j], 1 / (1 + zr)) - 1).max() < 2e-3

    rng = np.random.default_rng(43)
    L, npart = c3['L'], 60_000
    part = rng.uniform(0, L, (npart, 3))
    Snap = bfg.utils.ParticleSnapshot(x=part[:, 0], y=part[:, 1], z=part[:, 2], M=np.ones(npart), L=L, redshift=zr, cosmo=cos)
    smodel = PlainDisplacement()
    new = bfg.Runners.BaryonifySnapshot(HCat3, Snap, 5.0, smodel, verbose=False).process()
    sh = smodel._bfgx_tabulated[1]
    oras = G.baryonify_snapshot([part[:, k] for k in range(3)], L, cat3, zr, table_of(sh, sh.raw_input_d, 7.0), 5.0, bg)
    got = np.stack([new[k] for k in 'xyz'], axis=1)
    oras = np.stack(oras, axis=1)
    scale = np.abs(oras - part).max()
    assert scale > 1e-3 and np.abs(got - oras).max() <= max(1e-10 * scale, 1e-13 * L)
    with pytest.raises(TypeError):
        bfg.Runners.BaryonifyGrid(HCat2, GMap2, 5.0, object(), verbose=False).process()


@pytest.mark.parametrize('ndim,N,nh,chunks', [(3, 64, 700, 4), (3, 61, 500, 3), (3, 96, 900, 6), (2, 250, 60, 5), (2, 512, 300, 8)])
def test_grid_host_entry_in_plane_ranges_equals_one_pass(gpu, monkeypatch, ndim, N, nh, chunks):
    """BaryonifyGrid.process() streams the map in ranges of whole block rows (upload, copy + gather, download overlapped; the first block
    row gathered last because the grid is periodic) == the one-pass route == the oracle; plans and device maps are cached between calls"""
    import baryonification_amd as bfg
    c = _big_case(ndim, N, nh, 31)
    cat = c['cat']
    # halos on the first / last planes so that deposits cross the periodic face and the range boundaries
    cat['x'][:20] = np.float32(0.01 * c['L'] / N); cat['y'][20:40] = np.float32(c['L'] * (1 - 0.01 / N))
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(c['cosmo']), epsilon_max=8.0)
    model.set_table(c['z'], c['Mt'], c['r'], c['d'])
    kw = dict(x=cat['x'], y=cat['y'], M=cat['M'], redshift=c['redshift'], cosmo=c['cosmo'])
    if ndim == 3:
        kw['z'] = cat['z']
    HCat = bfg.utils.HaloNDCatalog(**kw)
    GMap = bfg.utils.GriddedMap(map=c['map'], redshift=c['redshift'], bins=c['bins'], cosmo=c['cosmo'])
    runner = bfg.Runners.BaryonifyGrid(HCat, GMap, 6.0, model, verbose=False)
    monkeypatch.setenv('BFGX_NO_PIPELINE', '1')
    one = runner.process().copy()
    monkeypatch.delenv('BFGX_NO_PIPELINE')
    monkeypatch.setenv('BFGX_PIPE_CHUNKS', str(chunks))
    piped = runner.process().copy()
    again = runner.process().copy()                                     # warm: cached plan, pooled pinned result
    assert np.abs(piped - one).max() <= 1e-12 * np.abs(one).max() and np.abs(again - one).max() <= 1e-12 * np.abs(one).max()
    assert not np.array_equal(one, c['map']) and np.isclose(piped.sum(), c['map'].sum())
    cos = dict(c['cosmo'], w0=-1.0)                                     # the grid runners drop w0 (Map2DRunner.py:456-459)
    ocat = {k: np.asarray(HCat.cat[k], dtype=np.float64) for k in (('M', 'x', 'y', 'z') if ndim == 3 else ('M', 'x', 'y'))}
    ocat.setdefault('z', np.zeros(nh))
    ora = H.grid_oracle_run(dict(kind='baryonify', tab_z=c['z'], tab_M=c['Mt'], tab_r=c['r'], tab_values=c['d'], rdelta=False, eps_model=8.0,
                                 map_in=c['map'], bins=c['bins'], cat=ocat, redshift=c['redshift'], eps_runner=6.0, cosmo_runner=cos,
                                 cosmo_model=c['cosmo'], rmat=None))
    assert np.abs(piped - ora).max() <= 1e-10 * np.abs(ora).max()


@pytest.mark.parametrize('ndim,N,nh,chunks,scale', [(3, 96, 300, 6, 3000.0), (2, 512, 200, 8, 6000.0)])
def test_grid_host_entry_in_plane_ranges_with_large_moves_along_the_first_axis(gpu, monkeypatch, ndim, N, nh, chunks, scale):
    """The streamed route gathers the map in ranges of planes of the first array axis, and a deposit may only travel 2^S - 1 cells (7 in 3-D, 15
    in 2-D) along that axis before it could land in planes that have not arrived yet or have already left (the mass check would not notice:
    its sums are arithmetic).  A table scaled until cells move by tens of cells: the gather kernel flags it and the call repeats the regrid in
    one pass on the uploaded map -- the result equals the one-pass route; a table with small moves does not take the fallback"""
    import baryonification_amd as bfg
    from baryonification_amd import _lib
    lib = _lib.load()
    c = _big_case(ndim, N, nh, 77)
    cat = c['cat']
    for sc, expect_fallback in ((scale, True), (1.0, False)):
        model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(c['cosmo']), epsilon_max=8.0)
        model.set_table(c['z'], c['Mt'], c['r'], sc * c['d'])
        kw = dict(x=cat['x'], y=cat['y'], M=cat['M'], redshift=c['redshift'], cosmo=c['cosmo'])
        if ndim == 3:
            kw['z'] = cat['z']
        HCat = bfg.utils.HaloNDCatalog(**kw)
        GMap = bfg.utils.GriddedMap(map=c['map'], redshift=c['redshift'], bins=c['bins'], cosmo=c['cosmo'])
        runner = bfg.Runners.BaryonifyGrid(HCat, GMap, 6.0, model, verbose=False)
        monkeypatch.setenv('BFGX_NO_PIPELINE', '1')
        one = runner.process().copy()
        monkeypatch.delenv('BFGX_NO_PIPELINE')
        monkeypatch.setenv('BFGX_PIPE_CHUNKS', str(chunks))
        n0 = lib.bfgx_debug_grid_pipe_fallbacks()
        piped = runner.process().copy()
        n1 = lib.bfgx_debug_grid_pipe_fallbacks()
        monkeypatch.delenv('BFGX_PIPE_CHUNKS')
        assert (n1 - n0 == 1) == expect_fallback, (sc, n0, n1)
        assert np.isclose(piped.sum(), c['map'].sum()) and not np.array_equal(one, c['map'])
        assert np.abs(piped - one).max() <= 1e-12 * np.abs(one).max()
        if expect_fallback:
            # the moves really are large: mass has travelled more than a block row from where it was
            moved = np.abs(one - c['map'])
            assert moved.max() > 0.5 * np.abs(c['map']).max()


def test_make_map_records_entry_equals_columns(gpu):
    """ParticleSnapshot.make_map hands the structured array to the library as it is (bfgx_deposit_particles_records: fields read at their
    stride on the device) == the column entry a catalog with an unaligned layout falls back to"""
    import baryonification_amd as bfg
    rng = np.random.default_rng(9)
    L, N, npart = 100.0, 96, 1_200_000
    xyz = rng.uniform(-0.01 * L, 1.01 * L, (npart, 3))
    mass = rng.uniform(0.5, 2.0, npart)
    Snap = bfg.utils.ParticleSnapshot(x=xyz[:, 0], y=xyz[:, 1], z=xyz[:, 2], M=mass, L=L, redshift=0.0, cosmo=H.load_grid_golden('grid2d_paint')['cosmo_runner'])
    a = Snap.make_map(N).copy()
    odd = np.zeros(npart, dtype=[('tag', np.float32), ('M', np.float64), ('x', np.float64), ('y', np.float64), ('z', np.float64)])
    for k in ('M', 'x', 'y', 'z'):
        odd[k] = Snap.cat[k]
    Snap.cat = odd
    b = Snap.make_map(N)
    assert np.array_equal(a != 0, b != 0) and np.abs(a - b).max() <= 1e-12 * a.max() and np.isclose(a.sum(), mass[(xyz >= 0).all(1) & (xyz <= L).all(1)].sum())


def test_make_map_refuses_nan_masses_as_the_reference(gpu):
    """io.py:636: `assert np.isnan(self.cat['M']).sum() == 0` -- on the records path the library makes the check on the device"""
    import baryonification_amd as bfg
    rng = np.random.default_rng(2)
    xyz = rng.uniform(0, 50.0, (5000, 3))
    m = np.ones(5000)
    m[4321] = np.nan
    Snap = bfg.utils.ParticleSnapshot(x=xyz[:, 0], y=xyz[:, 1], z=xyz[:, 2], M=m, L=50.0, redshift=0.0, cosmo=H.load_grid_golden('grid2d_paint')['cosmo_runner'])
    with pytest.raises(AssertionError, match="provide a value for the particle mass"):
        Snap.make_map(16)
    Snap.cat['M'][4321] = 1.0
    assert Snap.make_map(16).sum() == 5000
