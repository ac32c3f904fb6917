"""Parity of the particle-snapshot HIP path (csrc/bfgx_snapshot.hpp) through the drop-in BaryonifySnapshot runner
(-> ctypes -> C ABI) against the reference's outputs (tests/golden/snap*.npz) and the brute-force CPU oracle.
All fp64; differences are libm ulps and the order in which a particle's halos are summed  ->  |d| <= 1e-10 * max|offset|
(and 1e-13 of the box size on the positions themselves)."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _positions(new_cat, ndim):
    return np.stack([new_cat[k] for k in ('x', 'y', 'z')[:ndim]], axis=1)


@pytest.mark.parametrize('name', H.SNAPSHOT_CASES)
def test_snapshot_runner_vs_reference_golden(gpu, name):
    g = H.load_snapshot_golden(name)
    runner = H.snapshot_product_runner(g)
    new_cat = runner.process()
    assert new_cat.dtype == runner.ParticleSnapshot.cat.dtype and new_cat is not runner.ParticleSnapshot.cat
    out = _positions(new_cat, g['ndim'])
    exp = g['expected']
    assert np.array_equal(np.isnan(out), np.isnan(exp))
    scale = np.nanmax(np.abs(exp - g['part']))                   # largest displacement
    assert np.nanmax(np.abs(out - exp)) <= max(1e-10 * scale, 1e-13 * g['L'])
    moved = np.abs(out - g['part']).max(axis=1) > 0
    assert np.array_equal(np.nonzero(moved)[0], g['moved_idx'])  # exactly the reference's particles moved
    assert np.array_equal(new_cat['M'], runner.ParticleSnapshot.cat['M'])
    assert runner.last_stats['n_pairs'] >= g['moved_idx'].size


@pytest.mark.parametrize('cells', ['7', '64', '301'])
def test_snapshot_cell_grid_independence(gpu, cells):
    """the halo-cell grid is only an acceleration structure: any resolution gives the same particles"""
    g = H.load_snapshot_golden('snap3d_baryonify')
    os.environ['BFGX_SNAP_CELLS'] = cells
    try:
        out = _positions(H.snapshot_product_runner(g).process(), 3)
    finally:
        del os.environ['BFGX_SNAP_CELLS']
    assert np.nanmax(np.abs(out - g['expected'])) <= 1e-13 * g['L']


@pytest.mark.parametrize('ndim', [2, 3])
def test_snapshot_vs_oracle_larger(gpu, ndim):
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    from oracle import grid as G
    from oracle import oracle as O
    rng = np.random.default_rng(50 + ndim)
    L, nh, npart, zr = 300.0, 1500, 300_000, 0.2
    M = (10 ** rng.uniform(12.8, 15.0, nh)).astype(np.float32).astype(np.float64)
    hpos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
    part = rng.uniform(0, L, (npart, 3))
    part[:50_000] = (hpos[rng.integers(0, nh, 50_000)] + rng.normal(scale=1.0, size=(50_000, 3))) % L    # clustered around halos
    z, Mt, r = np.linspace(0.15, 0.25, 3), np.geomspace(10 ** 12.7, 10 ** 15.1, 8), np.geomspace(1e-3, 2e2, 200)
    d = syn.displacement_table(z, Mt, r)
    cos = dict(syn.COSMO)
    HCat = bfg.utils.HaloNDCatalog(x=hpos[:, 0], y=hpos[:, 1], z=hpos[:, 2] if ndim == 3 else None, M=M, redshift=zr, cosmo=cos)
    Snap = bfg.utils.ParticleSnapshot(x=part[:, 0], y=part[:, 1], z=part[:, 2] if ndim == 3 else None, M=np.ones(npart), L=L, redshift=zr, cosmo=cos)
    model = bfg.Profiles.Baryonification3D(None, None, bfg.utils.Cosmology.from_dict(cos), epsilon_max=8.0)
    model.set_table(z, Mt, r, d)
    runner = bfg.Runners.BaryonifySnapshot(HCat, Snap, 5.0, model, verbose=False)
    out = _positions(runner.process(), ndim)
    cat = {'M': M, 'x': hpos[:, 0], 'y': hpos[:, 1], 'z': hpos[:, 2] if ndim == 3 else np.zeros(nh)}
    tab = O.Table([np.log(1 + z), np.log(Mt), np.log(r)], d, False, 8.0)
    ora, pairs = G.baryonify_snapshot([part[:, k] for k in range(ndim)], L, cat, zr, tab, 5.0, G.grid_background(cos), return_pairs=True)
    ora = np.stack(ora, axis=1)
    scale = np.abs(ora - part[:, :ndim]).max()
    assert scale > 1e-3
    assert np.abs(out - ora).max() <= max(1e-10 * scale, 1e-13 * L)
    assert 0 < runner.last_stats['n_pairs'] <= pairs


def test_snapshot_device_resident_and_errors(gpu):
    import ctypes as C
    import torch
    import baryonification_amd as bfg
    from baryonification_amd import _lib, engine
    g = H.load_snapshot_golden('snap3d_baryonify')
    host = _positions(H.snapshot_product_runner(g).process(), 3)
    dev = torch.device('cuda:0')
    cat = g['cat']
    m, keep = engine.model_from_tables([np.log(1 + g['tab_z']), np.log(g['tab_M']), np.log(g['tab_r'])], g['tab_values'],
                                       dict(g['cosmo_runner'], w0=-1.0), g['eps_runner'], g['eps_model'], cosmo_model=g['cosmo_model'])
    t = {k: torch.tensor(cat[k], dtype=torch.float64, device=dev) for k in ('M', 'x', 'y', 'z')}
    lnM = torch.tensor(np.log(cat['M'].astype(np.float32)).astype(np.float64), device=dev)
    p = torch.tensor(np.ascontiguousarray(g['part'].T), device=dev)
    o = torch.empty_like(p)
    out = engine.baryonify_snapshot_device(m, _lib.make_grid_catalog_dev(cat['M'].size, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(),
                                                                        t['z'].data_ptr(), lnM.data_ptr()),
                                           (p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr()), g['npart'], g['L'], g['redshift'],
                                           (o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr()), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert out > 0
    assert np.nanmax(np.abs(o.cpu().numpy().T - host)) <= 1e-13 * g['L']
    # resident plan: same answer on repeated calls (workspace reuse), entry storage regrown when a later call needs more
    plan = engine.SnapshotPlan(m, keep, 3, g['L'], g['redshift'], cat['M'].size, 0, torch.cuda.current_stream().cuda_stream)
    dcat = _lib.make_grid_catalog_dev(cat['M'].size, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr(), lnM.data_ptr())
    for _ in range(2):
        o.zero_()
        assert plan.displace(dcat, g['npart'], (p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr()),
                             (o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())) == out
        torch.cuda.synchronize()
        assert np.nanmax(np.abs(o.cpu().numpy().T - host)) <= 1e-13 * g['L']
    plan.close()
    # particles outside the box are refused (scipy's periodic KDTree raises ValueError there too)
    bad = g['part'].copy()
    bad[5, 0] = -1.0
    g2 = dict(g, part=bad)
    with pytest.raises(ValueError):
        H.snapshot_product_runner(g2).process()


def test_snapshot_edge_cases(gpu):
    """no halos: particles come back unchanged; no particles: an empty catalog comes back"""
    import baryonification_amd as bfg
    g = H.load_snapshot_golden('snap2d_baryonify')
    runner = H.snapshot_product_runner(g)
    cos = g['cosmo_runner']
    runner.HaloNDCatalog = bfg.utils.HaloNDCatalog(x=np.zeros(0), y=np.zeros(0), M=np.zeros(0), redshift=g['redshift'], cosmo=cos)
    out = runner.process()
    assert np.array_equal(out['x'], g['part'][:, 0]) and np.array_equal(out['y'], g['part'][:, 1])
    runner2 = H.snapshot_product_runner(g)
    runner2.ParticleSnapshot = bfg.utils.ParticleSnapshot(x=np.zeros(0), y=np.zeros(0), z=None, M=np.zeros(0), L=g['L'], redshift=g['redshift'], cosmo=cos)
    assert runner2.process().size == 0


@pytest.mark.parametrize('ndim,chunks', [(3, 1), (3, 5), (2, 3), (3, 64)])
def test_snapshot_records_entry_streamed_equals_columns(gpu, monkeypatch, ndim, chunks):
    """bfgx_baryonify_snapshot_records (the structured array uploaded in chunks, displaced in place, downloaded into the new catalog)
    == the column entry (host gathers x, y, z, library displaces, host scatters them into a copy): same positions bit for bit, every other
    field untouched, the input catalog unchanged"""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    rng = np.random.default_rng(70 + ndim)
    L, nh, npart, zr = 250.0, 1200, 200_003, 0.1
    M = (10 ** rng.uniform(12.8, 15.0, nh)).astype(np.float32).astype(np.float64)
    hpos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
    part = rng.uniform(0, L, (npart, 3))
    part[:40_000] = (hpos[rng.integers(0, nh, 40_000)] + rng.normal(scale=1.0, size=(40_000, 3))) % L
    z, Mt, r = np.linspace(0.05, 0.15, 3), np.geomspace(10 ** 12.7, 10 ** 15.1, 8), np.geomspace(1e-3, 2e2, 200)
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=6.0)
    model.set_table(z, Mt, r, syn.displacement_table(z, Mt, r) * 20)
    kw = dict(x=hpos[:, 0], y=hpos[:, 1], M=M, redshift=zr, cosmo=syn.COSMO)
    pk = dict(x=part[:, 0], y=part[:, 1], M=rng.uniform(1, 2, npart), L=L, redshift=zr, cosmo=syn.COSMO)
    if ndim == 3:
        kw['z'] = hpos[:, 2]; pk['z'] = part[:, 2]
    HCat, Snap = bfg.utils.HaloNDCatalog(**kw), bfg.utils.ParticleSnapshot(**pk)
    before = Snap.cat.copy()
    runner = bfg.Runners.BaryonifySnapshot(HCat, Snap, 5.0, model, verbose=False)
    runner.use_records = False
    cols = runner.process()
    n_cols = runner.last_stats['n_pairs']
    runner.use_records = True
    monkeypatch.setenv('BFGX_PIPE_CHUNKS', str(chunks))
    recs = runner.process()
    assert recs.dtype == Snap.cat.dtype and recs is not Snap.cat and runner.last_stats['n_pairs'] == n_cols > 1000
    def same(a, b):          # (a particle inside several balls sums their offsets in the order the LDS adds arrive: last bits differ run to run)
        return np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b), initial=0.0) <= 1e-13 * L
    for k in ('x', 'y', 'z', 'M'):
        assert same(recs[k], cols[k]), k
    assert np.array_equal(Snap.cat, before) and np.array_equal(recs['M'], before['M'])
    assert 0.01 < np.mean(recs['x'] != before['x']) < 0.99             # (2-D: projected balls cover most of the box)
    again = runner.process()                                           # warm call: cached plan and device buffer, pooled result
    assert same(again['x'], cols['x'])
    # a catalog whose records the library cannot take as they are (a float32 field in front: 4-byte offsets) falls back to the columns
    odd = np.zeros(npart, dtype=[('tag', np.float32), ('M', np.float64), ('x', np.float64), ('y', np.float64), ('z', np.float64)])
    for k in ('M', 'x', 'y', 'z'):
        odd[k] = before[k]
    Snap.cat = odd
    out = runner.process()
    assert out.dtype == odd.dtype and same(out['x'], cols['x']) and np.array_equal(out['tag'], odd['tag'])


@pytest.mark.parametrize('ndim,n_grid,masses', [(3, 64, False), (3, 50, True), (2, 256, False), (2, 333, True)])
def test_snapshot_displace_deposit_fused_equals_process_then_make_map(gpu, ndim, n_grid, masses):
    """bfgx_snapshot_displace_deposit_device (the displacement kernel writes the deposit's sort keys; the displaced coordinates are never
    stored) == BaryonifySnapshot.process() followed by ParticleSnapshot.make_map(n_grid): cell for cell the map of the two-call route
    (SnapshotRunner.py:173-262, io.py:622-670), and np.histogramdd of the CPU oracle's displaced particles up to the handful of particles
    that sit within 1e-13 L of a cell edge"""
    import torch
    from baryonification_amd import _lib, engine
    from baryonification_amd import synthetic as syn
    from oracle import grid as G
    from oracle import oracle as O
    rng = np.random.default_rng(70 + ndim + n_grid)
    L, nh, npart, zr = 300.0, 1500, 300_000, 0.2
    M = (10 ** rng.uniform(12.8, 15.0, nh)).astype(np.float32).astype(np.float64)
    hpos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
    if ndim == 2:
        hpos[:, 2] = 0.0
    part = rng.uniform(0, L, (npart, 3))
    part[:50_000] = (hpos[rng.integers(0, nh, 50_000)] + rng.normal(scale=1.0, size=(50_000, 3))) % L
    part[50_000:50_010, 0] = [0.0, L] * 5                         # on the box faces: the closed last edge of np.histogramdd
    mass = rng.uniform(0.5, 2.0, npart) if masses else None
    z, Mt, r = np.linspace(0.15, 0.25, 3), np.geomspace(10 ** 12.7, 10 ** 15.1, 8), np.geomspace(1e-3, 2e2, 200)
    d = syn.displacement_table(z, Mt, r)
    cos = dict(syn.COSMO)
    dev = torch.device('cuda:0')
    stream = torch.cuda.current_stream().cuda_stream
    m, keep = engine.model_from_tables([np.log(1 + z), np.log(Mt), np.log(r)], d, dict(cos, w0=-1.0), 5.0, 8.0)
    t = {'M': torch.tensor(M, device=dev), 'x': torch.tensor(hpos[:, 0].copy(), device=dev), 'y': torch.tensor(hpos[:, 1].copy(), device=dev),
         'z': torch.tensor(hpos[:, 2].copy(), device=dev), 'lnM': torch.tensor(np.log(M.astype(np.float32)).astype(np.float64), device=dev)}
    dcat = _lib.make_grid_catalog_dev(nh, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr() if ndim == 3 else 0, t['lnM'].data_ptr())
    p = torch.tensor(np.ascontiguousarray(part[:, :ndim].T), device=dev)
    o = torch.empty_like(p)
    ptrs = [p[k].data_ptr() for k in range(ndim)] + [0] * (3 - ndim)
    optrs = [o[k].data_ptr() for k in range(ndim)] + [0] * (3 - ndim)
    d_mass = torch.tensor(mass, device=dev) if masses else None
    edges = np.linspace(0, L, n_grid + 1)
    d_edges = torch.tensor(edges, device=dev)
    two, one = torch.empty(n_grid ** ndim, dtype=torch.float64, device=dev), torch.full((n_grid ** ndim,), -1.0, dtype=torch.float64, device=dev)
    plan = engine.SnapshotPlan(m, keep, ndim, L, zr, nh, 0, stream)
    pairs = plan.displace(dcat, npart, ptrs, optrs)
    engine.deposit_particles_device(optrs[0], optrs[1], optrs[2], d_mass.data_ptr() if masses else 0, npart, n_grid, d_edges.data_ptr(), two.data_ptr(),
                                    ndim, 0, stream)
    for _ in range(2):                                            # (twice: the workspaces are reused)
        one.fill_(-1.0)
        assert plan.displace_deposit(dcat, npart, ptrs, d_mass.data_ptr() if masses else 0, n_grid, d_edges.data_ptr(), one.data_ptr()) == pairs
        torch.cuda.synchronize()
        if masses:           # (the sums inside a cell run in another order)
            assert np.abs(one.cpu().numpy() - two.cpu().numpy()).max() <= 1e-12 * two.max().item()
        else:
            assert torch.equal(one, two)
    plan.close()
    cat = {'M': M, 'x': hpos[:, 0], 'y': hpos[:, 1], 'z': hpos[:, 2]}
    tab = O.Table([np.log(1 + z), np.log(Mt), np.log(r)], d, False, 8.0)
    ora = G.baryonify_snapshot([part[:, k] for k in range(ndim)], L, cat, zr, tab, 5.0, G.grid_background(cos))
    ref = np.histogramdd(np.stack(ora, axis=1), bins=[edges] * ndim, weights=mass)[0].ravel()
    got = one.cpu().numpy()
    assert np.isclose(got.sum(), ref.sum(), rtol=1e-12) and np.abs(got - ref).sum() <= 8.0          # (at most four particles in a neighbouring cell)
    if not masses:
        assert np.count_nonzero(got != ref) <= 8
    assert np.abs(got - np.histogramdd(part[:, :ndim], bins=[edges] * ndim, weights=mass)[0].ravel()).sum() > 100      # (particles did move)


@pytest.mark.parametrize('name,n_grid', [('snap3d_baryonify', 32), ('snap3d_baryonify', 50), ('snap2d_baryonify', 128)])
def test_process_make_map_equals_process_then_make_map(gpu, name, n_grid):
    """BaryonifySnapshot.process_make_map(N) (bfgx_baryonify_snapshot_records_map: one upload, no displaced records) ==
    ParticleSnapshot(cat=runner.process()).make_map(N) of the product == np.histogramdd of the REFERENCE's displaced particles
    (tests/golden/snap*.npz) with the same weights (io.py:622-670)"""
    import baryonification_amd as bfg
    g = H.load_snapshot_golden(name)
    rng = np.random.default_rng(5)
    runner = H.snapshot_product_runner(g)
    runner.ParticleSnapshot.cat['M'] = rng.uniform(0.5, 2.0, runner.ParticleSnapshot.cat.size)
    mass = np.array(runner.ParticleSnapshot.cat['M'])
    ndim = g['ndim']
    fused = runner.process_make_map(n_grid)
    assert fused.shape == (n_grid,) * ndim and fused.dtype == np.float64 and runner.last_stats['n_pairs'] >= g['moved_idx'].size
    snap = runner.ParticleSnapshot
    two = bfg.utils.ParticleSnapshot(x=np.zeros(1), y=np.zeros(1), z=None if ndim == 2 else np.zeros(1), M=np.ones(1), L=g['L'], redshift=g['redshift'],
                                     cosmo=g['cosmo_runner'])
    two.cat = runner.process()
    two_map = two.make_map(n_grid)
    assert np.abs(fused - two_map).max() <= 1e-12 * two_map.max()
    edges = np.linspace(0, g['L'], n_grid + 1)
    exp = g['expected']
    ok = ~np.isnan(exp).any(axis=1)                                # (a particle exactly on a halo: NaN in the reference, dropped by histogramdd's edges)
    ref = np.histogramdd(exp[ok], bins=[edges] * ndim, weights=mass[ok])[0]
    assert np.abs(fused - ref).sum() <= 4 * 2.0 and np.isclose(fused.sum(), ref.sum(), rtol=1e-12)
    assert snap.cat is runner.ParticleSnapshot.cat and np.array_equal(snap.cat['M'], mass)          # the input snapshot is untouched
    # NaN masses: the reference's make_map asserts (io.py:636)
    runner.ParticleSnapshot.cat['M'][3] = np.nan
    with pytest.raises(AssertionError):
        runner.process_make_map(n_grid)
