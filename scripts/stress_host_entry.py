"""GPU box: randomised cross-check of the band-range route of the one-shot host entries (bfgx_baryonify_shell / bfgx_paint_shell) against
their one-pass route (BFGX_NO_PIPELINE): NSIDE, number of ranges, catalog size, displacement scale, halos on the poles, accumulator
precision.   python3 scripts/stress_host_entry.py [cases] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
import baryonification_amd as bfg
from baryonification_amd import synthetic as syn

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cosmo = bfg.utils.Cosmology.from_dict(syn.COSMO)
worst = 0.0
for c in range(cases):
    nside = int(rng.choice([256, 512, 1024]))
    N = int(rng.choice([500, 20_000, 200_000]))
    chunks = int(rng.integers(2, 17))
    scale = float(rng.choice([1.0, 1.0, 20.0, 80.0, 300.0]))
    paint = bool(rng.random() < 0.3)
    f64 = bool(rng.random() < 0.5)
    cat = syn.make_catalog(N, seed=int(rng.integers(1, 1 << 30)), logM_lo=12.5, logM_hi=15.0)
    npole = int(rng.choice([0, 50, 400]))
    if npole:
        k = min(npole, N)
        cat['dec'][:k] = np.where(np.arange(k) % 2 == 0, 90.0, -90.0) - np.sign(np.where(np.arange(k) % 2 == 0, 1, -1)) * rng.uniform(0, 0.3, k)
    z, M, r = syn.table_grid(cat, pad=1e-3)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    if paint:
        model = bfg.utils.TabulatedProfile(None, cosmo)
        model.set_table(z, M, r, syn.paint_table(z, M, r))
        runner = bfg.Runners.PaintProfilesShell(Catalog, bfg.utils.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=syn.COSMO), 10.0, model, verbose=False)
        runner.acc_f64 = True if f64 else 'mixed'
        tol = 1e-12 if f64 else 1e-12          # both routes run the same kernels on the same tiles
    else:
        model = bfg.Profiles.Baryonification2D(None, None, cosmo, epsilon_max=10.0)
        model.set_table(z, M, r, scale * syn.displacement_table(z, M, r))
        hmap = syn.make_map(nside, seed=int(rng.integers(1, 1 << 30)))
        hmap[::int(rng.integers(5, 50))] = 0.0
        runner = bfg.Runners.BaryonifyShell(Catalog, bfg.utils.LightconeShell(map=hmap, cosmo=syn.COSMO), 10.0, model, verbose=False)
        runner.acc_f64 = f64
        tol = 1e-12 if f64 else 2e-5 * max(1.0, scale)      # fp32 accumulators: the LDS adds of K1 come in any order, in either route
    os.environ['BFGX_PIPE_CHUNKS'] = str(chunks)
    os.environ.pop('BFGX_NO_PIPELINE', None)
    a = runner.process().copy()
    os.environ['BFGX_NO_PIPELINE'] = '1'
    b = runner.process().copy()
    ref = np.abs(b).max()
    d = np.abs(a - b).max() / ref
    ok = np.isfinite(a).all() and d <= tol and (paint or np.isclose(a.sum(), b.sum(), rtol=1e-10))
    worst = max(worst, d / tol)
    print("%2d nside %4d halos %6d ranges %2d %s scale %5.0f poles %3d %s  max|d|/max %.2e  %s" %
          (c, nside, N, chunks, 'paint' if paint else 'baryonify', scale, npole, 'f64' if f64 else ('mixed' if paint else 'f32'), d, 'ok' if ok else 'FAIL'), flush=True)
    if not ok:
        sys.exit(1)
print("all %d cases agree; worst difference / tolerance %.2f" % (cases, worst))
