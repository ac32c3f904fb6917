"""GPU box: wall time of BaryonifyShell.process() at config 2 (1e6 halos, NSIDE 1024), numpy in / numpy out (BFGX_NO_PIPELINE=1: one-pass route)"""
import sys, time, os
import numpy as np
sys.path.insert(0, '.')
import baryonification_amd as bfg
from baryonification_amd import synthetic as syn
N, nside = 1_000_000, 1024
cat = syn.make_catalog(N)
z, M, r = syn.table_grid(cat)
model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=10.0)
model.set_table(z, M, r, syn.displacement_table(z, M, r))
Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
hmap = syn.make_map(nside)
runner = bfg.Runners.BaryonifyShell(Catalog, bfg.utils.LightconeShell(map=hmap, cosmo=syn.COSMO), 10.0, model, verbose=False)
for _ in range(3): out = runner.process()
t = time.perf_counter()
K = 20
for _ in range(K): out = runner.process()
dt = (time.perf_counter() - t) / K * 1e3
print(os.environ.get('BFGX_NO_PIPELINE'), os.environ.get('BFGX_PIN_INPUT'), 'ms per process() %.3f' % dt, {k: round(v, 3) for k, v in runner.last_stats.items() if k.startswith('ms')}, np.isclose(out.sum(), hmap.sum()))
