"""GPU box: wall time of PaintProfilesShell.process() at config 3 (1e6 halos, NSIDE 2048), numpy in / numpy out"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import baryonification_amd as bfg
from baryonification_amd import synthetic as syn
N, nside = 1_000_000, int(os.environ.get('NSIDE', 2048))
cat = syn.make_catalog(N)
z, M, r = syn.table_grid(cat)
prof = bfg.utils.TabulatedProfile(None, bfg.utils.Cosmology.from_dict(syn.COSMO))
prof.set_table(z, M, r, syn.paint_table(z, M, r))
Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
runner = bfg.Runners.PaintProfilesShell(Catalog, bfg.utils.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=syn.COSMO), 10.0, prof, verbose=False)
for _ in range(2): out = runner.process()
t = time.perf_counter()
K = 8
for _ in range(K): out = runner.process()
print(os.environ.get('BFGX_NO_PIPELINE'), 'nside', nside, 'ms per process() %.3f' % ((time.perf_counter() - t) / K * 1e3),
      {k: round(v, 3) for k, v in runner.last_stats.items() if k.startswith('ms')}, out.max() > 0)
