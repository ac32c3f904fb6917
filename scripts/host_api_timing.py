#!/usr/bin/env python
"""PCIe-inclusive rate of the one-shot host API (numpy in / numpy out) at BASELINE config 2."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
import baryonification_amd as bfg
from baryonification_amd import synthetic as syn
N, nside = 1_000_000, 1024
cat = syn.make_catalog(N)
z, M, r = syn.table_grid(cat)
Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
Shell = bfg.utils.LightconeShell(map=syn.make_map(nside), cosmo=syn.COSMO)
model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=10.0)
model.set_table(z, M, r, syn.displacement_table(z, M, r))
runner = bfg.Runners.BaryonifyShell(Catalog, Shell, 10.0, model, verbose=False)
runner.process()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out = runner.process(); ts.append(time.perf_counter() - t0)
print(json.dumps({"what": "BaryonifyShell.process() one-shot host API, 1e6 halos, NSIDE 1024, pageable numpy inputs, page-locked result, cached plan",
                  "wall_s_best": min(ts), "halos_per_s_incl_pcie": N / min(ts), "stats_ms": runner.last_stats}))
