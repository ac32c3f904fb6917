"""GPU box: wall time of BaryonifySnapshot.process() on the notebook-10 sized snapshot (512^3 / 2 particles, 1e5 halos), numpy in / out.
usage: python scripts/e2e_snapshot_time.py [particles] [halos]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import baryonification_amd as bfg                      # noqa: E402
from baryonification_amd import synthetic as syn       # noqa: E402

npart = int(sys.argv[1]) if len(sys.argv) > 1 else 512 ** 3 // 2
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
L = 205.0 / syn.COSMO['h']
rng = np.random.default_rng(syn.SEED_CATALOG)
M = syn.make_catalog(nh, seed=syn.SEED_CATALOG)['M']
pos = rng.uniform(0, L, (nh, 3))
z, Mt, r = np.array([0.0, 0.01]), np.geomspace(0.99e12, 1.01e15, 10), np.geomspace(1e-3, 3e2, 500)
model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=5.0)
model.set_table(z, Mt, r, syn.displacement_table(z, Mt, r))
HCat = bfg.utils.HaloNDCatalog(x=pos[:, 0], y=pos[:, 1], z=pos[:, 2], M=M, redshift=0.0, cosmo=syn.COSMO)
prng = np.random.default_rng(syn.SEED_MAP)
Snap = bfg.utils.ParticleSnapshot(x=prng.uniform(0, L, npart), y=prng.uniform(0, L, npart), z=prng.uniform(0, L, npart), M=1.0, L=L, redshift=0.0,
                                  cosmo=syn.COSMO)
runner = bfg.Runners.BaryonifySnapshot(HCat, Snap, 5.0, model, verbose=False)
out = None
for _ in range(2):
    out = runner.process()
K = 3
t = time.perf_counter()
for _ in range(K):
    out = runner.process()
dt = (time.perf_counter() - t) / K * 1e3
st = runner.last_stats or {}
moved = float(np.mean(out['x'] != Snap.cat['x']))
print(os.environ.get('BFGX_NO_PIPELINE'), 'BaryonifySnapshot.process() %d particles, %d halos: %.1f ms per call' % (npart, nh, dt),
      {k: round(v, 3) for k, v in st.items() if k.startswith('ms')}, 'pairs', st.get('n_pairs'), 'moved fraction %.4f' % moved)
