"""GPU box: the notebook-10 route with numpy in / out at config 5's sizes -- BaryonifySnapshot.process() followed by
ParticleSnapshot.make_map(N) against the one-call BaryonifySnapshot.process_make_map(N) (bfgx_baryonify_snapshot_records_map)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import baryonification_amd as bfg                      # noqa: E402
from baryonification_amd import synthetic as syn       # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
npart = N ** 3 // 2
nh = 100_000
L = 205.0 / syn.COSMO['h']
prng = np.random.default_rng(syn.SEED_MAP)
Snap = bfg.utils.ParticleSnapshot(x=prng.uniform(0, L, npart), y=prng.uniform(0, L, npart), z=prng.uniform(0, L, npart), M=1.0, L=L, redshift=0.0,
                                  cosmo=syn.COSMO)
rng = np.random.default_rng(syn.SEED_CATALOG)
M = syn.make_catalog(nh, seed=syn.SEED_CATALOG)['M'].astype(np.float32).astype(np.float64)
pos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
z, Mt, r = np.array([0.0, 0.01]), np.geomspace(0.99e12, 1.01e15, 10), np.geomspace(1e-3, 3e2, 500)
model = bfg.Profiles.Baryonification3D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=5.0)
model.set_table(z, Mt, r, syn.displacement_table(z, Mt, r))
HCat = bfg.utils.HaloNDCatalog(x=pos[:, 0], y=pos[:, 1], z=pos[:, 2], M=M, redshift=0.0, cosmo=syn.COSMO)
runner = bfg.Runners.BaryonifySnapshot(HCat, Snap, 5.0, model, verbose=False)


def two_calls():
    new = bfg.utils.ParticleSnapshot.__new__(bfg.utils.ParticleSnapshot)
    new.__dict__.update(Snap.__dict__)
    new.cat = runner.process()
    return new.make_map(N)


for name, fn in (('process() + make_map(%d)' % N, two_calls), ('process_make_map(%d)' % N, lambda: runner.process_make_map(N))):
    for _ in range(2):
        m = fn()
    t = time.perf_counter()
    for _ in range(3):
        m = fn()
    print('%-32s %d particles, %d halos: %.1f ms per call, sum %.0f' % (name, npart, nh, (time.perf_counter() - t) / 3 * 1e3, m.sum()),
          {k: round(v, 3) for k, v in (runner.last_stats or {}).items() if k.startswith('ms')}, flush=True)
    if name.startswith('process()'):
        ref = m
print('max |one call - two calls| = %g' % np.abs(m - ref).max())
