"""GPU box: config 2 with a FOUR-axis table (one property axis: ParamTabulatedProfile / Baryonification2D with other_params) -- the generic
tile kernel runs every halo (no fast kernel for property axes).   python3 scripts/param_axis_time.py"""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import _lib, engine, synthetic as syn
dev = torch.device('cuda', 0)
N, nside = 1_000_000, 1024
cat = syn.make_catalog(N)
z, M, r = syn.table_grid(cat)
base = syn.displacement_table(z, M, r)
p = np.linspace(0.5, 1.5, 4)
rng = np.random.default_rng(1)
cat['c'] = rng.uniform(0.5, 1.5, N)
for name, axes, table, extra in (('3 axes', [np.log(1 + z), np.log(M), np.log(r)], base, ()),
                                 ('4 axes', [np.log(1 + z), np.log(M), np.log(r), p], base[:, :, None, :].repeat(4, 2).transpose(0, 1, 3, 2) * p[None, None, None, :] if False else None, ('c',))):
    if name == '4 axes':
        # device layout [z][M][p][r]; the ABI takes axes (z, M, r, p) and values [z][M][r][p]?  ask _lib.make_table
        table = base[:, :, :, None] * p[None, None, None, :]
    model, keep = engine.model_from_tables(axes, table, syn.COSMO, 10.0, 10.0)
    cols = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
    cd = _lib.make_catalog_dev(N, cols['M'].data_ptr(), cols['z'].data_ptr(), cols['ra'].data_ptr(), cols['dec'].data_ptr(), extra_ptrs=[cols[e].data_ptr() for e in extra])
    npix = 12 * nside ** 2
    plan = engine.ShellPlan(model, keep, nside, N, 0, torch.cuda.current_stream().cuda_stream)
    off = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
    for _ in range(3):
        plan.offsets(cd, off.data_ptr(), False)
    torch.cuda.synchronize()
    plan.status()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.offsets(cd, off.data_ptr(), False)
    e1.record()
    torch.cuda.synchronize()
    print("%s: K0 + binning + K1 %.3f ms" % (name, e0.elapsed_time(e1) / 10), flush=True)
    plan.close()
