"""GPU box: 20 000 halos at TWO positions (half of them next to the north pole): every wave of K0 names one tile, the tiles' fixed lists
overflow into the shared one, ten thousand discs add to the same pixels -- the fast kernel's forms against each other and the oracle (fp64)."""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import _lib, engine, synthetic as syn
from oracle import oracle as O
dev = torch.device('cuda', 0)
N, nside = 20000, 256
cat = syn.make_catalog(N)
cat['ra'][:] = 123.4; cat['dec'][:] = -33.3
cat['ra'][: N // 2] = 0.001; cat['dec'][: N // 2] = 89.9995           # half of them next to the north pole
z, M, r = syn.table_grid(cat)
table = syn.displacement_table(z, M, r)
axes = [np.log(1 + z), np.log(M), np.log(r)]
model, keep = engine.model_from_tables(axes, table, syn.COSMO, 10.0, 10.0)
cols = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
cd = _lib.make_catalog_dev(N, cols['M'].data_ptr(), cols['z'].data_ptr(), cols['ra'].data_ptr(), cols['dec'].data_ptr())
npix = 12 * nside ** 2
import os
for form in ('1', '2', '0'):
    os.environ['BFGX_K1_FLUID'] = form
    plan = engine.ShellPlan(model, keep, nside, N, 0, torch.cuda.current_stream().cuda_stream)
    off = torch.zeros(npix * 3, dtype=torch.float64, device=dev)
    plan.offsets(cd, off.data_ptr(), True)
    torch.cuda.synchronize()
    try:
        plan.status()
        st = 'ok'
    except Exception as e:
        st = str(e)[:80]
    got = off.cpu().numpy().reshape(npix, 3)
    if form == '1':
        bg = O.Background.from_dict(syn.COSMO)
        ora = O.baryonify_offsets(nside, cat, O.Table(axes, table, False, 10.0), 10.0, bg)
    print('form', form, 'status', st, 'pairs', plan.count_pairs(cd, True), 'max|off|', np.abs(got).max(), 'vs oracle', (np.abs(got - ora).max() / np.abs(ora).max()) if ora is not None else 'n/a')
    plan.close()
