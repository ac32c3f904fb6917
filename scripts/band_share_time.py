"""GPU box: K0 + K1 of ONE rank's share of a strong-scaling step (1 / W of the bands of config 2, the halos whose discs touch them), under the
fast kernel's two forms.   python3 scripts/band_share_time.py [W]"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import _lib, engine, synthetic as syn
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
N, nside = 1_000_000, 1024
cat = syn.make_catalog(N)
z, M, r = syn.table_grid(cat)
table = syn.displacement_table(z, M, r)
axes = [np.log(1 + z), np.log(M), np.log(r)]
for form in ('1', '2', '0'):
    os.environ['BFGX_K1_FLUID'] = form
    model, keep = engine.model_from_tables(axes, table, syn.COSMO, 10.0, 10.0)
    plan = engine.ShellPlan(model, keep, nside, N, 0, torch.cuda.current_stream().cuda_stream)
    bounds = plan.bands()
    nb = len(bounds) - 1
    b0, b1 = (nb // 2) - nb // (2 * W), (nb // 2) - nb // (2 * W) + nb // W        # an equatorial share
    cols = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
    cd_all = _lib.make_catalog_dev(N, cols['M'].data_ptr(), cols['z'].data_ptr(), cols['ra'].data_ptr(), cols['dec'].data_ptr())
    rings = torch.zeros((N, 2), dtype=torch.int32, device=dev)
    plan.disc_rings(cd_all, rings.data_ptr())
    rr = rings.cpu().numpy()
    r0, r1 = 1 + 32 * b0, 1 + 32 * b1
    sel = np.nonzero((rr[:, 0] < r1) & (rr[:, 1] >= r0))[0]
    sub = {k: torch.from_numpy(np.ascontiguousarray(v[sel])).to(dev) for k, v in cat.items()}
    cd = _lib.make_catalog_dev(sel.size, sub['M'].data_ptr(), sub['z'].data_ptr(), sub['ra'].data_ptr(), sub['dec'].data_ptr())
    sl = torch.zeros((int(bounds[b1]) - int(bounds[b0])) * 3, dtype=torch.float32, device=dev)
    for _ in range(5):
        plan.offsets_bands(cd, b0, b1, sl.data_ptr(), False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        plan.offsets_bands(cd, b0, b1, sl.data_ptr(), False)
    e1.record()
    torch.cuda.synchronize()
    plan.status()
    print("W %d  BFGX_K1_FLUID=%s  bands [%d, %d) of %d, %d halos: K0 + binning + K1 + wide %.1f us per call" % (W, form, b0, b1, nb, sel.size, e0.elapsed_time(e1) * 10.0))
    plan.close()
