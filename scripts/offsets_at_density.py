import sys, numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import _lib, engine, synthetic as syn
dev = torch.device('cuda', 0)
halos, nside = int(float(sys.argv[1])), int(sys.argv[2])
cat = syn.make_catalog(halos)
z, M, r = syn.table_grid(cat)
table = syn.s19_displacement_table(z, M, r)
axes = [np.log(1 + z), np.log(M), np.log(r)]
model, keep = engine.model_from_tables(axes, table, syn.COSMO, 10.0, 10.0)
t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
lnz, lnM = _lib.table_coords(cat['M'], cat['z'])
t['lnz'], t['lnM'] = torch.from_numpy(lnz).to(dev), torch.from_numpy(lnM).to(dev)
npix = 12 * nside ** 2
d_map = torch.from_numpy(syn.make_map(nside)).to(dev)
plan = engine.ShellPlan(model, keep, nside, halos, device=0, stream=torch.cuda.current_stream().cuda_stream)
cd = _lib.make_catalog_dev(halos, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr(), ln1pz_ptr=t['lnz'].data_ptr(), lnM_ptr=t['lnM'].data_ptr())
off = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
out = torch.zeros(npix, dtype=torch.float64, device=dev)
sums = torch.zeros(2, dtype=torch.float64, device=dev)
plan.baryonify(cd, d_map.data_ptr(), off.data_ptr(), out.data_ptr(), sums.data_ptr(), acc_f64=False)
torch.cuda.synchronize()
print('regrid stats', plan.regrid_stats())
o = off.view(npix, 3)
mag = torch.sqrt((o * o).sum(1)) / np.sqrt(4 * np.pi / npix)
q = torch.quantile(mag[::16].double(), torch.tensor([0.5, 0.9, 0.99, 0.999], dtype=torch.float64, device=dev))
print('|offset| / pixel: mean %.2f  p50 %.2f p90 %.2f p99 %.2f p99.9 %.2f max %.2f   beyond 15 rings: %.4f' % (mag.mean().item(), *q.tolist(), mag.max().item(), (mag > 15 * 0.8).float().mean().item()))
