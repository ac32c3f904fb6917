"""K2 on clustered displacement fields: config 2 with the closed-form table scaled up (usage: python scripts/k2_scaled_table.py <scale>).
Prints the largest displacement in pixels, the fraction of pixels beyond the one-ring reach and the regrid time per step."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from baryonification_amd import _lib, engine, synthetic as syn
scale = float(sys.argv[1])
dev = torch.device('cuda', 0)
nside, N = 1024, 1_000_000
npix = 12 * nside ** 2
cat = syn.make_catalog(N)
z, M, r = syn.table_grid(cat)
model, keep = engine.model_from_tables([np.log(1 + z), np.log(M), np.log(r)], syn.displacement_table(z, M, r) * scale, syn.COSMO, 10.0, 10.0)
t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
lnz, lnM = _lib.table_coords(cat['M'], cat['z'])
t['lnz'], t['lnM'] = torch.from_numpy(lnz).to(dev), torch.from_numpy(lnM).to(dev)
plan = engine.ShellPlan(model, keep, nside, N, device=0, stream=torch.cuda.current_stream().cuda_stream)
cd = _lib.make_catalog_dev(N, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr(), ln1pz_ptr=t['lnz'].data_ptr(), lnM_ptr=t['lnM'].data_ptr())
off = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
out = torch.zeros(npix, dtype=torch.float64, device=dev)
sums = torch.zeros(2, dtype=torch.float64, device=dev)
hmap = torch.from_numpy(syn.make_map(nside)).to(dev)
for _ in range(5): plan.baryonify(cd, hmap.data_ptr(), off.data_ptr(), out.data_ptr(), sums.data_ptr())
plan.timing_enable(True); torch.cuda.synchronize()
for _ in range(50): plan.baryonify(cd, hmap.data_ptr(), off.data_ptr(), out.data_ptr(), sums.data_ptr())
kt = plan.timing_read()
pix = np.sqrt(4 * np.pi / npix)
mag = off.view(-1, 3).norm(dim=1)
print("scale %g: max |o| %.2f px, frac > 0.65 px %.4f, regrid %.4f ms, sums %r" % (scale, mag.max().item() / pix, (mag > 0.65 * pix).float().mean().item(), kt['regrid'][0] / kt['regrid'][1], sums.tolist()))
plan.status()
