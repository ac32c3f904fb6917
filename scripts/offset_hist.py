"""Runs on the GPU box: how far does BaryonifyShell move pixels, in units of the ring spacing / the pixel width?
Decides the reach of the gathering regrid's apron (K2).  usage: python scripts/offset_hist.py [closed-form|s19] [nside]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baryonification_amd import _lib, engine, synthetic as syn       # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 's19'
nside = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
npix = 12 * nside ** 2
dev = torch.device('cuda', 0)
cat = syn.make_catalog(1_000_000)
z, M, r = syn.table_grid(cat)
table = syn.s19_displacement_table(z, M, r) if kind == 's19' else syn.displacement_table(z, M, r)
print("table |d| max %.3g Mpc, mean %.3g" % (np.abs(table).max(), np.abs(table).mean()))
axes = [np.log(1 + z), np.log(M), np.log(r)]
model, keep = engine.model_from_tables(axes, table, syn.COSMO, 10.0, 10.0)
t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
lnz, lnM = _lib.table_coords(cat['M'], cat['z'])
t['lnz'], t['lnM'] = torch.from_numpy(lnz).to(dev), torch.from_numpy(lnM).to(dev)
plan = engine.ShellPlan(model, keep, nside, cat['M'].size, device=0, stream=torch.cuda.current_stream().cuda_stream)
cat_dev = _lib.make_catalog_dev(cat['M'].size, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr(),
                                ln1pz_ptr=t['lnz'].data_ptr(), lnM_ptr=t['lnM'].data_ptr())
d_off = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
plan.offsets(cat_dev, d_off.data_ptr())
torch.cuda.synchronize()
o = d_off.view(-1, 3).double()
mag = torch.linalg.norm(o, dim=1)
pix = float(np.sqrt(4 * np.pi / npix))
print("pixel size %.3e rad; |offset| / pixel: mean %.3f  p50 %.3f  p90 %.3f  p99 %.3f  p99.9 %.3f  max %.3f" % (
    pix, *(float(x) for x in (mag.mean() / pix, *(torch.quantile(mag[::7], q) / pix for q in (0.5, 0.9, 0.99, 0.999)), mag.max() / pix))))
for k in (0.5, 0.75, 1, 1.5, 2, 3, 4, 6, 8, 12, 16):
    print("  fraction of pixels with |offset| > %5.2f pixels: %.5f" % (k, float((mag > k * pix).double().mean())))
d_map = torch.from_numpy(syn.make_map(nside)).to(dev)
d_out = torch.zeros(npix, dtype=torch.float64, device=dev)
plan.regrid(d_map.data_ptr(), d_off.data_ptr(), d_out.data_ptr())
torch.cuda.synchronize()
try:
    plan.status()
    print("status ok")
except Exception as e:       # noqa: BLE001
    print("status:", e)
