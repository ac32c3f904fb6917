#!/usr/bin/env python
"""GPU box: shader-clock accounting of the fluid K1 kernel's waves at config 2 (variant build with -DBFGX_K1F_PROF=1:
BFGX_LIB=baryonification_amd/csrc/variants/libbfgx_k1fprof.so python scripts/k1f_prof.py)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baryonification_amd import _lib, engine, synthetic as syn
dev = torch.device('cuda', 0)
paint = '--paint' in sys.argv
halos, nside = 1_000_000, (2048 if paint else 1024)
cat = syn.make_catalog(halos)
z, M, r = syn.table_grid(cat)
table = syn.paint_table(z, M, r) if paint else syn.displacement_table(z, M, r)
axes = [np.log(1 + z), np.log(M), np.log(r)]
with np.errstate(divide='ignore'):
    model, keep = engine.model_from_tables(axes, np.log(table) if paint else table, syn.COSMO, 10.0, 10.0, log_values=paint)
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
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 8)()
def step():
    if paint:
        plan.paint(cd, out.data_ptr(), acc_f64=2)
    else:
        plan.baryonify(cd, d_map.data_ptr(), off.data_ptr(), out.data_ptr(), sums.data_ptr(), acc_f64=False)


for it in range(3):
    step()
torch.cuda.synchronize()
lib.bfgx_debug_k1f_prof(buf, 1)

N = 10
for it in range(N):
    step()
torch.cuda.synchronize()
lib.bfgx_debug_k1f_prof(buf, 1)
wait, chunk, flush, total, nfl, nw, nchunk, maxsum = [float(x) for x in buf[:8]]
print("waves %d per launch, flushes %d per launch" % (nw / N, nfl / N))
print("per wave (clock ticks): total %.0f  wait for slot %.0f (%.1f %%)  chunks %.0f (%.1f %%)  flush + refill %.0f (%.1f %%; %.0f per flush)  other %.1f %%"
      % (total / nw, wait / nw, 100 * wait / total, chunk / nw, 100 * chunk / total, flush / nw, 100 * flush / total, flush / max(nfl, 1),
         100 * (total - wait - chunk - flush) / total))
print("chunks %d per launch (%.1f per tile): mean %.0f ticks, mean over tiles of the longest chunk %.0f ticks" % (nchunk / N, nchunk / max(nfl, 1), chunk / max(nchunk, 1), maxsum / max(nfl, 1)))
