#!/usr/bin/env python
"""GPU box: how far the default (fp32 pair math, fp32 pix_offsets) BaryonifyShell result at config 2 sits from the CPU oracle,
in units of the stated tolerance 1e-6 * mean(map)  (tests/test_gpu_fullsize.py asserts <= 1).   [--halos N --nside S]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--halos', type=int, default=1_000_000)
ap.add_argument('--nside', type=int, default=1024)
a = ap.parse_args()
import torch
from baryonification_amd import _lib, engine, synthetic as syn
from oracle import oracle as O
dev = torch.device('cuda', 0)
cat = syn.make_catalog(a.halos)
z, M, r = syn.table_grid(cat)
table = syn.displacement_table(z, M, r)
axes = [np.log(1 + z), np.log(M), np.log(r)]
model, keep = engine.model_from_tables(axes, table, syn.COSMO, 10.0, 10.0)
t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
lnz, lnM = _lib.table_coords(cat['M'], cat['z'])
t['lnz'], t['lnM'] = torch.from_numpy(lnz).to(dev), torch.from_numpy(lnM).to(dev)
npix = 12 * a.nside ** 2
hmap = syn.make_map(a.nside)
d_map = torch.from_numpy(hmap).to(dev)
plan = engine.ShellPlan(model, keep, a.nside, a.halos, device=0, stream=torch.cuda.current_stream().cuda_stream)
cd = _lib.make_catalog_dev(a.halos, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr(), ln1pz_ptr=t['lnz'].data_ptr(), lnM_ptr=t['lnM'].data_ptr())
off = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
out = torch.zeros(npix, dtype=torch.float64, device=dev)
sums = torch.zeros(2, dtype=torch.float64, device=dev)
plan.baryonify(cd, d_map.data_ptr(), off.data_ptr(), out.data_ptr(), sums.data_ptr(), acc_f64=False)
torch.cuda.synchronize()
bg = O.Background.from_dict(syn.COSMO)
tab = O.Table(axes, table, False, 10.0)
T = max(1, min(16, len(os.sched_getaffinity(0))))
ora, _, _, pairs = O.baryonify_shell_threads(a.nside, hmap, cat, tab, 10.0, bg, T)
got = out.cpu().numpy()
err = np.abs(got - ora).max()
print("pairs %d   max|hip - oracle| = %.3e = %.3f x the tolerance 1e-6 mean(map)   (mean %.3f)" % (pairs, err, err / (1e-6 * ora.mean()), ora.mean()))
