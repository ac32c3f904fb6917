"""Runs on the GPU box: one precision mode of the config-2 step (K0 + binning + K1 + K2) on SURVEY 8(d)'s table (ii) or (i) under the library
named by BFGX_LIB -- its time, its kernel times and its map against a map saved by another run (another mode, another build).

  python scripts/prec_modes.py --acc 1 --save /tmp/ref.npy                 # fp64 throughout: the 1e-10 parity path
  python scripts/prec_modes.py --acc 3 --cmp /tmp/ref.npy                  # another mode against it: max |d| / mean(map)
  BFGX_LIB=.../variants/libbfgx_x.so python scripts/prec_modes.py --acc 1 --cmp /tmp/ref.npy
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baryonification_amd import _lib, engine, synthetic as syn       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--acc', type=int, default=0, help='acc_f64 of bfgx_baryonify_device (0 f32, 1 f64 throughout, 3 parity-grade)')
ap.add_argument('--table', default='s19', choices=['s19', 'closed-form'])
ap.add_argument('--nside', type=int, default=1024)
ap.add_argument('--halos', type=int, default=1_000_000)
ap.add_argument('--steps', type=int, default=100)
ap.add_argument('--save', default='')
ap.add_argument('--cmp', default='')
ap.add_argument('--tag', default='')
ap.add_argument('--scale', type=float, default=1.0, help='the table times this')
args = ap.parse_args()

nside, npix = args.nside, 12 * args.nside ** 2
dev = torch.device('cuda', 0)
cat = syn.make_catalog(args.halos)
z, M, r = syn.table_grid(cat)
tpath = '/tmp/s19_table_%d.npy' % args.halos
if args.table == 's19':
    if os.path.exists(tpath):
        table = np.load(tpath)
    else:
        table = syn.s19_displacement_table(z, M, r)
        np.save(tpath, table)
else:
    table = syn.displacement_table(z, M, r)
table = args.scale * table
axes = [np.log(1 + z), np.log(M), np.log(r)]
model, keep = engine.model_from_tables(axes, table, syn.COSMO, 10.0, 10.0)
t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
lnz, lnM = _lib.table_coords(cat['M'], cat['z'])
t['lnz'], t['lnM'] = torch.from_numpy(lnz).to(dev), torch.from_numpy(lnM).to(dev)
stream = torch.cuda.current_stream().cuda_stream
plan = engine.ShellPlan(model, keep, nside, cat['M'].size, device=0, stream=stream)
cat_dev = _lib.make_catalog_dev(cat['M'].size, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr(),
                                ln1pz_ptr=t['lnz'].data_ptr(), lnM_ptr=t['lnM'].data_ptr())
d_off = torch.zeros(npix * 3, dtype=torch.float64, device=dev)           # (sized for the widest mode)
d_map = torch.from_numpy(syn.make_map(nside)).to(dev)
d_out = torch.zeros(npix, dtype=torch.float64, device=dev)
d_sums = torch.zeros(2, dtype=torch.float64, device=dev)


def step():
    plan.baryonify(cat_dev, d_map.data_ptr(), d_off.data_ptr(), d_out.data_ptr(), d_sums.data_ptr(), acc_f64=args.acc)


for _ in range(5):
    step()
torch.cuda.synchronize()
try:
    plan.status()
except Exception as e:       # noqa: BLE001  (ablation builds)
    print('status:', e)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.steps):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.steps
plan.timing_enable(True)
for _ in range(20):
    step()
kt = plan.timing_read()
plan.timing_enable(False)
kms = {k: round(v[0] / max(v[1], 1) * (v[1] / 20.0), 4) for k, v in kt.items() if v[1]}
out = d_out.cpu().numpy()
sums = d_sums.cpu().numpy()
line = "%s acc=%d table=%s: %.4f ms/step  kernels %s  sums %.6f / %.6f  stats %s" % (
    args.tag or os.path.basename(os.environ.get('BFGX_LIB', 'libbfgx.so')), args.acc, args.table, ms, kms, sums[0], sums[1], plan.regrid_stats())
if args.cmp:
    ref = np.load(args.cmp)
    d = np.abs(out - ref)
    line += "  max|d|/mean %.3e  rms/mean %.3e  (99.99th %.3e)" % (d.max() / ref.mean(), np.sqrt((d ** 2).mean()) / ref.mean(),
                                                                   np.quantile(d[::17], 0.9999) / ref.mean())
print(line)
if args.save:
    np.save(args.save, out)
