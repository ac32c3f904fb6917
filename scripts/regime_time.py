"""GPU box: the BaryonifyShell step outside the benchmark's regime -- per-kernel times, census and regrid statistics for a catalog of N halos
with z in [zlo, zhi], log10 M in [mlo, mhi] on an NSIDE shell (closed-form table, --s19, or --paint: PaintProfilesShell).
python3 scripts/regime_time.py N NSIDE zlo zhi mlo mhi [--s19 | --paint]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import _lib, engine, synthetic as syn
N, nside = int(float(sys.argv[1])), int(sys.argv[2])
zlo, zhi, mlo, mhi = [float(x) for x in sys.argv[3:7]]
dev = torch.device('cuda', 0)
cat = syn.make_catalog(N, z_lo=zlo, z_hi=zhi, logM_lo=mlo, logM_hi=mhi)
z, M, r = syn.table_grid(cat)
paint = '--paint' in sys.argv
table = syn.paint_table(z, M, r) if paint else (syn.s19_displacement_table(z, M, r) if '--s19' in sys.argv else syn.displacement_table(z, M, r))
axes = [np.log(1 + z), np.log(M), np.log(r)]
with np.errstate(divide='ignore'):
    model, keep = engine.model_from_tables(axes, np.log(table) if paint else table, syn.COSMO, 10.0, 10.0, log_values=paint)
t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
lnz, lnM = _lib.table_coords(cat['M'], cat['z'])
t['lnz'], t['lnM'] = torch.from_numpy(lnz).to(dev), torch.from_numpy(lnM).to(dev)
npix = 12 * nside ** 2
d_map = torch.from_numpy(syn.make_map(nside)).to(dev)
plan = engine.ShellPlan(model, keep, nside, N, device=0, stream=torch.cuda.current_stream().cuda_stream)
cd = _lib.make_catalog_dev(N, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr(), ln1pz_ptr=t['lnz'].data_ptr(), lnM_ptr=t['lnM'].data_ptr())
off = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
out = torch.zeros(npix, dtype=torch.float64, device=dev)
sums = torch.zeros(2, dtype=torch.float64, device=dev)
def step():
    if paint:
        plan.paint(cd, out.data_ptr(), acc_f64=2)
    else:
        plan.baryonify(cd, d_map.data_ptr(), off.data_ptr(), out.data_ptr(), sums.data_ptr(), acc_f64=False)


for _ in range(3):
    step()
torch.cuda.synchronize()
plan.status()
pairs = plan.count_pairs(cd, not paint)
K = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
plan.timing_enable(True)
for _ in range(K):
    step()
tm = {k: round(v[0] / K, 3) for k, v in plan.timing_read().items() if v[1]}
s = sums.cpu().numpy()
print("N %d NSIDE %d z [%.2f, %.2f] log M [%.1f, %.1f]%s: %d pairs (%.0f per halo), step %.3f ms = %.2e halos/s, kernels %s, mass %s, regrid %s"
      % (N, nside, zlo, zhi, mlo, mhi, ' paint' if paint else (' S19' if '--s19' in sys.argv else ''), pairs, pairs / N, ms, N / ms * 1e3, tm, (None if paint else bool(np.isclose(s[0], s[1]))), (None if paint else plan.regrid_stats())))
