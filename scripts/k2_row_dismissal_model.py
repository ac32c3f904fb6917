"""Runs on the GPU box: what would row-level dismissal buy the walking regrid (K2) on the S19 table?  Takes K1's pix_offsets at config 2,
looks at the equatorial belt (rings of 4 NSIDE pixels: tiles of 32 rings x 64 columns are aligned there) and counts, per tile, the source
pixels a gather has to evaluate under (a) the present per-tile apron (R rings, K columns from the neighbourhood's largest |o|), (b) rows
dismissed by the largest |o| of the (row, 64-column segment), (c) the same with 16-column segments.
usage: python scripts/k2_row_dismissal_model.py [closed-form|s19] [scale]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baryonification_amd import _lib, engine, synthetic as syn       # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 's19'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
nside = 1024
npix = 12 * nside ** 2
dev = torch.device('cuda', 0)
cat = syn.make_catalog(1_000_000)
z, M, r = syn.table_grid(cat)
table = (syn.s19_displacement_table(z, M, r) if kind == 's19' else syn.displacement_table(z, M, r)) * scale
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
ncap = 2 * nside * (nside - 1)
nr = 4 * nside
# belt rings nside .. 3 nside - 1 (2 nside rings), aligned to bands of 32 rings: band b holds rings 1 + 32 b ..; ring nside = 1 + 32 * 31 + 31 -> start at ring 1 + 32 * 32 = 1025
r0 = 1025
nrings = 2 * nside - 64
p0 = ncap + (r0 - nside) * nr
o = d_off.view(-1, 3)[p0:p0 + nrings * nr].double()
mag = torch.linalg.norm(o, dim=1).view(nrings, nr)
sp = 0.999 * 2.0 / (3.0 * nside)
pixw = 2 * np.pi / nr                       # column width in azimuth (x sin(theta) ~ 0.75 - 1 on the sky)
print("belt: |o| / ring spacing: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (
    float(mag.mean() / sp), *(float(torch.quantile(mag.flatten()[::11], q) / sp) for q in (0.5, 0.9, 0.99)), float(mag.max() / sp)))
BR, W = 32, 64
cap = min(0.02, 9.9 / nside)
magc = torch.clamp(mag, max=cap)            # what lies beyond the cap takes the far list
for SEG in (64, 16):
    nb, ns = nrings // BR, nr // SEG
    segmax = magc.view(nrings, ns, SEG).amax(dim=2)              # [ring][segment]
    tilemax = magc.view(nb, BR, nr // W, W).amax(dim=(1, 3))     # [band][tile]
    # (a) the present scheme: m = max over 3 x 3 tiles, R = floor(1.5 nside m) + 1, K ~ m / (sth - m) nr / 2pi + 3.5
    tm = tilemax
    nbh = torch.stack([torch.roll(torch.roll(tm, i, 0), j, 1) for i in (-1, 0, 1) for j in (-1, 0, 1)]).amax(0)
    R = torch.clamp((nbh * 1.5 * nside).floor() + 1, max=16)
    K = (nbh / 0.745 / pixw * 1.02 + 3.5).floor() + 1
    ev_a = ((BR + 2 * R) * (W + 2 * K)).mean() / (BR * W)
    # (b) rows dismissed by segment maxima: a segment of an apron row dr rings from the tile is evaluated iff its max >= (dr - 1) sp;
    #     own rows: the tile's own span always, the side aprons segment by segment iff max >= (dc - 1.5) column widths
    tot = 0.0
    spt = W // SEG                                                 # segments per tile width
    for dr in range(1, 17):
        need = (dr - 1) * sp
        for sign in (-1, 1):
            # rows at distance dr above (sign -1) / below (+1) the tile: ring index of the tile's first / last ring -+ dr
            # for every band b and every segment column s: is segmax[ring][s] >= need?  ring = b BR - dr (above) or b BR + BR - 1 + dr (below)
            ring = (torch.arange(nb, device=dev) * BR - dr) if sign < 0 else (torch.arange(nb, device=dev) * BR + BR - 1 + dr)
            ok = (ring >= 0) & (ring < nrings)
            sm = segmax[ring.clamp(0, nrings - 1)]                 # [band][segment]
            live = (sm >= need) & ok[:, None]
            # plus the side segments that could reach diagonally: count segments of the tile's span + one either side
            cnt = live.view(nb, nr // W, spt).sum(2).double()      # live segments over the tile's own span
            side = (torch.roll(live, 1, 1) | torch.roll(live, -1, 1)).view(nb, nr // W, spt)[:, :, [0, -1]].sum(2).double() * 0  # (ignored)
            tot += float((cnt * SEG).sum())
    # own rows: side aprons of K columns sized from the neighbouring segment's max
    tmrow = magc.view(nrings, nr // W, W).amax(2)                  # [ring][tile]: own-row maxima
    Kl = (torch.roll(tmrow, 1, 1) / 0.745 / pixw * 1.02 + 3.5).floor() + 1
    Kr = (torch.roll(tmrow, -1, 1) / 0.745 / pixw * 1.02 + 3.5).floor() + 1
    own = float((W + Kl + Kr).sum())
    ntile = nb * (nr // W)
    ev_b = (own + tot * 1.15) / (ntile * BR * W)                  # 1.15: side columns of the apron rows
    print("segments of %d columns: evaluations per stored pixel: present %.2f, row-dismissed %.2f (own rows %.2f, apron rows %.2f)" % (
        SEG, float(ev_a), ev_b, own / (ntile * BR * W), tot * 1.15 / (ntile * BR * W)))
print("R histogram (present):", torch.bincount(R.flatten().long()).tolist())
