"""GPU box: randomised cross-check of the fast kernel's two forms (fluid: BFGX_K1_FLUID=2, forced whatever the number of tiles; barrier per
tile: =0) -- NSIDE, catalog size and mass range, halos on the poles, tile shape, displacement / paint, whole sphere and band ranges; the same
census, outputs equal to the last bits of the fp64 LDS sums.   python3 scripts/stress_k1_forms.py [cases] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import _lib, engine, synthetic as syn

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda', 0)
worst = 0.0
for c in range(cases):
    nside = int(rng.choice([64, 128, 256, 512, 1024, 2048]))
    N = int(rng.choice([300, 5_000, 100_000, 600_000]))
    paint = bool(rng.random() < 0.4)
    f64 = bool(rng.random() < 0.4)                               # fp64 pair math (the fluid form then has twelve waves per workgroup)
    lo = float(rng.choice([11.5, 12.5, 13.5]))
    cat = syn.make_catalog(N, seed=int(rng.integers(1, 1 << 30)), logM_lo=lo, logM_hi=15.3, z_lo=float(rng.choice([0.05, 0.2, 0.8])), z_hi=1.0)
    npole = int(rng.choice([0, 20, 300]))
    if npole:
        k = min(npole, N)
        sgn = np.where(np.arange(k) % 2 == 0, 1.0, -1.0)
        cat['dec'][:k] = sgn * (90.0 - rng.uniform(0, 0.5, k))
    z, M, r = syn.table_grid(cat, pad=1e-3)
    table = syn.paint_table(z, M, r) if paint else syn.displacement_table(z, M, r)
    axes = [np.log(1 + z), np.log(M), np.log(r)]
    for k in ('BFGX_TILE_BR', 'BFGX_TILE_W'):
        os.environ.pop(k, None)
    if rng.random() < 0.3:
        os.environ['BFGX_TILE_BR'] = str(int(rng.choice([8, 16, 32])))
    if rng.random() < 0.3:
        os.environ['BFGX_TILE_W'] = str(int(rng.choice([16, 32, 64])))
    plans = []
    for form in ('2', '0'):
        os.environ['BFGX_K1_FLUID'] = form
        with np.errstate(divide='ignore'):
            model, keep = engine.model_from_tables(axes, np.log(table) if paint else table, syn.COSMO, 10.0, 10.0, log_values=paint)
        plans.append((engine.ShellPlan(model, keep, nside, N, 0, torch.cuda.current_stream().cuda_stream), keep))
    cols = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
    cd = _lib.make_catalog_dev(N, cols['M'].data_ptr(), cols['z'].data_ptr(), cols['ra'].data_ptr(), cols['dec'].data_ptr())
    npix = 12 * nside * nside
    comp = 1 if paint else 3
    outs, counts = [], []
    skip = False
    for pl, _ in plans:
        o = torch.zeros(npix * comp, dtype=torch.float64 if (paint or f64) else torch.float32, device=dev)
        if paint:
            pl.paint(cd, o.data_ptr(), acc_f64=(1 if f64 else 2))
        else:
            pl.offsets(cd, o.data_ptr(), f64)
        torch.cuda.synchronize()
        try:
            pl.status()
        except ValueError as e:           # (huge discs at low z: more (halo, tile) entries than a resident plan of this size holds -- not what is tested here)
            if 'overflowed' not in str(e):
                raise
            skip = True
            break
        outs.append(o)
        counts.append(pl.count_pairs(cd, not paint))
    if skip:
        print("case %2d  nside %4d  N %6d  skipped: the entry list of a resident plan of this size overflows" % (c, nside, N), flush=True)
        for pl, _ in plans:
            pl.close()
        continue
    scale = max(outs[1].abs().max().item(), 1e-300)
    d = (outs[0] - outs[1]).abs().max().item() / scale
    bounds = plans[0][0].bands()
    nb = len(bounds) - 1
    b0 = int(rng.integers(0, nb)); b1 = int(rng.integers(b0, nb + 1))
    db = 0.0
    if b1 > b0:
        sl = []
        for pl, _ in plans:
            t = torch.full(((int(bounds[b1]) - int(bounds[b0])) * comp,), 3.0, dtype=outs[0].dtype, device=dev)
            if paint:
                pl.paint_bands(cd, b0, b1, t.data_ptr(), acc_f64=(1 if f64 else 2))
            else:
                pl.offsets_bands(cd, b0, b1, t.data_ptr(), f64)
            torch.cuda.synchronize()
            pl.status()
            sl.append(t)
        db = (sl[0] - sl[1]).abs().max().item() / scale
    # (fp32 pix_offsets: each form rounds a pixel once, and the wide pass -- shared -- rounds a polar / low-z tile again at every visit, from
    # starting values that may differ by an ulp: a few 1e-7 of the scale for catalogs with many wide discs)
    # (fp64 sums in a different order: 1e-16 of the partial sums, which near the poles -- hundreds of discs on a pixel, offsets that cancel -- are
    # tens of times the result)
    tol = 1e-13 if paint else (1e-11 if f64 else 1e-6)
    ok = counts[0] == counts[1] and d <= tol and db <= tol
    worst = max(worst, d / tol, db / tol)
    print("case %2d  nside %4d  N %6d  %s  tile %s x %s  poles %3d  pairs %10d  bands [%d, %d) of %d   max diff / scale %.1e (bands %.1e)  %s"
          % (c, nside, N, ('paint' if paint else 'offsets') + (' f64' if f64 else ''), os.environ.get('BFGX_TILE_BR', '-'), os.environ.get('BFGX_TILE_W', '-'), npole, counts[0], b0, b1, nb, d, db,
             'ok' if ok else 'MISMATCH'), flush=True)
    for pl, _ in plans:
        pl.close()
    if not ok:
        sys.exit(1)
print("all %d cases agree; worst difference %.2f of the tolerance" % (cases, worst))
