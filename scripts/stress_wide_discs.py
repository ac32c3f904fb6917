"""GPU box: randomised cross-check of the two routes of WIDE discs (a pole inside, pixels beyond 0.40 rad of the halo's azimuth, fallback pixels on the
first rings): through the fast kernel's second copy of the row / pair phases (default) and through the generic kernel's wide pass (BFGX_K1_WIDE=0 at
plan creation) -- NSIDE, catalog size, tile shape, displacement / paint, fp32 / fp64 pair math, both forms of K1, halos on and around the poles and
discs of 0.05 .. pi rad; the same census, outputs equal to the rounding of the two pair arithmetics.   python3 scripts/stress_wide_discs.py [cases] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import _lib, engine, synthetic as syn

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda', 0)
worst32 = worst64 = 0.0
for c in range(cases):
    nside = int(rng.choice([16, 64, 128, 256, 512, 1024]))
    N = int(rng.choice([200, 5_000, 100_000]))
    paint = bool(rng.random() < 0.4)
    f64 = bool(rng.random() < 0.4)
    fluid = str(rng.choice(['0', '1', '2']))
    cat = syn.make_catalog(N, seed=int(rng.integers(1, 1 << 30)), logM_lo=12.0, logM_hi=15.3, z_lo=0.1, z_hi=0.6)
    npole = int(rng.choice([4, 40, 150]))
    k = min(npole, N // 2)
    sgn = np.where(np.arange(k) % 2 == 0, 1.0, -1.0)
    cat['dec'][:k] = sgn * (90.0 - rng.uniform(0, 1.0, k) ** 3 * 4.0)
    cat['dec'][:2] = [90.0 - 1e-8, -90.0 + 1e-8]
    nbig = int(rng.choice([0, 3, 10]))                               # large discs anywhere: the heaviest halo at small z
    if nbig:
        cat['z'][k:k + nbig] = np.exp(rng.uniform(np.log(0.0008), np.log(0.05), nbig))
        cat['M'][k:k + nbig] = cat['M'].max()
    z, M, r = syn.table_grid(cat, pad=1e-3)
    table = syn.paint_table(z, M, r) if paint else syn.displacement_table(z, M, r)
    axes = [np.log(1 + z), np.log(M), np.log(r)]
    for kk in ('BFGX_TILE_BR', 'BFGX_TILE_W'):
        os.environ.pop(kk, None)
    if rng.random() < 0.3:
        os.environ['BFGX_TILE_BR'] = str(int(rng.choice([8, 16, 32])))
    if rng.random() < 0.3:
        os.environ['BFGX_TILE_W'] = str(int(rng.choice([16, 32, 64])))
    os.environ['BFGX_K1_FLUID'] = fluid
    plans = []
    for wide in ('1', '0'):
        os.environ['BFGX_K1_WIDE'] = wide
        with np.errstate(divide='ignore'):
            model, keep = engine.model_from_tables(axes, np.log(table) if paint else table, syn.COSMO, 10.0, 10.0, log_values=paint)
        plans.append((engine.ShellPlan(model, keep, nside, N, 0, torch.cuda.current_stream().cuda_stream), keep))
    cols = {kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in cat.items()}
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
        except ValueError as e:           # (whole-sphere discs: more (halo, tile) entries than a resident plan of this size holds -- not what is tested here)
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
    b0 = int(rng.choice([0, int(rng.integers(0, nb))])); b1 = int(rng.integers(b0, nb + 1))
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
    tol = 1e-9 if f64 else 1e-5
    ok = counts[0] == counts[1] and d <= tol and db <= tol
    if f64:
        worst64 = max(worst64, d, db)
    else:
        worst32 = max(worst32, d, db)
    print("case %2d  nside %4d  N %6d  %s  %s  fluid=%s  poles %3d  big %2d  tile %s x %s  pairs %11d  |fast - pass| / max %.1e  bands [%d, %d) %.1e  %s"
          % (c, nside, N, 'paint' if paint else 'displ', 'f64' if f64 else 'f32', fluid, k, nbig, os.environ.get('BFGX_TILE_BR', '-'), os.environ.get('BFGX_TILE_W', '-'),
             counts[0], d, b0, b1, db, 'ok' if ok else 'MISMATCH (census %d / %d)' % (counts[0], counts[1])), flush=True)
    for pl, _ in plans:
        pl.close()
    if not ok:
        sys.exit(1)
print("all %d cases agree; worst relative difference fp32 %.1e, fp64 %.1e" % (cases, worst32, worst64))
