"""GPU box: pix_offsets (fp64 throughout) of catalogs with many WIDE discs against the CPU oracle, for both routes of the wide discs (the fast
kernel: default; the generic kernel's wide pass: BFGX_K1_WIDE=0).   python3 scripts/wide_vs_oracle.py [cases] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import _lib, engine, synthetic as syn
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dev = torch.device('cuda', 0)
worst = [0.0, 0.0]
for c in range(cases):
    nside = int(rng.choice([16, 64, 128]))
    N = int(rng.choice([200, 2000]))
    cat = syn.make_catalog(N, seed=int(rng.integers(1, 1 << 30)), logM_lo=12.0, logM_hi=15.3, z_lo=0.1, z_hi=0.6)
    k = min(int(rng.choice([40, 100])), N // 2)
    sgn = np.where(np.arange(k) % 2 == 0, 1.0, -1.0)
    cat['dec'][:k] = sgn * (90.0 - rng.uniform(0, 1.0, k) ** 3 * 4.0)
    cat['dec'][:2] = [90.0 - 1e-8, -90.0 + 1e-8]
    nbig = int(rng.choice([0, 3]))
    if nbig:
        cat['z'][k:k + nbig] = np.exp(rng.uniform(np.log(0.002), np.log(0.05), nbig))
        cat['M'][k:k + nbig] = cat['M'].max()
    z, M, r = syn.table_grid(cat, pad=1e-3)
    table = syn.displacement_table(z, M, r)
    axes = [np.log(1 + z), np.log(M), np.log(r)]
    ora = O.baryonify_offsets(nside, cat, O.Table(axes, table, False, 10.0), 10.0, O.Background.from_dict(syn.COSMO)).ravel()
    # (the reference's (v + e) / |v + e| - v carries 1e-16 of absolute rounding whatever e is; the GPU's series in e does not.  Where every offset is
    # small the relative figure below is that residue over the scale: 1e-10 at offsets of 1e-6; a catalog whose offsets are all below 1e-12 -- NSIDE 16:
    # every disc takes the 4 fallback pixels, degrees away -- is residue only and is skipped)
    scale = np.abs(ora).max()
    if scale < 1e-12:
        print("case %2d  nside %4d  N %5d  poles %3d  big %d   every offset of the oracle is below 1e-12 (rounding residue): skipped" % (c, nside, N, k, nbig), flush=True)
        continue
    cols = {kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in cat.items()}
    lnz, lnM = _lib.table_coords(cat['M'], cat['z'])
    cols['lnz'], cols['lnM'] = torch.from_numpy(lnz).to(dev), torch.from_numpy(lnM).to(dev)
    cd = _lib.make_catalog_dev(N, cols['M'].data_ptr(), cols['z'].data_ptr(), cols['ra'].data_ptr(), cols['dec'].data_ptr(),
                               ln1pz_ptr=cols['lnz'].data_ptr(), lnM_ptr=cols['lnM'].data_ptr())
    res = []
    for i, wide in enumerate(('1', '0')):
        os.environ['BFGX_K1_WIDE'] = wide
        model, keep = engine.model_from_tables(axes, table, syn.COSMO, 10.0, 10.0)
        pl = engine.ShellPlan(model, keep, nside, N, 0, torch.cuda.current_stream().cuda_stream)
        o = torch.zeros(12 * nside * nside * 3, dtype=torch.float64, device=dev)
        pl.offsets(cd, o.data_ptr(), True)
        torch.cuda.synchronize()
        pl.status()
        got = o.cpu().numpy()
        d = np.abs(got - ora).max() / scale
        if d > 1e-6 and os.environ.get('BFGX_DIAG'):
            bad = np.flatnonzero(np.abs(got - ora) > 1e-6 * scale)
            pix = np.unique(bad // 3)
            print("   route", wide, "differs at pixels", pix[:20], "of", 12 * nside * nside)
            for q in pix[:6]:
                print("     pixel", q, "gpu", got[3 * q:3 * q + 3], "oracle", ora[3 * q:3 * q + 3])
            _, cnts = O.baryonify_offsets(nside, cat, O.Table(axes, table, False, 10.0), 10.0, O.Background.from_dict(syn.COSMO), return_counts=True)
            print("     oracle pair counts: min %d max %d, halos with < 4: %d; gpu census %d vs oracle %d" % (cnts.min(), cnts.max(), (cnts < 4).sum(), pl.count_pairs(cd, True), cnts.sum()))
            # which halos touch the first bad pixel?  (brute force through single-halo oracle calls)
            for j in range(N):
                one = {kk: v[j:j + 1] for kk, v in cat.items()}
                oo = O.baryonify_offsets(nside, one, O.Table(axes, table, False, 10.0), 10.0, O.Background.from_dict(syn.COSMO)).ravel()
                if np.any(oo[3 * pix[0]:3 * pix[0] + 3] != 0):
                    o1 = torch.zeros(12 * nside * nside * 3, dtype=torch.float64, device=dev)
                    c1 = {kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in one.items()}
                    l1, l2 = _lib.table_coords(one['M'], one['z'])
                    c1['a'], c1['b'] = torch.from_numpy(l1).to(dev), torch.from_numpy(l2).to(dev)
                    cd1 = _lib.make_catalog_dev(1, c1['M'].data_ptr(), c1['z'].data_ptr(), c1['ra'].data_ptr(), c1['dec'].data_ptr(), ln1pz_ptr=c1['a'].data_ptr(), lnM_ptr=c1['b'].data_ptr())
                    pl.offsets(cd1, o1.data_ptr(), True); torch.cuda.synchronize()
                    g1 = o1.cpu().numpy()
                    print("     halo", j, "ra dec", cat['ra'][j], cat['dec'][j], "M z", cat['M'][j], cat['z'][j], " oracle pixels", np.unique(np.flatnonzero(oo) // 3), " gpu pixels", np.unique(np.flatnonzero(g1) // 3),
                          " max diff", np.abs(g1 - oo).max() / scale)
        worst[i] = max(worst[i], d)
        res.append(d)
        pl.close()
    print("case %2d  nside %4d  N %5d  poles %3d  big %d   |fast-kernel route - oracle| / max %.1e   |wide-pass route - oracle| / max %.1e" % (c, nside, N, k, nbig, res[0], res[1]), flush=True)
print("worst: fast-kernel route %.1e, wide-pass route %.1e" % tuple(worst))
