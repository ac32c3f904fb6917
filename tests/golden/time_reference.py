#!/usr/bin/env python
"""
SURVEY 8(d)(2): wall-clock of the UNMODIFIED reference's BaryonifyShell.process() in the build container (1 process),
under the oracle/refshim stand-ins (healpy -> numpy/ctypes HEALPix, pyccl -> background only, numba.njit -> identity, so
the post-loop regrid runs as plain Python).  Not comparable with a real healpy/numba install; recorded next to the
fixtures as the task asks.  Writes tests/golden/reference_timing.json.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle.refshim import install  # noqa: E402

install.install()
import BaryonForge as bfg  # noqa: E402
from baryonification_amd import synthetic as syn  # noqa: E402
import make_golden as MG  # noqa: E402


def run(nhalo, nside, ncat):
    cat = syn.make_catalog(ncat)
    ax = syn.table_grid(cat, pad=1e-9)
    sub = {k: v[:nhalo] for k, v in cat.items()}
    Catalog = bfg.utils.HaloLightConeCatalog(ra=sub['ra'], dec=sub['dec'], M=sub['M'], z=sub['z'], cosmo=syn.COSMO)
    Shell = bfg.utils.LightconeShell(map=syn.make_map(nside), cosmo=syn.COSMO)
    model = MG.ref_displacement_model(*ax, syn.displacement_table(*ax), False, 10.0, syn.COSMO)
    runner = bfg.Runners.BaryonifyShell(Catalog, Shell, 10.0, model, verbose=False)
    t0 = time.time()
    runner.process()
    return time.time() - t0


out = {"note": "reference BaryonifyShell.process() under oracle/refshim stand-ins, 1 process, build container (8 cores, no GPU)"}
t = run(1000, 128, 1000)
out["C1_1e3_halos_nside128"] = {"seconds": t, "halos_per_s": 1000 / t}
print(out, flush=True)
t = run(10000, 1024, 1_000_000)
out["C2_slice_1e4_halos_nside1024"] = {"seconds": t, "halos_per_s": 10000 / t,
                                         "remark": "includes the full NSIDE=1024 regrid (12.6M pixels) running as plain Python"}
print(out, flush=True)
json.dump(out, open(os.path.join(HERE, 'reference_timing.json'), 'w'), indent=1)
