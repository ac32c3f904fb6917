"""Runs on the GPU box: builds SURVEY 8(d)'s benchmark table (ii) with the device table builders and saves it, so that the
build container (no GPU) can study it.  usage: python scripts/dump_s19_table.py [out.npz]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baryonification_amd import synthetic as syn       # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/s19_table.npz'
cat = syn.make_catalog(1_000_000)
z, M, r = syn.table_grid(cat)
table = syn.s19_displacement_table(z, M, r)
np.savez_compressed(out, z=z, M=M, r=r, table=table)
print("saved", out, table.shape, "max |d|", np.abs(table).max())
