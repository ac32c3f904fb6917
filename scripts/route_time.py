"""GPU box: time of the routing kernels (bfgx_route_count_device / bfgx_route_pack_device) for 1e6 scattered halos and `world` destinations
on ONE GPU (what every rank of an N-GPU step runs before its all_to_all).   python3 scripts/route_time.py [world]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import engine, synthetic as syn
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
n, nside = 1_000_000, 1024
rng = np.random.default_rng(3)
first = rng.integers(1, 4 * nside - 12, n)
last = first + rng.integers(4, 12, n)
bounds = np.linspace(1, 4 * nside, world + 1).astype(np.int64)
cols = [torch.from_numpy(rng.normal(size=n)).to(dev) for _ in range(6)]
rings = torch.from_numpy(np.stack([first, last], axis=1).astype(np.int32)).to(dev)
cat = syn.make_catalog(1000)
z, M, r = syn.table_grid(cat)
model, keep = engine.model_from_tables([np.log(1 + z), np.log(M), np.log(r)], syn.displacement_table(z, M, r), syn.COSMO, 10.0, 10.0)
plan = engine.ShellPlan(model, keep, nside, 1000, device=0, stream=torch.cuda.current_stream().cuda_stream)
counts = torch.empty(world, dtype=torch.int32, device=dev)
cursor = torch.empty(world, dtype=torch.int32, device=dev)
capb = int(1.2 * n / world) + 4096
blocks = torch.empty((world, 6, capb), dtype=torch.float64, device=dev)
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
ptrs = [x.data_ptr() for x in cols]
for name, fn in (('count', lambda: plan.route_count(n, rings.data_ptr(), bounds, counts.data_ptr())),
                 ('pack', lambda: plan.route_pack(n, rings.data_ptr(), bounds, capb, ptrs, cursor.data_ptr(), blocks.data_ptr(), ovf.data_ptr()))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("world %d  route_%s: %.1f us per call   (overflow flag %d, rows %d)" % (world, name, e0.elapsed_time(e1) * 20.0, int(ovf.item()), int(counts.sum().item())))
