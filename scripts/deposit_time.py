"""GPU box: ParticleSnapshot.make_map on device-resident columns -- unit masses / per-particle masses, random / coarse-cell order, 3-D and 2-D.
python3 scripts/deposit_time.py [npart] [ngrid]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from baryonification_amd import engine
npart = int(float(sys.argv[1])) if len(sys.argv) > 1 else 67_108_864
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device('cuda', 0)
L = 300.0
torch.manual_seed(5)
part = torch.rand((3, npart), dtype=torch.float64, device=dev) * L
mass = torch.rand(npart, dtype=torch.float64, device=dev) + 0.5
edges = torch.from_numpy(np.linspace(0, L, N + 1)).to(dev)
st = torch.cuda.current_stream().cuda_stream
for ndim in (3, 2):
    out = torch.empty(N ** ndim, dtype=torch.float64, device=dev)
    for order in ('random', 'coarse cells'):
        p = part
        m = mass
        if order != 'random':
            key = ((part[0] * (64 / L)).long().clamp_(0, 63) * 64 + (part[1] * (64 / L)).long().clamp_(0, 63)) * 64 + (part[2] * (64 / L)).long().clamp_(0, 63)
            o = torch.argsort(key)
            p = part[:, o].contiguous(); m = mass[o].contiguous()
            del key, o
        for name, mp in (('unit masses', 0), ('masses', m.data_ptr())):
            def run():
                engine.deposit_particles_device(p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr() if ndim == 3 else 0, mp, npart, N, edges.data_ptr(), out.data_ptr(), ndim=ndim, device=0, stream=st)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            tot = out.sum().item()
            print("%d-D %d^%d, %.1e particles, %-12s %-11s: %.3f ms per call   (sum %.6e)" % (ndim, N, ndim, npart, order, name, e0.elapsed_time(e1) / 10, tot), flush=True)
