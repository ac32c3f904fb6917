"""GPU box: time of bfgx_power_spectrum_device on a resident N^3 map (torch events around the call), and a check of the binned sums
against a torch.fft restatement.   python3 scripts/pk_time.py [N] [reps]   (BFGX_LIB selects a library variant)"""
import sys
import numpy as np
import torch

sys.path.insert(0, '.')
from baryonification_amd import engine          # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
Nk, L = 180, 1000.0
dev = torch.device('cuda:0')
torch.manual_seed(5)
m = torch.rand(N ** 3, dtype=torch.float64, device=dev)
work = torch.empty(engine.power_spectrum_work_doubles(N), dtype=torch.float64, device=dev)
pk, ks = torch.zeros(Nk, dtype=torch.float64, device=dev), torch.zeros(Nk, dtype=torch.float64, device=dev)
cnt = torch.zeros(Nk, dtype=torch.int64, device=dev)
call = lambda: engine.power_spectrum_device(m.data_ptr(), N, L, Nk, work.data_ptr(), pk.data_ptr(), ks.data_ptr(), cnt.data_ptr(), 0, 0)
for _ in range(3):
    call()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    call()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
# restatement: |rfftn|^2 with the mirrored half weighted 2, linear bins between the fundamental and Nyquist (cells 12, 15)
F = torch.fft.rfftn(m.view(N, N, N))
P = (F.real ** 2 + F.imag ** 2)
kl = torch.fft.fftfreq(N, d=L / N, device=dev, dtype=torch.float64) * 2 * np.pi
kz = kl[:N // 2 + 1].abs()
kz[-1] = abs(float(kl[N // 2]))
kk = torch.sqrt(kl[:, None, None] ** 2 + kl[None, :, None] ** 2 + kz[None, None, :] ** 2)
w = torch.full((N // 2 + 1,), 2.0, dtype=torch.float64, device=dev)
w[0] = 1.0
w[-1] = 1.0
edges = torch.linspace(2 * np.pi / L, np.pi * N / L, Nk + 1, dtype=torch.float64, device=dev)
idx = torch.bucketize(kk.reshape(-1), edges, right=True) - 1
ok = (idx >= 0) & (idx < Nk)
ref = torch.zeros(Nk, dtype=torch.float64, device=dev).index_add_(0, idx[ok], (P * w[None, None, :]).reshape(-1)[ok])
got = pk.cpu().numpy()
refn = ref.cpu().numpy()
good = refn > 0
# (bin edges: modes that sit on an edge may fall either side in the restatement; the sum over all bins is the robust check)
print("N %d  pk call %.4f ms   total |F|^2 rel diff %.2e   bins within 1e-9: %d / %d" %
      (N, ms, abs(got.sum() / refn.sum() - 1), int((np.abs(got[good] / refn[good] - 1) < 1e-9).sum()), int(good.sum())))
