#!/usr/bin/env python
"""Functional rehearsal of config 5 over N ranks on a ONE-GPU box: every rank uses device 0, the process group is gloo and
utils/GridSlabs stages its collectives through the host.  Checks that the slab pipeline with the HIP backend (particle
routing -> slab deposit -> clipped halo loop -> regrid + apron exchange -> FFT with transpose -> reduced P(k)) returns what the
single-GPU pipeline on the full grid returns.  Launch:

    GLOO_SOCKET_IFNAME=lo python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29578 scripts/rehearse_grid_slabs.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    from baryonification_amd import _lib, engine, synthetic as syn
    from baryonification_amd.utils import GridSlabs as GS
    os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device('cuda', 0)
    N, L, nh, Nk, npart = 128, 150.0, 3000, 60, 2_000_000
    rng = np.random.default_rng(11)
    bins = (np.arange(N) + 0.5) * (L / N)
    M = (10 ** rng.uniform(12.5, 14.9, nh)).astype(np.float32).astype(np.float64)
    pos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
    z, Mt, r = np.array([0.0, 0.01]), np.geomspace(0.99e12, 1.01e15, 10), np.geomspace(1e-3, 3e2, 500)
    model, keep = engine.model_from_tables([np.log(1 + z), np.log(Mt), np.log(r)], syn.displacement_table(z, Mt, r) * 120.0,
                                           dict(syn.COSMO, w0=-1.0), 5.0, 5.0)
    t = {k: torch.from_numpy(v.copy()).to(dev) for k, v in (('M', M), ('x', pos[:, 0]), ('y', pos[:, 1]), ('z', pos[:, 2]))}
    lnM = torch.from_numpy(np.log(M.astype(np.float32)).astype(np.float64)).to(dev)
    cat_dev = _lib.make_grid_catalog_dev(nh, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr(), lnM.data_ptr())
    all_rows = torch.from_numpy(np.concatenate([rng.uniform(0, L, (3, npart)), rng.uniform(0.5, 2.0, (1, npart))], axis=0))
    rows = all_rows[:, rank::world].contiguous().to(dev)
    be = GS.HipBackend(model, keep, bins, 0.0, nh, device=0)
    new, kc, pk, cnt, sums = GS.slab_step(be, rows, cat_dev, N, L, Nk)
    gathered = [None] * world
    dist.all_gather_object(gathered, new.cpu().numpy())
    if rank == 0:
        edges = torch.from_numpy(np.linspace(0, L, N + 1)).to(dev)
        full_map = be.deposit(all_rows.to(dev), edges, 0, N)
        full_off = be.offsets(cat_dev, 0, N)
        ref, s0, missed = be.regrid(full_map, full_off, 0, 0, N)
        amax = float(torch.nan_to_num(full_off[..., 1]).abs().max().item())
        kc0, pk0, cnt0 = engine.power_spectrum(ref.cpu().numpy(), L, Nk)
        got = np.concatenate(gathered, axis=0)
        ref = ref.cpu().numpy()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        ok = cnt0 > 0
        perr = np.abs(pk[ok] / pk0[ok] - 1).max()
        print("ranks %d  grid %d^3  particles %d  halos %d  max |offset| along the slab axis %.2f cells" % (world, N, npart, nh, amax))
        print("slabs vs full grid: max |d map| / max = %.2e   max |d P(k)| / P(k) = %.2e   counts equal: %s   sums %r" % (
            err, perr, np.array_equal(cnt, cnt0), sums.tolist()))
        assert err <= 1e-12 and perr <= 1e-10 and np.array_equal(cnt, cnt0) and np.isclose(sums[0], sums[1], rtol=1e-12)
        print("OK")
    be.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
