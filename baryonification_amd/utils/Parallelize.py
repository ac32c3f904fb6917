"""
Halo-sharded multi-GPU execution: the MI355X counterpart of BaryonForge/utils/Parallelize.py
(`SplitJoinParallel` :116-320, `SimpleParallel` :8-113).

The reference splits the (shuffled) halo catalog into `njobs` sub-runners, runs them in joblib/loky
worker processes and sums the returned maps (Parallelize.py:255-273, :312-318).  Here a "job" is one
GPU: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI), each rank runs the
per-halo kernels on its shard and the partial accumulators are summed with ONE collective:

  * PaintProfilesShell : reduce(sum) of the painted map                       (== Parallelize.py:318)
  * BaryonifyShell     : reduce(sum) of pix_offsets BEFORE the regrid, which rank 0 then runs once.
    (The reference refuses Baryonify runners in SplitJoinParallel, Parallelize.py:206-209, because it can
    only sum final maps; summing the offsets is exact since every halo's contribution is computed
    against the undisplaced grid, HealpixRunner.py:312-331.)

Entry points
  shard_slices(n, world)                        -- the reference's ceil(N/njobs) contiguous split
  distributed_process(runner, kind, ...)        -- call from every rank of an initialised process group
  SplitJoinParallel(runner, njobs, seed).process()  -- drop-in: spawns one process per GPU and returns the map

`compute` is injectable so the sharding / collective logic can be exercised on CPU ranks (gloo) in the
test-suite; the product default is the HIP engine and there is no CPU fallback.
"""
import os

import numpy as np

__all__ = ['SplitJoinParallel', 'SimpleParallel', 'shard_slices', 'distributed_process']


def shard_slices(n, world):
    """Parallelize.py:250-266: Npersplit = ceil(N / Nsplits); shard i = [i*Npersplit, (i+1)*Npersplit)"""
    per = int(np.ceil(n / world)) if n > 0 else 0
    return [slice(min(i * per, n), min((i + 1) * per, n)) for i in range(world)]


def shuffled_order(n, seed=42):
    """Parallelize.py:255: default_rng(seed).choice(N, size=N, replace=False)"""
    return np.random.default_rng(seed).choice(n, size=n, replace=False)


def _hip_compute(runner, kind, cat_cols, device):
    """Per-rank partial accumulator on `device` (torch tensor): pix_offsets (f32 [npix*3]) or painted map."""
    import torch
    from .. import _lib, engine
    from ..Runners._model import build_model
    model, p_keys, keep = build_model(runner, 'displacement' if kind == 'baryonify' else 'projected')
    nside = int(runner.LightconeShell.NSIDE)
    npix = 12 * nside * nside
    dev = torch.device('cuda', device)
    n = cat_cols['M'].size
    t = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(dev) for k, v in cat_cols.items()}
    plan = engine.ShellPlan(model, keep, nside, n, device=device, stream=torch.cuda.current_stream(dev).cuda_stream)
    cd = _lib.make_catalog_dev(n, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr(),
                               [t[k].data_ptr() for k in p_keys])
    if kind == 'baryonify':
        acc = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
        plan.offsets(cd, acc.data_ptr(), acc_f64=False)
    else:
        acc = torch.zeros(npix, dtype=torch.float64, device=dev)
        plan.paint(cd, acc.data_ptr(), acc_f64=True)
    torch.cuda.synchronize(dev)
    return acc, plan


def _hip_regrid(runner, plan, acc, device):
    import torch
    dev = torch.device('cuda', device)
    hmap = torch.from_numpy(np.ascontiguousarray(runner.LightconeShell.map, dtype=np.float64)).to(dev)
    out = torch.zeros_like(hmap)
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    plan.regrid(hmap.data_ptr(), acc.data_ptr(), out.data_ptr(), sums.data_ptr(), acc_f64=False)
    torch.cuda.synchronize(dev)
    s = sums.cpu().numpy()
    assert np.isclose(s[1], s[0]), "ERROR in pixel regridding, sum(new_map) [%0.14e] != sum(oldmap) [%0.14e]" % (s[1], s[0])
    return out.cpu().numpy()


def distributed_process(runner, kind, seed=42, device=None, compute=None, regrid=None):
    """Run `runner` (holding the FULL catalog on every rank) halo-sharded over the ranks of the default
    torch.distributed group.  Returns the final map on rank 0, None elsewhere.

    compute(runner, kind, cat_cols, device) -> (tensor accumulator, ctx) and regrid(runner, ctx, acc, device)
    -> ndarray default to the HIP engine."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    compute = compute or _hip_compute
    regrid = regrid or _hip_regrid
    cat = runner.HaloLightConeCatalog.cat
    order = shuffled_order(cat.size, seed)
    mine = order[shard_slices(cat.size, world)[rank]]
    cols = {k: np.ascontiguousarray(cat[k][mine]) for k in cat.dtype.names}
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', rank))
    acc, ctx = compute(runner, kind, cols, device)
    dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)          # the one exchange step of the path
    if rank != 0:
        return None
    if kind == 'baryonify':
        return regrid(runner, ctx, acc, device)
    return acc.cpu().numpy().astype(np.float64)


def _spawn_worker(rank, world, port, runner, kind, seed, backend, out_path):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend == 'nccl':
        torch.cuda.set_device(rank)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        out = distributed_process(runner, kind, seed=seed, device=rank)
        if rank == 0:
            np.save(out_path, out)
    finally:
        dist.destroy_process_group()


class SplitJoinParallel(object):
    """Drop-in for Parallelize.py:116-320 where `njobs` = number of GPUs.  `process()` blocks and returns
    the summed map.  Unlike the reference it also accepts BaryonifyShell (offsets are reduced, not maps)."""

    def __init__(self, Runner, njobs=-1, seed=42):
        from ..Runners import BaryonifyShell, PaintProfilesShell
        assert isinstance(Runner, (BaryonifyShell, PaintProfilesShell)), \
            f"Runner of type {type(Runner)} is not supported for SplitJoinParallel."
        self.Runner, self.seed = Runner, seed
        if njobs == -1:
            from .. import _lib
            njobs = max(1, _lib.load().bfgx_device_count())
        self.njobs = njobs
        self.kind = 'baryonify' if isinstance(Runner, BaryonifyShell) else 'paint'

    def process(self):
        if self.njobs == 1:
            return self.Runner.process()
        import tempfile
        import torch.multiprocessing as mp
        import socket
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        with tempfile.TemporaryDirectory() as d:
            out_path = os.path.join(d, 'map.npy')
            mp.spawn(_spawn_worker, args=(self.njobs, port, self.Runner, self.kind, self.seed, 'nccl', out_path),
                     nprocs=self.njobs, join=True)
            return np.load(out_path)


class SimpleParallel(object):
    """Parallelize.py:8-113: independent runners, outputs in input order.  Runners are assigned to GPUs
    round-robin and executed back to back (each `process()` call is a few milliseconds of GPU time)."""

    def __init__(self, Runner_list, njobs=-1):
        self.Runner_list = Runner_list
        from .. import _lib
        ndev = max(1, _lib.load().bfgx_device_count())
        self.njobs = ndev if njobs == -1 else min(njobs, max(ndev, 1))

    def process(self):
        outputs = []
        for i, r in enumerate(self.Runner_list):
            r.device = i % self.njobs
            outputs.append(r.process())
        return outputs
