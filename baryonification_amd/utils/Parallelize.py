"""
Halo-sharded multi-GPU execution: the MI355X counterpart of BaryonForge/utils/Parallelize.py
(`SplitJoinParallel` :116-320, `SimpleParallel` :8-113).

The reference splits the (shuffled) halo catalog into `njobs` sub-runners, runs them in joblib/loky
worker processes and sums the returned maps (Parallelize.py:255-273, :312-318).  Here a "job" is one
GPU: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI), each rank runs the
per-halo kernels on its shard of halos; the partial accumulators are then summed.

xGMI is a point-to-point mesh (one link per peer), so a ring reduce of the whole accumulator to one rank
(151 MB of pix_offsets at NSIDE 1024) is bound by ONE link.  The default exchange therefore works on pixel slices
(`exchange='slices'`): the sphere's ring bands are dealt out to the ranks in contiguous pixel ranges,
  1. all_to_all: every rank sends slice j of its accumulator straight to rank j (all 7 links busy, 1/N of the
     data per link) and sums the N slices it receives -- a reduce-scatter;
  2. BaryonifyShell only: every rank regrids the source pixels of ITS bands (K2 on 1/N of the map) into a window
     that is a few rings wider than its slice (deposits cross the band boundary by at most a pixel or two);
  3. all_to_all with empty splits except towards rank 0: the windows (slices for PaintProfilesShell) travel to
     rank 0 over separate links and are added into the final map.
`exchange='reduce'` keeps the single `reduce(sum)` to rank 0 (== Parallelize.py:318; for BaryonifyShell on
pix_offsets BEFORE the regrid, which rank 0 then runs once).  Summing offsets is exact because every halo's
contribution is computed against the undisplaced grid (HealpixRunner.py:312-331); the reference refuses Baryonify
runners in SplitJoinParallel (Parallelize.py:206-209) only because it can sum nothing but final maps.

Entry points
  shard_slices(n, world)                        -- the reference's ceil(N/njobs) contiguous split
  band_partition(first_pixel, world)            -- ring bands -> ranks, balanced by pixel count
  sliced_reduce / gather_windows                -- the two all_to_all steps above (any backend)
  distributed_process(runner, kind, ...)        -- call from every rank of an initialised process group
  SplitJoinParallel(runner, njobs, seed).process()  -- drop-in: spawns one process per GPU and returns the map

`compute`, `bounds` and `regrid_slice` are injectable so the sharding / collective logic can be exercised on CPU
ranks (gloo) in the test-suite; the product default is the HIP engine and there is no CPU fallback.
"""
import os

import numpy as np

__all__ = ['SplitJoinParallel', 'SimpleParallel', 'shard_slices', 'distributed_process', 'band_partition', 'sliced_reduce',
           'gather_windows']


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


def band_partition(first_pixel, world):
    """Deals the ring bands (first_pixel[b] = first RING pixel of band b, first_pixel[-1] = npix) out to `world` ranks
    in contiguous runs balanced by pixel count.  Returns the band index bounds [world + 1]."""
    first_pixel = np.asarray(first_pixel, dtype=np.int64)
    npix = int(first_pixel[-1])
    cuts = [0]
    for r in range(1, world):
        b = int(np.argmin(np.abs(first_pixel - npix * r / world)))
        cuts.append(max(b, cuts[-1]))
    cuts.append(first_pixel.size - 1)
    return np.asarray(cuts, dtype=np.int64)


def sliced_reduce(acc, bounds, width, recv=None):
    """Reduce-scatter over contiguous element ranges with ONE all_to_all: rank j ends up with the sum over ranks of
    acc[bounds[j] * width : bounds[j + 1] * width].  Returns this rank's summed slice."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    ins = [int(bounds[j + 1] - bounds[j]) * width for j in range(world)]
    mylen = ins[rank]
    if recv is None:
        recv = acc.new_empty(world * mylen)
    if acc.is_cuda and dist.get_backend() == 'gloo':        # rehearsal on a box without RCCL peers: stage through the host
        r = acc.new_empty(world * mylen, device='cpu')
        dist.all_to_all_single(r, acc.cpu(), output_split_sizes=[mylen] * world, input_split_sizes=ins)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, acc, output_split_sizes=[mylen] * world, input_split_sizes=ins)
    return recv.view(world, mylen).sum(0)


def gather_windows(win, windows, npix, recv=None, out=None):
    """Every rank's window (pixels [wlo_j, whi_j) of the output map, possibly overlapping its neighbours') travels to
    rank 0 in ONE all_to_all whose splits are empty except towards rank 0, which adds them into the full map."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    lens = [int(hi - lo) for lo, hi in windows]
    ins = [lens[rank]] + [0] * (world - 1)
    outs = lens if rank == 0 else [0] * world
    if recv is None:
        recv = win.new_empty(sum(outs))
    if win.is_cuda and dist.get_backend() == 'gloo':
        r = win.new_empty(sum(outs), device='cpu')
        dist.all_to_all_single(r, win.cpu(), output_split_sizes=outs, input_split_sizes=ins)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, win, output_split_sizes=outs, input_split_sizes=ins)
    if rank != 0:
        return None
    full = out if out is not None else win.new_zeros(npix)
    if out is not None:
        full.zero_()
    o = 0
    for (lo, hi), n in zip(windows, lens):
        full[int(lo):int(hi)] += recv[o:o + n]
        o += n
    return full


def _hip_bounds(runner, plan, world):
    first = plan.bands()
    cuts = band_partition(first, world)
    return cuts, first[cuts]


def _hip_regrid_slice(runner, plan, my_off, b0, b1, wlo, whi, device):
    """K2 on the bands [b0, b1) this rank owns; returns the window tensor (pixels [wlo, whi))"""
    import torch
    dev = torch.device('cuda', device)
    hmap = torch.from_numpy(np.ascontiguousarray(runner.LightconeShell.map, dtype=np.float64)).to(dev)
    win = torch.zeros(int(whi - wlo), dtype=torch.float64, device=dev)
    plan.regrid_bands(b0, b1, hmap.data_ptr(), my_off.data_ptr(), win.data_ptr(), wlo, whi, acc_f64=False)
    plan.status()
    return win


def window_margin(nside):
    """pixels by which a rank's output window exceeds its slice on either side: more than 4 rings anywhere"""
    return 16 * int(nside)


def distributed_process(runner, kind, seed=42, device=None, compute=None, regrid=None, exchange='slices', bounds=None,
                        regrid_slice=None):
    """Run `runner` (holding the FULL catalog on every rank) halo-sharded over the ranks of the default
    torch.distributed group.  Returns the final map on rank 0, None elsewhere.

    compute(runner, kind, cat_cols, device) -> (tensor accumulator, ctx); exchange='reduce': regrid(runner, ctx, acc,
    device) -> ndarray on rank 0; exchange='slices': bounds(runner, ctx, world) -> (band cuts, pixel bounds) and
    regrid_slice(runner, ctx, my_offsets, b0, b1, wlo, whi, device) -> window tensor.  All default to the HIP engine."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    compute = compute or _hip_compute
    cat = runner.HaloLightConeCatalog.cat
    order = shuffled_order(cat.size, seed)
    mine = order[shard_slices(cat.size, world)[rank]]
    cols = {k: np.ascontiguousarray(cat[k][mine]) for k in cat.dtype.names}
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', rank))
    acc, ctx = compute(runner, kind, cols, device)
    if exchange == 'reduce':
        regrid = regrid or _hip_regrid
        dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)
        if rank != 0:
            return None
        if kind == 'baryonify':
            return regrid(runner, ctx, acc, device)
        return acc.cpu().numpy().astype(np.float64)
    assert exchange == 'slices', "exchange must be 'slices' or 'reduce'"
    nside = int(runner.LightconeShell.NSIDE)
    npix = 12 * nside * nside
    cuts, pb = (bounds or _hip_bounds)(runner, ctx, world)
    if kind != 'baryonify':
        mine_sum = sliced_reduce(acc, pb, 1)
        full = gather_windows(mine_sum, [(pb[j], pb[j + 1]) for j in range(world)], npix)
        return None if rank != 0 else full.cpu().numpy().astype(np.float64)
    my_off = sliced_reduce(acc, pb, 3)
    m = window_margin(nside)
    wins = [(max(0, int(pb[j]) - m), min(npix, int(pb[j + 1]) + m)) for j in range(world)]
    win = (regrid_slice or _hip_regrid_slice)(runner, ctx, my_off, int(cuts[rank]), int(cuts[rank + 1]), wins[rank][0], wins[rank][1], device)
    full = gather_windows(win, wins, npix)
    if rank != 0:
        return None
    new_map = full.cpu().numpy().astype(np.float64)
    new_sum, old_sum = new_map.sum(), np.sum(runner.LightconeShell.map)
    assert np.isclose(new_sum, old_sum), "ERROR in pixel regridding, sum(new_map) [%0.14e] != sum(oldmap) [%0.14e]" % (new_sum, old_sum)
    return new_map


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
