"""
Halo-sharded multi-GPU execution: the MI355X counterpart of BaryonForge/utils/Parallelize.py
(`SplitJoinParallel` :116-320, `SimpleParallel` :8-113).

The reference splits the (shuffled) halo catalog into `njobs` sub-runners, runs them in joblib/loky
worker processes and sums the returned maps (Parallelize.py:255-273, :312-318).  Here a "job" is one
GPU: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI), each rank runs the
per-halo kernels on its shard of halos; the partial accumulators are then summed.

The default (`exchange='spatial'`) moves NO accumulator at all: the sphere's ring bands are dealt out to the ranks in
contiguous pixel ranges and every rank takes, out of the catalog it holds (the reference's workers all hold the full catalog;
a catalog read in chunks is routed with `route_halos`, 48 B per halo), the halos whose discs can touch ITS bands, computes
pix_offsets / the painted map for its own pixels only (K0 + K1 / K3 on its tiles), exchanges the few apron rings the regrid
needs with its two neighbours, regrids its bands and sends its slice of the map to rank 0.  The halo-sharded forms below remain
(`exchange='slices' | 'reduce'`): they are what the reference's SplitJoinParallel does, at the price of summing full-sky
accumulators across the ranks.

xGMI is a point-to-point mesh (one link per peer), so a ring reduce of the whole accumulator to one rank
(151 MB of pix_offsets at NSIDE 1024) is bound by ONE link.  The halo-sharded exchange therefore works on pixel slices
(`exchange='slices'`): the sphere's ring bands are dealt out to the ranks in contiguous pixel ranges,
  1. all_to_all: every rank sends slice j of its accumulator straight to rank j (all 7 links busy, 1/N of the
     data per link) and sums the N slices it receives -- a reduce-scatter;
  2. BaryonifyShell only: a halo exchange of ONE ring with the two neighbouring ranks (the gathering regrid evaluates
     the displaced pixels of the ring either side of its bands), then every rank regrids ITS bands: K2 on 1/N of the
     map, every output pixel of the slice stored exactly once (no window margins, no overlap to sum); the rare deposits
     that need the generic route (pole caps, moves of several pixels) are listed with global pixel numbers and added by
     whichever rank holds the pixel;
  3. the disjoint slices travel to rank 0 in one all_to_all with empty splits except towards rank 0 (`result='root'`),
     or to every rank with one all_gather (`result='all'`).
Bytes per link at N = 8, NSIDE 1024 (1e6-halo shards): step 1 moves 151 MB / 8 = 18.9 MB of f32 pix_offsets to each of
7 peers, i.e. per direction of one ~153 GB/s xGMI link ~0.12 ms; step 2 a 49 KB ring to 2 peers; step 3 12.6 MB of f64
map to rank 0 over 7 separate links (~0.08 ms) or 7 x 12.6 MB into every rank for `result='all'`.
`exchange='reduce'` keeps the single `reduce(sum)` to rank 0 (== Parallelize.py:318; for BaryonifyShell on
pix_offsets BEFORE the regrid, which rank 0 then runs once).  Summing offsets is exact because every halo's
contribution is computed against the undisplaced grid (HealpixRunner.py:312-331); the reference refuses Baryonify
runners in SplitJoinParallel (Parallelize.py:206-209) only because it can sum nothing but final maps.

Entry points
  shard_slices(n, world)                        -- the reference's ceil(N/njobs) contiguous split
  band_partition(first_pixel, world)            -- ring bands -> ranks, balanced by pixel count
  sliced_reduce / halo_exchange / gather_slices -- the collective steps above (any backend)
  distributed_process(runner, kind, ...)        -- call from every rank of an initialised process group
  SplitJoinParallel(runner, njobs, seed).process()  -- drop-in: all GPUs of the node from ONE process through the C ABI
                                                   (bfgx_*_shell_multi), or one process per GPU with backend='nccl'

`compute`, `bounds` and `regrid_slice` are injectable so the sharding / collective logic can be exercised on CPU
ranks (gloo) in the test-suite; the product default is the HIP engine and there is no CPU fallback.
"""
import os

import numpy as np

__all__ = ['SplitJoinParallel', 'SimpleParallel', 'shard_slices', 'distributed_process', 'band_partition', 'sliced_reduce',
           'halo_exchange', 'gather_slices', 'gather_slices_start', 'route_halos', 'route_halos_fixed', 'route_step_fixed', 'band_ring_bounds']


def shard_slices(n, world):
    """Parallelize.py:250-266: Npersplit = ceil(N / Nsplits); shard i = [i*Npersplit, (i+1)*Npersplit)"""
    per = int(np.ceil(n / world)) if n > 0 else 0
    return [slice(min(i * per, n), min((i + 1) * per, n)) for i in range(world)]


def shuffled_order(n, seed=42):
    """Parallelize.py:255: default_rng(seed).choice(N, size=N, replace=False)"""
    return np.random.default_rng(seed).choice(n, size=n, replace=False)


def _paint_mode(runner):
    """PaintProfilesShell.acc_f64 as the single-GPU runner reads it: None / 2 / 'mixed' = fp32 pair math into the fp64 map, True = fp64
    throughout (the map exchanged between ranks is fp64 either way)"""
    a = getattr(runner, 'acc_f64', None)
    return 2 if (a is None or a in (2, 'mixed')) else (1 if a else 2)


def _wide_offsets(runner, plan):
    """Do the multi-GPU paths of this runner need more than fp32 pair math?  runner.acc_f64 as BaryonifyShell.process() reads it (None: the plan
    picks from the table, include/bfgx.h BFGX_ACC_AUTO).  The rank-to-rank exchanges move ONE array of pix_offsets, so the parity-grade mode runs
    as fp64 throughout here (the band-restricted entries resolve it the same way): a table that moves pixels holds SURVEY 8(d)'s 1e-6 mean(map)
    on N GPUs as on one."""
    from .. import _lib
    return plan.precision(_lib.acc_mode(getattr(runner, 'acc_f64', None)))[0] != _lib.ACC_F32


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
    lnz, lnM = _lib.table_coords(cat_cols['M'], cat_cols['z'])        # numpy's np.log(1/a), np.log(M): table-edge halos as the reference
    t['_lnz'], t['_lnM'] = torch.from_numpy(lnz).to(dev), torch.from_numpy(lnM).to(dev)
    plan = engine.ShellPlan(model, keep, nside, n, device=device, stream=torch.cuda.current_stream(dev).cuda_stream)
    cd = _lib.make_catalog_dev(n, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr(),
                               [t[k].data_ptr() for k in p_keys], ln1pz_ptr=t['_lnz'].data_ptr(), lnM_ptr=t['_lnM'].data_ptr())
    if kind == 'baryonify':
        wide = _wide_offsets(runner, plan)
        acc = torch.zeros(npix * 3, dtype=torch.float64 if wide else torch.float32, device=dev)
        plan.offsets(cd, acc.data_ptr(), acc_f64=int(wide))
    else:
        acc = torch.zeros(npix, dtype=torch.float64, device=dev)
        plan.paint(cd, acc.data_ptr(), acc_f64=_paint_mode(runner))
    plan.status()            # blocking: the halo -> tile entry list did not overflow (a resident plan does not regrow it)
    return acc, plan


def route_halos(cols, rings, ring_bounds, plan=None):
    """Spatial sharding of halos that start out scattered over the ranks (a catalog read in chunks): `cols` = list of 1-D float64
    tensors (this rank's halos, one per catalog column), `rings` int32 [n][2] = ring range [first, last] every disc can touch
    (engine.ShellPlan.disc_rings), `ring_bounds[j]` = first ring owned by rank j (world + 1 entries, ascending).  A halo goes to
    EVERY rank whose rings it can touch -- a contiguous run of ranks, almost always one or two (first > last: to none).
    Returns the received columns as one tensor [k][m].  With `plan` (a ShellPlan on the tensors' GPU) counting and packing are two
    libbfgx kernels; either way ONE all_to_all_single of counts and one of packed rows."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    n, k = int(rings.shape[0]), len(cols)
    dev = rings.device
    if plan is not None and rings.is_cuda:
        counts = torch.empty(world, dtype=torch.int32, device=dev)
        plan.route_count(n, rings.data_ptr(), ring_bounds, counts.data_ptr())
        t_in = counts.to(torch.int64)
    else:
        rbt = torch.as_tensor(np.asarray(ring_bounds, dtype=np.int64), device=dev)
        first, last = rings[:, 0].long(), rings[:, 1].long()
        j_lo = (torch.bucketize(first, rbt, right=True) - 1).clamp_(0, world - 1)
        j_hi = (torch.bucketize(last, rbt, right=True) - 1).clamp_(0, world - 1)
        span = torch.where(first <= last, j_hi - j_lo + 1, torch.zeros_like(j_lo))
        src, dst = [], []
        for d in range(int(span.max().item()) if n else 0):                # d-th destination of the halos that have one
            sel = torch.nonzero(span > d, as_tuple=False).reshape(-1)
            src.append(sel); dst.append(j_lo[sel] + d)
        src = torch.cat(src) if src else j_lo[:0]
        dst = torch.cat(dst) if dst else j_lo[:0]
        t_in = torch.bincount(dst, minlength=world)[:world].to(torch.int64)
    t_out = torch.empty_like(t_in)
    _a2a(t_out, t_in, None, None)
    ins, outs = [[int(c) for c in row] for row in torch.stack([t_in, t_out]).tolist()]      # one read-back for both
    if plan is not None and rings.is_cuda:
        rows = torch.empty((sum(ins), k), dtype=torch.float64, device=dev)
        cursor = torch.empty(world, dtype=torch.int32, device=dev)
        start = np.concatenate([[0], np.cumsum(ins)[:-1]]).astype(np.int64)
        plan.route_fill(n, rings.data_ptr(), ring_bounds, start, [c.data_ptr() for c in cols], cursor.data_ptr(), rows.data_ptr())
    else:
        order = torch.argsort(dst, stable=True)
        rows = torch.stack(cols, dim=1)[src[order]] if n else torch.zeros((0, k), dtype=torch.float64, device=dev)
    recv = rows.new_empty((sum(outs), k))
    _a2a(recv.view(-1), rows.contiguous().view(-1), [o * k for o in outs], [i * k for i in ins])
    return recv.t().contiguous()


def route_halos_fixed(cols, rings, ring_bounds, blockcap, plan=None, work=None):
    """route_halos for a resident step: nothing is read back by the host.  Every (source, destination) pair gets a block of `blockcap`
    rows -- the equal splits of ONE all_to_all_single --, rows a destination does not receive carry M = NaN (column 0), which K0 drops
    as invalid halos.  Returns (columns [k][world * blockcap], overflow): `overflow` is an int32 tensor that is non-zero on a rank one
    of whose blocks was too small (check it once, after the steps; route_halos() has no such limit).  `work`: dict reused between
    calls (send / receive buffers).  With `plan` (a ShellPlan on the tensors' GPU) the packing is one libbfgx kernel."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    n, k = int(rings.shape[0]), len(cols)
    dev = rings.device
    work = work if work is not None else {}
    key = (world, k, int(blockcap), str(dev))
    if work.get('key') != key:
        work.clear()
        work.update(key=key, send=torch.empty((world, k, blockcap), dtype=torch.float64, device=dev),
                    recv=torch.empty((world, k, blockcap), dtype=torch.float64, device=dev),
                    out=torch.empty((k, world * blockcap), dtype=torch.float64, device=dev),
                    cursor=torch.empty(world, dtype=torch.int32, device=dev), overflow=torch.zeros(1, dtype=torch.int32, device=dev))
    send, recv, out = work['send'], work['recv'], work['out']
    if plan is not None and rings.is_cuda:
        plan.route_pack(n, rings.data_ptr(), ring_bounds, blockcap, [c.data_ptr() for c in cols], work['cursor'].data_ptr(), send.data_ptr(),
                        work['overflow'].data_ptr())
    else:
        send[:, 0, :] = float('nan')
        rbt = torch.as_tensor(np.asarray(ring_bounds, dtype=np.int64), device=dev)
        first, last = rings[:, 0].long(), rings[:, 1].long()
        j_lo = (torch.bucketize(first, rbt, right=True) - 1).clamp_(0, world - 1)
        j_hi = (torch.bucketize(last, rbt, right=True) - 1).clamp_(0, world - 1)
        stacked = torch.stack(cols, dim=0)
        for d in range(world):
            sel = torch.nonzero((first <= last) & (j_lo <= d) & (j_hi >= d), as_tuple=False).reshape(-1)
            if sel.numel() > blockcap:
                work['overflow'].fill_(1)
                sel = sel[:blockcap]
            send[d, :, :sel.numel()] = stacked[:, sel]
    _a2a(recv.view(-1), send.view(-1), None, None)
    out.view(k, world, blockcap).copy_(recv.permute(1, 0, 2))         # [source][column][row] -> contiguous columns
    return out, work['overflow']


def route_step_fixed(cat_dev, cols, ring_bounds, blockcap, plan, work):
    """The routing of a resident step as route_halos_fixed does it, in ONE library call and one collective, nothing read back, no transpose:
    plan.route_step finds every local halo's ring range and packs its rows by destination into fixed-capacity blocks [columns][blockcap]; the
    rows bound for this rank itself are written straight into the LAST block of the receive buffer and never enter the collective, the other
    world - 1 blocks travel in ONE all_to_all_single whose split towards oneself is empty (so the received blocks sit in rank order in front
    of it).  Returns (recv [world][k][blockcap] -- K0 reads it as a blocked catalog: plan.set_catalog_blocks(blockcap, k * blockcap) --,
    overflow int32 tensor).  `cat_dev`: bfgx_catalog of the local halos (M, z, dec are what the ring range needs)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    k = len(cols)
    dev = cols[0].device
    key = ('step', world, k, int(blockcap), str(dev))
    if work.get('key') != key:
        work.clear()
        work.update(key=key, send=torch.empty((max(world - 1, 1), k, blockcap), dtype=torch.float64, device=dev),
                    recv=torch.empty((world, k, blockcap), dtype=torch.float64, device=dev),
                    cursor=torch.empty(world, dtype=torch.int32, device=dev), overflow=torch.zeros(1, dtype=torch.int32, device=dev))
    send, recv = work['send'], work['recv']
    plan.route_step(cat_dev, rank, ring_bounds, blockcap, [c.data_ptr() for c in cols], work['cursor'].data_ptr(),
                    send.data_ptr() if world > 1 else 0, recv.data_ptr(), work['overflow'].data_ptr())
    blk = k * blockcap
    splits = [blk if d != rank else 0 for d in range(world)]
    _a2a(recv.view(-1)[:(world - 1) * blk], send.view(-1)[:(world - 1) * blk], splits, splits)
    return recv, work['overflow']


def band_ring_bounds(cuts, rings_per_band, nside):
    """first ring of every rank's run of bands (+ 4 nside for the end): band b = rings [1 + b R, 1 + (b + 1) R)"""
    return np.minimum(1 + rings_per_band * np.asarray(cuts, dtype=np.int64), 4 * nside)


def _hip_compute_spatial(runner, kind, cat_cols, device, world, rank):
    """Spatial sharding: this rank takes, out of the FULL catalog every rank holds, the halos whose discs can touch its run of
    ring bands and computes pix_offsets / the painted map for ITS pixels only (K0 + K1 / K3 on its tiles).  Nothing is summed
    across ranks afterwards.  Returns (slice tensor on `device`, plan)."""
    import torch
    from .. import _lib, engine
    from ..Runners._model import build_model
    model, p_keys, keep = build_model(runner, 'displacement' if kind == 'baryonify' else 'projected')
    nside = int(runner.LightconeShell.NSIDE)
    dev = torch.device('cuda', device)
    stream = torch.cuda.current_stream(dev).cuda_stream
    n_all = cat_cols['M'].size
    names = ['M', 'z', 'ra', 'dec'] + list(p_keys)
    lnz, lnM = _lib.table_coords(cat_cols['M'], cat_cols['z'])        # numpy's np.log(1/a), np.log(M): table-edge halos as the reference
    host = [np.ascontiguousarray(cat_cols[k], dtype=np.float64) for k in names] + [lnz, lnM]
    probe = engine.ShellPlan(model, keep, nside, 16, device=device, stream=stream)      # bands + ring ranges need no per-halo workspace
    first = probe.bands()
    cuts = band_partition(first, world)
    rb = band_ring_bounds(cuts, probe.tile_shape()[0], nside)
    t_all = [torch.from_numpy(h).to(dev) for h in host]
    cd_all = _lib.make_catalog_dev(n_all, t_all[0].data_ptr(), t_all[1].data_ptr(), t_all[2].data_ptr(), t_all[3].data_ptr())
    rings = torch.empty((n_all, 2), dtype=torch.int32, device=dev)
    probe.disc_rings(cd_all, rings.data_ptr())
    mine = (rings[:, 0].long() < int(rb[rank + 1])) & (rings[:, 1].long() >= int(rb[rank])) & (rings[:, 0] <= rings[:, 1])
    t = [c[mine].contiguous() for c in t_all]
    probe.close()
    n = int(t[0].numel())
    plan = engine.ShellPlan(model, keep, nside, max(n, 1), device=device, stream=stream)
    cd = _lib.make_catalog_dev(n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                               [t[4 + i].data_ptr() for i in range(len(p_keys))], ln1pz_ptr=t[-2].data_ptr(), lnM_ptr=t[-1].data_ptr())
    p0, p1 = int(first[cuts[rank]]), int(first[cuts[rank + 1]])
    if kind == 'baryonify':
        wide = _wide_offsets(runner, plan)
        sl = torch.zeros((p1 - p0) * 3, dtype=torch.float64 if wide else torch.float32, device=dev)
        plan.offsets_bands(cd, int(cuts[rank]), int(cuts[rank + 1]), sl.data_ptr(), acc_f64=int(wide))
    else:
        sl = torch.zeros(p1 - p0, dtype=torch.float64, device=dev)
        plan.paint_bands(cd, int(cuts[rank]), int(cuts[rank + 1]), sl.data_ptr(), acc_f64=_paint_mode(runner))
    plan.status()            # blocking: the halo -> tile entry list did not overflow
    plan._spatial_keep = t   # the catalog columns live as long as the plan
    return sl, plan


def _hip_regrid(runner, plan, acc, device):
    import torch
    dev = torch.device('cuda', device)
    hmap = torch.from_numpy(np.ascontiguousarray(runner.LightconeShell.map, dtype=np.float64)).to(dev)
    out = torch.zeros_like(hmap)
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    plan.regrid(hmap.data_ptr(), acc.data_ptr(), out.data_ptr(), sums.data_ptr(), acc_f64=int(acc.dtype == torch.float64))
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


def _a2a(recv, send, outs, ins):
    import torch.distributed as dist
    if send.is_cuda and dist.get_backend() == 'gloo':       # rehearsal on a box without RCCL peers: stage through the host
        r = send.new_empty(recv.numel(), device='cpu')
        dist.all_to_all_single(r, send.cpu(), output_split_sizes=outs, input_split_sizes=ins)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, output_split_sizes=outs, input_split_sizes=ins)
    return recv


def halo_exchange(mine, pb, needs, width, full=None):
    """After the reduce-scatter rank j holds the summed values of pixels [pb[j], pb[j+1]) (`mine`, `width` numbers per
    pixel); it needs the pixels needs[j] = (lo_j, hi_j), a superset that reaches into its neighbours' slices.  One
    all_to_all whose splits are empty except towards the ranks that need a piece of this rank's slice.  Returns the
    tensor of pixels [lo_rank, hi_rank).  `full`: that tensor, preallocated, with `mine` already a view of its middle part
    (the slice was computed in place): only the apron pieces are written."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()

    def piece(owner, taker):
        """pixel range of `owner`'s slice that `taker` needs (taker != owner): below or above the taker's own slice"""
        lo, hi = needs[taker]
        if owner < taker:
            a, b = max(int(pb[owner]), int(lo)), min(int(pb[owner + 1]), int(pb[taker]))
        else:
            a, b = max(int(pb[owner]), int(pb[taker + 1])), min(int(pb[owner + 1]), int(hi))
        return (a, b) if b > a else (0, 0)

    p0 = int(pb[rank])
    send_ranges = [piece(rank, i) if i != rank else (0, 0) for i in range(world)]
    recv_ranges = [piece(j, rank) if j != rank else (0, 0) for j in range(world)]
    ins = [(b - a) * width for a, b in send_ranges]
    outs = [(b - a) * width for a, b in recv_ranges]
    send = torch.cat([mine[(a - p0) * width:(b - p0) * width] for a, b in send_ranges]) if sum(ins) else mine.new_empty(0)
    recv = _a2a(mine.new_empty(sum(outs)), send, outs, ins)
    lo, hi = int(needs[rank][0]), int(needs[rank][1])
    if full is None:
        full = mine.new_empty((hi - lo) * width)
        full[(p0 - lo) * width:(int(pb[rank + 1]) - lo) * width] = mine
    else:
        assert full.numel() == (hi - lo) * width and mine.data_ptr() == full.data_ptr() + (p0 - lo) * width * full.element_size()
    o = 0
    for (a, b), n in zip(recv_ranges, outs):
        if n:
            full[(a - lo) * width:(b - lo) * width] = recv[o:o + n]
            o += n
    return full


def gather_slices(mine, pb, npix, result='root', recv=None, out=None, root_in_place=False):
    """The ranks' disjoint slices (pixels [pb[j], pb[j+1])) -> the full map on rank 0 (`result='root'`: one all_to_all with
    empty splits except towards rank 0, 7 separate links) or on every rank (`result='all'`: one all_gather of slices
    padded to the longest).  Returns the map or None.  root_in_place: rank 0 has written its own slice into `out` already (it
    regridded straight into the final map): only the other ranks' slices travel."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    lens = [int(pb[j + 1] - pb[j]) for j in range(world)]
    if result == 'all':
        L = max(lens)
        pad = mine if lens[rank] == L else torch.cat([mine, mine.new_zeros(L - lens[rank])])
        buf = mine.new_empty(world * L)
        if mine.is_cuda and dist.get_backend() == 'gloo':
            b = buf.cpu()
            dist.all_gather_into_tensor(b, pad.cpu())
            buf.copy_(b)
        else:
            dist.all_gather_into_tensor(buf, pad)
        full = out if out is not None else mine.new_empty(npix)
        for j in range(world):
            full[int(pb[j]):int(pb[j + 1])] = buf[j * L:j * L + lens[j]]
        return full
    assert result == 'root', "result must be 'root' or 'all'"
    if root_in_place:
        ins = [0] * world if rank == 0 else [lens[rank]] + [0] * (world - 1)
        outs = [0] + lens[1:] if rank == 0 else [0] * world
        tail = out[int(pb[1]):] if rank == 0 else mine.new_empty(0)
        _a2a(tail, mine.new_empty(0) if rank == 0 else mine, outs, ins)
        return out if rank == 0 else None
    ins = [lens[rank]] + [0] * (world - 1)
    outs = lens if rank == 0 else [0] * world
    if rank == 0 and out is not None and recv is None:
        recv = out                                        # the slices arrive in rank order = pixel order: receive in place
    if recv is None:
        recv = mine.new_empty(sum(outs))
    _a2a(recv, mine, outs, ins)
    if rank != 0:
        return None
    return recv


class _Done(object):
    """the handle of a collective that has already completed (gloo rehearsal: staged through the host, synchronous)"""

    def wait(self):
        return True


def gather_slices_start(mine, pb, npix, out=None, group=None):
    """gather_slices(..., result='root', root_in_place=True) as an ASYNCHRONOUS collective on `group` (a process group of its own: RCCL runs it
    on that communicator's stream, so the caller's stream goes on with the NEXT pass while the slices travel to rank 0 -- at 2 / 4 / 8 GPUs
    a rank's 50 / 25 / 12.6 MB of map cross ONE xGMI link each, 1.0 / 0.5 / 0.26 ms against 0.6 / 0.3 / 0.15 ms of kernels).  Rank 0 has written its
    own slice into `out` already.  Returns a handle: .wait() before `mine` / `out` are written again."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    lens = [int(pb[j + 1] - pb[j]) for j in range(world)]
    ins = [0] * world if rank == 0 else [lens[rank]] + [0] * (world - 1)
    outs = [0] + lens[1:] if rank == 0 else [0] * world
    recv = out[int(pb[1]):] if rank == 0 else mine.new_empty(0)
    send = mine.new_empty(0) if rank == 0 else mine
    if mine.is_cuda and dist.get_backend(group) == 'gloo':          # rehearsal: through the host, done when this returns
        r = send.new_empty(recv.numel(), device='cpu')
        dist.all_to_all_single(r, send.cpu(), output_split_sizes=outs, input_split_sizes=ins, group=group)
        recv.copy_(r)
        return _Done()
    return dist.all_to_all_single(recv, send, output_split_sizes=outs, input_split_sizes=ins, group=group, async_op=True)


def _hip_bounds(runner, plan, world):
    first = plan.bands()
    cuts = band_partition(first, world)
    needs = [plan.band_apron(int(cuts[j]), int(cuts[j + 1])) for j in range(world)]
    return cuts, first[cuts], needs


def _hip_reach(runner, plan, my_off, bands=None):
    """collective: the rings of apron the gathering regrid needs follow the largest |offset| of the SUMMED pix_offsets over
    all ranks; every rank sets the same value on its plan.  `bands` = (b0, b1): `my_off` is the slice plan.offsets_bands() has
    just written for those bands (spatial sharding, nothing added since) -- the maximum then comes from the per-tile maxima K1
    left behind instead of a pass over the slice."""
    import torch
    import torch.distributed as dist
    if my_off.is_cuda and bands is not None:
        m2 = torch.empty(1, dtype=torch.float32, device=my_off.device)
        plan.bands_max_offset2(int(bands[0]), int(bands[1]), m2.data_ptr())
    elif my_off.is_cuda:
        m2 = torch.empty(1, dtype=torch.float32, device=my_off.device)
        plan.max_offset2(my_off.data_ptr(), my_off.numel() // 3, m2.data_ptr(), acc_f64=(my_off.dtype == torch.float64))
    else:
        m2 = my_off.view(-1, 3).float().square().sum(1).max().reshape(1) if my_off.numel() else my_off.new_zeros(1, dtype=torch.float32)
    if dist.get_backend() == 'gloo' and m2.is_cuda:
        m2 = m2.cpu()
    dist.all_reduce(m2, op=dist.ReduceOp.MAX)
    plan.set_band_reach(plan.reach_rings(float(m2.item()) ** 0.5))


def _hip_regrid_slice(runner, plan, off_apron, olo, ohi, b0, b1, p0, p1, device):
    """K2 for the output pixels [p0, p1) of the bands [b0, b1) this rank owns; returns (slice tensor, far pixels, far values,
    [sum of the rank's source pixels, sum of its deposits])"""
    import torch
    dev = torch.device('cuda', device)
    hmap = torch.from_numpy(np.ascontiguousarray(runner.LightconeShell.map, dtype=np.float64)).to(dev)
    out = torch.empty(int(p1 - p0), dtype=torch.float64, device=dev)
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    plan.regrid_bands(b0, b1, hmap.data_ptr(), off_apron.data_ptr(), olo, ohi, out.data_ptr(), sums.data_ptr(), acc_f64=int(off_apron.dtype == torch.float64))
    pix, val = plan.far_fetch()
    return out, pix, val, sums.cpu().numpy()


def distributed_process(runner, kind, seed=42, device=None, compute=None, regrid=None, exchange='spatial', bounds=None,
                        regrid_slice=None, result='root', reach=None, compute_spatial=None):
    """Run `runner` (holding the FULL catalog on every rank) halo-sharded over the ranks of the default
    torch.distributed group.  Returns the final map on rank 0 (`result='root'`; None elsewhere) or on every rank
    (`result='all'`, exchange='slices').

    compute(runner, kind, cat_cols, device) -> (tensor accumulator, ctx); exchange='reduce': regrid(runner, ctx, acc,
    device) -> ndarray on rank 0; exchange='slices': bounds(runner, ctx, world) -> (band cuts, pixel bounds, needs) and
    regrid_slice(runner, ctx, offsets_with_apron, olo, ohi, b0, b1, p0, p1, device) -> (slice tensor, far pixels, far
    values, sums); reach(runner, ctx, my_summed_offsets) (collective) fixes the apron before `bounds` is asked for the
    ranges to exchange.  exchange='spatial': no accumulator travels at all -- every rank takes the halos whose discs can touch
    ITS run of ring bands and computes its own pixels only, compute_spatial(runner, kind, full_catalog_cols, device, world, rank)
    -> (slice tensor, ctx); the regrid then proceeds as for 'slices'.  All default to the HIP engine."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    compute = compute or _hip_compute
    cat = runner.HaloLightConeCatalog.cat
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', rank))
    if exchange == 'spatial':
        return _distributed_spatial(runner, kind, cat, device, compute_spatial or _hip_compute_spatial, bounds, regrid_slice, reach, result)
    order = shuffled_order(cat.size, seed)
    mine = order[shard_slices(cat.size, world)[rank]]
    cols = {k: np.ascontiguousarray(cat[k][mine]) for k in cat.dtype.names}
    # a failure on ONE rank (device out of memory, entry-list overflow) must not leave the others waiting in a collective:
    # the ranks agree on success before every exchange step and raise together
    err = None
    acc = ctx = None
    try:
        acc, ctx = compute(runner, kind, cols, device)
    except Exception as e:        # noqa: BLE001
        err = e
    _agree(err, acc.device if acc is not None else None)
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
    cuts, pb, needs = (bounds or _hip_bounds)(runner, ctx, world)
    if kind != 'baryonify':
        mine_sum = sliced_reduce(acc, pb, 1)
        full = gather_slices(mine_sum, pb, npix, result)
        return None if full is None else full.cpu().numpy().astype(np.float64)
    my_off = sliced_reduce(acc, pb, 3)
    return _regrid_own_slices(runner, ctx, my_off, cuts, pb, needs, device, bounds, regrid_slice, reach, result)


def _regrid_own_slices(runner, ctx, my_off, cuts, pb, needs, device, bounds, regrid_slice, reach, result):
    """collective tail of BaryonifyShell over pixel slices: every rank holds the COMPLETE pix_offsets of its own pixels; agree on
    the reach, exchange the apron rings, regrid the own bands, route the far deposits, gather the disjoint slices"""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    nside = int(runner.LightconeShell.NSIDE)
    npix = 12 * nside * nside
    if reach is None and bounds is None:
        reach = _hip_reach
    if reach is not None:
        err = None
        try:
            reach(runner, ctx, my_off)
        except Exception as e:        # noqa: BLE001
            err = e
        _agree(err, my_off.device)
        cuts, pb, needs = (bounds or _hip_bounds)(runner, ctx, world)          # the apron ranges follow the reach
    off_apron = halo_exchange(my_off, pb, needs, 3)
    err, sl, fpix, fval, sums = None, None, np.zeros(0, dtype=np.int64), np.zeros(0), np.zeros(2)
    try:
        sl, fpix, fval, sums = (regrid_slice or _hip_regrid_slice)(runner, ctx, off_apron, int(needs[rank][0]), int(needs[rank][1]),
                                                                  int(cuts[rank]), int(cuts[rank + 1]), int(pb[rank]), int(pb[rank + 1]), device)
    except Exception as e:        # noqa: BLE001
        err = e
    _agree(err, my_off.device)
    # far deposits (global pixel numbers, almost always none): every rank learns all of them and adds those in its slice
    gathered = [None] * world
    dist.all_gather_object(gathered, (np.asarray(fpix), np.asarray(fval), np.asarray(sums)))
    p0, p1 = int(pb[rank]), int(pb[rank + 1])
    for fp, fv, _ in gathered:
        if len(fp):
            m = (fp >= p0) & (fp < p1)
            if m.any():
                sl.index_add_(0, torch.from_numpy(fp[m] - p0).to(sl.device), torch.from_numpy(fv[m]).to(sl.device))
    full = gather_slices(sl, pb, npix, result)
    if full is None:
        return None
    new_map = full.cpu().numpy().astype(np.float64)
    new_sum = sum(float(g[2][1]) for g in gathered)              # (a rank's deposit sum includes the deposits it listed as far)
    old_sum = sum(float(g[2][0]) for g in gathered)
    assert np.isclose(new_sum, old_sum), "ERROR in pixel regridding, sum(new_map) [%0.14e] != sum(oldmap) [%0.14e]" % (new_sum, old_sum)
    return new_map


def _distributed_spatial(runner, kind, cat, device, compute_spatial, bounds, regrid_slice, reach, result):
    """exchange='spatial' (see distributed_process)"""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    cols = {k: np.ascontiguousarray(cat[k]) for k in cat.dtype.names}
    err, mine, ctx = None, None, None
    try:
        mine, ctx = compute_spatial(runner, kind, cols, device, world, rank)
    except Exception as e:        # noqa: BLE001
        err = e
    _agree(err, mine.device if mine is not None else None)
    nside = int(runner.LightconeShell.NSIDE)
    npix = 12 * nside * nside
    cuts, pb, needs = (bounds or _hip_bounds)(runner, ctx, world)
    if kind != 'baryonify':
        full = gather_slices(mine, pb, npix, result)
        return None if full is None else full.cpu().numpy().astype(np.float64)
    if reach is None and bounds is None:
        import functools
        reach = functools.partial(_hip_reach, bands=(int(cuts[rank]), int(cuts[rank + 1])))     # K1's per-tile maxima: no pass over the slice
    return _regrid_own_slices(runner, ctx, mine, cuts, pb, needs, device, bounds, regrid_slice, reach, result)


def _agree(err, device):
    """collective: raise on every rank if any rank failed"""
    import torch
    import torch.distributed as dist
    use_cuda = dist.get_backend() == 'nccl'
    flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=(device if (use_cuda and device is not None) else
                                                                                  ('cuda' if use_cuda else 'cpu')))
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if err is not None:
        raise err
    if int(flag.item()):
        raise RuntimeError("another rank failed in the halo-sharded run (see its log)")


def _spawn_worker(rank, world, port, runner, kind, seed, backend, pipe):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend == 'nccl':
        torch.cuda.set_device(rank)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        try:
            out = distributed_process(runner, kind, seed=seed, device=rank)
        except BaseException as e:        # noqa: BLE001  the parent re-raises
            out = e
        if rank == 0 and pipe is not None:
            pipe.send(out)
            pipe.close()
    finally:
        dist.destroy_process_group()


class SplitJoinParallel(object):
    """Drop-in for Parallelize.py:116-320 where `njobs` = number of GPUs.  `process()` blocks and returns the summed map.
    Unlike the reference it also accepts BaryonifyShell (offsets are reduced before the regrid, not final maps).

    backend='capi' (default): ONE process drives all GPUs through the C ABI (bfgx_*_shell_multi: halo shards, peer-to-peer
    slice exchange over xGMI, every device regrids and returns its own slice of the map).  backend='nccl': one process per
    GPU over torch.distributed / RCCL (distributed_process); the result comes back through a pipe, not a file.
    `devices`: explicit device list (default range(njobs)); a device may be named more than once."""

    def __init__(self, Runner, njobs=-1, seed=42, backend='capi', devices=None):
        from ..Runners import BaryonifyShell, PaintProfilesShell
        assert isinstance(Runner, (BaryonifyShell, PaintProfilesShell)), \
            f"Runner of type {type(Runner)} is not supported for SplitJoinParallel."
        assert backend in ('capi', 'nccl'), "backend must be 'capi' or 'nccl'"
        self.Runner, self.seed, self.backend = Runner, seed, backend
        if njobs == -1:
            from .. import _lib
            njobs = max(1, _lib.load().bfgx_device_count()) if devices is None else len(devices)
        self.njobs = njobs
        self.devices = list(range(njobs)) if devices is None else [int(d) for d in devices]
        assert len(self.devices) == self.njobs, "len(devices) must equal njobs"
        self.kind = 'baryonify' if isinstance(Runner, BaryonifyShell) else 'paint'
        self.last_stats = None

    def process(self):
        if self.njobs == 1 and self.devices == [0]:
            out = self.Runner.process()
            self.last_stats = self.Runner.last_stats
            return out
        if self.backend == 'capi':
            return self._process_capi()
        import torch.multiprocessing as mp
        import socket
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        ctx = mp.get_context('spawn')
        recv, send = ctx.Pipe(duplex=False)
        procs = [ctx.Process(target=_spawn_worker, args=(r, self.njobs, port, self.Runner, self.kind, self.seed, 'nccl', send if r == 0 else None))
                 for r in range(self.njobs)]
        for p in procs:
            p.start()
        send.close()
        try:
            out = recv.recv()
        except EOFError:
            out = None
        for p in procs:
            p.join()
        if out is None or any(p.exitcode != 0 for p in procs):
            raise RuntimeError("a rank of the halo-sharded run failed (exit codes %r)" % ([p.exitcode for p in procs],))
        if isinstance(out, BaseException):
            raise out
        return out

    def _process_capi(self):
        import ctypes as C
        from .. import _lib
        from ..Runners._model import build_model
        runner = self.Runner
        model, p_keys, keep = build_model(runner, 'displacement' if self.kind == 'baryonify' else 'projected')
        cat = runner.HaloLightConeCatalog.cat
        order = shuffled_order(cat.size, self.seed)                                 # Parallelize.py:255
        cols = [np.ascontiguousarray(cat[k][order], dtype=np.float64) for k in ['M', 'z', 'ra', 'dec'] + list(p_keys)]
        c, keep_cols = _lib.make_catalog_host(cols[0], cols[1], cols[2], cols[3], cols[4:])
        nside = int(runner.LightconeShell.NSIDE)
        npix = 12 * nside * nside
        new_map = _lib.pinned_empty(npix)
        devs = (C.c_int32 * self.njobs)(*self.devices)
        stats = _lib.bfgx_stats()
        acc64 = getattr(runner, 'acc_f64', None)
        if self.kind == 'baryonify':
            opts = _lib.bfgx_opts(0, _lib.acc_mode(acc64), 1, 1, 1, 0)
            orig_map = _lib.f8(runner.LightconeShell.map)
            rc = _lib.load().bfgx_baryonify_shell_multi(C.byref(c), C.byref(model), nside, orig_map.ctypes.data, new_map.ctypes.data,
                                                        self.njobs, devs, C.byref(opts), C.byref(stats))
        else:
            opts = _lib.bfgx_opts(0, 0, 1, 0, 1, 0)
            rc = _lib.load().bfgx_paint_shell_multi(C.byref(c), C.byref(model), nside, new_map.ctypes.data, self.njobs, devs,
                                                    C.byref(opts), C.byref(stats))
        _lib.check(rc)
        self.last_stats = {k: getattr(stats, k) for k, _ in stats._fields_}
        del keep, keep_cols
        return new_map


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
