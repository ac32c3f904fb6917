"""
BASELINE config 5 over N GPUs: slab decomposition of the periodic grid along its FIRST array axis.

The reference does this flow on one node with numpy (`ParticleSnapshot.make_map`, utils/io.py:622-670 -> `BaryonifyGrid.process`,
Runners/Map2DRunner.py:476-605 -> the notebook's P(k), examples/10_Reproduce_Schneider_deltaPk.ipynb cells 12 + 15); it has no
multi-process form of it.  Here rank r of W owns the planes [r n/W, (r + 1) n/W) of every n^3 array:

  1. particles are routed to the rank that owns their plane (one all_to_all_single of packed rows) and deposited there
     (`bfgx_deposit_particles_slab_device`): no grid data moves;
  2. every rank runs the halo loop for ITS planes over the whole (replicated) halo catalog, cutouts clipped to the slab
     (`bfgx_grid_plan_set_slab` + `bfgx_grid_offsets_device`): no exchange;
  3. the ranks agree on the apron A = ceil(max |offset along the first axis|) + 1 planes (all_reduce MAX), every rank regrids
     its source cells into its slab + A planes either side (`bfgx_grid_regrid_slab_device`) and hands the two aprons to its
     neighbours, which add them (one all_to_all_single, 2 A n^2 values per rank);
  4. P(k): transforms along the last two axes on the slab, one all_to_all transpose (n^2 (n/2+1) / W complex values per rank,
     1/W of them to each peer) so that every rank holds all planes of n/W columns of the middle axis, the transform along
     the first axis + binning there, all_reduce(SUM) of the 3 x Nk bin sums.

The per-rank compute is injected (`backend`): `HipBackend` is the product (libbfgx through `engine`); tests drive the same
collective logic on CPU/gloo with a numpy backend (tests/test_distributed_gloo.py).
"""
import numpy as np


def slab_bounds(n, world, rank):
    """planes [lo, lo + cnt) of the first axis owned by `rank` (n must divide evenly: the transpose needs equal blocks)"""
    if n % world:
        raise ValueError("the grid side (%d) must be a multiple of the number of ranks (%d)" % (n, world))
    cnt = n // world
    return rank * cnt, cnt


def _a2a(recv, send, outs, ins):
    import torch.distributed as dist
    if send.is_cuda and dist.get_backend() == 'gloo':          # functional rehearsal on one GPU: stage through the host
        r = recv.cpu()
        dist.all_to_all_single(r, send.cpu(), outs, ins)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, outs, ins)
    return recv


def _allreduce(t, op):
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == 'gloo':
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
    return t


def _route_particles_hip(cols, edges, n, world, backend):
    """route_particles through libbfgx (two kernels: owners + counts, then the columns packed by destination)"""
    import torch
    eng, dev = backend.engine, cols.device
    w, m = int(cols.shape[0]), int(cols.shape[1])
    owner = torch.empty(max(m, 1), dtype=torch.uint8, device=dev)
    counts = torch.empty(world, dtype=torch.int32, device=dev)
    eng.route_particles_count_device(cols[0].data_ptr(), m, n, edges.data_ptr(), world, owner.data_ptr(), counts.data_ptr(),
                                     device=backend.device, stream=backend.stream)
    t_in = counts.to(torch.int64)
    t_out = torch.empty_like(t_in)
    _a2a(t_out, t_in, None, None)                    # how many particles every rank sends me
    ins, outs = [[int(c) for c in row] for row in torch.stack([t_in, t_out]).tolist()]
    total = sum(ins)
    packed = cols.new_empty((w, total))
    cursor = torch.empty(world, dtype=torch.int32, device=dev)
    start = np.concatenate([[0], np.cumsum(ins)[:-1]]).astype(np.int64)
    eng.route_particles_fill_device([cols[i].data_ptr() for i in range(w)], m, owner.data_ptr(), world, start, total, cursor.data_ptr(),
                                    packed.data_ptr(), device=backend.device, stream=backend.stream)
    recv = cols.new_empty((w, sum(outs)))
    for i in range(w):
        _a2a(recv[i], packed[i], outs, ins)
    return recv


def route_particles(cols, edges, n, world, backend=None):
    """cols [w][m] (rows x, y, z[, mass]; columns = particles held by this rank) -> [w][m'] the particles whose first-axis bin
    lies in this rank's slab.  Bin rule of np.histogramdd: edges[b] <= x < edges[b + 1], the last edge inclusive, anything
    else dropped.  One all_to_all_single per coordinate (the blocks per destination are contiguous in a sorted column).
    With a HipBackend and columns on its GPU the owners, counts and the packing are libbfgx kernels instead of a torch sort."""
    import torch
    if backend is not None and getattr(backend, 'engine', None) is not None and cols.is_cuda and cols.is_contiguous() and world <= 255 \
            and cols.shape[0] <= 4 and cols.dtype == torch.float64:
        return _route_particles_hip(cols, edges, n, world, backend)
    cnt = n // world
    x = cols[0]
    b = torch.bucketize(x, edges, right=True) - 1
    b = torch.where(x == edges[-1], torch.full_like(b, n - 1), b)
    inside = (b >= 0) & (b < n)
    owner = torch.where(inside, torch.div(b, cnt, rounding_mode='floor'), torch.full_like(b, world)).to(torch.uint8 if world < 255 else torch.int32)
    order = torch.argsort(owner, stable=True)                                                              # `world` = dropped: sorted last
    counts = torch.bincount(owner.to(torch.int64), minlength=world + 1)[:world]
    ins = [int(c) for c in counts.tolist()]
    keep = order[:sum(ins)]
    t_in = counts.to(torch.int64).contiguous()
    t_out = torch.empty_like(t_in)
    _a2a(t_out, t_in, None, None)                    # how many particles every rank sends me
    outs = [int(c) for c in t_out.tolist()]
    recv = cols.new_empty((cols.shape[0], sum(outs)))
    for i in range(cols.shape[0]):
        _a2a(recv[i], cols[i][keep].contiguous(), outs, ins)
    return recv


def exchange_aprons(buf, apron, cnt, world):
    """buf [cnt + 2 apron][...]: the regridded slab with its aprons -> the slab [cnt][...] with the neighbours' aprons added.
    My lower apron belongs to rank - 1, my upper one to rank + 1 (periodic); with world == 2 both go to the same peer."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    own = buf[apron:apron + cnt].clone()
    if apron == 0:
        return own
    if apron > cnt:
        raise ValueError("apron of %d planes exceeds the slab thickness %d: use fewer ranks" % (apron, cnt))
    lower, upper = buf[:apron].contiguous(), buf[apron + cnt:].contiguous()
    if world == 1:                                   # periodic wrap onto myself
        own[cnt - apron:] += lower
        own[:apron] += upper
        return own
    prev_r, next_r = (rank - 1) % world, (rank + 1) % world
    plane = lower[0].numel()
    parts = {r: [] for r in range(world)}
    parts[prev_r].append(lower.view(-1))             # fixed order per destination: lower first, then upper
    parts[next_r].append(upper.view(-1))
    send = torch.cat([torch.cat(parts[r]) if parts[r] else buf.new_empty(0) for r in range(world)])
    ins = [sum(p.numel() for p in parts[r]) for r in range(world)]
    outs = [0] * world
    outs[next_r] += apron * plane                    # rank + 1 sends me ITS lower apron (my last planes)
    outs[prev_r] += apron * plane                    # rank - 1 sends me ITS upper apron (my first planes)
    recv = _a2a(buf.new_empty(sum(outs)), send, outs, ins)
    off = [0]
    for o in outs:
        off.append(off[-1] + o)
    shape = (apron,) + tuple(buf.shape[1:])
    if prev_r == next_r:                             # one peer: its message is [its lower, its upper]
        m = recv[off[prev_r]:off[prev_r + 1]]
        own[cnt - apron:] += m[:apron * plane].view(shape)
        own[:apron] += m[apron * plane:].view(shape)
    else:
        own[cnt - apron:] += recv[off[next_r]:off[next_r + 1]].view(shape)
        own[:apron] += recv[off[prev_r]:off[prev_r + 1]].view(shape)
    return own


def transpose_planes_to_columns(work, n, world):
    """work [cnt][n][nz] complex (this rank's planes, all columns) -> [n][cnt][nz] (all planes, this rank's columns of the
    middle axis): one all_to_all_single with equal blocks"""
    import torch
    cnt = n // world
    nz = work.shape[2]
    send = torch.view_as_real(work.view(cnt, world, cnt, nz).permute(1, 0, 2, 3).contiguous())      # [dest][plane][col][nz][2]
    recv = torch.empty_like(send)
    blk = cnt * cnt * nz * 2
    _a2a(recv.view(-1), send.view(-1), [blk] * world, [blk] * world)
    return torch.view_as_complex(recv).reshape(n, cnt, nz)                                           # source rank r holds planes r cnt ...


class HipBackend(object):
    """per-rank compute through libbfgx (engine.GridPlan and the slab entry points); tensors live on `device`"""

    def __init__(self, model, keep, bins, redshift, max_halos, device=0):
        import torch
        from .. import engine
        self.torch, self.engine = torch, engine
        self.dev = torch.device('cuda', device)
        self.device = device
        self.n = len(bins)
        self.stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.plan = engine.GridPlan(model, keep, bins, 3, redshift, max_halos, device=device, stream=self.stream)

    def deposit(self, cols, edges, lo, cnt):
        t = self.torch
        assert cols.is_contiguous()
        out = t.empty((cnt, self.n, self.n), dtype=t.float64, device=self.dev)
        self.engine.deposit_particles_slab_device(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), cols[3].data_ptr() if cols.shape[0] > 3 else 0,
                                                  cols.shape[1], self.n, edges.data_ptr(), lo, cnt, out.data_ptr(), ndim=3, device=self.device,
                                                  stream=self.stream)
        return out

    def offsets(self, cat_dev, lo, cnt):
        t = self.torch
        self.plan.set_slab(lo, cnt)
        off = t.empty((cnt, self.n, self.n, 3), dtype=t.float64, device=self.dev)
        self.pairs = self.plan.offsets(cat_dev, off.data_ptr())
        return off

    def regrid(self, slab, off, apron, lo, cnt):
        t = self.torch
        self.plan.set_slab(lo, cnt)
        buf = t.empty((cnt + 2 * apron, self.n, self.n), dtype=t.float64, device=self.dev)
        sums = t.zeros(2, dtype=t.float64, device=self.dev)
        missed = t.zeros(1, dtype=t.int32, device=self.dev)
        self.plan.regrid_slab(slab.data_ptr(), off.data_ptr(), apron, buf.data_ptr(), sums.data_ptr(), missed.data_ptr())
        return buf, sums, missed

    def fft_planes(self, slab):
        t = self.torch
        cnt = slab.shape[0]
        work = t.zeros((cnt, self.n, self.engine.fft_pitch(self.n)), dtype=t.complex128, device=self.dev)     # rows padded to 128-byte lines
        self.engine.fft_slab_planes_device(slab.data_ptr(), self.n, cnt, work.data_ptr(), device=self.device, stream=self.stream)
        return work

    def fft_axis0_pk(self, work, col0, L, nk):
        t = self.torch
        sums = t.zeros((2, nk), dtype=t.float64, device=self.dev)
        cnt = t.zeros(nk, dtype=t.int64, device=self.dev)
        self.engine.fft_slab_axis0_pk_device(work.data_ptr(), self.n, work.shape[1], col0, L, nk, sums[0].data_ptr(), sums[1].data_ptr(),
                                             cnt.data_ptr(), device=self.device, stream=self.stream)
        return sums, cnt

    def close(self):
        self.plan.close()


def slab_step(backend, rows, cat, n, L, nk, timers=None):
    """One pass of config 5 on this rank's slab (collective): particles `rows` [3|4][m] (x, y, z[, mass] as rows) held by this rank, halo catalog `cat`
    (whatever the backend's offsets() takes: replicated on every rank).  Returns (new slab [cnt][n][n], k_cen, Pk, counts,
    [sum of the deposited map, sum of the regridded map] over all ranks)."""
    import time
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, cnt = slab_bounds(n, world, rank)
    t_last = [time.perf_counter()]

    def lap(name):                                   # optional per-stage wall times (synchronising: diagnostics only)
        if timers is not None:
            if rows.is_cuda:
                torch.cuda.synchronize()
            now = time.perf_counter()
            timers[name] = timers.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now

    edges = torch.from_numpy(np.linspace(0, L, n + 1)).to(rows.device)                         # io.py:651 np.linspace(0, L, N_grid + 1)
    mine = route_particles(rows, edges, n, world, backend)
    lap('route')
    slab = backend.deposit(mine, edges, lo, cnt)                                               # ParticleSnapshot.make_map
    lap('deposit')
    off = backend.offsets(cat, lo, cnt)                                                        # BaryonifyGrid halo loop
    lap('offsets')
    o1 = off[..., 1]
    amax = torch.nan_to_num(o1, nan=0.0, posinf=0.0, neginf=0.0).abs().max().reshape(1) if o1.numel() else off.new_zeros(1)
    _allreduce(amax, dist.ReduceOp.MAX)
    apron = min(int(np.ceil(float(amax.item()))) + 1, (n - cnt) // 2) if world > 1 else 0
    lap('apron')
    buf, sums, missed = backend.regrid(slab, off, apron, lo, cnt)                              # Map2DRunner.py:577-605
    flag = missed.to(torch.int32).reshape(1).clone()
    _allreduce(flag, dist.ReduceOp.MAX)
    if int(flag.item()):
        raise RuntimeError("a regridded cell fell outside the slab + apron (%d planes): displacement larger than the slab allows" % apron)
    lap('regrid')
    new = exchange_aprons(buf, apron, cnt, world)
    sums = sums.clone()
    _allreduce(sums, dist.ReduceOp.SUM)
    lap('aprons')
    work = backend.fft_planes(new)                                                             # notebook 10, cells 12 + 15
    lap('fft_planes')
    cols = transpose_planes_to_columns(work, n, world) if world > 1 else work
    lap('transpose')
    psum, pcnt = backend.fft_axis0_pk(cols.contiguous(), lo, L, nk)
    psum, pcnt = psum.clone(), pcnt.clone()
    _allreduce(psum, dist.ReduceOp.SUM)
    _allreduce(pcnt, dist.ReduceOp.SUM)
    lap('fft_axis0_pk')
    with np.errstate(divide='ignore', invalid='ignore'):
        c = pcnt.cpu().numpy()
        pk = psum[0].cpu().numpy() / c
        kc = psum[1].cpu().numpy() / c
    return new, kc, pk, c, sums.cpu().numpy()
