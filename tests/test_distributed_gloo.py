"""N>1 path on CPU: two gloo ranks run the product's sharding + collective logic
(baryonification_amd.utils.Parallelize.distributed_process) with the CPU oracle injected as the per-shard
compute; the result must equal the single-process answer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from baryonification_amd.utils.Parallelize import band_partition, distributed_process, shard_slices, shuffled_order
from helpers import load_golden, oracle_run, product_runner


def _oracle_compute(runner, kind, cols, device):
    from oracle import oracle as O
    g = runner._golden
    axes = [np.log(1 + g['tab_z']), np.log(g['tab_M']), np.log(g['tab_r'])]
    bg_r, bg_m = O.Background.from_dict(g['cosmo_runner']), O.Background.from_dict(g['cosmo_model'])
    if kind == 'baryonify':
        tab = O.Table(axes, g['tab_values'], g['rdelta'], g['eps_model'])
        acc = O.baryonify_offsets(g['nside'], cols, tab, g['eps_runner'], bg_r, bg_m).reshape(-1)
    else:
        with np.errstate(divide='ignore'):
            tab = O.Table(axes, np.log(g['tab_values']))
        acc = O.paint_shell(g['nside'], cols, tab, g['eps_runner'], bg_r)
    return torch.from_numpy(acc), None


def _oracle_regrid(runner, ctx, acc, device):
    from oracle import oracle as O
    g = runner._golden
    return O.regrid(g['nside'], g['map_in'], acc.numpy().reshape(-1, 3))


def _ring_bands(nside, BR=8):
    """first RING pixel of bands of BR rings (what ShellPlan.bands() reports for the HIP tiling)"""
    first = []
    for ring in list(range(1, 4 * nside, BR)):
        if ring < nside:
            first.append(2 * ring * (ring - 1))
        elif ring < 3 * nside:
            first.append(2 * nside * (nside - 1) + (ring - nside) * 4 * nside)
        else:
            q = 4 * nside - ring
            first.append(12 * nside * nside - 2 * q * (q + 1))
    return np.array(first + [12 * nside * nside], dtype=np.int64)


APRON_RINGS = 4          # the oracle regrids by scattering: sources up to 4 rings outside a slice are evaluated for it


def _ring_first(nside, ring):
    if ring < 1:
        return 0
    if ring >= 4 * nside:
        return 12 * nside * nside
    if ring < nside:
        return 2 * ring * (ring - 1)
    if ring < 3 * nside:
        return 2 * nside * (nside - 1) + (ring - nside) * 4 * nside
    q = 4 * nside - ring
    return 12 * nside * nside - 2 * q * (q + 1)


def _oracle_bounds(runner, ctx, world, BR=8):
    nside = runner._golden['nside']
    first = _ring_bands(nside, BR)
    cuts = band_partition(first, world)
    needs = [(_ring_first(nside, 1 + BR * int(cuts[j]) - APRON_RINGS), _ring_first(nside, 1 + BR * int(cuts[j + 1]) + APRON_RINGS))
             for j in range(world)]
    return cuts, first[cuts], needs


def _oracle_regrid_slice(runner, ctx, off_apron, olo, ohi, b0, b1, p0, p1, device):
    """The output pixels [p0, p1) of this rank's bands: the oracle scatters the source pixels [olo, ohi) it was handed (its
    own + the apron) and keeps what lands in the slice.  Returns (slice, far pixels, far values, [sum in, sum out])."""
    from oracle import oracle as O
    g = runner._golden
    npix = 12 * g['nside'] ** 2
    assert olo <= p0 and p1 <= ohi and off_apron.numel() == 3 * (ohi - olo)
    off = np.zeros((npix, 3))
    off[olo:ohi] = off_apron.numpy().reshape(-1, 3)
    new_map = np.zeros(npix)
    O.lib().bfgo_regrid_range(g['nside'], olo, ohi, O._ptr(O._f8(g['map_in'])), O._ptr(off), O._ptr(new_map))
    own = np.zeros(npix)
    O.lib().bfgo_regrid_range(g['nside'], p0, p1, O._ptr(O._f8(g['map_in'])), O._ptr(off), O._ptr(own))
    assert own[:olo].sum() == 0 and own[ohi:].sum() == 0                 # the apron is wide enough for every deposit
    sl = new_map[p0:p1].copy()
    return torch.from_numpy(sl), np.zeros(0, dtype=np.int64), np.zeros(0), np.array([g['map_in'][p0:p1].sum(), sl.sum()])


def _oracle_compute_spatial(runner, kind, cols, device, world, rank):
    """exchange='spatial' with the oracle: the rank's own pixels of the full-catalog result (which halos a rank takes is the HIP
    side's business, tests/test_gpu_fullsize.py::test_band_restricted_pass_with_selected_halos_equals_full_pass)"""
    acc, _ = _oracle_compute(runner, kind, cols, device)
    cuts, pb, _ = _oracle_bounds(runner, None, world)
    w = 3 if kind == 'baryonify' else 1
    return acc[int(pb[rank]) * w:int(pb[rank + 1]) * w].clone(), None


def _worker(rank, world, port, name, out_path, exchange='slices', result='root'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = load_golden(name)
        runner = product_runner(g)
        runner._golden = g
        kind = 'baryonify' if g['kind'] == 'baryonify' else 'paint'
        out = distributed_process(runner, kind, seed=42, device=rank, compute=_oracle_compute, regrid=_oracle_regrid, exchange=exchange,
                                  bounds=_oracle_bounds, regrid_slice=_oracle_regrid_slice, result=result,
                                  compute_spatial=_oracle_compute_spatial)
        if result == 'all':
            np.save(out_path + '.%d.npy' % rank, out)            # every rank holds the map
        elif rank == 0:
            np.save(out_path, out)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def _failing_compute(runner, kind, cols, device):
    if dist.get_rank() == 1:
        raise MemoryError("rank 1 ran out of device memory")
    return _oracle_compute(runner, kind, cols, device)


def _failing_compute_spatial(runner, kind, cols, device, world, rank):
    if rank == 1:
        raise MemoryError("rank 1 ran out of device memory")
    return _oracle_compute_spatial(runner, kind, cols, device, world, rank)


def _worker_one_rank_fails(rank, world, port, name, out_path, exchange='slices'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = load_golden(name)
        runner = product_runner(g)
        runner._golden = g
        try:
            distributed_process(runner, 'baryonify', seed=42, device=rank, compute=_failing_compute, bounds=_oracle_bounds, exchange=exchange,
                                regrid_slice=_oracle_regrid_slice, compute_spatial=_failing_compute_spatial)
            raised = 'none'
        except MemoryError:
            raised = 'own'
        except RuntimeError as e:
            raised = 'peer' if 'another rank failed' in str(e) else 'other'
        with open(out_path + '.%d.txt' % rank, 'w') as f:
            f.write(raised)
    finally:
        dist.destroy_process_group()


def test_shard_slices_cover_catalog_like_reference():
    for n, w in ((10, 3), (1000, 8), (7, 8), (0, 2), (16, 4)):
        sl = shard_slices(n, w)
        idx = np.concatenate([np.arange(n)[s] for s in sl]) if n else np.zeros(0, dtype=int)
        assert np.array_equal(idx, np.arange(n)) and len(sl) == w
        per = int(np.ceil(n / w)) if n else 0
        assert all((s.stop - s.start) <= per for s in sl)          # Parallelize.py:253
    o = shuffled_order(100, 42)
    assert np.array_equal(np.sort(o), np.arange(100))
    assert np.array_equal(o, np.random.default_rng(42).choice(100, size=100, replace=False))


def test_band_partition_balances_pixels():
    for nside, world in ((64, 2), (64, 3), (128, 8), (16, 5)):
        first = _ring_bands(nside)
        cuts = band_partition(first, world)
        assert cuts[0] == 0 and cuts[-1] == first.size - 1 and np.all(np.diff(cuts) >= 0) and cuts.size == world + 1
        share = np.diff(first[cuts]) / first[-1]
        assert abs(share.sum() - 1) < 1e-15 and np.all(np.abs(share - 1 / world) < 0.35 / world + 8 * 4 * nside / first[-1])


@pytest.mark.parametrize('world,exchange', [(2, 'slices'), (3, 'slices'), (2, 'reduce'), (2, 'spatial'), (4, 'spatial')])
@pytest.mark.parametrize('name', ['lowz_baryonify', 'lowz_paint'])
def test_multi_rank_gloo_equals_single_process(tmp_path, name, world, exchange):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / 'out.npy')
    mp.spawn(_worker, args=(world, port, name, out_path, exchange), nprocs=world, join=True)
    out = np.load(out_path)
    ref = oracle_run(load_golden(name))
    # summation order differs (shuffle + two partial sums): float64 round-off only
    assert np.abs(out - ref).max() <= 1e-11 * np.abs(ref).max()


def test_five_rank_slice_exchange(tmp_path):
    """more ranks than a fixture usually sees: interior ranks exchange a ring apron with both neighbours"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / 'out.npy')
    mp.spawn(_worker, args=(5, port, 'lowz_baryonify', out_path, 'slices'), nprocs=5, join=True)
    out = np.load(out_path)
    ref = oracle_run(load_golden('lowz_baryonify'))
    assert np.abs(out - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize('name', ['lowz_baryonify', 'lowz_paint'])
def test_eight_rank_slice_exchange_both_runners(tmp_path, name):
    """the node size the scaling bench uses (8 ranks), both runners, result on every rank (all_gather of the slices)"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / 'out')
    mp.spawn(_worker, args=(8, port, name, out_path, 'slices', 'all'), nprocs=8, join=True)
    ref = oracle_run(load_golden(name))
    for rank in range(8):
        out = np.load(out_path + '.%d.npy' % rank)
        assert np.abs(out - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize('exchange', ['slices', 'spatial'])
def test_one_failing_rank_raises_on_every_rank(tmp_path, exchange):
    """a rank that fails before a collective must not leave its peers blocked in it: all ranks raise"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / 'flag')
    mp.spawn(_worker_one_rank_fails, args=(3, port, 'lowz_baryonify', out_path, exchange), nprocs=3, join=True)
    got = [open(out_path + '.%d.txt' % r).read() for r in range(3)]
    assert got == ['peer', 'own', 'peer']


def _worker_route(rank, world, port, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from baryonification_amd.utils.Parallelize import route_halos
        rng = np.random.default_rng(100 + rank)
        n = 5000 + 100 * rank
        first = rng.integers(1, 400, n)
        last = np.minimum(first + rng.integers(0, 30, n), 399)
        first[:5], last[:5] = 1, 399                                    # every rank
        first[5:9], last[5:9] = 50, 49                                  # nobody
        ident = (rank * 1_000_000 + np.arange(n)).astype(np.float64)
        cols = [torch.from_numpy(ident), torch.from_numpy(rng.normal(size=n)), torch.from_numpy(first.astype(np.float64)), torch.from_numpy(last.astype(np.float64))]
        bounds = np.linspace(1, 400, world + 1).astype(np.int64)
        got = route_halos(cols, torch.from_numpy(np.stack([first, last], axis=1).astype(np.int32)), bounds)
        np.save(out_path + '.%d.npy' % rank, got.numpy())
        # the fixed-capacity form (equal splits, nothing read back): the same halos, NaN-padded; a block that is too small raises the flag
        from baryonification_amd.utils.Parallelize import route_halos_fixed
        rings_t = torch.from_numpy(np.stack([first, last], axis=1).astype(np.int32))
        fix, ovf = route_halos_fixed(cols, rings_t, bounds, 4000)
        assert fix.shape == (4, world * 4000) and int(ovf.item()) == 0
        keepm = ~torch.isnan(fix[0])
        assert np.array_equal(np.sort(fix[0][keepm].numpy()), np.sort(got[0].numpy()))
        order_a, order_b = np.argsort(fix[0][keepm].numpy()), np.argsort(got[0].numpy())
        assert np.array_equal(fix[:, keepm].numpy()[:, order_a], got.numpy()[:, order_b])
        _, ovf2 = route_halos_fixed(cols, rings_t, bounds, 50)
        assert int(ovf2.item()) == 1
    finally:
        dist.destroy_process_group()


def test_route_halos_gloo_every_halo_reaches_every_rank_it_touches(tmp_path):
    """spatial sharding of scattered halos (torch path of route_halos): rank j receives exactly the halos, from all ranks, whose
    ring range touches its rings"""
    world = 3
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / 'route')
    mp.spawn(_worker_route, args=(world, port, out_path), nprocs=world, join=True)
    bounds = np.linspace(1, 400, world + 1).astype(np.int64)
    got = [np.load(out_path + '.%d.npy' % r) for r in range(world)]
    total = 0
    for j in range(world):
        g = got[j]
        assert g.shape[0] == 4
        first, last = g[2], g[3]
        assert np.all((first < bounds[j + 1]) & (last >= bounds[j]) & (first <= last))         # only halos that touch my rings
        assert np.unique(g[0]).size == g.shape[1]                                               # each once
        total += g.shape[1]
    # nothing is lost: regenerate the inputs and count the (halo, rank) incidences
    want = 0
    for rank in range(world):
        rng = np.random.default_rng(100 + rank)
        n = 5000 + 100 * rank
        first = rng.integers(1, 400, n)
        last = np.minimum(first + rng.integers(0, 30, n), 399)
        first[:5], last[:5] = 1, 399
        first[5:9], last[5:9] = 50, 49
        for j in range(world):
            want += int(((first < bounds[j + 1]) & (last >= bounds[j]) & (first <= last)).sum())
    assert total == want


def _worker_overlapped_gather(rank, world, port, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from baryonification_amd.utils.Parallelize import gather_slices, gather_slices_start
        pb = np.array([0, 700, 1500, 1501, 4000][:world + 1] if world == 4 else np.linspace(0, 3000, world + 1).astype(np.int64))
        npix = int(pb[-1])
        group = dist.new_group(backend='gloo')                         # the gathers travel on a communicator of their own
        # bench.py's pattern: pass i computes into buffer pair i % 2 and starts its gather; the pair is written again only after that
        # gather has been waited for; a fence waits for what is in flight.  Every pass's assembled map is checked.
        slices = [torch.zeros(int(pb[rank + 1] - pb[rank]), dtype=torch.float64) for _ in (0, 1)]
        fins = [torch.full((npix,), -1.0, dtype=torch.float64) if rank == 0 else None for _ in (0, 1)]
        pending, seen = [None, None], []
        for i in range(7):
            k = i % 2
            if pending[k] is not None:
                pending[k][0].wait()
                if rank == 0:
                    seen.append((pending[k][1], fins[k].clone()))
                pending[k] = None
            val = torch.arange(int(pb[rank]), int(pb[rank + 1]), dtype=torch.float64) + 10000.0 * i
            if rank == 0:
                fins[k][int(pb[0]):int(pb[1])] = val                    # rank 0 regrids straight into the final map
            else:
                slices[k].copy_(val)
            pending[k] = (gather_slices_start(slices[k], pb, npix, out=fins[k], group=group), i)
            dist.all_reduce(torch.ones(1))                              # (the step's own collectives go on on the default group meanwhile)
        for k in (0, 1):
            if pending[k] is not None:
                pending[k][0].wait()
                if rank == 0:
                    seen.append((pending[k][1], fins[k].clone()))
        if rank == 0:
            assert sorted(i for i, _ in seen) == list(range(7))
            for i, m in seen:
                assert torch.equal(m, torch.arange(npix, dtype=torch.float64) + 10000.0 * i), i
            # the in-line form gives the same map
            full = torch.full((npix,), -1.0, dtype=torch.float64)
            full[:int(pb[1])] = torch.arange(int(pb[1]), dtype=torch.float64)
            gather_slices(slices[0], pb, npix, 'root', out=full, root_in_place=True)
            np.save(out_path, np.array([1.0]))
        else:
            slices[0].copy_(torch.arange(int(pb[rank]), int(pb[rank + 1]), dtype=torch.float64))
            gather_slices(slices[0], pb, npix, 'root', out=None, root_in_place=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_overlapped_gather_two_buffers_own_group(tmp_path, world):
    """Parallelize.gather_slices_start: the slices of pass i travel to rank 0 asynchronously on a process group of their own while the ranks
    go on (bench.py's N > 1 loop); with two buffer pairs and a wait before reuse every pass's map arrives whole (one rank owns a single pixel)"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / 'ok.npy')
    mp.spawn(_worker_overlapped_gather, args=(world, port, out_path), nprocs=world, join=True)
    assert os.path.exists(out_path)
