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


def _oracle_bounds(runner, ctx, world):
    first = _ring_bands(runner._golden['nside'])
    cuts = band_partition(first, world)
    return cuts, first[cuts]


def _oracle_regrid_slice(runner, ctx, my_off, b0, b1, wlo, whi, device):
    """regrid of the source pixels of this rank's slice only (oracle bfgo_regrid_range), returned as the window"""
    from oracle import oracle as O
    g = runner._golden
    first = _ring_bands(g['nside'])
    p0, p1 = int(first[b0]), int(first[b1])
    npix = 12 * g['nside'] ** 2
    off = np.zeros((npix, 3))
    off[p0:p1] = my_off.numpy().reshape(-1, 3)
    new_map = np.zeros(npix)
    O.lib().bfgo_regrid_range(g['nside'], p0, p1, O._ptr(O._f8(g['map_in'])), O._ptr(off), O._ptr(new_map))
    assert new_map[:wlo].sum() == 0 and new_map[whi:].sum() == 0         # deposits stay inside the window
    return torch.from_numpy(new_map[wlo:whi].copy())


def _worker(rank, world, port, name, out_path, exchange='slices'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = load_golden(name)
        runner = product_runner(g)
        runner._golden = g
        kind = 'baryonify' if g['kind'] == 'baryonify' else 'paint'
        out = distributed_process(runner, kind, seed=42, device=rank, compute=_oracle_compute, regrid=_oracle_regrid, exchange=exchange,
                                  bounds=_oracle_bounds, regrid_slice=_oracle_regrid_slice)
        if rank == 0:
            np.save(out_path, out)
        else:
            assert out is None
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


@pytest.mark.parametrize('world,exchange', [(2, 'slices'), (3, 'slices'), (2, 'reduce')])
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
    """more ranks than a fixture usually sees: interior ranks have windows overlapping both neighbours"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / 'out.npy')
    mp.spawn(_worker, args=(5, port, 'lowz_baryonify', out_path, 'slices'), nprocs=5, join=True)
    out = np.load(out_path)
    ref = oracle_run(load_golden('lowz_baryonify'))
    assert np.abs(out - ref).max() <= 1e-11 * np.abs(ref).max()
