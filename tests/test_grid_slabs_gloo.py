"""Config 5 over N ranks on CPU: the collective logic of baryonification_amd.utils.GridSlabs (particle routing, apron
exchange of the slab regrid, FFT transpose, reductions) runs over gloo with the CPU oracle injected as the per-rank compute;
the assembled slabs and the P(k) must equal the single-process oracle pipeline
(ParticleSnapshot.make_map -> BaryonifyGrid -> notebook-10 P(k))."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from baryonification_amd.utils import GridSlabs as GS

N, L, NK, NH, NPART = 32, 90.0, 12, 60, 40_000


def _case():
    from baryonification_amd import synthetic as syn
    rng = np.random.default_rng(5)
    M = (10 ** rng.uniform(13.2, 14.9, NH)).astype(np.float32).astype(np.float64)
    pos = rng.uniform(0, L, (NH, 3)).astype(np.float32).astype(np.float64)
    pos[:3] = [[0.4, 40.0, 40.0], [L - 0.3, 10.0, 80.0], [L / 2 + 0.01, 45.0, 45.0]]          # balls across the box face and a slab face
    z, Mt, r = np.array([0.0, 0.01]), np.geomspace(0.99e12, 1.01e15, 10), np.geomspace(1e-3, 3e2, 500)
    table = syn.displacement_table(z, Mt, r) * 250.0                                           # up to ~2 cells
    rows = np.concatenate([rng.uniform(-0.01 * L, 1.01 * L, (NPART, 3)), rng.uniform(0.5, 2.0, (NPART, 1))], axis=1)
    rows[0, :3], rows[1, :3] = 0.0, L
    cat = {'M': M, 'x': pos[:, 0].copy(), 'y': pos[:, 1].copy(), 'z': pos[:, 2].copy()}
    return dict(cat=cat, z=z, Mt=Mt, r=r, table=table, rows=rows, bins=(np.arange(N) + 0.5) * (L / N), cosmo=dict(syn.COSMO, w0=-1.0))


class OracleBackend(object):
    """per-rank compute by the CPU oracle (test infrastructure): the slab is cut out of full-grid oracle results"""

    def __init__(self, c):
        from oracle import grid as G
        from oracle import oracle as O
        self.G, self.c = G, c
        self.tab = O.Table([np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])], c['table'], False, 5.0)
        self.bg = G.grid_background(c['cosmo'])

    def deposit(self, rows, edges, lo, cnt):
        r = rows.numpy()
        full = self.G.make_map([r[0], r[1], r[2]], r[3], L, N)              # the routed rows all lie in my planes
        assert full[:lo].sum() == 0 and full[lo + cnt:].sum() == 0
        return torch.from_numpy(full[lo:lo + cnt].copy())

    def offsets(self, cat, lo, cnt):
        off = self.G.baryonify_grid_offsets((N,) * 3, self.c['bins'], cat, 0.0, self.tab, 5.0, self.bg)
        return torch.from_numpy(off.reshape(N, N, N, 3)[lo:lo + cnt].copy())

    def regrid(self, slab, off, apron, lo, cnt):
        full_map, full_off = np.zeros((N,) * 3), np.zeros((N, N, N, 3))
        full_map[lo:lo + cnt], full_off[lo:lo + cnt] = slab.numpy(), off.numpy()
        out = self.G.regrid_offsets(full_map, full_off.reshape(-1, 3))
        planes = np.arange(lo - apron, lo + cnt + apron) % N
        rest = np.ones(N, dtype=bool)
        rest[planes] = False
        missed = torch.tensor([int(np.abs(out[rest]).sum() > 0)], dtype=torch.int32)
        return torch.from_numpy(out[planes].copy()), torch.tensor([full_map.sum(), out.sum()]), missed

    def fft_planes(self, slab):
        return torch.from_numpy(np.fft.fft(np.fft.rfft(slab.numpy(), axis=2), axis=1))

    def fft_axis0_pk(self, work, col0, L_, nk):
        F = np.fft.fft(work.numpy(), axis=0)                                            # [N][cols][nz]
        klin = np.fft.fftfreq(N, 1 / (2 * np.pi / L_) / N)
        kb = np.linspace(2 * np.pi / L_, 2 * np.pi / L_ * N / 2, nk + 1)
        nz, ncol = F.shape[2], F.shape[1]
        k = np.sqrt(klin[:, None, None] ** 2 + klin[None, None, :nz] ** 2 + klin[None, col0:col0 + ncol, None] ** 2)
        k[:, :, nz - 1] = np.sqrt(klin[:, None] ** 2 + klin[N // 2] ** 2 + klin[None, col0:col0 + ncol] ** 2)     # c = N/2 is klin[N/2] (negative)
        kind = np.floor((k - kb[0]) / (kb[1] - kb[0])).astype(int)
        ok = (kind >= 0) & (kind < nk)
        mult = np.where((np.arange(nz) > 0) & (np.arange(nz) < N // 2), 2, 1)[None, None, :] * np.ones(k.shape, dtype=int)
        P = (F.conj() * F).real
        sums = np.stack([np.bincount(kind[ok], weights=(mult * P)[ok], minlength=nk), np.bincount(kind[ok], weights=(mult * k)[ok], minlength=nk)])
        cnt = np.bincount(kind[ok], weights=mult[ok], minlength=nk).astype(np.int64)
        return torch.from_numpy(sums), torch.from_numpy(cnt)


def _worker(rank, world, port, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        c = _case()
        rows = torch.from_numpy(np.ascontiguousarray(c['rows'][rank::world].T))         # every rank holds a share of the snapshot, [4][m]
        new, kc, pk, cnt, sums = GS.slab_step(OracleBackend(c), rows, c['cat'], N, L, NK)
        np.savez(out_path + '.%d.npz' % rank, new=new.numpy(), kc=kc, pk=pk, cnt=cnt, sums=sums)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_slab_pipeline_gloo_equals_single_process(tmp_path, world):
    from oracle import grid as G
    from oracle import oracle as O
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / 'slab')
    mp.spawn(_worker, args=(world, port, out_path), nprocs=world, join=True)
    c = _case()
    r = c['rows']
    m0 = G.make_map([r[:, 0], r[:, 1], r[:, 2]], r[:, 3], L, N)
    tab = O.Table([np.log(1 + c['z']), np.log(c['Mt']), np.log(c['r'])], c['table'], False, 5.0)
    off = G.baryonify_grid_offsets((N,) * 3, c['bins'], c['cat'], 0.0, tab, 5.0, G.grid_background(c['cosmo']))
    assert 1.0 < np.abs(np.nan_to_num(off.reshape(-1, 3)[:, 1])).max() < 6.0            # aprons really carry deposits
    ref = G.regrid_offsets(m0, off)
    kc0, pk0, cnt0 = G.power_spectrum(ref, L, NK)
    got = [np.load(out_path + '.%d.npz' % rk) for rk in range(world)]
    new = np.concatenate([g['new'] for g in got], axis=0)
    assert new.shape == ref.shape and np.abs(new - ref).max() <= 1e-12 * np.abs(ref).max()
    for g in got:                                                                        # every rank holds the reduced summary
        assert np.array_equal(g['cnt'], cnt0)
        ok = cnt0 > 0
        assert np.allclose(g['pk'][ok], pk0[ok], rtol=1e-10) and np.allclose(g['kc'][ok], kc0[ok], rtol=1e-12)
        assert np.isclose(g['sums'][0], m0.sum(), rtol=1e-12) and np.isclose(g['sums'][1], ref.sum(), rtol=1e-12)


def test_slab_bounds_and_apron_limits():
    assert GS.slab_bounds(64, 4, 3) == (48, 16)
    with pytest.raises(ValueError, match='multiple'):
        GS.slab_bounds(30, 4, 0)
