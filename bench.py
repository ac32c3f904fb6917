#!/usr/bin/env python
"""
bench.py -- BaryonifyShell hot path on MI355X: halos/s for a 1e6-halo synthetic catalog into an
NSIDE=1024 HEALPix shell (BASELINE.json configs[1]), inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]     # N > 1 without a launcher: the N ranks are started as a CHILD
                                                            # `python -m torch.distributed.run` (this process never touches HIP)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    ... bench.py --gpus N                             # STRONG scaling (north_star): ONE 1e6-halo catalog split over the N GPUs;
                                                      # the line carries the weak number (1e6 halos per GPU) as `value_weak`
    ... bench.py --gpus N --scaling weak              # weak scaling only
    ... bench.py --gpus 8 --config 4                  # BASELINE config 4: 1e7 halos, NSIDE 2048, over the GPUs given

One step = one full pass of the hot path: [N>1, default --exchange spatial: every halo's catalog row is routed (RCCL
all_to_all) to the ranks whose ring bands its disc touches ->] K0 halo_prep (per-halo scalars, halo -> tile binning) -> K1
tile_scatter2 (pix_offsets; tile-owned LDS accumulation, no global atomics) -> [N>1: all_reduce(MAX) of the largest
|offset|, exchange of that many apron rings ->] K2 tile_regrid3 (gathering regrid: every output pixel stored once) -> the
two sums of the mass-conservation check [-> N>1: disjoint slices to rank 0].  --exchange slices / reduce keep the halo
shards where they are and exchange pix_offsets instead.  N>1 is STRONG scaling by default (ONE 1e6-halo catalog, shuffled as
Parallelize.py:255 does, rank r holds the r-th slice); value = all halos / max-over-ranks time.  `value_weak` = the same step
with 1e6 halos PER GPU on the same shell.
--mode paint (BASELINE config 3 with --nside 2048): K0 -> K3, the pair phase in fp32 accumulated in fp64 into the fp64 map
(--acc-f64: fp64 throughout; `value_acc_f64` carries that number on the default line).  --mode grid3d / snapshot: config 5.

Prints ONE JSON line (rank 0).  `value` / `ms_per_step` come from a timed region of exactly --steps steps between
barrier + synchronize fences WITHOUT per-kernel events; the same K steps are then repeated with HIP events around every
kernel on the launch stream (bfgx_plan_timing_*) for `kernel_ms` and `roofline` (dominant kernel by measured time;
`ms_per_step_with_kernel_events` shows what the events cost).  At N = 1 the line also carries `value_acc_f64` (the same
step in fp64 throughout), `end_to_end` (the drop-in BaryonifyShell.process() from numpy arrays, PCIe-inclusive) and
`cpu_baseline` (the CPU oracle -- oracle/, the checker, never the product -- on the box's host cores).
Diagnostics (stderr, never in `value`): BFGX_BENCH_STAGE_TIMES=1 adds one untimed, stage-by-stage synchronised step of the N > 1 path and
prints its stage times; BFGX_BENCH_ENQUEUE_TIME=1 prints how much of a step is host-side enqueue time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# precision modes of the displacement path (include/bfgx.h BFGX_ACC_*)
PRECISION_NAMES = {0: "f32", 1: "f64", 3: "parity"}
DTYPES = {0: "f64 ring-row geometry + f64 LDS accumulation / map; f32 pair math, f32 pix_offsets, f32 regrid geometry",
          1: "f64 throughout (fp64 pair math, fp64 pix_offsets, fp64 regrid): the reference's own arithmetic",
          3: "f64 throughout in the parity-grade mode: f64 ring-row geometry, f64 pair math with 1e-11 elementary functions (fp32 seeds + one Newton step), "
             "f64 LDS accumulation, pix_offsets stored as two f32 arrays (hi + lo), f64 regrid geometry, f64 map"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=600, help='timed steps (default 600: a timed region of ~0.5 s at config 2)')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='strong',
                    help="N > 1: 'strong' (default) = ONE --halos catalog split over the GPUs (north_star's metric; the weak number rides "
                         "along as `value_weak`), 'weak' = --halos per GPU")
    ap.add_argument('--config', type=int, default=0, choices=[0, 4],
                    help='4 = BASELINE config 4: 1e7 halos, NSIDE 2048, BaryonifyShell, strong scaling over the GPUs given (8 in the config)')
    ap.add_argument('--catalog-order', choices=('shuffled', 'sky', 'mass'), default='shuffled',
                    help="order of the halos in the catalog (strong scaling / N = 1): 'shuffled' as Parallelize.py:255, 'mass' = heaviest first, 'sky' = patch by patch")
    ap.add_argument('--no-extras', action='store_true', help='skip value_acc_f64 and end_to_end (N = 1 only)')
    ap.add_argument('--halos', type=int, default=1_000_000, help='halos per GPU (default: BASELINE config 2)')
    ap.add_argument('--nside', type=int, default=1024)
    ap.add_argument('--eps', type=float, default=10.0)
    ap.add_argument('--acc-f64', action='store_true', help='fp64 throughout (= --precision f64; paint: fp64 pair math)')
    ap.add_argument('--precision', choices=['auto', 'f32', 'f64', 'parity'], default='auto',
                    help="baryonify: 'auto' (default) = what the drop-in runners do: the plan picks fp32 pair math or the parity-grade mode from its "
                         "table (include/bfgx.h BFGX_ACC_*: within SURVEY 8(d)'s 1e-6 mean(map) of the fp64 reference either way); 'f32' / 'f64' / "
                         "'parity' force one")
    ap.add_argument('--algo', type=int, default=1, help='1 = LDS tiles (default), 0 = per-halo global atomics')
    ap.add_argument('--mode', choices=['baryonify', 'paint', 'grid3d', 'snapshot'], default='baryonify',
                    help="'paint' = PaintProfilesShell (BASELINE config 3 with --nside 2048); 'grid3d' = BASELINE config 5 "
                         "(particle deposit + BaryonifyGrid on an --ngrid^3 periodic grid + FFT P(k)); 'snapshot' = the flow of the "
                         "reference's notebook 10 (BaryonifySnapshot of the particles + deposit + FFT P(k)); none of these is the headline metric")
    ap.add_argument('--ngrid', type=int, default=512, help='grid3d: cells per side (power of two)')
    ap.add_argument('--particles', type=int, default=0, help='grid3d: particles in the snapshot (default ngrid^3 / 2)')
    ap.add_argument('--grid-halos', type=int, default=100_000, help='grid3d: halos per GPU')
    ap.add_argument('--separate-deposit', action='store_true',
                    help='grid3d, one GPU: deposit and BaryonifyGrid as two calls (the deposit then does not leave map_out / the sum behind and '
                         'BaryonifyGrid copies and sums the map itself); default: the fused bfgx_grid_deposit_baryonify_device.  snapshot: '
                         'displacement and deposit as two calls (the moved coordinates are stored and read again); default: the fused '
                         'bfgx_snapshot_displace_deposit_device')
    ap.add_argument('--sorted-particles', action='store_true',
                    help='grid3d / snapshot: order the synthetic particles by coarse cell (as snapshot files stored along a space-filling '
                         'curve are) instead of uniformly random order')
    ap.add_argument('--table', choices=['closed-form', 's19'], default='s19',
                    help="'s19' (default since round 5): SURVEY 8(d) table (ii), the BENCHMARK table -- built by the GPU table builders (K4-K6) from the "
                         "Schneider19 one-halo profiles with the reference's default_config parameters; 'closed-form': table (i), the plumbing table "
                         "(displacements < 0.2 pixels; the headline of rounds 1-4, `value_closed_form` on the default line); baryonify mode only")
    ap.add_argument('--exchange', choices=['spatial', 'slices', 'reduce'], default='spatial',
                    help="N > 1: 'spatial' (default) = halos routed to the owners of the ring bands their discs touch, no accumulator "
                         "crosses a link; 'slices' = halo shards + all_to_all reduce-scatter of pix_offsets by pixel slices + banded regrid; "
                         "'reduce' = halo shards + one reduce(sum) of the whole accumulator to rank 0")
    ap.add_argument('--no-check', action='store_true',
                    help="N > 1 (strong scaling): skip the comparison of the assembled map with ONE single-GPU pass over the whole catalog "
                         "(outside the timed region; the line's `check` object, non-zero exit above 2e-6 of the map's scale)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true', help='do not bracket the kernels with HIP events in the timed region (no roofline object)')
    ap.add_argument('--cpu-sample', type=int, default=100_000, help='halos in the CPU-oracle sample (single-thread baseline)')
    ap.add_argument('--cpu-threads', type=int, default=0, help='threads of the CPU baseline (0 = all available cores, at most 16)')
    return ap.parse_args()


def cpu_baseline(args, cat, hmap, axes, table):
    """Oracle (C restatement of the reference loop) on the host cores of the GPU box.  With --cpu-threads T > 1 (default:
    all cores the box gives us, at most 16) the FULL catalog is split into T halo slices (private pix_offsets buffers,
    summed) and the regrid into T pixel ranges; with T = 1 the halo loop runs on a bounded sample and is scaled."""
    from oracle import oracle as O
    from baryonification_amd import synthetic as syn
    bg = O.Background.from_dict(syn.COSMO)
    tab = O.Table(axes, table, False, args.eps)
    T = args.cpu_threads if args.cpu_threads > 0 else max(1, min(16, len(os.sched_getaffinity(0))))
    N = cat['M'].size
    if T > 1:
        _, t_loop, t_rg, pairs = O.baryonify_shell_threads(args.nside, hmap, cat, tab, args.eps, bg, T)
        return {"value": N / (t_loop + t_rg), "unit": "halos/s", "cores": T, "kind": "port",
                "sample": "full workload: %d halos (%d pairs) in %d halo slices (%.2f s incl. summing the private pix_offsets) + "
                          "full NSIDE=%d regrid in %d pixel ranges (%.2f s); scalar C restatement per thread" % (
                              N, pairs, T, t_loop, args.nside, T, t_rg),
                "loop_halos_per_s": N / t_loop, "regrid_pix_per_s": hmap.size / t_rg}
    n = min(args.cpu_sample, N)
    sub = {k: v[:n] for k, v in cat.items()}
    t0 = time.time()
    off, counts = O.baryonify_offsets(args.nside, sub, tab, args.eps, bg, return_counts=True)
    t1 = time.time()
    O.regrid(args.nside, hmap, off)
    t2 = time.time()
    t_loop_full = (t1 - t0) * N / n
    t_full = t_loop_full + (t2 - t1)
    return {"value": N / t_full, "unit": "halos/s", "cores": 1, "kind": "port",
            "sample": "halo loop on first %d of %d halos (%.2f s, %d pairs) + full NSIDE=%d regrid (%.2f s); "
                      "loop time scaled to the full catalog" % (n, N, t1 - t0, int(counts.sum()), args.nside, t2 - t1),
            "loop_halos_per_s": n / (t1 - t0), "regrid_pix_per_s": hmap.size / (t2 - t1)}


def cpu_baseline_grid(args, cat, bins, zr, axes, table, eps, hmap_sample_pixels=1 << 22):
    """Oracle (C restatement, 1 thread): the full halo loop + the regrid of a bounded pixel sample, scaled."""
    from oracle import grid as G
    from oracle import oracle as O
    from baryonification_amd import synthetic as syn
    N = args.ngrid
    tab = O.Table(axes, table, False, eps)
    t0 = time.time()
    off, pairs = G.baryonify_grid_offsets((N, N, N), bins, cat, zr, tab, eps, G.grid_background(syn.COSMO), return_pairs=True)
    t1 = time.time()
    ns = min(hmap_sample_pixels, N ** 3)
    rng = np.random.default_rng(1)
    pos = off[:ns] + rng.uniform(0, N, (ns, 3))           # same arithmetic per pixel as the real regrid
    G.regrid_pixels(np.zeros((N, N, N)), pos, np.ones(ns))
    t2 = time.time()
    t_full = (t1 - t0) + (t2 - t1) * N ** 3 / ns
    return {"value": N ** 3 / t_full, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "full halo loop (%d halos, %d cutout pixels, %.2f s) + regrid of %d of %d pixels (%.2f s) scaled to the "
                      "grid; particle deposit and FFT not included" % (cat['M'].size, pairs, t1 - t0, ns, N ** 3, t2 - t1)}


def main_grid_slabs(args):
    """BASELINE config 5 as stated (N GPUs): the ONE 512^3 problem slab-decomposed over the ranks (utils/GridSlabs.py): particles
    routed to the owner of their plane, halo loop clipped to the slab, regrid with apron exchange, P(k) through one all_to_all
    transpose.  Strong scaling: the grid, the particle count and the halo catalog are those of the single-GPU line."""
    import torch
    import torch.distributed as dist
    from baryonification_amd import _lib, engine, synthetic as syn
    from baryonification_amd.utils import GridSlabs as GS

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0)) % max(1, torch.cuda.device_count())
    assert world == args.gpus, "launch with --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py needs a GPU (libbfgx has no CPU fallback)"
    backend = os.environ.get('BFGX_DIST_BACKEND', 'nccl')        # gloo: functional rehearsal with all ranks on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
    N, nh, eps, zr, Nk = args.ngrid, args.grid_halos, 5.0, 0.0, 180
    L = 205.0 / syn.COSMO['h']
    npart = args.particles or N ** 3 // 2
    bins = (np.arange(N) + 0.5) * (L / N)
    rng = np.random.default_rng(syn.SEED_CATALOG)                 # the SAME halo catalog on every rank (replicated)
    M = syn.make_catalog(nh, seed=syn.SEED_CATALOG)['M'].astype(np.float32).astype(np.float64)
    pos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
    z, Mt, r = np.array([0.0, 0.01]), np.geomspace(0.99e12, 1.01e15, 10), np.geomspace(1e-3, 3e2, 500)
    model, keep = engine.model_from_tables([np.log(1 + z), np.log(Mt), np.log(r)], syn.displacement_table(z, Mt, r), dict(syn.COSMO, w0=-1.0), eps, eps)
    t = {k: torch.from_numpy(v.copy()).to(dev) for k, v in (('M', M), ('x', pos[:, 0]), ('y', pos[:, 1]), ('z', pos[:, 2]))}
    lnM = torch.from_numpy(np.log(M.astype(np.float32)).astype(np.float64)).to(dev)
    cat_dev = _lib.make_grid_catalog_dev(nh, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr(), lnM.data_ptr())
    mine = npart // world + (1 if rank < npart % world else 0)   # this rank's share of the snapshot (as read from its file chunks)
    torch.manual_seed(syn.SEED_MAP + rank)
    rows = torch.rand((3, mine), dtype=torch.float64, device=dev) * L
    be = GS.HipBackend(model, keep, bins, zr, nh, device=local_rank)

    def fence():
        dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = GS.slab_step(be, rows, cat_dev, N, L, Nk)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = GS.slab_step(be, rows, cat_dev, N, L, Nk)
    fence()
    t1 = time.perf_counter()
    stages = {}
    GS.slab_step(be, rows, cat_dev, N, L, Nk, timers=stages)     # one more pass with synchronising stage timers (not in `value`)
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    new, kc, pk, cnt, sums = out
    if rank == 0:
        alg = N ** 3 * (3 * 8 + 8 + 8 * 8 + 8) + npart * (3 * 8 + 8) + N ** 3 * 8 + N ** 3 * 8 + 5 * N * N * (N // 2 + 1) * 16
        ach = alg / (elapsed / args.steps) / 1e9
        print(json.dumps({
            "metric": "grid cells/sec for particle deposit + BaryonifyGrid + FFT P(k) on a %d^3 periodic grid" % N,
            "value": N ** 3 / elapsed * args.steps, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 5: %d^3 grid of a %.1f Mpc box at z=0, %d uniform particles, %d halos (replicated catalog), "
                                   "epsilon_max=%g, closed-form displacement table, P(k) in %d linear bins" % (N, L, npart, nh, eps, Nk),
                       "ngrid": N, "particles": npart, "halos": nh,
                       "parallelism": "slabs of %d planes x%d: particle routing (all_to_all), halo loop clipped per slab, regrid + apron exchange, "
                                      "FFT with one all_to_all transpose, all_reduce of the bin sums (%s)" % (N // world, world, backend)},
            "mass_conserved": bool(np.isclose(sums[1], sums[0])), "pk_finite_bins": int(np.isfinite(pk).sum()),
            "stage_ms_rank0": {k: round(v, 3) for k, v in stages.items()},
            "roofline": {"kernel": "whole step (deposit + halo loop + regrid + FFT)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS * world,
                         "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * world), "traffic": None, "algorithmic_bytes_per_launch": alg}}), flush=True)
    be.close()
    dist.destroy_process_group()


def main_grid(args):
    """BASELINE config 5 on one GPU: ParticleSnapshot.make_map -> BaryonifyGrid -> P(k) (N > 1: main_grid_slabs)."""
    import torch
    import torch.distributed as dist
    from baryonification_amd import _lib, engine, synthetic as syn

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if args.mode == 'grid3d' and (world > 1 or os.environ.get('BFGX_FORCE_SLABS') == '1'):
        return main_grid_slabs(args)
    assert world == args.gpus, "launch with --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py needs a GPU (libbfgx has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    N, nh, eps, zr, Nk = args.ngrid, args.grid_halos, 5.0, 0.0, 180
    L = 205.0 / syn.COSMO['h']                                   # TNG300 box of the reference's notebook 10
    npart = args.particles or N ** 3 // 2
    bins = (np.arange(N) + 0.5) * (L / N)
    edges = np.linspace(0, L, N + 1)
    rng = np.random.default_rng(syn.SEED_CATALOG + rank)
    M = syn.make_catalog(nh, seed=syn.SEED_CATALOG + rank)['M'].astype(np.float32).astype(np.float64)   # HaloNDCatalog is float32
    pos = rng.uniform(0, L, (nh, 3)).astype(np.float32).astype(np.float64)
    cat = {'M': M, 'x': pos[:, 0].copy(), 'y': pos[:, 1].copy(), 'z': pos[:, 2].copy()}
    z, Mt, r = np.array([0.0, 0.01]), np.geomspace(0.99e12, 1.01e15, 10), np.geomspace(1e-3, 3e2, 500)
    table = syn.displacement_table(z, Mt, r)
    axes = [np.log(1 + z), np.log(Mt), np.log(r)]
    model, keep = engine.model_from_tables(axes, table, dict(syn.COSMO, w0=-1.0), eps, eps)
    stream = torch.cuda.current_stream().cuda_stream
    plan = engine.GridPlan(model, keep, bins, 3, zr, nh, device=local_rank, stream=stream)
    t = {k: torch.from_numpy(v).to(dev) for k, v in cat.items()}
    lnM = torch.from_numpy(np.log(M.astype(np.float32)).astype(np.float64)).to(dev)
    cat_dev = _lib.make_grid_catalog_dev(nh, t['M'].data_ptr(), t['x'].data_ptr(), t['y'].data_ptr(), t['z'].data_ptr(), lnM.data_ptr())
    torch.manual_seed(syn.SEED_MAP)
    part = torch.rand((3, npart), dtype=torch.float64, device=dev) * L       # synthetic snapshot, unit particle mass
    if args.sorted_particles:            # setup, untimed: raster order of 64^3 coarse cells
        key = ((part[0] * (64 / L)).long().clamp_(0, 63) * 64 + (part[1] * (64 / L)).long().clamp_(0, 63)) * 64 + (part[2] * (64 / L)).long().clamp_(0, 63)
        part = part[:, torch.argsort(key)].contiguous()
        del key
    d_edges = torch.from_numpy(edges).to(dev)
    d_map = torch.empty(N ** 3, dtype=torch.float64, device=dev)
    d_off = torch.empty(N ** 3 * 3, dtype=torch.float64, device=dev)
    d_out = torch.empty(N ** 3, dtype=torch.float64, device=dev)
    d_work = torch.empty(engine.power_spectrum_work_doubles(N), dtype=torch.float64, device=dev)
    d_sums = torch.zeros(2, dtype=torch.float64, device=dev)
    d_pk, d_ks = torch.zeros(Nk, dtype=torch.float64, device=dev), torch.zeros(Nk, dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(Nk, dtype=torch.int64, device=dev)
    snapshot = args.mode == 'snapshot'
    if snapshot:
        assert world == 1, "--mode snapshot is single-GPU (particles are not sharded)"
        part_out = torch.empty_like(part)
        splan = engine.SnapshotPlan(model, keep, 3, L, zr, nh, device=local_rank, stream=stream)
    fused = world == 1 and not snapshot and not args.separate_deposit
    snap_fused = snapshot and not args.separate_deposit
    ev = {k: [] for k in ((('displace+deposit', 'pk') if snap_fused else ('displace', 'deposit', 'pk')) if snapshot else
                          (('deposit+baryonify', 'pk') if fused else ('deposit', 'pk')))}
    pairs = [0]

    def timed(kind, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ev[kind].append((a, b))

    def step_snapshot():
        if snap_fused:
            # process() + make_map(N) in one call: the displacement kernel writes the deposit's sort keys, the moved coordinates are never stored
            def both():
                pairs[0] = splan.displace_deposit(cat_dev, npart, (part[0].data_ptr(), part[1].data_ptr(), part[2].data_ptr()), 0, N,
                                                  d_edges.data_ptr(), d_out.data_ptr())
            timed('displace+deposit', both)
            timed('pk', lambda: engine.power_spectrum_device(d_out.data_ptr(), N, L, Nk, d_work.data_ptr(), d_pk.data_ptr(),
                                                             d_ks.data_ptr(), d_cnt.data_ptr(), local_rank, stream))
            return

        def displace():
            pairs[0] = splan.displace(cat_dev, npart, (part[0].data_ptr(), part[1].data_ptr(), part[2].data_ptr()),
                                      (part_out[0].data_ptr(), part_out[1].data_ptr(), part_out[2].data_ptr()))
        timed('displace', displace)
        timed('deposit', lambda: engine.deposit_particles_device(part_out[0].data_ptr(), part_out[1].data_ptr(), part_out[2].data_ptr(), 0, npart, N,
                                                                 d_edges.data_ptr(), d_out.data_ptr(), 3, local_rank, stream))
        timed('pk', lambda: engine.power_spectrum_device(d_out.data_ptr(), N, L, Nk, d_work.data_ptr(), d_pk.data_ptr(),
                                                         d_ks.data_ptr(), d_cnt.data_ptr(), local_rank, stream))

    def step():
        if snapshot:
            return step_snapshot()
        d_sums.zero_()
        if world == 1 and fused:
            # make_map + BaryonifyGrid in one call: the deposit's last kernel also stores the start value of map_out and the map's sum
            def both():
                pairs[0] = plan.deposit_baryonify(cat_dev, npart, part[0].data_ptr(), part[1].data_ptr(), part[2].data_ptr(), 0, d_edges.data_ptr(),
                                                  d_map.data_ptr(), d_out.data_ptr(), d_sums.data_ptr())
            timed('deposit+baryonify', both)
            timed('pk', lambda: engine.power_spectrum_device(d_out.data_ptr(), N, L, Nk, d_work.data_ptr(), d_pk.data_ptr(),
                                                             d_ks.data_ptr(), d_cnt.data_ptr(), local_rank, stream))
            return
        if rank == 0:
            timed('deposit', lambda: engine.deposit_particles_device(part[0].data_ptr(), part[1].data_ptr(), part[2].data_ptr(), 0, npart, N,
                                                                     d_edges.data_ptr(), d_map.data_ptr(), 3, local_rank, stream))
        if world == 1:
            # halo loop + regrid as one cell-owned pass: no pix_offsets array
            pairs[0] = plan.baryonify(cat_dev, d_map.data_ptr(), d_out.data_ptr(), d_sums.data_ptr())
        else:
            pairs[0] = plan.offsets(cat_dev, d_off.data_ptr())
            dist.reduce(d_off, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                plan.regrid(d_map.data_ptr(), d_off.data_ptr(), d_out.data_ptr(), d_sums.data_ptr())
        if rank == 0:
            timed('pk', lambda: engine.power_spectrum_device(d_out.data_ptr(), N, L, Nk, d_work.data_ptr(), d_pk.data_ptr(),
                                                             d_ks.data_ptr(), d_cnt.data_ptr(), local_rank, stream))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    for k in ev:
        ev[k].clear()
    plan.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kt = plan.timing_read()
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    ev_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in lst])) for k, lst in ev.items() if lst}      # (the timed region's events)
    sorted_companion = None
    if snapshot and not args.sorted_particles and not args.no_extras:
        # the same flow on the same particles in the order snapshot files stored along a space-filling curve have (raster order of 64^3
        # coarse cells): the displacement's look-ups then fall into lines the neighbouring lanes touch too.  BaryonifySnapshot returns the
        # particles in the caller's order, so this costs the caller nothing but is not something the library may do to unsorted input.
        key = ((part[0] * (64 / L)).long().clamp_(0, 63) * 64 + (part[1] * (64 / L)).long().clamp_(0, 63)) * 64 + (part[2] * (64 / L)).long().clamp_(0, 63)
        part = part[:, torch.argsort(key)].contiguous()
        del key
        for k in ev:
            ev[k].clear()
        for _ in range(3):
            step()
        fence()
        for k in ev:
            ev[k].clear()
        ns = max(5, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(ns):
            step()
        fence()
        els = time.perf_counter() - t0
        sorted_companion = {"value": npart / els * ns, "unit": "particles/s", "ms_per_step": els / ns * 1e3, "steps": ns,
                            "particle_order": "coarse-cell raster (64^3)",
                            "kernel_ms": {k: float(np.mean([a.elapsed_time(b) for a, b in lst])) for k, lst in ev.items() if lst}}
    if rank == 0:
        kernels = {k: ms / n for k, (ms, n) in kt.items() if n}
        kernels.update(ev_ms)
        if fused:        # the deposit's share of the fused call: what the plan's own kernel timers (prep, lists, gather, sums) do not cover
            kernels['deposit'] = kernels['deposit+baryonify'] - sum(kernels.get(k, 0.0) for k in ('prep', 'bin', 'offsets', 'sum', 'regrid'))
        sums = d_sums.cpu().numpy()
        pk = (d_pk / d_cnt).cpu().numpy()
        # algorithmic bytes per launch.  world == 1: 'regrid' is the copy map_out = map_in (every cell deposits into itself unless it is
        # moved), 'offsets' the cell-owned BaryonifyGrid pass (the halo lists; per moved cell its value and 9 read-modify-writes of
        # map_out; no pix_offsets array); 'pk': the real map read, the padded half spectrum written by the r2c pass,
        # read + written by the middle pass and read by the last one (binned from LDS, nothing written back)
        spec = N * N * engine.fft_pitch(N) * 16
        alg = {'regrid': (0 if fused else N ** 3 * 16) if world == 1 else N ** 3 * (3 * 8 + 8 + 8 * 8 + 8),
               'offsets': (pairs[0] * (8 + 9 * 16) + nh * 8 * (4 + 344)) if world == 1 else pairs[0] * 3 * 8 + nh * 32,
               'deposit': npart * (3 * 8 + 8) + N ** 3 * (16 if fused else 8), 'pk': N ** 3 * 8 + 4 * spec, 'displace': npart * 48 + nh * 32,
               'displace+deposit': npart * 24 + nh * 32 + N ** 3 * 8}
        dom = max(alg, key=lambda k: kernels.get(k) or 0.0)
        ach = alg[dom] / (kernels[dom] * 1e-3) / 1e9
        g_traffic, _, g_src = committed_traffic({"ngrid": N, "particles": npart, "particle_order": "coarse-cell raster" if args.sorted_particles else "random",
                                                 "halos_per_gpu": nh} if world == 1 else {"ngrid": -1}, dom,
                                                metric_has="BaryonifySnapshot" if snapshot else "BaryonifyGrid")
        if snapshot:
            sums = np.array([float(npart), float(d_out.sum().item())])       # every particle lands in the box
        out = {"metric": ("particles/sec for BaryonifySnapshot + deposit + FFT P(k) on a %d^3 grid" % N) if snapshot else
                         ("grid cells/sec for particle deposit + BaryonifyGrid + FFT P(k) on a %d^3 periodic grid" % N),
               "value": (npart if snapshot else N ** 3) / elapsed * args.steps, "unit": "particles/s" if snapshot else "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE config 5 (single-node form): %d^3 grid of a %.1f Mpc box at z=0, %d uniform particles, %d halos "
                                      "per GPU (SURVEY 8d mass function, float32 catalog), epsilon_max=%g, closed-form displacement table, "
                                      "P(k) in %d linear bins" % (N, L, npart, nh, eps, Nk),
                          "particle_order": "coarse-cell raster" if args.sorted_particles else "random", "ngrid": N, "particles": npart, "halos_per_gpu": nh, "contributing_pairs_per_gpu": pairs[0],
                          "parallelism": "halo shards x%d + RCCL reduce(pix_offsets) -> rank 0 regrid + P(k)" % world if world > 1 else "single GPU"},
               "halos_per_s": nh * world / elapsed * args.steps, "kernel_ms": kernels,
               "fused_deposit": bool(fused or snap_fused),
               "kernel_ms_note": ("single GPU, fused call (bfgx_grid_deposit_baryonify_device): the deposit's last kernel stores map_in, the start value of "
                                  "map_out and the map's sum together (no grid_copy_sum_kernel); 'deposit' = the fused call minus the plan-timed kernels, "
                                  "'offsets' = grid_gather_regrid_kernel (halo loop AND the regrid of the moved cells in one cell-owned pass), "
                                  "'bin' = the per-block halo lists (count, scan, fill)") if fused else
                                 ("single GPU: 'regrid' = grid_copy_sum_kernel (map_out = map_in: a cell that is not moved deposits into itself), "
                                  "'offsets' = grid_gather_regrid_kernel (halo loop AND the regrid of the moved cells in one cell-owned pass), "
                                  "'bin' = the per-block halo lists (count, scan, fill)") if world == 1 and not snapshot else None,
               "mass_conserved": bool(np.isclose(sums[1], sums[0])), "pk_finite_bins": int(np.isfinite(pk).sum()),
               "roofline": {"kernel": {"regrid": "grid_copy_sum_kernel" if world == 1 else "grid_regrid_kernel<3>",
                                       "offsets": "grid_gather_regrid_kernel<3> (halo loop + regrid, cell-owned)" if world == 1 else "grid_scatter_kernel<3,OFFSETS>",
                                       "deposit": "deposit_keys + 2 x deposit_split + deposit_tiles (bfgx_deposit.hpp)",
                                       "pk": "fft_r2c_lines + fft_c2c_strided + fft_c2c_strided<bins>", "displace": "snap_displace_kernel<3>",
                                       "displace+deposit": "snap_displace_kernel<3, keys> + 2 x deposit_split + deposit_tiles"}[dom],
                            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": g_traffic,
                            "traffic_source": g_src, "algorithmic_bytes_per_launch": alg[dom]}}
        if sorted_companion is not None:
            out["value_sorted_input"] = sorted_companion
        if world == 1 and not args.no_cpu_baseline and not snapshot:
            out["cpu_baseline"] = cpu_baseline_grid(args, cat, bins, zr, axes, table, eps)
        print(json.dumps(out), flush=True)
    plan.close()
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher (RANK unset): start the N ranks as a CHILD `python -m torch.distributed.run`
    (one process per GPU over RCCL), relay its output and return code.  This parent has touched neither torch nor HIP, and it does
    not exec: a process that has initialised the GPU must never be replaced.  One call drives all workers, as
    SplitJoinParallel(runner, njobs).process() does (Parallelize.py:191-320)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL across processes needs it on this image
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dist_context(args):
    """(rank, world, local_rank, device, backend) of this rank; checks the launch and the GPU.  BFGX_DIST_BACKEND=gloo: functional
    rehearsal of the N > 1 path on a one-GPU box (all ranks share device 0, collectives staged through the host; timings of such a
    run mean nothing).  BFGX_BENCH_STOP_AFTER_INIT=1 (launcher test, no GPU): rendezvous over gloo, rank 0 prints a stub line, exit."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node == --gpus, or without a launcher)" % (world, args.gpus))
    backend = os.environ.get('BFGX_DIST_BACKEND', 'nccl')
    if os.environ.get('BFGX_BENCH_STOP_AFTER_INIT') == '1':
        dist.init_process_group('gloo')
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launcher_ok": True, "n_gpus": int(t.item()), "scaling": args.scaling}), flush=True)
        dist.destroy_process_group()
        sys.exit(0)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libbfgx has no CPU fallback)")
    local_rank = int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    return rank, world, local_rank, dev, backend


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args))
    if args.mode in ('grid3d', 'snapshot'):
        return main_grid(args)
    import torch.distributed as dist
    rank, world, local_rank, dev, backend = ctx = dist_context(args)
    if world > 1 or os.environ.get('BFGX_FORCE_EXCHANGE') == '1':
        if 'RANK' not in os.environ:          # BFGX_FORCE_EXCHANGE=1 from a plain `python bench.py`: a one-rank group
            import socket
            with socket.socket() as sck:
                sck.bind(('127.0.0.1', 0))
                port = sck.getsockname()[1]
            os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
    scaling = 'strong' if args.config == 4 else args.scaling
    out = shell_line(args, ctx, scaling, brief=False)
    if world > 1 and scaling == 'strong' and not args.no_extras and args.config != 4:
        # the weak-scaling number of the same step (--halos per GPU on the same shell) rides along; `value` stays north_star's metric
        w = shell_line(args, ctx, 'weak', brief=True)
        if rank == 0:
            out["value_weak"] = {k: w[k] for k in ("value", "unit", "ms_per_step", "steps", "scaling", "mass_conserved")}
            out["value_weak"]["workload"] = w["config"]["workload"]
    rc = 0
    if rank == 0:
        print(json.dumps(out), flush=True)
        if out.get("check") is not None and not out["check"]["ok"]:
            print("bench: the assembled map differs from the single-GPU pass by %.3e (> %.3e): exit 4" % (
                out["check"]["max_abs_diff_vs_single_gpu"], out["check"]["tolerance"]), file=sys.stderr, flush=True)
            rc = 4
    if dist.is_initialized():
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


def shell_line(args, ctx, scaling, brief):
    """One bench line of the HEALPix-shell path (rank 0 returns the dict, the others None).  `brief`: the timed region only (no
    per-kernel events, fp64 / end-to-end extras or CPU baseline): the weak-scaling companion of a strong line."""
    import torch
    import torch.distributed as dist
    from baryonification_amd import _lib, engine, synthetic as syn
    rank, world, local_rank, dev, backend = ctx

    paint = args.mode == 'paint'
    if args.config == 4:            # BASELINE config 4: 1e7 halos, BaryonifyShell, NSIDE 2048, halo-sharded over 8 GPUs
        args.nside = 2048
        args.halos = 10_000_000 if args.halos == 1_000_000 else args.halos
    nside, npix = args.nside, 12 * args.nside ** 2
    strong = scaling == 'strong'
    total_halos = args.halos if strong else args.halos * world
    if strong:
        # ONE catalog (BASELINE seeds), shuffled as Parallelize.py:255 does, rank r takes the r-th ceil(N/world) slice
        from baryonification_amd.utils.Parallelize import shard_slices, shuffled_order
        full = syn.make_catalog(total_halos)
        mine = shuffled_order(total_halos, 42)[shard_slices(total_halos, world)[rank]]
        if args.catalog_order == 'mass':                 # heaviest halos first, as halo finders write them
            mine = mine[np.argsort(-full['M'][mine], kind='stable')]
        if args.catalog_order == 'sky':
            # a catalog written patch by patch (what a lightcone pipeline usually leaves): (band of 32 rings, azimuth) order.  The default
            # is the reference's own order for parallel runs -- shuffled with seed 42 (Parallelize.py:255)
            zc = np.sin(np.radians(full['dec'][mine]))
            nsd = args.nside
            ring = np.where(np.abs(zc) <= 2 / 3, nsd * (2 - 1.5 * zc), np.where(zc > 0, nsd * np.sqrt(3 * (1 - np.abs(zc))), 4 * nsd - nsd * np.sqrt(3 * (1 - np.abs(zc)))))
            key = (ring.astype(np.int64) // 32) * 4096 + (full['ra'][mine] / 360.0 * 4096).astype(np.int64)
            mine = mine[np.argsort(key, kind='stable')]
        cat = {k: np.ascontiguousarray(v[mine]) for k, v in full.items()}
        z, M, r = syn.table_grid(full)                   # README.md:78-80: edges = catalog min/max
        del full
    else:
        cat = syn.make_catalog(args.halos, seed=syn.SEED_CATALOG + rank)
        if world == 1:
            z, M, r = syn.table_grid(cat)
        else:                                            # shards differ: analytic support of the catalog
            z, M, r = np.geomspace(0.2, 0.3, 10), np.geomspace(1e12, 1e15, 10), np.geomspace(1e-3, 3e2, 500)
    nh = cat['M'].size
    if args.table == 's19' and not paint:
        table = syn.s19_displacement_table(z, M, r)
    else:
        table = syn.paint_table(z, M, r) if paint else syn.displacement_table(z, M, r)
    axes = [np.log(1 + z), np.log(M), np.log(r)]
    hmap = np.zeros(npix) if paint else syn.make_map(nside)
    with np.errstate(divide='ignore'):       # the closed-form profile underflows to 0 at the largest radii: ln -> -inf
        values = np.log(table) if paint else table
    model, keep = engine.model_from_tables(axes, values, syn.COSMO, args.eps, args.eps, log_values=paint)

    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cat.items()}
    lnz, lnM = _lib.table_coords(cat['M'], cat['z'])      # numpy's np.log(1/a), np.log(M): exact table-edge classification
    t['lnz'], t['lnM'] = torch.from_numpy(lnz).to(dev), torch.from_numpy(lnM).to(dev)
    d_map = torch.from_numpy(hmap).to(dev)
    d_out = torch.zeros(npix, dtype=torch.float64, device=dev)
    d_sums = torch.zeros(2, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    plan = engine.ShellPlan(model, keep, nside, nh, device=local_rank, stream=stream)
    cat_dev = _lib.make_catalog_dev(nh, t['M'].data_ptr(), t['z'].data_ptr(), t['ra'].data_ptr(), t['dec'].data_ptr(),
                                    ln1pz_ptr=t['lnz'].data_ptr(), lnM_ptr=t['lnM'].data_ptr())
    plan.set_algo(args.algo)
    n_pairs = plan.count_pairs(cat_dev, fallback4=not paint)
    plan.status()
    # precision of the displacement path (include/bfgx.h BFGX_ACC_*): what the request resolves to on this plan (auto: from the table)
    acc_req = _lib.ACC_F64 if args.acc_f64 else {'auto': _lib.ACC_AUTO, 'f32': _lib.ACC_F32, 'f64': _lib.ACC_F64, 'parity': _lib.ACC_PARITY}[args.precision]
    if paint:
        acc_res, disp_px = (1 if args.acc_f64 else 2), None
    elif args.algo == 1:
        acc_res, disp_px = plan.precision(acc_req)
    else:
        acc_res, disp_px = (_lib.ACC_F32 if acc_req in (_lib.ACC_AUTO, _lib.ACC_F32) else _lib.ACC_F64), None

    # N > 1: slice exchange (utils/Parallelize.py): all_to_all reduce-scatter of the accumulator by pixel slices, one-ring
    # halo exchange, every rank regrids the OUTPUT pixels of its bands, disjoint slices travel to rank 0.
    # --exchange reduce keeps the single reduce(sum) to rank 0.
    # BFGX_FORCE_EXCHANGE=1: run the N > 1 exchange code with a single rank too (exercises the RCCL calls on a one-GPU box)
    force_x = os.environ.get('BFGX_FORCE_EXCHANGE') == '1'
    slices = (world > 1 or force_x) and args.exchange == 'slices' and args.algo == 1
    d_fin = torch.zeros(npix if ((world > 1 or force_x) and rank == 0) else 0, dtype=torch.float64, device=dev)
    d_foreign = torch.zeros(1, dtype=torch.int64, device=dev)          # far deposits that belong to another rank's slice
    route_far = [False]                # set (on every rank) when the untimed trial step finds such deposits
    spatial = (world > 1 or force_x) and args.exchange == 'spatial' and args.algo == 1
    if slices or spatial:
        from baryonification_amd.utils.Parallelize import (_hip_reach, band_partition, band_ring_bounds, gather_slices, gather_slices_start,
                                                           halo_exchange, route_halos, route_halos_fixed, route_step_fixed, sliced_reduce)
        first = plan.bands()
        cuts = band_partition(first, world)
        pb = first[cuts]
        needs = [plan.band_apron(int(cuts[j]), int(cuts[j + 1])) for j in range(world)]
        p0, p1 = int(pb[rank]), int(pb[rank + 1])
        d_slice = torch.zeros(p1 - p0, dtype=torch.float64, device=dev)
    if spatial:
        # spatial sharding: the halos this rank holds (its chunk of the catalog) are routed to the ranks whose ring bands their discs
        # can touch; every rank then computes ITS pixels only -- no accumulator crosses a link
        rb = band_ring_bounds(cuts, plan.tile_shape()[0], nside)
        # fixed-capacity routing: every (source, destination) pair owns a block of `blockcap` rows, the equal splits of ONE all_to_all, so
        # that no count is read back by the host inside a step (the catalog is shuffled, Parallelize.py:255: a destination receives about
        # nh / world halos from every source; 1.75 x that + 2048 leaves room for the halos of the two neighbouring bands too).  An overflow is detected on the device, checked after the
        # untimed trial step and answered with the variable-split routing (route_halos: one read-back per step)
        # The capacity comes from ONE untimed count pass (bfgx_route_count_device) with the widest routing a step may use (every halo one
        # band further: the bands next to a rank's own are computed locally when the reach fits, below): the largest (source,
        # destination) count over all ranks, rounded up -- round 3 guessed 1.75 x the mean and K0 ran over 1.9 padded rows per real one
        d_probe = torch.empty((nh, 2), dtype=torch.int32, device=dev)
        cnt = torch.zeros(world, dtype=torch.int32, device=dev)
        plan.set_route_margin(0 if paint else plan.tile_shape()[0])
        plan.disc_rings(cat_dev, d_probe.data_ptr())
        plan.route_count(nh, d_probe.data_ptr(), rb, cnt.data_ptr())
        plan.set_route_margin(0)
        cnt64 = cnt.to(torch.int64) if backend == 'nccl' else cnt.to(torch.int64).cpu()
        rows_mat = [torch.zeros_like(cnt64) for _ in range(world)]
        if dist.is_initialized():
            dist.all_gather(rows_mat, cnt64)
        else:
            rows_mat = [cnt64]
        rows_mat = torch.stack(rows_mat).cpu().numpy()                       # [source][destination], the same on every rank
        blockcap = (int(rows_mat.max()) + 255) // 256 * 256 + 256
        routing_rows = {"real_rank0": int(rows_mat[:, 0].sum()), "padded_per_rank": world * blockcap,
                        "real_max_rank": int(rows_mat.sum(axis=0).max()), "blockcap": blockcap,
                        "sized_by": "one untimed count pass (route margin of one band), max over (source, destination) + 256, in 256s"}
        del d_probe
        cap = max(world * blockcap, int(rows_mat.sum(axis=0).max()) + 4096)
        plan_sp = engine.ShellPlan(model, keep, nside, cap, device=local_rank, stream=stream)
        d_rings = torch.empty((nh, 2), dtype=torch.int32, device=dev)
        cols_local = [t[k] for k in ('M', 'z', 'ra', 'dec', 'lnz', 'lnM')]
        route_work = {}
        sp_state = {'fixed': True, 'reach_known': False, 'nd': None, 'cd': None}
        stage_t = {}
        # The slices of pass i travel to rank 0 WHILE pass i + 1 computes: the gather runs on a process group of its own (its own RCCL
        # communicator and stream) out of one of two slice buffers into one of two final maps; a buffer is written again only after the
        # gather that read it has been waited for, and fence() waits for what is still in flight, so the timed region ends with every map
        # assembled.  (BFGX_BENCH_OVERLAP_GATHER=0: the gather inside the step, on the step's own communicator, as in rounds 3 - 4.)
        # (a single forced rank has nothing to overlap with and pays ~12 us for the second communicator: off unless asked for with =1)
        ov_env = os.environ.get('BFGX_BENCH_OVERLAP_GATHER')
        ov = {'on': (ov_env == '1' or (ov_env != '0' and world > 1)) and not paint, 'k': 0, 'last': 0, 'pending': [None, None],
              'slice': [d_slice, torch.zeros_like(d_slice)], 'fin': [d_fin, torch.zeros_like(d_fin)], 'group': None}
        if ov['on'] and dist.is_initialized():
            ov['group'] = dist.new_group(backend=backend)

        def mark(name):
            """BFGX_BENCH_STAGE_TIMES=1: one extra, untimed step with a synchronisation after every stage (rank 0 prints the stage times)"""
            if sp_state.get('staging'):
                torch.cuda.synchronize()
                now = time.perf_counter()
                stage_t[name] = stage_t.get(name, 0.0) + (now - sp_state['t_last']) * 1e3
                sp_state['t_last'] = now

    def run_steps(acc):
        """returns a closure doing one full pass of the hot path in the given precision: painting: True = fp64 pair math, False = the mixed mode;
        displacement: BFGX_ACC_F32 / F64 / PARITY (24 bytes of pix_offsets per pixel unless fp32; the band-restricted entries of the N > 1
        paths keep ONE array of pix_offsets and serve the parity-grade mode with fp64 throughout)"""
        acc_f64 = bool(acc) if paint else (acc if (world == 1 and not slices and not spatial) else (1 if acc != _lib.ACC_F32 else 0))
        acc_dtype = torch.float64 if acc_f64 else torch.float32
        d_off = torch.zeros(npix * 3, dtype=acc_dtype, device=dev)
        x_recv = torch.empty(world * (p1 - p0) * (1 if paint else 3), dtype=torch.float64 if paint else acc_dtype, device=dev) if slices else None

        def step_paint():
            if args.algo == 0:
                d_out.zero_()
            plan.paint(cat_dev, d_out.data_ptr(), acc_f64=(1 if acc_f64 else 2))    # 2: f32 pair math, f64 accumulation and map
            if slices:
                mine = sliced_reduce(d_out, pb, 1, recv=x_recv)
                gather_slices(mine, pb, npix, 'root', out=d_fin if rank == 0 else None)
            elif world > 1:
                if backend == 'nccl':
                    dist.reduce(d_out, dst=0, op=dist.ReduceOp.SUM)     # Parallelize.py:318
                else:                                                    # gloo rehearsal: staged through the host
                    h_out = d_out.cpu()
                    dist.reduce(h_out, dst=0, op=dist.ReduceOp.SUM)
                    if rank == 0:
                        d_out.copy_(h_out)

        def step_spatial():
            if sp_state['fixed']:
                # ring ranges + packing in ONE call (two launches); the rows bound for this rank itself skip the collective; K0 reads the
                # received blocks as they are (no transpose)
                recv, _ = route_step_fixed(cat_dev, cols_local, rb, blockcap, plan_sp, route_work)          # [world][6][blockcap], NaN-padded
                mark('route')
                if sp_state['cd'] is None:                     # (the receive buffers are reused: the descriptor is built once)
                    B8 = blockcap * 8
                    bp = recv.data_ptr()
                    sp_state['cd'] = _lib.make_catalog_dev(world * blockcap, bp, bp + B8, bp + 2 * B8, bp + 3 * B8, ln1pz_ptr=bp + 4 * B8, lnM_ptr=bp + 5 * B8)
                    plan_sp.set_catalog_blocks(blockcap, 6 * blockcap)
                cd = sp_state['cd']
            else:
                plan_sp.set_catalog_blocks(0, 0)
                plan_sp.disc_rings(cat_dev, d_rings.data_ptr())
                mark('disc_rings')
                got = route_halos(cols_local, d_rings, rb, plan=plan_sp)       # [6][n]: the halos whose discs can touch my ring bands
                n = int(got.shape[1])
                assert n <= cap, "rank %d received %d halos, more than the plan holds (%d): a strongly clustered sky" % (rank, n, cap)
                cd = _lib.make_catalog_dev(n, got[0].data_ptr(), got[1].data_ptr(), got[2].data_ptr(), got[3].data_ptr(),
                                           ln1pz_ptr=got[4].data_ptr(), lnM_ptr=got[5].data_ptr())
            mark('route')
            b0, b1 = int(cuts[rank]), int(cuts[rank + 1])
            if paint:
                plan_sp.paint_bands(cd, b0, b1, d_slice.data_ptr(), acc_f64=(1 if acc_f64 else 2))
                mark('K0+K3')
                gather_slices(d_slice, pb, npix, 'root', out=d_fin if rank == 0 else None)
                mark('gather')
                return
            if sp_state.get('local_apron'):
                # the bands next to mine are computed here as well (their halos were routed along: set_route_margin), so the apron rows of
                # the regrid need no exchange with the neighbours: one collective less per step
                B0, B1, wlo, whi = sp_state['wide']
                if sp_state.get('full') is None or sp_state['full'].dtype != acc_dtype or sp_state['full'].numel() != (whi - wlo) * 3:
                    sp_state['full'] = torch.empty((whi - wlo) * 3, dtype=acc_dtype, device=dev)
                if not route_far[0]:
                    # rank 0's pixels are regridded straight into the final map (they never enter the gather), K0 .. K2 + far deposits + sums in
                    # ONE enqueue-only call (bfgx_offsets_regrid_bands_device)
                    k = ov['k'] if ov['on'] else 0
                    if ov['pending'][k] is not None:          # the gather that read this pair of buffers two passes ago
                        ov['pending'][k].wait()
                        ov['pending'][k] = None
                    fin_k, slice_k = ov['fin'][k], ov['slice'][k]
                    out_ptr = fin_k[p0:p1].data_ptr() if rank == 0 else slice_k.data_ptr()
                    # (nothing of pix_offsets leaves the device on this route: the parity-grade mode stays what it is on one GPU -- fp32 high
                    # and low halves in the 24 bytes per pixel the fp64 buffer has)
                    plan_sp.offsets_regrid_bands(cd, B0, B1, sp_state['full'].data_ptr(), b0, b1, d_map.data_ptr(), out_ptr, d_sums.data_ptr(),
                                                 d_foreign.data_ptr(), acc_f64=(acc if (acc == _lib.ACC_PARITY and not paint) else acc_f64))
                    mark('K0+K1+K2')
                    if ov['on'] and dist.is_initialized():
                        ov['pending'][k] = gather_slices_start(slice_k, pb, npix, out=fin_k if rank == 0 else None, group=ov['group'])
                        if sp_state.get('staging'):
                            ov['pending'][k].wait()
                            ov['pending'][k] = None
                        ov['k'], ov['last'] = k ^ 1, k
                    else:
                        gather_slices(slice_k, pb, npix, 'root', out=fin_k if rank == 0 else None, root_in_place=True)
                        ov['last'] = k
                    mark('gather')
                    return
                plan_sp.offsets_bands(cd, B0, B1, sp_state['full'].data_ptr(), acc_f64=acc_f64)
                mark('K0+K1')
                plan_sp.regrid_bands(b0, b1, d_map.data_ptr(), sp_state['full'].data_ptr(), wlo, whi, d_slice.data_ptr(), d_sums.data_ptr(), acc_f64=acc_f64)
                mark('K2')
                fp, fv = plan_sp.far_fetch()
                lists = [None] * world
                dist.all_gather_object(lists, (fp, fv))
                for qp, qv in lists:
                    m = (qp >= p0) & (qp < p1)
                    if m.any():
                        d_slice.index_add_(0, torch.from_numpy(qp[m] - p0).to(dev), torch.from_numpy(qv[m]).to(dev))
                mark('far')
                gather_slices(d_slice, pb, npix, 'root', out=d_fin if rank == 0 else None)
                mark('gather')
                return
            # once the reach is known the slice is computed in place inside the buffer that also holds the apron rings
            if sp_state['reach_known']:
                lo_, hi_ = sp_state['nd'][rank]
                if sp_state.get('full') is None or sp_state['full'].dtype != acc_dtype:
                    sp_state['full'] = torch.empty((hi_ - lo_) * 3, dtype=acc_dtype, device=dev)
                my_off = sp_state['full'][(p0 - lo_) * 3:(p1 - lo_) * 3]
            else:
                my_off = d_off[:(p1 - p0) * 3]
            plan_sp.offsets_bands(cd, b0, b1, my_off.data_ptr(), acc_f64=acc_f64)
            mark('K0+K1')
            if not sp_state['reach_known']:
                # the reach of the gathering regrid (rings of apron every rank exchanges and gathers from) is agreed ONCE, in the untimed
                # trial step: all_reduce(MAX) of the largest |offset| (K1's per-tile maxima) + one read-back.  The timed steps reuse it:
                # a source pixel that moves further than the reach covers is never lost -- it takes the far-deposit route, and a far
                # deposit that lands in another rank's slice is counted on the device (d_foreign, checked after the steps)
                if dist.is_initialized():
                    _hip_reach(None, plan_sp, my_off, bands=(b0, b1))
                sp_state['nd'] = [plan_sp.band_apron(int(cuts[j]), int(cuts[j + 1])) for j in range(world)]
                sp_state['reach_known'] = True
                # do the apron rows fit into ONE band either side of every rank's own (reach <= 16 rings, a band is 32 at NSIDE 1024)?  Then
                # the timed steps route every halo one band further and compute those bands locally instead of exchanging apron rows
                nbands = len(first) - 1
                B0, B1 = (max(b0 - 1, 0), min(b1 + 1, nbands)) if b1 > b0 else (b0, b0)
                fits = (b1 == b0) or (int(first[B0]) <= sp_state['nd'][rank][0] and int(first[B1]) >= sp_state['nd'][rank][1])
                flag = torch.tensor([1 if fits and os.environ.get('BFGX_BENCH_APRON_EXCHANGE') != '1' else 0], dtype=torch.int32,
                                    device=dev if backend == 'nccl' else 'cpu')
                if dist.is_initialized():
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()):
                    sp_state['local_apron'] = True
                    sp_state['wide'] = (B0, B1, int(first[B0]), int(first[B1]))
                    plan_sp.set_route_margin(plan_sp.tile_shape()[0])
                    sp_state['cd'] = None
            nd = sp_state['nd']
            off_apron = halo_exchange(my_off, pb, nd, 3, full=sp_state.get('full') if my_off.data_ptr() != d_off.data_ptr() else None)
            mark('apron exchange')
            plan_sp.regrid_bands(b0, b1, d_map.data_ptr(), off_apron.data_ptr(), nd[rank][0], nd[rank][1], d_slice.data_ptr(), d_sums.data_ptr(),
                                 acc_f64=acc_f64)
            mark('K2')
            if route_far[0]:
                fp, fv = plan_sp.far_fetch()
                lists = [None] * world
                dist.all_gather_object(lists, (fp, fv))
                for qp, qv in lists:
                    m = (qp >= p0) & (qp < p1)
                    if m.any():
                        d_slice.index_add_(0, torch.from_numpy(qp[m] - p0).to(dev), torch.from_numpy(qv[m]).to(dev))
            else:
                plan_sp.far_apply(d_slice.data_ptr(), p0, p1, d_foreign.data_ptr())
            mark('far')
            gather_slices(d_slice, pb, npix, 'root', out=d_fin if rank == 0 else None)
            mark('gather')

        def step():
            if spatial:
                return step_spatial()
            if paint:
                return step_paint()
            if args.algo == 0:
                d_off.zero_()                      # algo 1 stores every element of pix_offsets exactly once
                d_out.zero_()
            if world == 1 and not slices:
                # K0 + K1 + K2 in one call (bfgx_baryonify_device): K1's flush hands K2 the largest displacement of every tile
                plan.baryonify(cat_dev, d_map.data_ptr(), d_off.data_ptr(), d_out.data_ptr(), d_sums.data_ptr(), acc_f64=acc_f64)
                return
            plan.offsets(cat_dev, d_off.data_ptr(), acc_f64=acc_f64)
            if slices:
                my_off = sliced_reduce(d_off, pb, 3, recv=x_recv)
                if dist.is_initialized():
                    _hip_reach(None, plan, my_off)         # collective: rings of apron from the largest summed |offset|
                needs = [plan.band_apron(int(cuts[j]), int(cuts[j + 1])) for j in range(world)]
                off_apron = halo_exchange(my_off, pb, needs, 3)
                plan.regrid_bands(int(cuts[rank]), int(cuts[rank + 1]), d_map.data_ptr(), off_apron.data_ptr(), needs[rank][0], needs[rank][1],
                                  d_slice.data_ptr(), d_sums.data_ptr(), acc_f64=acc_f64)
                if route_far[0]:
                    # far deposits (pole caps, moves beyond the reach) may belong to another rank's slice: every rank learns all
                    # of them (host lists, a few KB) and adds those in its slice -- what distributed_process() always does
                    fp, fv = plan.far_fetch()
                    lists = [None] * world
                    dist.all_gather_object(lists, (fp, fv))
                    for qp, qv in lists:
                        m = (qp >= p0) & (qp < p1)
                        if m.any():
                            d_slice.index_add_(0, torch.from_numpy(qp[m] - p0).to(dev), torch.from_numpy(qv[m]).to(dev))
                else:
                    plan.far_apply(d_slice.data_ptr(), p0, p1, d_foreign.data_ptr())
                gather_slices(d_slice, pb, npix, 'root', out=d_fin if rank == 0 else None)
                return
            if args.algo == 0:
                d_out.zero_()                      # algo 1: the gathering regrid stores every pixel of the map exactly once
            if world > 1:
                if backend == 'nccl':
                    dist.reduce(d_off, dst=0, op=dist.ReduceOp.SUM)     # Parallelize.py:318 counterpart, before the regrid
                else:                                                    # gloo rehearsal: staged through the host
                    h_off = d_off.cpu()
                    dist.reduce(h_off, dst=0, op=dist.ReduceOp.SUM)
                    if rank == 0:
                        d_off.copy_(h_off)
            if rank == 0:
                plan.regrid(d_map.data_ptr(), d_off.data_ptr(), d_out.data_ptr(), d_sums.data_ptr(), acc_f64=acc_f64)
        return step

    def fence():
        if spatial:
            for q in (0, 1):                       # gathers still in flight belong to the passes before this fence
                if ov['pending'][q] is not None:
                    ov['pending'][q].wait()
                    ov['pending'][q] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, events, tp=None):
        tp = tp if tp is not None else (plan_sp if spatial else plan)            # the plan whose kernels run in the step
        tp.timing_enable(events)
        fence()
        prof = None
        if os.environ.get('BFGX_BENCH_HOST_PROFILE') == '1' and rank == 0:     # where the host time of a step goes (cProfile, stderr)
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        t_enq = time.perf_counter() - t0
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats('tottime').print_stats(25)
        fence()
        el = time.perf_counter() - t0
        if os.environ.get('BFGX_BENCH_ENQUEUE_TIME') == '1' and rank == 0:        # how much of a step is host-side enqueue time
            print("bench: %d steps enqueued in %.3f ms per step, done in %.3f ms per step" % (steps, t_enq / steps * 1e3, el / steps * 1e3),
                  file=sys.stderr, flush=True)
        kt = tp.timing_read() if events else None
        tp.timing_enable(False)
        if world > 1:
            te = torch.tensor([el], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        return el, kt

    step = run_steps(args.acc_f64 if paint else acc_res)
    if slices or spatial:
        # one untimed trial step.  A rank that fails (out of memory, an overflowing entry list) may have left its peers inside a
        # collective, so nothing is agreed on afterwards and no fallback runs on the same communicator: the rank reports and exits
        # non-zero, torch.distributed.run then ends the other ranks, and the launcher (or the user) starts a fresh run.
        try:
            step()
            fence()
            (plan_sp if spatial else plan).status()
        except Exception as e:        # noqa: BLE001
            import traceback
            traceback.print_exc()
            print("bench[rank %d]: the multi-rank step failed (%s: %s); try --exchange reduce" % (rank, type(e).__name__, e), file=sys.stderr, flush=True)
            sys.stderr.flush()
            os._exit(3)
        if spatial:
            if sp_state.get('local_apron'):
                # the trial step routed with margin 0 and switched the wider routing on afterwards: one more untimed step so that the
                # overflow flag below has seen the routing the timed steps use (the flag is sticky)
                step()
                fence()
            # did a routing block overflow on any rank (a catalog that is not shuffled, a sky patch)?  then route with variable splits
            ovf = route_work['overflow'].clone() if backend == 'nccl' else route_work['overflow'].cpu()
            if dist.is_initialized():
                dist.all_reduce(ovf, op=dist.ReduceOp.MAX)
            if int(ovf.item()):
                sp_state['fixed'] = False
                if rank == 0:
                    print("bench: a fixed-capacity routing block overflowed: routing with variable splits (one read-back per step)", file=sys.stderr, flush=True)
                step()
                fence()
        if not paint:
            # did a far deposit land in another rank's slice?  then every step routes the lists (collective decision)
            nf = d_foreign.clone() if backend == 'nccl' else d_foreign.cpu()
            if dist.is_initialized():
                dist.all_reduce(nf, op=dist.ReduceOp.MAX)
            if int(nf.item()):
                route_far[0] = True
                d_foreign.zero_()
                if rank == 0:
                    print("bench: far deposits cross band boundaries: routing the far lists in every step", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()
    # the timed region proper: EXACTLY --steps passes, no per-kernel events (they cost ~0.05 ms per step); `value` comes from it
    elapsed, _ = timed(step, args.steps, False)
    # the same K steps again with HIP events around every kernel on the launch stream (bfgx_plan_timing_*): kernel_ms, roofline
    elapsed_ev, kt = timed(step, args.steps, True) if not (args.no_kernel_events or brief) else (None, None)
    if spatial and os.environ.get('BFGX_BENCH_STAGE_TIMES') == '1':
        fence()
        sp_state['staging'], sp_state['t_last'] = True, time.perf_counter()
        step()
        sp_state['staging'] = False
        if rank == 0:
            print("bench: stage times of one synchronised step [ms]: %s" % json.dumps({k: round(v, 3) for k, v in stage_t.items()}), file=sys.stderr, flush=True)
    if os.environ.get('BFGX_BENCH_NOSTATUS') != '1':   # (timing-only ablation builds produce meaningless offsets)
        (plan_sp if spatial else plan).status()   # entry-list capacity, far-deposit list
    if spatial and sp_state['fixed']:
        # the overflow flag is sticky: a block that overflowed in ANY warm-up or timed step dropped halos silently (the map would still
        # conserve mass): never report a number from such a run
        ovf = route_work['overflow'].clone() if backend == 'nccl' else route_work['overflow'].cpu()
        if dist.is_initialized():
            dist.all_reduce(ovf, op=dist.ReduceOp.MAX)
        if int(ovf.item()):
            print("bench[rank %d]: a fixed-capacity routing block overflowed inside the timed steps: halos were dropped, no number reported" % rank,
                  file=sys.stderr, flush=True)
            os._exit(5)
    if (slices or spatial) and not paint:
        assert int(d_foreign.item()) == 0, "far deposits crossed a band boundary: use distributed_process(), which routes them"
    check = None
    want_check = (world > 1 and not args.no_check and not brief) or os.environ.get('BFGX_BENCH_CHECK') == '1'
    if want_check and strong and (slices or spatial) and rank == 0:
        # self-check (tests): the map the ranks assembled on rank 0 against ONE single-GPU pass over the whole catalog (fp32 pair math
        # on both sides: they agree to the stated fp32 tolerance, most pixels exactly)
        full = syn.make_catalog(total_halos)
        tf = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in full.items()}
        lz, lm = _lib.table_coords(full['M'], full['z'])
        tf['lnz'], tf['lnM'] = torch.from_numpy(lz).to(dev), torch.from_numpy(lm).to(dev)
        pf = engine.ShellPlan(model, keep, nside, total_halos, device=local_rank, stream=stream)
        cdf = _lib.make_catalog_dev(total_halos, tf['M'].data_ptr(), tf['z'].data_ptr(), tf['ra'].data_ptr(), tf['dec'].data_ptr(),
                                    ln1pz_ptr=tf['lnz'].data_ptr(), lnM_ptr=tf['lnM'].data_ptr())
        ref = torch.zeros(npix, dtype=torch.float64, device=dev)
        if paint:
            pf.paint(cdf, ref.data_ptr(), acc_f64=(1 if args.acc_f64 else 2))
        else:
            woff = torch.zeros(npix * 3, dtype=torch.float64, device=dev)
            pf.baryonify(cdf, d_map.data_ptr(), woff.data_ptr(), ref.data_ptr(), 0, acc_f64=acc_res)
        torch.cuda.synchronize()
        pf.status()
        got = (ov['fin'][ov['last']] if spatial else d_fin) if (slices or spatial) else d_out
        scale = float(ref.abs().max().item()) if paint else float(ref.mean().item())
        diff = float((got - ref).abs().max().item())
        check = {"max_abs_diff_vs_single_gpu": diff, "scale": scale, "scale_is": "max |map|" if paint else "mean(map)",
                 "tolerance": 2e-6 * scale, "ok": bool(diff <= 2e-6 * scale),
                 "what": "the map the ranks assembled on rank 0 against ONE single-GPU pass over the whole catalog on rank 0's GPU (same "
                         "arithmetic on both sides; outside the timed region)"}
        pf.close()
        del tf, ref

    extra = {}
    if world == 1 and paint and not args.acc_f64 and args.algo == 1 and not args.no_extras and not brief:
        step64 = run_steps(True)                 # fp64 pair math too (the 1e-10 parity path)
        for _ in range(3):
            step64()
        n64 = max(20, args.steps // 4)
        el64, _ = timed(step64, n64, False)
        extra["value_acc_f64"] = {"value": total_halos / el64 * n64, "unit": "halos/s", "ms_per_step": el64 / n64 * 1e3, "steps": n64,
                                  "dtype": "f64 throughout (fp64 pair math, fp64 LDS accumulation, fp64 map)"}
        del step64
    if world == 1 and not paint and args.algo == 1 and not args.no_extras and not brief:
        def side_line(pl, acc, nsteps, table_name, what):
            """the same step in another precision / on the other table: the timed region, then the same steps with kernel events (kernel_ms, roofline)"""
            res, _ = pl.precision(acc)
            w_off = torch.zeros(npix * 3, dtype=torch.float32 if res == _lib.ACC_F32 else torch.float64, device=dev)

            def st():
                pl.baryonify(cat_dev, d_map.data_ptr(), w_off.data_ptr(), d_out.data_ptr(), d_sums.data_ptr(), acc_f64=res)
            for _ in range(3):
                st()
            el, _ = timed(st, nsteps, False, tp=pl)
            el_e, kt_s = timed(st, nsteps, True, tp=pl)
            pl.status()
            ks = {k: (ms / n if n else None) for k, (ms, n) in kt_s.items() if n}
            sm = d_sums.cpu().numpy()
            o = {"value": total_halos / el * nsteps, "unit": "halos/s", "ms_per_step": el / nsteps * 1e3, "steps": nsteps, "what": what,
                 "precision": PRECISION_NAMES[res], "dtype": DTYPES[res], "kernel_ms": ks, "ms_per_step_with_kernel_events": el_e / nsteps * 1e3,
                 "mass_conserved": bool(np.isclose(sm[1], sm[0])), "regrid": pl.regrid_stats(),
                 "roofline": roofline(args, ks, n_pairs, nh, npix, paint, table=table_name, acc=res),
                 "roofline_regrid": roofline(args, ks, n_pairs, nh, npix, paint, table=table_name, acc=res, force='regrid')}
            del w_off
            return o
        nside_steps = max(20, args.steps // 4)
        if acc_res != _lib.ACC_F64:
            # the reference's own arithmetic on the SAME table: fp64 throughout, the 1e-10 parity path -- with its kernel times and roofline
            extra["value_acc_f64"] = side_line(plan, _lib.ACC_F64, nside_steps, args.table, "the headline step with fp64 throughout (--precision f64): fp64 pair math, fp64 pix_offsets, fp64 regrid")
        if acc_res != _lib.ACC_F32:
            # fp32 pair math on the same table: what rounds 1-4 ran by default.  Its error grows with displacement / pixel
            extra["value_f32"] = side_line(plan, _lib.ACC_F32, nside_steps, args.table, "the headline step with fp32 pair math, fp32 pix_offsets and fp32 regrid geometry (--precision f32)")
            extra["value_f32"]["tolerance_note"] = ("against the fp64 oracle on the Schneider19 table at config 2: 2.1e-5 mean(map) measured (the error of a bilinear deposit grows "
                                                    "with displacement / pixel: 20 pixels on this table); SURVEY 8(d) states 1e-6, which --precision auto / parity / f64 meet "
                                                    "(tests/test_gpu_fullsize.py::test_config2_s19_benchmark_table_full_size_vs_oracle)")
        torch.cuda.empty_cache()
        # the OTHER table of SURVEY 8(d) in the default precision: (i) the closed-form plumbing table (displacements < 0.2 pixels: K2 runs its lean
        # one-ring kernel, the plan picks fp32 pair math) -- the headline of rounds 1-4 -- or (ii) the Schneider19 benchmark table
        other = 'closed-form' if args.table == 's19' else 's19'
        table2 = syn.displacement_table(z, M, r) if other == 'closed-form' else syn.s19_displacement_table(z, M, r)
        model2, keep2 = engine.model_from_tables(axes, table2, syn.COSMO, args.eps, args.eps)
        plan2 = engine.ShellPlan(model2, keep2, nside, nh, device=local_rank, stream=stream)
        plan2.set_algo(1)
        key2 = "value_closed_form" if other == 'closed-form' else "value_s19"
        extra[key2] = side_line(plan2, acc_req, max(20, args.steps // 2), other,
                                "the same catalog, shell and step on SURVEY 8(d) table (i), the closed-form plumbing table (rounds 1-4 quoted this as `value`)"
                                if other == 'closed-form' else
                                "the same catalog, shell and step with the 10x10x500 Schneider19 one-halo displacement table built by K4-K6 (SURVEY 8d table (ii))")
        extra[key2]["table_disp_pixels"] = plan2.precision(acc_req)[1]
        if plan2.precision(acc_req)[0] != _lib.ACC_F64:
            f64o = side_line(plan2, _lib.ACC_F64, nside_steps, other, "fp64 throughout on that table")
            extra[key2]["acc_f64"] = {k: f64o[k] for k in ("value", "unit", "ms_per_step", "steps", "dtype", "kernel_ms")}
        plan2.close()
        del plan2
        step()                                   # (d_out / d_sums hold the headline result again)
        torch.cuda.synchronize()
        # the drop-in call from numpy arrays (BaryonifyShell.process(): PCIe both ways, plan cache warm after the first call)
        extra["end_to_end"] = end_to_end(args, cat, hmap, z, M, r, table)

    if (slices or spatial) and not paint and dist.is_initialized():
        # a rank's sums are {its source pixels, the deposits that landed in ITS slice + the far ones it listed}: only the totals match
        tot = d_sums.clone() if backend == 'nccl' else d_sums.cpu()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        d_sums.copy_(tot)
    out = None
    if rank == 0:
        sums = d_sums.cpu().numpy()
        ms_step = elapsed / args.steps * 1e3
        out = {
            "metric": "halos/sec for %s NSIDE=%d (%d-halo synthetic catalog%s)" % (
                "PaintProfilesShell" if paint else "BaryonifyShell", nside, total_halos if strong else args.halos,
                "" if world == 1 else (", ONE catalog split over the GPUs" if strong else " per GPU")),
            "value": total_halos / elapsed * args.steps, "unit": "halos/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": (("f64" if args.acc_f64 else "f64 ring-row geometry + f64 LDS accumulation + f64 map; f32 pair math (chord, ln r, read-out, exp)") if paint else
                      DTYPES[1 if (acc_res != _lib.ACC_F32 and (world > 1 or slices or spatial) and not (spatial and sp_state.get('local_apron') and not route_far[0])) else acc_res]),
            "data": "synthetic",
            "config": {"workload": "%s: %d-halo synthetic catalog %s (SURVEY 8d seeds), %s, "
                                   "NSIDE=%d shell, epsilon_max=%g, 10x10x500 %s %s table (edges = catalog min/max)" % (
                                       config_label(args, paint, nside, total_halos if strong else args.halos), total_halos if strong else args.halos,
                                       "split over the GPUs" if strong else "per GPU", "PaintProfilesShell" if paint else "BaryonifyShell", nside, args.eps,
                                       "Schneider19 (K4-K6 built)" if (args.table == 's19' and not paint) else "closed-form",
                                       "profile" if paint else "displacement"),
                       "halos_per_gpu": nh, "nside": nside, "npix": npix, "pairs_per_gpu": n_pairs, "table": ("closed-form" if paint else args.table),
                       **({"catalog_order": {"sky": "sky: (band of 32 rings, azimuth) order", "mass": "mass: heaviest halos first"}[args.catalog_order]} if args.catalog_order != 'shuffled' else {}),
                       "accumulators": "f64 LDS tiles; global " + ("f64" if paint else {0: "f32 pix_offsets / f64 map", 1: "f64 pix_offsets / f64 map", 3: "split f32 hi + lo pix_offsets / f64 map"}[acc_res]),
                       **({} if paint else {"precision": PRECISION_NAMES[acc_res], "precision_requested": ("f64" if args.acc_f64 else args.precision),
                                            "table_disp_pixels": disp_px}),
                       "parallelism": ("single GPU" if world == 1 else
                                       "spatial sharding x%d: halos routed (RCCL all_to_all of catalog columns) to the ranks whose ring bands their discs "
                                       "touch, every rank computes and regrids its own pixels (%s), disjoint slices -> rank 0"
                                       % (world, "apron rows computed locally: halos routed one band further, no exchange with the neighbours"
                                          if (not paint and sp_state.get('local_apron')) else ("no aprons" if paint else "apron rings exchanged"))
                                       if spatial else
                                       "halo shards x%d + RCCL all_to_all reduce-scatter by pixel slices, apron-ring exchange, banded gathering regrid on "
                                       "every rank, disjoint slices -> rank 0" % world
                                       if slices else "halo shards x%d + RCCL reduce(accumulator) -> rank 0" % world)},
            "map_pixels_per_s": npix / elapsed * args.steps,
            "mass_conserved": None if paint else bool(np.isclose(sums[1], sums[0])),
            "backend": (backend if dist.is_initialized() else None), "world_size_seen": (dist.get_world_size() if dist.is_initialized() else 1),
        }
        if spatial:
            out["routing_rows"] = routing_rows
            if sp_state['fixed'] and sp_state.get('local_apron') and not route_far[0] and not paint:
                # what one timed step enqueues on a rank (two library calls + two collectives; nothing is read back by the host)
                out["step_launches"] = ["route_prepare_kernel (NaN into column 0 of the routing blocks, cursors zeroed)",
                                        "route_step_kernel (ring range per halo + rows packed by destination; this rank's own rows straight into the receive buffer)",
                                        "all_to_all_single (halo rows; the split towards oneself is empty)",
                                        "memset (every control word of the binning and the regrid)", "halo_prep_kernel (blocked catalog: no transpose)",
                                        "tile_scan_kernel", "tile_place_kernel", "tile_scatter2f_kernel", "tile_apron_kernel", "tile_regrid3_kernel",
                                        "regrid_far_local_kernel (far deposits + the two sums)",
                                        "all_to_all_single (map slices -> rank 0; rank 0 regrids its own slice in place)" +
                                        (" -- asynchronous, on a communicator of its own: it overlaps the next pass" if ov['on'] else "")]
                out["gather_overlapped"] = bool(ov['on'])
        if kt is not None:
            kernels = {k: (ms / n if n else None) for k, (ms, n) in kt.items() if n}
            out["ms_per_step_with_kernel_events"] = elapsed_ev / args.steps * 1e3
            out["kernel_ms"] = kernels
            out["roofline"] = roofline(args, kernels, n_pairs, nh, npix, paint, acc=(None if paint else acc_res))
        out.update(extra)
        if check is not None:
            out["check"] = check
        if world == 1 and not args.no_cpu_baseline and not paint and not brief:
            out["cpu_baseline"] = cpu_baseline(args, cat, hmap, axes, table)
    plan.close()
    if 'plan_sp' in locals():
        plan_sp.close()
    return out if rank == 0 else None


def config_label(args, paint, nside, halos):
    """which BASELINE.json config a line is -- only when the sizes ARE that config's"""
    if args.config == 4 or (not paint and nside == 2048 and halos == 10_000_000):
        return "BASELINE config 4"
    if paint and nside == 2048 and halos == 1_000_000:
        return "BASELINE config 3"
    if not paint and nside == 1024 and halos == 1_000_000:
        return "BASELINE config 2"
    if not paint and nside == 2048 and halos == 1_250_000:
        return "one GPU's share of BASELINE config 4 (1e7 halos / 8)"
    return "not a BASELINE config (%s sizes)" % ("paint" if paint else "baryonify")


def committed_traffic(match, key, metric_has=None):
    """HBM bytes per launch (and SQ_INSTS_VALU) of kernel group `key` from the committed rocprofv3 --pmc passes of THIS configuration:
    profiles/traffic_<tag>.json, written by scripts/traffic_pmc.sh (one counter per pass, FETCH_SIZE doubled as the gfx950 guide
    prescribes).  `match`: the config keys that must agree with the file's bench line.  Returns (traffic, valu, file) or (None, None, None)."""
    import glob
    for f in sorted(glob.glob(os.path.join(HERE, 'profiles', 'traffic_*.json'))):
        try:
            tj = json.load(open(f))
            c = tj.get('config', {})
            if metric_has is not None and metric_has not in (tj.get('metric') or ''):
                continue
            if all(c.get(k, {'table': 'closed-form', 'precision': 'f32'}.get(k)) == v for k, v in match.items()) and key in tj.get('kernels', {}):      # (files of rounds 1-4: closed-form unless said, fp32 pair math)
                return tj['kernels'][key], tj.get('valu_wave_insts', {}).get(key), os.path.basename(f)
        except Exception:        # noqa: BLE001
            continue
    return None, None, None


def roofline(args, kernels, n_pairs, nh, npix, paint, table=None, force=None, acc=None):
    """The dominant kernel (by measured time) against the HBM roof SURVEY 8(d) defines: algorithmic bytes per launch / average
    launch duration (HIP events on the launch stream over K steps).  traffic / VALU counts come from the committed rocprofv3 --pmc
    passes of this same configuration (profiles/traffic_latest.json)."""
    if acc is None:
        acc = 1 if args.acc_f64 else 0
    acc_b = 4 if acc == 0 else 8                 # (the parity-grade mode stores hi + lo: 8 bytes per component, like fp64)
    # K1 12 B/pair (24 B with fp64 accumulators) + 32 B/halo; K2 60 B per map pixel (3 acc + 8 + 4 x 8 + 8); K3 8 B/pair
    alg = {'offsets': n_pairs * 3 * acc_b + nh * 32, 'regrid': npix * (3 * acc_b + 8 + 4 * 8 + 8), 'paint': n_pairs * 8 + nh * 32 + npix * 8}
    table = 'closed-form' if paint else (table or args.table)          # (painting has one synthetic table; --table is the displacement's)
    dom = force or ('paint' if paint else max(('offsets', 'regrid'), key=lambda k: kernels.get(k) or 0.0))
    real = 'float' if acc == 0 else 'double'
    areal = 'double' if acc == 1 else 'float'    # the type pix_offsets are stored in
    # (the fluid form: one workgroup per CU with two tile slots, 16 waves for fp32 pair math, 12 for fp64; shells of < 512 tiles or fewer than 6e5 halos per sphere: the barrier form)
    k1 = "tile_scatter2f_kernel" if (12 * args.nside ** 2 >= 2 * 256 * 2048 and nh >= 600000) else "tile_scatter2_kernel"
    names = {"offsets": ("%s<OFFSETS, %s, %s%s>" % (k1, areal, real, ", parity" if acc == 3 else "")) if args.algo == 1 else "halo_scatter_kernel<OFFSETS>",
             # (<.., 0>: the lean gather, reach of one ring; <.., 2>: the walking kernel -- every tile on the S19 table)
             "regrid": ("tile_regrid3_kernel<%s, %s, %d%s>" % (areal, real, 2 if table == 's19' else 0, ", split" if acc == 3 else "")) if args.algo == 1 else "regrid_kernel",
             "paint": ("%s<PAINT, double, %s>" % (k1, real)) if args.algo == 1 else "halo_scatter_kernel<PAINT>"}
    ach = alg[dom] / (kernels[dom] * 1e-3) / 1e9
    traffic = valu = src = None
    if args.algo == 1:
        match = {"halos_per_gpu": nh, "nside": args.nside, "pairs_per_gpu": n_pairs, "table": table}
        if not paint:
            match["precision"] = PRECISION_NAMES[acc]
        traffic, valu, src = committed_traffic(match, dom, metric_has="PaintProfilesShell" if paint else "BaryonifyShell")
    r = {"kernel": names[dom], "algo": args.algo, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": alg[dom],
         "launch_ms": kernels[dom]}
    if valu:
        # what binds the kernel: vector issue.  Measured on MI355X (scripts/ubench/valu_rate.hip, profiles/r02_ubench_valu_rate.txt):
        # one wave64 fp32 instruction per ~1.0 ns per SIMD with >= 2 waves resident, fp64 / packed fp32 2.0 ns; 1024 SIMDs
        r["valu_issue"] = {"wave_insts_per_launch": valu, "achieved_per_s": valu / (kernels[dom] * 1e-3),
                           "peak_per_s": 1024 / 1.0e-9, "frac": valu / (kernels[dom] * 1e-3) / (1024 / 1.0e-9)}
    if dom == 'regrid' and args.algo == 1:
        r["note"] = ("gathering regrid: every output pixel owned by one workgroup, deposits summed in LDS, the tile stored once; on the S19 table "
                     "the walking kernel scans 2.7 window pixels per stored one (two |o|^2 thresholds each) and evaluates the ~1.3 that can "
                     "reach the tile on full waves: bounded by the scan's loads and vector issue, not by HBM (DESIGN.md section 4)")
        return r
    r["note"] = ("tile-owned LDS accumulation, no global atomics, every output element stored once: the algorithmic HBM traffic is 12 B per "
                 "(halo, pixel) pair against ~115 vector instructions per pair (fp32 pair math; the fp64 ring-row phase adds ~350 per 64 rows), so the "
                 "kernel sits below the HBM roof and is bounded by vector issue + latency at 4 waves/SIMD; since round 4 (fluid form) no wave waits "
                 "at a barrier between tiles: 74 % of a wave's time is chunk work, 17 % waiting for a tile slot (DESIGN.md section 4)") if args.algo == 1 else \
                ("scatter-add path: the applicable ceiling for the atomic share is ~1300 GB/s (gfx950 memory-side float atomics), not the 8 TB/s stream peak")
    return r


def end_to_end(args, cat, hmap, z, M, r, table, calls=5):
    """bfg.Runners.BaryonifyShell(...).process() from numpy arrays: what a user of the drop-in hits (H2D of catalog + map, kernels,
    D2H of the map).  The first call builds and caches the plan; the median of the following calls is reported."""
    import baryonification_amd as bfg
    from baryonification_amd import synthetic as syn
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=syn.COSMO)
    Shell = bfg.utils.LightconeShell(map=hmap, cosmo=syn.COSMO)
    model = bfg.Profiles.Baryonification2D(None, None, bfg.utils.Cosmology.from_dict(syn.COSMO), epsilon_max=args.eps)
    model.set_table(z, M, r, table)
    runner = bfg.Runners.BaryonifyShell(Catalog, Shell, args.eps, model, verbose=False)
    ts, st = [], None
    for i in range(calls + 1):
        t0 = time.perf_counter()
        out = runner.process()
        ts.append(time.perf_counter() - t0)
        st = runner.last_stats
    med = float(np.median(ts[1:]))
    return {"what": "BaryonifyShell.process() from numpy arrays, PCIe-inclusive (never `value`); the catalog stays on the device between calls "
                    "(bfgx_opts.catalog_token), the map goes up / is regridded / comes back in band ranges; phases: ms_h2d = until the last byte of "
                    "the map has arrived (K0 + K1 and the first ranges' regrid and download run underneath), ms_kernels = what is left of the "
                    "kernels after that, ms_d2h = until the last range is back",
            "ms_per_call": med * 1e3,
            "ms_first_call": ts[0] * 1e3, "halos_per_s": cat['M'].size / med, "mass_conserved": bool(np.isclose(out.sum(), hmap.sum())),
            "phases_ms": {k: st[k] for k in ('ms_h2d', 'ms_kernels', 'ms_d2h')} if st else None}


if __name__ == '__main__':
    main()
