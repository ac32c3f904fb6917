"""bench.py prints ONE JSON line with the fields the driver reads (task contract): run at a small size, every mode."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_common(d, n_gpus=1):
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == n_gpus and d['higher_is_better'] is True and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert isinstance(d['config'].get('workload'), str) and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and r['peak'] > 0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert d['value'] > 0 and d['ms_per_step'] > 0


def test_default_mode_line_small(gpu):
    d = _run('--halos', '20000', '--nside', '128', '--steps', '5', '--warmup', '2', '--cpu-threads', '2')
    _check_common(d)
    assert d['unit'] == 'halos/s' and d['steps'] == 5 and d['warmup'] == 2 and d['scaling'] == 'strong' and d['mass_conserved'] is True
    assert abs(d['value'] - 20000 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0
    assert 'value_acc_f64' in d and 'end_to_end' in d and d['end_to_end']['mass_conserved'] is True
    assert set(d['kernel_ms']) >= {'prep', 'offsets', 'regrid'}
    # the headline runs SURVEY 8(d) table (ii), the Schneider19 benchmark table, in the default precision (the plan picks from the table and says so)
    assert d['backend'] is None and d['world_size_seen'] == 1 and '20000-halo' in d['metric'] and d['config']['table'] == 's19'
    assert d['config']['precision'] in ('f32', 'parity') and d['config']['precision_requested'] == 'auto' and d['config']['table_disp_pixels'] > 0
    assert d['roofline']['kernel'].startswith('tile_') and d['roofline']['launch_ms'] > 0          # (at this size K2 may be the dominant one)
    # fp64 throughout (the reference's own arithmetic) rides along WITH its kernel times and roofline
    f64 = d['value_acc_f64']
    assert f64['precision'] == 'f64' and f64['value'] > 0 and set(f64['kernel_ms']) >= {'prep', 'offsets', 'regrid'} and f64['mass_conserved'] is True
    assert 'double, double' in f64['roofline']['kernel'] and 'tile_regrid3_kernel<double, double' in f64['roofline_regrid']['kernel']
    # table (i), the closed-form plumbing table (rounds 1-4 quoted it as `value`), in the default precision
    cf = d['value_closed_form']
    assert cf['value'] > 0 and cf['mass_conserved'] is True and cf['acc_f64']['value'] > 0 and cf['precision'] in ('f32', 'parity')
    assert cf['regrid']['far_overflowed'] is False and cf['regrid']['max_reach_rings'] >= 1
    for r in (cf['roofline'], cf['roofline_regrid'], f64['roofline'], f64['roofline_regrid']):
        assert r['bound'] == 'hbm' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    # the other table as the headline, one precision forced
    d = _run('--halos', '20000', '--nside', '128', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--table', 'closed-form', '--precision', 'parity')
    _check_common(d)
    assert d['config']['table'] == 'closed-form' and d['config']['precision'] == 'parity' and 'value_s19' in d and 'value_f32' in d and d['mass_conserved'] is True


def test_paint_and_grid_lines_small(gpu):
    d = _run('--mode', 'paint', '--halos', '20000', '--nside', '128', '--steps', '3', '--warmup', '1', '--no-cpu-baseline')
    _check_common(d)
    assert 'PaintProfilesShell' in d['metric'] and 'f32 pair math' in d['dtype'] and 'f64 map' in d['dtype']
    assert d['value_acc_f64']['value'] > 0 and d['roofline']['kernel'].endswith('float>')
    d = _run('--mode', 'paint', '--halos', '20000', '--nside', '128', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--acc-f64')
    assert d['dtype'] == 'f64' and 'value_acc_f64' not in d
    d = _run('--mode', 'grid3d', '--ngrid', '64', '--grid-halos', '500', '--steps', '2', '--warmup', '1', '--no-cpu-baseline')
    _check_common(d)
    assert d['unit'] == 'cells/s' and d['mass_conserved'] is True and d['pk_finite_bins'] > 0
    d = _run('--mode', 'snapshot', '--ngrid', '64', '--grid-halos', '500', '--steps', '2', '--warmup', '1', '--no-cpu-baseline')
    _check_common(d)
    assert d['unit'] == 'particles/s' and d['mass_conserved'] is True


def test_two_ranks_self_launched_strong_and_weak(gpu):
    """`python bench.py --gpus 2` as the driver types it (no launcher): the ranks are started as a child torch.distributed.run; both share
    device 0 over gloo (a one-GPU box), the product's HIP kernels + every exchange step of the spatial sharding run for real"""
    env = dict(os.environ, BFGX_DIST_BACKEND='gloo')          # (no BFGX_BENCH_CHECK: N > 1 checks itself by default)
    env.pop('BFGX_BENCH_CHECK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--halos', '20001', '--nside', '128', '--steps', '3',
                          '--warmup', '1'], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    _check_common(d, n_gpus=2)
    assert d['scaling'] == 'strong' and d['mass_conserved'] is True and d['config']['halos_per_gpu'] == 10001      # (shards that differ by one halo: every rank must still use the same routing block size)
    w = d['value_weak']
    assert w['scaling'] == 'weak' and w['value'] > 0 and w['mass_conserved'] is True
    # the map the two ranks assembled (fixed-capacity routing with NaN-padded rows, band-restricted K0 + K1, apron exchange, banded
    # regrid, gather) == one single-GPU pass over the whole catalog, to the stated fp32 tolerance
    c = d['check']
    assert c['ok'] is True and c['max_abs_diff_vs_single_gpu'] <= c['tolerance'] == 2e-6 * c['scale'], c
    # the line says what ran: backend, the world size torch.distributed saw, the real halo count, and how much padding the
    # fixed-capacity routing blocks carry (sized by one untimed count pass, not by a guess)
    assert d['backend'] == 'gloo' and d['world_size_seen'] == 2
    assert '20001-halo' in d['metric'] and 'not a BASELINE config' in d['config']['workload']
    rr = d['routing_rows']
    assert rr['padded_per_rank'] == 2 * rr['blockcap'] and 0 < rr['real_rank0'] <= rr['real_max_rank'] <= rr['padded_per_rank']
    assert rr['padded_per_rank'] <= 1.3 * rr['real_max_rank'] + 4 * 512, rr
    # --no-check skips the comparison
    out2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--halos', '20001', '--nside', '128', '--steps', '2',
                           '--warmup', '1', '--no-check', '--no-extras'], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out2.returncode == 0, out2.stderr[-3000:]
    d2 = json.loads([l for l in out2.stdout.strip().splitlines() if l.startswith('{')][0])
    assert 'check' not in d2 and d2['world_size_seen'] == 2


def test_forced_single_rank_exchange_over_rccl_with_the_overlapped_gather(gpu):
    """BFGX_FORCE_EXCHANGE=1: the N > 1 code of the shell path with ONE RCCL rank (the only RCCL run a one-GPU box allows), here with the
    asynchronous gather on its own communicator forced on (BFGX_BENCH_OVERLAP_GATHER=1: the default for N > 1): two slice buffers, two final
    maps, waits before a buffer is reused and at the fence -- and the assembled map still equals a single-GPU pass"""
    env = dict(os.environ, BFGX_FORCE_EXCHANGE='1', BFGX_BENCH_OVERLAP_GATHER='1', BFGX_BENCH_CHECK='1')
    env.pop('BFGX_DIST_BACKEND', None)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--halos', '30000', '--nside', '256', '--steps', '7', '--warmup', '2',
                          '--no-extras', '--no-cpu-baseline'], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith('{')][0])
    assert d['backend'] == 'nccl' and d['world_size_seen'] == 1 and d['gather_overlapped'] is True and d['mass_conserved'] is True
    assert d['check']['ok'] is True and 'asynchronous' in d['step_launches'][-1]
