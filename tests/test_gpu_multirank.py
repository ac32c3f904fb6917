"""HIP kernels + the exchange steps together, under the driver's `-m gpu` run: the rehearsals of the N > 1 product paths with two ranks
that share device 0 over gloo (a one-GPU box; collectives staged through the host).  utils/Parallelize.distributed_process with the HIP
engine (exchange = 'slices' and 'spatial', both runners, halos on the pole caps) == the single-process runner; utils/GridSlabs with the HIP
backend (config 5's slab decomposition) == the single-GPU pipeline.  One launch each; the CPU/gloo suites cover the exchange logic for more
ranks with the oracle as per-rank compute."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(script, world):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GLOO_SOCKET_IFNAME='lo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'scripts', script)]
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)


def test_distributed_process_hip_two_ranks(gpu):
    out = _launch('rehearse_multi_gpu.py', 2)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith('rehearsal')]
    assert len(lines) == 6 and all(l.rstrip().endswith('OK') for l in lines), out.stdout[-2000:]       # 3 cases x 2 exchanges


def test_grid_slabs_hip_two_ranks(gpu):
    out = _launch('rehearse_grid_slabs.py', 2)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert 'OK' in out.stdout and 'FAIL' not in out.stdout, out.stdout[-2000:]
