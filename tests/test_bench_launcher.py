"""`python bench.py --gpus N` must start its own ranks: the parent spawns `python -m torch.distributed.run` as a child
before it touches torch or HIP.  No GPU here: BFGX_BENCH_STOP_AFTER_INIT=1 stops every rank after the gloo rendezvous."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = dict(os.environ, BFGX_BENCH_STOP_AFTER_INIT='1', **(env or {}))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], cwd=ROOT, capture_output=True, text=True, timeout=300, env=e)


def test_gpus_2_self_launches_and_defaults_to_strong():
    out = _bench('--gpus', '2', '--steps', '2', '--warmup', '1')
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d == {"launcher_ok": True, "n_gpus": 2, "scaling": "strong"}


def test_world_size_mismatch_is_an_error_not_an_assert():
    e = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999', BFGX_BENCH_STOP_AFTER_INIT='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], cwd=ROOT, capture_output=True, text=True, timeout=120, env=e)
    assert out.returncode != 0 and 'WORLD_SIZE=1 but --gpus 2' in out.stderr


def test_parent_does_not_import_torch_before_spawning():
    """the launching parent must not have initialised anything GPU-side: it only needs the standard library (+ numpy at module import)"""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    body = src[src.index('def launch_ranks'):src.index('def dist_context')]
    assert 'import torch' not in body and 'os.exec' not in body and 'subprocess.run' in body
    head = src[:src.index('def parse')]
    assert 'import torch' not in head
