"""CPU: the N>1 path with world_size 2 over gloo (torch.distributed), as the driver's multi-GPU launch does it."""
import os
import socket
import subprocess
import sys
from conftest import ROOT


def free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_sharding_and_allgather(oracle_lib):
    worker = os.path.join(ROOT, 'tests', '_gloo_worker.py')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), worker]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'GLOO_OK' in out.stdout
