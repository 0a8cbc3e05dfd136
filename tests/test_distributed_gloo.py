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


def test_two_ranks_over_the_torch_free_control_plane(oracle_lib, tmp_path):
    """VERDICT r02 #8a: rendezvous, barriers, reductions, the unique-id broadcast and gathers through a shared directory
    (cage_challenge_4_amd.distributed.FilePlane) -- two plain processes, no launcher, PyTorch never imported."""
    worker = os.path.join(ROOT, 'tests', '_plane_worker.py')
    procs = []
    for r in range(2):
        env = dict(os.environ, OMP_NUM_THREADS='1', RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()),
                   CC4_CONTROL_PLANE_KEY=f'test_{os.getpid()}', CC4_CONTROL_PLANE_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, worker], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[0][-1500:] + o[1][-1500:] for o in outs)
    assert 'PLANE_OK' in outs[0][0]
    assert not [f for f in os.listdir(tmp_path) if os.listdir(os.path.join(tmp_path, f))]      # the ranks cleaned up behind them
