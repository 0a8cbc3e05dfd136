"""world_size-2 worker (CPU) of the torch-free control plane (cage_challenge_4_amd.distributed.FilePlane): drives bench.py's OWN
sharding and timing functions (plan_shard, timed_regions, summarise) with the CPU oracle as the step function, exchanges the
128-byte id, gathers per-rank records and the observations, and compares on rank 0 with an unsharded run."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench                                                   # noqa: E402
from cage_challenge_4_amd import distributed as D              # noqa: E402
from oracle_binding import OracleVecEnv, random_actions        # noqa: E402


def main():
    assert 'torch' not in sys.modules
    plane = D.control_plane()
    rank, world = plane.rank, plane.world
    assert plane.kind == 'file' and world == 2
    total, seed0, K, W = 6, 4242, 4, 2
    lo, hi = bench.plan_shard(total, rank, world)
    env = OracleVecEnv(hi - lo, steps=50)
    env.reset(seeds=np.uint64(seed0) + np.arange(lo, hi, dtype=np.uint64))
    last = {}

    def run_k(t0, k, timed):
        for t in range(t0, t0 + k):
            last['obs'], last['rew'], _, _ = env.step(random_actions(seed0 + lo, t, hi - lo))
        return float(rank + 1) if timed else 0.0               # a rank-dependent "kernel time": the reduction must return the max

    secs, kms = bench.timed_regions(run_k, K, W, 0.0, lambda: None, plane.barrier, lambda v: plane.allreduce(v, 'max'), max_regions=3)
    assert len(secs) == 1 and kms == [float(world)]
    ident = plane.bcast_bytes(bytes(range(128)) if rank == 0 else b'', src=0)          # the RCCL unique id travels like this
    assert ident == bytes(range(128))
    assert plane.allreduce([rank + 1.0, 5.0], 'sum') == [3.0, 10.0] and plane.allreduce([float(rank)], 'min') == [0.0]
    recs = plane.gather_obj({'rank': rank, 'envs': hi - lo})
    assert [r['rank'] for r in recs] == [0, 1] and sum(r['envs'] for r in recs) == total
    rows = plane.exchange(np.ascontiguousarray(last['obs']).tobytes())
    rews = plane.exchange(np.ascontiguousarray(last['rew']).tobytes())
    if rank == 0:
        g_obs = np.concatenate([np.frombuffer(b, np.int32).reshape(-1, 578) for b in rows])
        g_rew = np.concatenate([np.frombuffer(b, np.float32) for b in rews])
        full = OracleVecEnv(total, steps=50)
        full.reset(seeds=np.uint64(seed0) + np.arange(total, dtype=np.uint64))
        for t in range(W + K):
            fo, fr, fd, _ = full.step(random_actions(seed0, t, total))
        assert np.array_equal(g_obs, fo) and np.array_equal(g_rew, fr), 'sharded + gathered results differ from the unsharded batch'
        print('PLANE_OK')
    for _ in range(200):                                       # many operations in a row: file reuse / clean-up
        plane.barrier()
    plane.close()
    assert 'torch' not in sys.modules


if __name__ == '__main__':
    main()
