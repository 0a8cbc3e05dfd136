"""world_size-2 gloo worker (CPU): drives bench.py's OWN sharding and timing functions (plan_shard, timed_regions,
summarise) with the CPU oracle as the step function, all-gathers the flat observations through the product's control-plane
helper, and compares on rank 0 with an unsharded run."""
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
    rank, world, _ = D.init_control_plane('gloo')
    import torch
    import torch.distributed as dist
    total, seed0, K, W = 6, 4242, 4, 2
    lo, hi = bench.plan_shard(total, rank, world)              # the bench's sharding
    assert (lo, hi) == D.shard_seeds(seed0, total, rank, world)[:2]
    env = OracleVecEnv(hi - lo, steps=50)
    env.reset(seeds=np.uint64(seed0) + np.arange(lo, hi, dtype=np.uint64))
    last = {}

    def run_k(t0, k, timed):                                   # what CC4VecEnv.run_random_steps does on the device
        for t in range(t0, t0 + k):
            last['obs'], last['rew'], _, _ = env.step(random_actions(seed0 + lo, t, hi - lo))   # env-indexed action stream == slice of the global one
        return float(rank + 1) if timed else 0.0               # a rank-dependent "kernel time": the reduction must return the max

    def reduce_max(v):
        t = torch.tensor(v, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(a) for a in t]

    secs, kms = bench.timed_regions(run_k, K, W, 0.0, lambda: None, dist.barrier, reduce_max, max_regions=3)
    # min_seconds 0 -> exactly one region; every rank sees the same (max-reduced) numbers and stops together
    assert len(secs) == 1 and kms == [float(world)]
    res = bench.summarise(secs, kms, K, total)
    assert res['regions'] == 1 and abs(res['value'] - 5.0 * total * K / secs[0]) < 1e-6 * res['value']
    T = W + K
    g_obs = D.allgather_host(last['obs'], world)
    g_rew = D.allgather_host(last['rew'], world)
    if rank == 0:
        full = OracleVecEnv(total, steps=50)
        full.reset(seeds=np.uint64(seed0) + np.arange(total, dtype=np.uint64))
        for t in range(T):
            fo, fr, fd, _ = full.step(random_actions(seed0, t, total))
        assert g_obs.shape == (total, 578)
        assert np.array_equal(g_obs, fo), 'sharded + all-gathered observations differ from the unsharded batch'
        assert np.array_equal(g_rew, fr)
        print('GLOO_OK')
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
