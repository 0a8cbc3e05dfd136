"""world_size-2 gloo worker (CPU): shard the global env batch, step each shard with the CPU oracle, all-gather the flat
observations through the product's control-plane helper, compare on rank 0 with an unsharded run."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cage_challenge_4_amd import distributed as D
from oracle_binding import OracleVecEnv, random_actions


def main():
    rank, world, _ = D.init_control_plane('gloo')
    total, seed0, T = 6, 4242, 12
    lo, hi, seeds = D.shard_seeds(seed0, total, rank, world)
    env = OracleVecEnv(hi - lo, steps=50)
    env.reset(seeds=seeds)
    for t in range(T):
        a = random_actions(seed0 + lo, t, hi - lo)      # env-indexed action stream == slice of the global one
        obs, rew, done, _ = env.step(a)
    g_obs = D.allgather_host(obs, world)
    g_rew = D.allgather_host(rew, world)
    # max-over-ranks timing reduction used by bench.py
    import torch, torch.distributed as dist
    tm = torch.tensor([float(rank + 1)])
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    if rank == 0:
        full = OracleVecEnv(total, steps=50)
        full.reset(seeds=np.uint64(seed0) + np.arange(total, dtype=np.uint64))
        for t in range(T):
            fo, fr, fd, _ = full.step(random_actions(seed0, t, total))
        assert g_obs.shape == (total, 578)
        assert np.array_equal(g_obs, fo), 'sharded + all-gathered observations differ from the unsharded batch'
        assert np.array_equal(g_rew, fr)
        assert float(tm) == float(world)
        print('GLOO_OK')
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
