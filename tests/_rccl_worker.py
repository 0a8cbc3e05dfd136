"""One rank of a REAL multi-rank RCCL job (GPU): this rank's shard of the episode batch on device LOCAL_RANK, cc4_comm_init over the
torch-free control plane, the exchange from inside the one-launch kernel and from per-step launches, and every step's gathered rows
against the CPU oracle's UNSHARDED batch (test infrastructure: the oracle is the checker).  Launched by
tests/test_hip_parity.py::test_two_real_ranks_gather_the_unsharded_batch, which skips unless two devices are visible."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
from cage_challenge_4_amd import CC4VecEnv, distributed as D       # noqa: E402
from oracle_binding import OracleVecEnv, random_actions            # noqa: E402


def pack(obs):
    """[M, 578] values 0..2 -> [M, 148] bytes, value i in bits 2 * (i & 3) of byte i >> 2 (include/cc4.h CC4_OBS_PACKED_BYTES)."""
    v = np.zeros((obs.shape[0], 592), np.uint8)
    v[:, :578] = obs
    v = v.reshape(obs.shape[0], 148, 4)
    return (v[:, :, 0] | (v[:, :, 1] << 2) | (v[:, :, 2] << 4) | (v[:, :, 3] << 6)).astype(np.uint8)


def main():
    plane = D.control_plane()
    rank, world = plane.rank, plane.world
    total, seed0, steps = int(os.environ.get('CC4_TEST_TOTAL', 2048)), 515, 60
    lo, hi = D.shard_range(total, rank, world)
    n = hi - lo
    dev = CC4VecEnv(n, steps=steps, rng_mode=1, autoreset=True, device_id=int(os.environ.get('LOCAL_RANK', rank)))
    dev.reset(seeds=np.uint64(seed0) + np.arange(lo, hi, dtype=np.uint64))
    D.init_rccl(dev, rank, world, plane)
    info = dev.comm_info()
    infos = plane.gather_obj(info)
    assert all(i['nccl_comm_count'] == world for i in infos), infos
    assert len({i['device_uuid'] for i in infos}) == world, 'the ranks must sit on different devices: ' + repr(infos)
    ora = OracleVecEnv(total, steps=steps, rng_mode=1, autoreset=True)
    ora.reset_batch(seed0)
    t = 0
    for inkernel in (True, False):
        if not inkernel:
            os.environ['CC4_EXCHANGE_INKERNEL'] = '0'            # a second handle on the per-step exchange, continuing the same episodes
            snap = [dev.snapshot(i) for i in range(n)]
            dev.close()
            dev = CC4VecEnv(n, steps=steps, rng_mode=1, autoreset=True, device_id=int(os.environ.get('LOCAL_RANK', rank)))
            dev.reset(seeds=np.uint64(seed0) + np.arange(lo, hi, dtype=np.uint64))
            for i in range(n):
                dev.restore(i, snap[i])
            D.init_rccl(dev, rank, world, plane)
        assert dev.exchange_info()['in_kernel'] == inkernel
        for K in (20, 41, 12):
            if inkernel:
                dev.gather_log(K)
            # the action key is seed0 + the shard's first global episode: episode e of the batch draws from key seed0 + e on any sharding
            dev.run_random_steps(seed0 + lo, t, K, timed=False)
            want = []
            for k in range(K):
                o = ora.step_batch(random_actions(seed0, t + k, total))
                want.append(pack(o[0].astype(np.uint8)))
            t += K
            dev.synchronize()
            if inkernel:
                got = dev.get_gather_log(world, 0, K)            # [K, world * n, 148]: rank-major rows = the unsharded batch's order
                for k in range(K):
                    assert np.array_equal(got[k], want[k]), f'rank {rank}: gathered rows of step {t - K + k} differ from the unsharded batch'
            last = D.allgathered_obs_host(dev, world)
            assert np.array_equal(last, o[0].astype(np.uint8)), f'rank {rank}: last gathered step differs (in-kernel exchange: {inkernel})'
            dev._fetch()
            assert np.array_equal(dev._obs, o[0][lo:hi]) and np.array_equal(dev._rew, o[1][lo:hi])
        assert dev.exchange_info()['watchdog_timeouts'] == 0
    plane.barrier()
    if rank == 0:
        print('RCCL_OK', world, [i['pci'] for i in infos])
    dev.close()
    plane.close()


if __name__ == '__main__':
    main()
