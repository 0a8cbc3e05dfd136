#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: aggregate agent-env steps/s (5 blue agents x envs) of the CC4 step engine.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch: device-side uniform random blue actions -> k_step over the
rank's shard (reset / full SimulationController.step transition / reward / flat observations) [-> RCCL all-gather of
the observations when N>1].  Workload at every N: BASELINE configs[1] per GPU = 1024 concurrent episodes per GPU
(weak scaling: 8 GPUs = configs[3], 8192 episodes), EnterpriseScenarioGenerator(steps=500), FiniteStateRedAgent red,
EnterpriseGreenAgent green, autoreset on done so the timed region includes the scenario regeneration of finished
episodes.  Inputs (state, actions) are resident in HBM.  RNG mode: BASELINE.md section 3 quotes the GPU runs in the
counter-based Philox mode (lane-parallel kernel k_step_philox; bit-exact with the CPU oracle, distribution-checked
against the PCG mode); the numpy-PCG64 mode that is bit-exact with the reference itself (serial kernel k_step) is
measured in the same run and reported under "alt_rng".

Prints ONE JSON line on rank 0.  `roofline` = algorithmic bytes per k_step launch / mean launch duration from HIP
events recorded on the launch stream inside the timed region; `cpu_baseline` = the CPU oracle (kind "port": the
host build of the same restatement, OpenMP over episodes) on a bounded sample of the same workload, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on these hosts needs dmabuf IPC (the driver has no legacy IPC): set before any HIP / RCCL library loads
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(envs, seed0, budget_s=12.0):
    """Oracle (oracle/liboracle.so) on the host cores: same seeds, same action generator, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import ctypes
    import numpy as np
    from oracle_binding import OracleVecEnv, random_actions
    ora = OracleVecEnv(envs, steps=500)
    ora.reset(seeds=seed0)
    cores = int(ora.lib.cc4o_num_threads())
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:  # cgroup v2 CPU quota of the container ("max" or "<quota> <period>")
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            cores = max(1, min(cores, int(int(q) / int(p))))
    except Exception:
        pass
    ora.lib.cc4o_set_threads(cores)

    def run(t0, k):
        acts = [np.ascontiguousarray(random_actions(seed0, t0 + i, envs)) for i in range(k)]
        t = time.perf_counter()
        for a in acts:
            ora.lib.cc4o_step_all(ora._h, a.ctypes.data_as(ctypes.c_void_p))
        return time.perf_counter() - t

    probe = run(0, 10)
    k = int(max(10, min(480, budget_s / max(probe / 10, 1e-6))))
    dt = run(10, k)
    ora.close()
    return {'value': 5.0 * envs * k / dt, 'unit': 'agent-env steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'{envs} episodes x {k} steps (steps 10..{10 + k} of the same seeded workload), OpenMP over episodes'}


def load_pmc_traffic():
    """HBM bytes per k_step launch from the committed rocprofv3 --pmc passes (profiles/r01_pmc.json), or None."""
    p = os.path.join(ROOT, 'profiles', 'r01_pmc.json')
    try:
        with open(p) as f:
            return json.load(f).get('hbm_bytes_per_launch_1024env')
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1500)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--envs-per-gpu', type=int, default=1024)
    ap.add_argument('--episode-steps', type=int, default=500)
    ap.add_argument('--rng', choices=['pcg64', 'philox'], default='philox')
    ap.add_argument('--no-alt', action='store_true', help='skip the second measurement in the other RNG mode')
    ap.add_argument('--seed0', type=int, default=1000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from cage_challenge_4_amd import CC4VecEnv, RNG_PCG64, RNG_PHILOX
    from cage_challenge_4_amd import distributed as D

    rank, world, local = D.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})')
    dist_on = world > 1 or os.environ.get('CC4_BENCH_FORCE_DIST') == '1'   # the env var drives the N>1 code path at world 1
    if dist_on:
        # gloo / RCCL print banners on fd 1 from C++; keep stdout for the one JSON line
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        D.init_control_plane('gloo', force=True)
        import torch
        import torch.distributed as dist
    dev_id = local if world > 1 else 0
    if dist_on and world > 1:
        ndev = torch.cuda.device_count()      # a launcher may expose one GPU per rank (then it is device 0) or all of them
        if ndev > 0:
            dev_id = local % ndev
    dev_id = int(os.environ.get('CC4_BENCH_DEVICE', dev_id))   # override: several ranks on one GPU (tests the N>1 plumbing on a 1-GPU box)
    n_local = args.envs_per_gpu
    total_envs = n_local * world
    env = CC4VecEnv(n_local, steps=args.episode_steps, rng_mode=RNG_PCG64 if args.rng == 'pcg64' else RNG_PHILOX,
                    device_id=dev_id, autoreset=True)
    lo = rank * n_local
    import numpy as np
    env.reset(seeds=np.uint64(args.seed0) + np.arange(lo, lo + n_local, dtype=np.uint64))
    exchange_note = None
    if dist_on:
        # RCCL setup + first collective under a watchdog: if any rank fails or stalls, every rank drops the exchange and
        # the run is reported with "exchange": "none (...)" instead of dying without a number
        import threading
        res = {}

        def _setup():
            try:
                if os.environ.get('CC4_RCCL_SETUP_FAIL') == '1':      # exercises the fallback below
                    raise RuntimeError('CC4_RCCL_SETUP_FAIL=1')
                D.init_rccl(env, rank, world)
                env.run_random_steps(args.seed0 + lo, 0, 1, timed=False)   # first collective (RCCL prints its banner lazily)
                env.synchronize()
                res['ok'] = True
            except Exception as ex:      # noqa: BLE001 - reported below
                res['err'] = repr(ex)
        th = threading.Thread(target=_setup, daemon=True)
        th.start()
        th.join(float(os.environ.get('CC4_RCCL_SETUP_TIMEOUT', '240')))
        ok = torch.tensor([1 if res.get('ok') else 0], dtype=torch.int32)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0:
            exchange_note = 'none (RCCL setup failed or timed out on a rank: %s)' % res.get('err', 'ok here' if res.get('ok') else 'timeout')
            print('bench.py: ' + exchange_note, file=sys.stderr)
            env.close = lambda: None                                                # the old handle is abandoned, not destroyed
            env = CC4VecEnv(n_local, steps=args.episode_steps, rng_mode=RNG_PCG64 if args.rng == 'pcg64' else RNG_PHILOX,
                            device_id=dev_id, autoreset=True)
        env.reset(seeds=np.uint64(args.seed0) + np.arange(lo, lo + n_local, dtype=np.uint64))
        import ctypes
        ctypes.CDLL(None).fflush(None)           # C stdio buffers written while fd 1 pointed at stderr
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    seed_actions = args.seed0 + lo            # action key = seed0 + global episode index

    def timed_run(e):
        e.run_random_steps(seed_actions, 0, args.warmup, timed=False)
        e.synchronize()
        if dist_on:
            dist.barrier()
        t0 = time.perf_counter()
        ms_k = e.run_random_steps(seed_actions, args.warmup, args.steps, timed=True)   # syncs the stream at the end
        e.synchronize()
        d = time.perf_counter() - t0      # this rank's K steps; the slowest rank decides (MAX below), the control-plane
        if dist_on:                       # barrier that closes the bracket is not part of the steps
            dist.barrier()
        if dist_on:
            t = torch.tensor([d, ms_k], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d, ms_k = float(t[0]), float(t[1])
        return d, ms_k

    dt, ms_kernels = timed_run(env)
    env._fetch()
    err_any = bool(env.err.any())

    alt = None
    if not args.no_alt and not dist_on:      # single-GPU runs also time the other RNG mode
        other = 'pcg64' if args.rng == 'philox' else 'philox'
        env2 = CC4VecEnv(n_local, steps=args.episode_steps, rng_mode=RNG_PCG64 if other == 'pcg64' else RNG_PHILOX,
                         device_id=dev_id, autoreset=True)
        env2.reset(seeds=np.uint64(args.seed0) + np.arange(lo, lo + n_local, dtype=np.uint64))
        if world > 1:
            D.init_rccl(env2, rank, world)
        dt2, ms2 = timed_run(env2)
        alt = {'rng': other, 'kernel': 'k_step' if other == 'pcg64' else 'k_step_philox',
               'value': 5.0 * total_envs * args.steps / dt2, 'unit': 'agent-env steps/s', 'ms_per_step': dt2 / args.steps * 1e3,
               'launch_ms': ms2 / args.steps,
               'note': 'pcg64 = numpy Generator(PCG64) stream, bit-exact with the reference under the same seed' if other == 'pcg64'
                       else 'philox = counter-based streams per (agent, phase, step, episode)'}
        env2.close()
    uni = None
    if not args.no_alt and not dist_on and args.rng == 'philox':
        # BASELINE configs 2-4 vs 5: the same workload with ONE topology shared by all episodes (dynamics still keyed per
        # episode); the headline run above randomises the topology per episode and per reset, as the reference does
        env3 = CC4VecEnv(n_local, steps=args.episode_steps, rng_mode=RNG_PHILOX, device_id=0, autoreset=True, topology_seed=args.seed0)
        env3.reset(seeds=np.uint64(args.seed0) + np.arange(lo, lo + n_local, dtype=np.uint64))
        dt3, ms3 = timed_run(env3)
        uni = {'value': 5.0 * total_envs * args.steps / dt3, 'unit': 'agent-env steps/s', 'ms_per_step': dt3 / args.steps * 1e3,
               'launch_ms': ms3 / args.steps, 'note': 'uniform topology: every episode draws its scenario from one shared key (cc4_config.topology_seed)'}
        env3.close()
    if rank == 0:
        bytes_per_env = int(env.lib.cc4_algorithmic_bytes_per_env_step())
        launch_ms = ms_kernels / args.steps
        achieved = bytes_per_env * n_local / (launch_ms * 1e-3) / 1e9
        traffic = load_pmc_traffic() if n_local == 1024 else None
        out = {
            'metric': 'agent-env steps/sec (5 blue agents x N envs)',
            'value': 5.0 * total_envs * args.steps / dt,
            'unit': 'agent-env steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'int32', 'data': 'synthetic',
            'config': {
                'workload': f'{n_local} vectorised envs per GPU ({total_envs} total), uniform random blue actions '
                            f'(82/82/82/82/242 incl. invalid slots), FiniteStateRedAgent, EnterpriseGreenAgent, '
                            f'EnterpriseScenarioGenerator(steps={args.episode_steps}), autoreset incl. scenario regeneration, topology randomised per episode and reset',
                'envs_per_gpu': n_local, 'total_envs': total_envs, 'rng': args.rng,
                'exchange': exchange_note or ('RCCL all-gather of the observations of every step (2 bits per value, 148 B per episode) on a second stream, overlapped with the next step' if dist_on else 'none'),
                'env_steps_per_sec': total_envs * args.steps / dt, 'engine_error_flags': err_any,
            },
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'kernel': 'k_step', 'launch_ms': launch_ms, 'algorithmic_bytes_per_launch': bytes_per_env * n_local},
        }
        out['roofline']['kernel'] = 'k_step_philox' if args.rng == 'philox' else 'k_step'
        if alt is not None:
            out['alt_rng'] = alt
        if uni is not None:
            out['uniform_topology'] = uni
        if not dist_on and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(n_local, args.seed0)
        print(json.dumps(out), flush=True)
    env.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if exchange_note:                 # a stalled RCCL setup thread must not keep the process alive
        sys.stdout.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
