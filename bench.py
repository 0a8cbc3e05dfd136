#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: aggregate agent-env steps/s (5 blue agents x envs) of the CC4 step engine.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch: device-side uniform random blue actions -> k_step over the
rank's shard (reset / full SimulationController.step transition / reward / flat observations) [-> RCCL all-gather of
the observations when N>1].  Workload at every N: **8192 concurrent episodes in total** -- BASELINE configs[2] on one
GPU (the configuration the metric's target is quoted on), configs[3] on eight (1024 per GPU): strong scaling.
EnterpriseScenarioGenerator(steps=500), FiniteStateRedAgent red, EnterpriseGreenAgent green, topology randomised per
episode and per reset, autoreset on done.  Inputs (state, actions) are resident in HBM.  RNG mode: the counter-based
Philox mode (agents resolved side by side: kernel k_step_philox1, one wavefront per episode, for batches of more than eight
episodes per CU, k_step_philox, four per episode, for smaller ones -- cc4_create picks, `roofline.kernel` names it;
bit-exact with the CPU oracle, distribution-checked against the PCG mode); the numpy-PCG64 mode that is bit-exact with
the reference itself (kernel k_step) is measured in the same run and reported under "alt_rng", and the 1024-episode batch
(configs[1], the per-GPU share of an 8-GPU job) under "envs_1024".

Timing: W untimed warm-up steps, then regions of EXACTLY K steps, each bracketed by barrier + device synchronise on
both sides, repeated back to back (the episodes keep running across regions) until at least --min-seconds of timed
work has accumulated, so that a small K still averages over whole episodes including their scenario regenerations.
Region time = MAX over ranks.  `value` = steps of all regions / sum of the region times; the median region is
reported beside it.

Prints ONE JSON line on rank 0.  `roofline` = algorithmic bytes per k_step launch / mean launch duration from HIP
events recorded on the launch stream inside the timed regions; `cpu_baseline` = the CPU oracle (kind "port": the
host build of the same restatement) on a bounded sample of the same workload, rank 0, N=1 only.
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
TOTAL_ENVS = 8192           # BASELINE configs[2] / configs[3]
# reference Python on one core (survey container, Intel Xeon 2.10 GHz, CPython 3.10): a stated
# constant -- the reference never travels to the GPU box
REFERENCE_PYTHON = {'value': 172.0, 'unit': 'agent-env steps/s', 'cores': 1, 'kind': 'reference',
                    'sample': 'BASELINE.md section 2: EnterpriseScenarioGenerator seed 123, 500 steps, measured in the build container, not on this box'}


def plan_shard(total_envs, rank, world):
    """This rank's contiguous share [lo, hi) of the global episode batch (cage_challenge_4_amd.vec_env.shard_range)."""
    from cage_challenge_4_amd.vec_env import shard_range
    return shard_range(total_envs, rank, world)


def timed_regions(run_k, steps, warmup, min_seconds, sync, barrier=None, reduce_max=None, max_regions=4000):
    """The timing protocol of this bench (also driven by tests/_gloo_worker.py with a CPU step function).

    run_k(t0, k, timed) runs k steps starting at action-time t0 and returns the on-stream kernel milliseconds (or 0.0).
    Returns (region_seconds, region_kernel_ms): one entry per K-step region, each already MAX-reduced over ranks."""
    barrier = barrier or (lambda: None)
    reduce_max = reduce_max or (lambda v: v)
    run_k(0, warmup, False)
    sync()
    t = warmup
    secs, kms = [], []
    while True:
        barrier()
        sync()
        t0 = time.perf_counter()
        ms = run_k(t, steps, True)
        sync()
        d = time.perf_counter() - t0
        barrier()
        d, ms = reduce_max([d, ms])
        secs.append(d)
        kms.append(ms)
        t += steps
        if sum(secs) >= min_seconds or len(secs) >= max_regions:      # `secs` is rank-reduced: every rank stops together
            return secs, kms


def summarise(secs, kms, steps, total_envs):
    import statistics
    n = len(secs)
    tot = sum(secs)
    med = statistics.median(secs)
    return {'value': 5.0 * total_envs * steps * n / tot, 'ms_per_step': tot / (n * steps) * 1e3,
            'ms_per_step_median_region': med / steps * 1e3, 'regions': n,
            'region_spread': (max(secs) - min(secs)) / med if n > 1 else 0.0,
            # the five longest regions as (index, seconds / median): where a short-region run loses time (episodes regenerate in one region
            # out of steps-per-episode / K; anything else that long is the host's or the box's)
            'slowest_regions': [[i, round(secs[i] / med, 2)] for i in sorted(range(n), key=lambda i: -secs[i])[:5]],
            'launch_ms': sum(kms) / (n * steps)}


def cpu_baseline(envs, seed0, budget_s=10.0):
    """Oracle (oracle/liboracle.so) on the host cores: same seeds, same action generator, bounded sample; all cores
    (OpenMP over episodes) and one core."""
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

    def run(t0, k):
        acts = [np.ascontiguousarray(random_actions(seed0, t0 + i, envs)) for i in range(k)]
        t = time.perf_counter()
        for a in acts:
            ora.lib.cc4o_step_all(ora._h, a.ctypes.data_as(ctypes.c_void_p))
        return time.perf_counter() - t

    ora.lib.cc4o_set_threads(cores)
    probe = run(0, 10)
    k = int(max(10, min(400, budget_s / max(probe / 10, 1e-6))))
    dt = run(10, k)
    out = {'value': 5.0 * envs * k / dt, 'unit': 'agent-env steps/s', 'cores': cores, 'kind': 'port', 'cpu': cpu_model(),
           'sample': f'{envs} episodes x {k} steps (steps 10..{10 + k} of the same seeded workload: the first {envs} episodes of the headline batch -- episodes '
                     f'are independent, so the rate per episode is the 8192-episode rate), OpenMP over episodes on {cores} threads of: {cpu_model()}'}
    # one core: the rest of the same episodes' life, a shorter stretch
    ora.lib.cc4o_set_threads(1)
    t1 = 10 + k
    probe1 = run(t1, 2)
    k1 = int(max(2, min(489 - t1 - 2, (budget_s / 2) / max(probe1 / 2, 1e-6))))
    if k1 >= 2:
        dt1 = run(t1 + 2, k1)
        out['one_core'] = {'value': 5.0 * envs * k1 / dt1, 'unit': 'agent-env steps/s', 'cores': 1, 'kind': 'port',
                           'sample': f'{envs} episodes x {k1} steps (steps {t1 + 2}..{t1 + 2 + k1}), one thread'}
    out['reference_python'] = REFERENCE_PYTHON
    ora.close()
    return out


def host_api_rates(seed=123, steps=300, eval_eps=3):
    """The host-API path beside the reference's own single-environment rate (34 env-steps/s = 172 agent-env
    steps/s in reference Python): one episode stepped through the drop-in wrapper -- BlueFlatWrapper(CybORG(...)).step(dict of five
    action indices) -> dict observations: one launch, one host synchronisation and four small copies per step -- and the
    reference's evaluation loop (run_evaluation, mode 'sequential': one episode at a time, a Python agent call per blue agent)."""
    import numpy as np
    from cage_challenge_4_amd import (CybORG, EnterpriseScenarioGenerator, SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent,
                                      BlueFlatWrapper)
    from cage_challenge_4_amd.evaluation import run_evaluation
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent,
                                     red_agent_class=FiniteStateRedAgent, steps=500)
    env = BlueFlatWrapper(CybORG(sg, seed=seed))
    obs, info = env.reset()
    rng = np.random.default_rng(seed)
    valid = {a: np.nonzero(info[a]['action_mask'])[0] for a in env.agents}
    for _ in range(20):
        env.step({a: int(valid[a][rng.integers(len(valid[a]))]) for a in env.possible_agents})
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step({a: int(valid[a][rng.integers(len(valid[a]))]) for a in env.possible_agents})
    dt = time.perf_counter() - t0
    env.close()
    out = {'single_env_facade': {'value': 5.0 * steps / dt, 'unit': 'agent-env steps/s', 'env_steps_per_sec': steps / dt, 'us_per_step': dt / steps * 1e6,
                                 'sample': f'BlueFlatWrapper(CybORG(EnterpriseScenarioGenerator(steps=500), seed={seed})).step x {steps}, random valid actions, numpy-stream mode (bit-exact with the reference)',
                                 'reference_python': REFERENCE_PYTHON['value']}}

    class Agent:
        def __init__(self, k):
            self.sleep = 145 if k == 4 else 49

        def get_action(self, obs, action_space):
            return self.sleep

    class Submission:
        NAME, TEAM, TECHNIQUE = 'bench', 'cc4-amd', 'sleep'
        AGENTS = {f'blue_agent_{k}': Agent(k) for k in range(5)}

        @staticmethod
        def wrap(e):
            return BlueFlatWrapper(e)
    import contextlib
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):        # the harness prints its summary like the reference's; stdout is for the one JSON line
        scores = run_evaluation(Submission, None, max_eps=eval_eps, seed=seed, mode='sequential', write_to_file=False)
    dt = time.perf_counter() - t0
    n_steps = eval_eps * 499          # the step that raises `done` ends the episode (evaluation.py:108-110)
    out['eval_sequential'] = {'value': 5.0 * n_steps / dt, 'unit': 'agent-env steps/s', 'env_steps_per_sec': n_steps / dt, 'seconds': dt,
                              'sample': f"run_evaluation(mode='sequential', max_eps={eval_eps}): {eval_eps} x 500-step episodes, Sleep blue agents, reference's per-episode protocol",
                              'mean_score': float(sum(scores) / len(scores)), 'reference_python': REFERENCE_PYTHON['value']}
    return out


def load_pmc_traffic(envs, kernel):
    """HBM bytes per STEP from the committed rocprofv3 --pmc passes of this bench command (profiles/r06_pmc.json, tools/profile_round.sh),
    with their source -- only if they were taken on the very kernel the timed regions launched (`kernel`: the one-launch forms are
    profiled as what they are: bytes of a launch / steps of the launch); else (None, None)."""
    for name in ('r06_pmc.json',):
        p = os.path.join(ROOT, 'profiles', name)
        try:
            with open(p) as f:
                d = json.load(f)
            v = d.get(f'hbm_bytes_per_step_{envs}env')
            if v is None or d.get(f'kernel_{envs}env') != kernel:      # never another kernel's figure
                continue
            spl = d.get(f'steps_per_launch_{envs}env')
            how = (f'one launch = {spl} steps; bytes of the launch / {spl}' if spl else f"{d.get(f'launches_per_step_{envs}env')} serialised launches per step")
            return v, f'profiles/{name} (rocprofv3 --pmc passes of this bench command on {kernel}: {how}; a committed figure, not measured in this run)'
        except Exception:
            pass
    return None, None


# "valu_issue" when the committed issue-slot passes of the timed kernel show the vector ALUs at least this busy over the WHOLE launch (the one-launch kernel:
# 82-86 % in 100-step launches, 77 % in the driver's 20-step launches, whose last round idles a sixth of the slots; the latency-bound small batches: 34 %)
VALU_BOUND_AT = 0.7
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 4.0      # wave64 VALU instructions per ns the chip can issue: 256 CUs x 4 SIMDs, one per 4 cycles, 2.4 GHz = 614.4 G/s


def load_valu_issue(envs, kernel, steps_per_launch):
    """VALU issue-slot use of the timed kernel from the committed rocprofv3 --pmc passes (profiles/r06_pmc_valu_busy.json, tools/valu_busy.sh): the entry
    of this very kernel at this batch size whose launch length is nearest; None if there is none."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r06_pmc_valu_busy.json')) as f:
            d = json.load(f)
        c = [v for v in d.values() if isinstance(v, dict) and v.get('kernel') == kernel and v.get('valu_busy')]
        same = [v for v in c if v.get('episodes') == envs]
        if not same and envs >= 8192:
            # the one-launch kernel was counted at 8192 and 32768 episodes (every residency slot of the chip busy in both, the same busy fraction):
            # a batch in between or above takes the nearer one's figures, and says so
            near = min((v['episodes'] for v in c if v['episodes'] >= 8192), key=lambda n: abs(n - envs), default=None)
            same = [v for v in c if v.get('episodes') == near]
        if not same:
            return None
        v = min(same, key=lambda v: abs(v['steps_per_launch'] - steps_per_launch))
        return {'valu_busy': v['valu_busy'], 'salu_busy': v.get('salu_busy'), 'lds_busy': v.get('lds_busy'),
                'valu_instructions_per_episode_step': v.get('valu_per_episode_step'), 'active_lanes_per_valu_instruction': v.get('active_lanes_per_valu_instruction'),
                'steps_per_launch_of_the_counter_pass': v['steps_per_launch'], 'episodes_of_the_counter_pass': v['episodes'],
                'source': 'profiles/r06_pmc_valu_busy.json (rocprofv3 --pmc passes of this kernel: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles of the launch); '
                          'a committed figure, not measured in this run)'}
    except Exception:
        return None


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown CPU'


def exchange_world1(make_env, measure, D, args, lo, timeout_s=90.0):
    """The same timed regions with the exchange ON, on a one-rank RCCL communicator: the configuration every rank of
    BASELINE configs[3] runs -- 1024 episodes -- and the headline batch.  With a communicator cc4_run_random_steps stays one launch: the
    communication stream follows the kernel's per-step counters and all-gathers every step's packed rows (include/cc4.h
    cc4_exchange_info).  RCCL's first initialisation on a fresh box pages in a 570 MB library (seconds to minutes): the whole leg runs
    under a watchdog and reports 'skipped' instead of holding the line up."""
    import threading
    import numpy as np
    from cage_challenge_4_amd import RNG_PHILOX
    out, box = {}, {}
    sub_seconds = max(0.2, args.min_seconds / 5.0)

    def leg():
        try:
            for n in (1024, 8192):
                e = make_env(n, RNG_PHILOX, lo if n == 8192 else 0)
                D.init_rccl(e, 0, 1)
                e.reset(seeds=np.uint64(args.seed0) + np.arange(n, dtype=np.uint64))
                r = measure(e, 0, n, min_seconds=sub_seconds)
                xi = e.exchange_info()
                hs = e.host_stats()
                r.update({'unit': 'agent-env steps/s', 'total_envs': n, 'exchange': xi, 'allgathers_issued': hs['gathers'],
                          'note': 'RCCL all-gather of every step\'s observations (2 bits per value) on a one-rank communicator, issued from the communication '
                                  'stream behind the one-launch kernel\'s per-step counters' if xi['in_kernel'] else 'per-step launches, an all-gather behind each'})
                out[f'envs_{n}'] = r
                e.close()
            box['ok'] = True
        except Exception as ex:      # noqa: BLE001
            box['err'] = repr(ex)
    # RCCL prints its banner on fd 1 from C++ (lazily, and flushes it when the process exits): stdout is for the one JSON line
    import ctypes
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    th = threading.Thread(target=leg, daemon=True)
    th.start()
    th.join(timeout_s)
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    os.dup2(saved_fd, 1)
    os.close(saved_fd)
    if not box.get('ok'):
        out['skipped'] = box.get('err') or f'RCCL setup did not finish within {timeout_s:.0f} s on this box (library paging in)'
        out['_stalled'] = th.is_alive()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--total-envs', type=int, default=TOTAL_ENVS)
    ap.add_argument('--envs-per-gpu', type=int, default=0, help='override: total = envs-per-gpu x gpus (weak-scaled run)')
    ap.add_argument('--episode-steps', type=int, default=500)
    ap.add_argument('--min-seconds', type=float, default=2.0, help='timed regions of the headline are repeated until this much timed work has accumulated (the sub-entries take a fifth of it each, 0.2 s at least)')
    ap.add_argument('--rng', choices=['pcg64', 'philox'], default='philox')
    ap.add_argument('--no-alt', action='store_true', help='skip the sub-entries (other RNG mode, 1024 episodes, uniform topology)')
    ap.add_argument('--seed0', type=int, default=1000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from cage_challenge_4_amd import CC4VecEnv, RNG_PCG64, RNG_PHILOX
    from cage_challenge_4_amd import distributed as D
    import numpy as np

    rank, world, local = D.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})')
    dist_on = world > 1 or os.environ.get('CC4_BENCH_FORCE_DIST') == '1'   # the env var drives the N>1 code path at world 1
    plane = D.control_plane(force=dist_on)      # rendezvous / barriers / reductions: a shared directory, no PyTorch (distributed.py)
    if dist_on:
        # RCCL prints banners on fd 1 from C++; keep stdout for the one JSON line
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
    dev_id = local if world > 1 else 0
    if dist_on and world > 1:
        from cage_challenge_4_amd import _lib
        ndev = int(_lib.load().cc4_device_count())      # a launcher may expose one GPU per rank (then it is device 0) or all of them
        if ndev > 0:
            dev_id = local % ndev
    dev_id = int(os.environ.get('CC4_BENCH_DEVICE', dev_id))   # override: several ranks on one GPU (tests the N>1 plumbing on a 1-GPU box)
    total_envs = args.envs_per_gpu * world if args.envs_per_gpu else args.total_envs
    scaling = 'weak' if args.envs_per_gpu else 'strong'
    lo, hi = plan_shard(total_envs, rank, world)
    n_local = hi - lo
    mode = RNG_PCG64 if args.rng == 'pcg64' else RNG_PHILOX

    def make_env(n, rng_mode, first, **kw):
        e = CC4VecEnv(n, steps=args.episode_steps, rng_mode=rng_mode, device_id=dev_id, autoreset=True, strict=False, **kw)
        e.reset(seeds=np.uint64(args.seed0) + np.arange(first, first + n, dtype=np.uint64))
        return e

    env = make_env(n_local, mode, lo)
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
                D.init_rccl(env, rank, world, plane)
                env.run_random_steps(args.seed0 + lo, 0, 1, timed=False)   # first collective (RCCL prints its banner lazily)
                env.synchronize()
                res['ok'] = True
            except Exception as ex:      # noqa: BLE001 - reported below
                res['err'] = repr(ex)
        th = threading.Thread(target=_setup, daemon=True)
        th.start()
        # (a fresh box pages librccl.so in on first use: ncclCommInitRank has been seen to take 200-250 s there, seconds afterwards --
        # profiles/r03_exchange_streams_ab.txt; a fallback taken while the setup thread is still alive also runs beside its streams)
        th.join(float(os.environ.get('CC4_RCCL_SETUP_TIMEOUT', '900')))
        ok = plane.allreduce([1.0 if res.get('ok') else 0.0], 'min')
        if int(ok[0]) == 0:
            exchange_note = 'none (RCCL setup failed or timed out on a rank: %s)' % res.get('err', 'ok here' if res.get('ok') else 'timeout')
            print('bench.py: ' + exchange_note, file=sys.stderr)
            env.close = lambda: None                                                # the old handle is abandoned, not destroyed
            env = make_env(n_local, mode, lo)
        else:
            env.reset(seeds=np.uint64(args.seed0) + np.arange(lo, hi, dtype=np.uint64))
        import ctypes
        ctypes.CDLL(None).fflush(None)           # C stdio buffers written while fd 1 pointed at stderr
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)

    def reduce_max(v):
        return plane.allreduce(v, 'max')

    def measure(e, first, n_total, runner=None, min_seconds=None):
        key = args.seed0 + first            # action key = seed0 + global episode index
        run_kernel = e.run_kernel_for(args.steps)     # (asked before the timed regions: a persistent kernel's one-off census happens here)
        runner = runner or (lambda t0, k, timed: e.run_random_steps(key, t0, k, timed=timed))
        secs, kms = timed_regions(runner, args.steps, args.warmup,
                                  args.min_seconds if min_seconds is None else min_seconds, e.synchronize, plane.barrier, reduce_max)
        out = summarise(secs, kms, args.steps, n_total)
        t_end = args.warmup + len(secs) * args.steps        # launches so far; episodes regenerate on every (episode_steps)-th
        out['autoreset_launches_in_timed_regions'] = t_end // args.episode_steps - args.warmup // args.episode_steps
        out['launches_per_step'] = e.launches_per_step      # of the per-step kernel (`kernel`); see run_kernel
        # what cc4_run_random_steps launches: the step kernel once per step and group, or -- a batch the chip holds at once -- ONE launch
        # of the multi-step kernel per timed region (k_run_philox: every block loops over the steps of its episode, the row stays in LDS)
        out['run_kernel'] = run_kernel
        return out

    sub_seconds = max(0.2, args.min_seconds / 5.0)      # the sub-entries' share of timed work
    main_res = measure(env, lo, total_envs)
    # what this rank's host did (cc4_host_stats): a multi-GPU curve is explained by these -- the slowest rank's kernel period, the
    # host time per step spent enqueueing launches and all-gathers, and whether an all-gather ever held a step up
    hs = env.host_stats()
    per_rank = plane.gather_obj({'rank': rank, 'envs': n_local, 'kernel': env.step_kernel, 'launches_per_step': env.launches_per_step,
                                 'host_launch_us_per_step': hs['launch_us'] / max(hs['steps'], 1), 'host_allgather_enqueue_us_per_step': hs['gather_us'] / max(hs['steps'], 1),
                                 'allgathers_issued': hs['gathers'], 'steps_that_waited_for_an_allgather': hs['gather_stalls'],
                                 # RCCL's own view: a scaling line is only what it says if every rank reports nccl_comm_count == n_gpus and
                                 # the ranks sit on different devices (uuid / pci)
                                 **env.comm_info()})
    xinfo = env.exchange_info() if dist_on and not exchange_note else {}
    env._fetch()
    err_any = bool(env.err.any())
    # a sharding-independent digest of where the batch stands after the run (episodes are seeded and driven by their GLOBAL
    # index): the last step's rewards and done flags summed over all ranks -- equal for any world size at equal step counts
    digest = [float(env._rew.astype(np.float64).sum()), float(env._done.sum()), float(err_any)]
    if dist_on and world > 1:
        digest = plane.allreduce(digest, 'sum')
        err_any = digest[2] > 0
    mean_hosts = float(np.mean([int(env.topology(i)[27::2].sum()) for i in range(0, n_local, max(1, n_local // 64))]))

    subs = {}
    if dist_on and not args.envs_per_gpu and not exchange_note and os.environ.get('CC4_BENCH_NO_WEAK') != '1':
        # The same job WEAK-scaled in the same run: the strong line above cuts BASELINE configs[3]'s 8192 episodes into world shares --
        # 1024 per GPU at 8 GPUs, the latency regime of this engine --; here every GPU keeps the full single-GPU batch (8192 per rank,
        # world x 8192 in total) with the exchange of every step's observations across all ranks, which is the efficient way to use N GPUs.
        import threading
        wbox = {}

        def weak_leg():
            try:
                per = TOTAL_ENVS
                lo_w = rank * per
                ew = make_env(per, mode, lo_w)
                D.init_rccl(ew, rank, world, plane)
                ew.reset(seeds=np.uint64(args.seed0) + np.arange(lo_w, lo_w + per, dtype=np.uint64))
                rw = measure(ew, lo_w, per * world, min_seconds=max(0.2, args.min_seconds / 2.0))
                rw.update({'unit': 'agent-env steps/s', 'scaling': 'weak', 'envs_per_gpu': per, 'total_envs': per * world, 'kernel': ew.step_kernel,
                           'exchange_info': ew.exchange_info(), 'note': 'every rank keeps 8192 episodes; RCCL all-gather of every step\'s packed observations across all ranks'})
                ew.close()
                wbox['res'] = rw
            except Exception as ex:      # noqa: BLE001
                wbox['res'] = {'skipped': repr(ex)}
        # (under a watchdog: a rank that fails here must not take the strong line, which is already measured, down with it)
        sys.stdout.flush()
        saved_fd2 = os.dup(1)
        os.dup2(2, 1)
        th = threading.Thread(target=weak_leg, daemon=True)
        th.start()
        th.join(float(os.environ.get('CC4_BENCH_WEAK_TIMEOUT', '180')))
        import ctypes as _ct
        _ct.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(saved_fd2, 1)
        os.close(saved_fd2)
        subs['weak_scaled'] = wbox.get('res') or {'skipped': 'the weak-scaled leg did not finish within its time limit on this rank', '_stalled': True}
    if not args.no_alt and not dist_on:      # single-GPU runs also time the other RNG mode and the small batch
        other = 'pcg64' if args.rng == 'philox' else 'philox'
        e2 = make_env(n_local, RNG_PCG64 if other == 'pcg64' else RNG_PHILOX, lo)
        r2 = measure(e2, lo, total_envs, min_seconds=max(sub_seconds, args.min_seconds / 2.0))
        k2 = e2.step_kernel
        e2.close()
        r2.update({'rng': other, 'kernel': k2, 'unit': 'agent-env steps/s', 'total_envs': total_envs,
                   'note': 'pcg64 = numpy Generator(PCG64) stream, bit-exact with the reference under the same seed' if other == 'pcg64'
                           else 'philox = counter-based streams per (agent, phase, step, episode)'})
        hot_b, row_b = int(env.lib.cc4_hot_bytes()), int(env.lib.cc4_state_bytes())
        # contract bytes of the kernel that serves that mode: the numpy-stream kernel stages the agent part and visits the host table in place
        b2 = (2 * hot_b + (row_b - hot_b) + 4 * 578 + 29) if other == 'pcg64' else int(env.lib.cc4_algorithmic_bytes_per_env_step())
        a2 = b2 * n_local / (r2['launch_ms'] * 1e-3) / 1e9
        v2 = load_valu_issue(n_local, r2['run_kernel'], args.steps)
        r2['roofline'] = {'bound': 'valu_issue' if (v2 and v2['valu_busy'] >= VALU_BOUND_AT) else 'latency', 'valu_issue': v2, 'roofline': 'hbm', 'achieved': a2, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': a2 / HBM_PEAK_GBPS, 'traffic': None,
                          'kernel': r2['run_kernel'], 'step_ms': r2['launch_ms'], 'algorithmic_bytes_per_step': b2 * n_local,
                          'note': 'algorithmic bytes per step (contract figure of that kernel) / kernel time per step against the HBM peak; the walking lane of the '
                                  'numpy-stream kernel is a dependency chain, not a stream of bytes -- at 24 waves per CU the chains of the resident episodes fill the '
                                  'vector issue slots (valu_issue: twice the instructions of the counter-mode step per episode-step)'}
        subs['alt_rng'] = r2
        if total_envs != 1024:
            e4 = make_env(1024, mode, 0)
            r4 = measure(e4, 0, 1024, min_seconds=sub_seconds)
            k4 = e4.step_kernel
            e4.close()
            r4['roofline'] = ({'bound': 'latency', 'us_per_step': r4['launch_ms'] * 1e3, 'valu_issue': load_valu_issue(1024, r4['run_kernel'], args.steps),
                               'note': f"{r4['run_kernel']} keeps the episode's row in LDS from the first step of a launch to the last: the bytes of the contract "
                                       'figure do not move, so no HBM fraction is quoted -- the figure is the mean step time of an episode'}
                              if r4['run_kernel'] in ('k_run_philox', 'k_run_philox8') else None)
            r4.update({'rng': args.rng, 'kernel': k4, 'unit': 'agent-env steps/s', 'total_envs': 1024,
                       'note': 'BASELINE configs[1]: 1024 episodes on one GPU = the per-GPU share of the 8-GPU job (latency regime: 4 blocks per CU, one round); '
                               'without the exchange a timed region is ONE launch of the multi-step kernel (run_kernel k_run_philox): the batch advances at the '
                               'mean step time of its episodes, not at the slowest one\'s'})
            subs['envs_1024'] = r4
        if args.rng == 'philox':
            # BASELINE configs 2-4 vs 5: the same workload with ONE topology shared by all episodes (dynamics still keyed per
            # episode); the headline run above randomises the topology per episode and per reset, as the reference does
            e3 = make_env(n_local, RNG_PHILOX, lo, topology_seed=args.seed0)
            r3 = measure(e3, lo, total_envs, min_seconds=sub_seconds)
            e3.close()
            r3.update({'unit': 'agent-env steps/s', 'note': 'uniform topology: every episode draws its scenario from one shared key (cc4_config.topology_seed)'})
            subs['uniform_topology'] = r3
        if args.rng == 'philox':
            # What a learner's loop pays (VERDICT r04 missing #2): the actions are ON the device, written by a kernel of the caller's (here
            # k_random_actions standing in for a policy network's argmax), and cc4_step_device consumes them -- a launch of the step
            # kernel per step and episode group, no host synchronisation inside the region, observations readable after every step
            e5 = make_env(n_local, RNG_PHILOX, lo)
            # r06: the rollout form -- ONE launch of the persistent kernel per K-step region, the stand-in policy (k_random_actions' draws, per policy group)
            # on the handle's policy stream between gates and publishes (include/cc4.h cc4_rollout_begin)
            r7 = None
            if e5.run_kernel_for(args.steps) == 'k_run_philox1':
                try:
                    r7 = measure(e5, lo, total_envs, runner=lambda t0, k, timed: e5.run_rollout(k, 'random', args.seed0 + lo, t0, native=True), min_seconds=sub_seconds)
                    r7.update({'unit': 'agent-env steps/s', 'kernel': e5.step_kernel, 'run_kernel': 'k_run_philox1r', 'launches_per_step': 1.0 / args.steps,
                               'policy_ops_per_step': 8,
                               'note': 'cc4_rollout_begin .. cc4_rollout_end: one launch of k_run_philox1 per region; per step and policy group (4) two stream operations: '
                                       'cc4_rollout_sync (the last pass\'s publish + the gate on this group\'s packed observations of the last step) and the stand-in policy kernel -- the steps wait for the publishes'})
                except Exception as ex:      # noqa: BLE001
                    r7 = {'skipped': repr(ex)}
            r5 = measure(e5, lo, total_envs, runner=lambda t0, k, timed: e5.run_policy_steps(args.seed0 + lo, t0, k), min_seconds=sub_seconds)
            r5.update({'unit': 'agent-env steps/s', 'kernel': e5.step_kernel, 'run_kernel': e5.step_kernel, 'launches_per_step': 2,
                       'note': 'per step: k_random_actions -> the handle\'s device action buffer, then cc4_step_device; a policy over the WHOLE batch orders every '
                               'episode group behind it, so the library steps the batch with ONE launch on the main stream instead of a fork and a join across the '
                               'group streams per step (r05: 334-393 -> 583 M); `grouped` is the same policy applied per group'})
            r6 = measure(e5, lo, total_envs, runner=lambda t0, k, timed: e5.run_policy_steps_grouped(args.seed0 + lo, t0, k), min_seconds=sub_seconds)
            r6.update({'unit': 'agent-env steps/s', 'kernel': e5.step_kernel, 'run_kernel': e5.step_kernel, 'launches_per_step': 2 * e5.launches_per_step,
                       'note': 'the same with the policy applied per episode group on the group\'s own stream (cc4_group_info / cc4_step_group_device): a policy is '
                               'batch-independent, so nothing orders the groups against each other and their launches keep overlapping across steps'})
            r5['grouped'] = r6
            e5.close()
            if r7 is not None:
                # the rollout form (ONE launch of the persistent kernel per region) beside the launch-per-step forms: at this batch size the protocol of
                # gates and publishes costs more than the launches it saves -- the entry's own value stays the whole-batch launch-per-step rate
                r5['rollout'] = r7
            subs['policy_in_loop'] = r5
            subs['exchange_world1'] = exchange_world1(make_env, measure, D, args, lo)
    if rank == 0:
        bytes_per_env = int(env.lib.cc4_algorithmic_bytes_per_env_step())
        hot = int(env.lib.cc4_hot_bytes())
        state_bytes = int(env.lib.cc4_state_bytes())
        if args.rng == 'pcg64':   # k_step stages the agent part only; of the host table it touches the rows it visits
            bytes_per_env = 2 * hot + (state_bytes - hot) + 4 * 578 + 29
        launch_ms = main_res['launch_ms']
        # A step of a large batch is `lps` launches, one per group of episodes, on separate streams (include/cc4.h
        # cc4_launches_per_step): they run concurrently, so the bytes of all of them move within one launch duration.
        # launch_ms = average launch-to-launch period on the slowest group's stream (HIP events, cc4_run_random_steps) = the
        # kernel's average duration in rocprofv3 --stats; achieved = lps x algorithmic bytes of one launch / launch_ms
        lps = main_res['launches_per_step']
        # one-launch forms (run_kernel k_run_philox / k_run_philox8 / k_run_philox1m / k_run_philox1: cc4_run_random_steps issues the K steps of
        # a timed region as ONE launch): launch_ms from the library is that launch's duration / K = the kernel time per step; the launch
        # itself (what rocprofv3 --stats averages) lasts steps_per_launch times as long and moves steps_per_launch times the bytes
        one_launch = main_res['run_kernel'] != env.step_kernel
        spl = args.steps if one_launch else 1
        achieved = bytes_per_env * n_local / (launch_ms * 1e-3) / 1e9
        # live bytes: the agent part + the 64-byte rows of the hosts that exist in the episode (the grid has 137 positions)
        useful = 2 * (hot + 64.0 * mean_hosts) + 4 * 578 + 29
        traffic, traffic_src = load_pmc_traffic(n_local, main_res['run_kernel'])
        hbm_ctr_frac = (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None
        valu = load_valu_issue(n_local, main_res['run_kernel'], spl)
        if valu and valu.get('valu_instructions_per_episode_step'):
            # the committed instruction count of an episode-step against THIS run's kernel time: wave-instructions per ns the chip issued / can issue
            valu['achieved_ginstr_per_s'] = valu['valu_instructions_per_episode_step'] * n_local / (launch_ms * 1e-3) / 1e9
            valu['peak_ginstr_per_s'] = VALU_PEAK_GINSTR
            valu['frac'] = valu['achieved_ginstr_per_s'] / VALU_PEAK_GINSTR
        bound = 'hbm' if (hbm_ctr_frac or 0.0) >= 0.6 else 'valu_issue' if (valu and valu['valu_busy'] >= VALU_BOUND_AT) else 'latency'
        out = {
            'metric': 'agent-env steps/sec (5 blue agents x N envs)',
            'value': main_res['value'],
            'unit': 'agent-env steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': main_res['ms_per_step'],
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': 'int32', 'data': 'synthetic',
            'config': {
                'workload': f'{total_envs} vectorised envs in total ({n_local} per GPU), uniform random blue actions '
                            f'(82/82/82/82/242 incl. invalid slots), FiniteStateRedAgent, EnterpriseGreenAgent, '
                            f'EnterpriseScenarioGenerator(steps={args.episode_steps}), autoreset incl. scenario regeneration, topology randomised per episode and reset',
                'envs_per_gpu': n_local, 'total_envs': total_envs, 'rng': args.rng,
                'exchange': exchange_note or (('RCCL all-gather of the observations of every step (2 bits per value, 148 B per episode) on a second stream, '
                                               + ('issued behind the per-step counters of the one-launch kernel (a ring of 32 step slabs, gathered in chunks of 8; include/cc4.h cc4_exchange_info)'
                                                  if xinfo.get('in_kernel') else 'overlapped with the next step\'s launch')) if dist_on else 'none'),
                'exchange_info': xinfo if dist_on else None,
                'env_steps_per_sec': main_res['value'] / 5.0, 'engine_error_flags': err_any,
                'last_step_reward_sum': digest[0], 'last_step_done_count': digest[1],
                'steps_run': args.warmup + main_res['regions'] * args.steps,
                'timed_regions': main_res['regions'], 'region_steps': args.steps, 'ms_per_step_median_region': main_res['ms_per_step_median_region'],
                'region_spread': main_res['region_spread'], 'slowest_regions': main_res['slowest_regions'],
                'autoreset_in_timed_region': main_res['autoreset_launches_in_timed_regions'] > 0,
                'autoreset_launches_in_timed_regions': main_res['autoreset_launches_in_timed_regions'],
            },
            # `bound`: what the counters of THIS kernel say, not a constant.  The figure of merit stays the HBM roofline (integer / byte work, no MFMA):
            # `achieved` = contract bytes per step / kernel time per step.  But the memory system carries a fifth of 8 TB/s (`traffic`: inside a run of steps
            # an episode's agent part stays in LDS), while the issue-slot counters of the same kernel show the vector ALUs busy nine cycles in ten at 24 waves
            # per CU on ~5 active lanes per instruction: "valu_issue" -- the step time is (vector instructions of an episode-step) / (issue slots), and what
            # moves it now is a shorter instruction stream, not more waves.  "hbm" only if the HBM counters say so; "latency" where neither unit
            # is busy (the small batches: 1024 episodes keep the VALUs 38 % busy).
            'roofline': {'bound': bound, 'roofline': 'hbm',
                         'bound_evidence': ('HBM counters of this kernel: %.0f %% of peak' % (100.0 * hbm_ctr_frac) if traffic
                                            else 'no HBM counter pass of this kernel at this batch size is committed') +
                                           ('; VALU issue slots of this kernel: %.0f %% busy (%.0f wave instructions per episode-step on %.1f active lanes): the '
                                            'step is bound by the NUMBER of vector instructions of an episode, which more resident waves no longer hide'
                                            % (100.0 * valu['valu_busy'], valu['valu_instructions_per_episode_step'] or 0.0, valu['active_lanes_per_valu_instruction'] or 0.0)
                                            if valu else '; no issue-slot counter pass of this kernel at this batch size is committed') +
                                           '; waves per CU: profiles/kernel_resources.txt, DESIGN 3.4',
                         'valu_issue': valu,
                         'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS,
                         # the contract figure (the packed row in and out + the outputs, per episode-step) is what a step-per-launch schedule
                         # must move.  The one-launch kernel keeps an episode's agent part in LDS for a run of 4-8 steps and finds the host table in L2 /
                         # Infinity Cache, so at long regions the contract bytes per second pass the HBM peak: `frac` near or above 1 says the HBM roofline
                         # of the streaming schedule is no longer the one that binds -- `traffic` (what the HBM counters saw) and `valu_issue` (the roofline
                         # that does bind: vector issue slots) are the measured ones.
                         'frac_note': ('contract bytes per second (row in + out + outputs every step) against 8 TB/s; the one-launch kernel does not move them -- '
                                       'the agent part stays in LDS across a run of steps, the HBM counters see `traffic` = %s of the contract bytes -- so this '
                                       'fraction can pass 1; the roofline that binds is `valu_issue`' % (('%.2f x' % (traffic / (bytes_per_env * n_local))) if traffic else 'a fraction')
                                       if one_launch else None),
                         # the same algorithmic bytes against the WALL clock of the timed regions (ms_per_step: what `value` is made of) --
                         # below `frac` by what lies between and around the launches of a region
                         'frac_wall': bytes_per_env * n_local / (main_res['ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         # the bytes the HBM counters saw per step (traffic) against the same launch duration: what fraction of the
                         # 8 TB/s the memory system actually carried
                         'hbm_counter_frac': hbm_ctr_frac,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': main_res['run_kernel'], 'step_kernel': env.step_kernel, 'run_kernel': main_res['run_kernel'],
                         'step_ms': launch_ms, 'steps_per_launch': spl, 'launch_ms': launch_ms * spl,
                         'launches_per_step': (1.0 / spl) if one_launch else lps,
                         'episodes_per_launch': n_local if one_launch else n_local / lps,
                         'algorithmic_bytes_per_launch': bytes_per_env * n_local * spl if one_launch else bytes_per_env * n_local / lps,
                         'algorithmic_bytes_per_step': bytes_per_env * n_local,
                         'note': (f'a timed region of {spl} steps = ONE launch of {main_res["run_kernel"]} over all {n_local} episodes (the per-step kernel of this '
                                  f'handle is {env.step_kernel}); launch_ms = that launch, step_ms = launch_ms / steps_per_launch; '
                                  'achieved = algorithmic_bytes_per_launch / launch_ms; traffic is per step (PMC passes of this very kernel: bytes of a launch / steps of the launch)'
                                  if one_launch else
                                  f'a step = {lps} concurrent launch(es) of {env.step_kernel} on separate streams, {n_local / lps:.0f} episodes each; '
                                  'achieved = launches_per_step x algorithmic_bytes_per_launch / launch_ms; traffic is per step too'),
                         'useful_bytes_per_step': useful * n_local, 'useful_frac': useful * n_local / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         'useful_note': f'live bytes only: agent part {hot} B + 64 B x {mean_hosts:.1f} existing hosts (of 137 grid positions), in and out'},
        }
        out['config']['per_rank'] = per_rank
        out.update(subs)
        if not dist_on and not args.no_alt:
            out.update(host_api_rates())
        if not dist_on and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(4096, args.seed0)   # ~15 s of CPU work (16 cores x ~0.8 s + one core x ~1.5 s)
        print(json.dumps(out), flush=True)
    env.close()
    plane.close()
    if exchange_note or subs.get('exchange_world1', {}).get('_stalled') or subs.get('weak_scaled', {}).get('_stalled'):                 # a stalled RCCL setup thread must not keep the process alive
        sys.stdout.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
