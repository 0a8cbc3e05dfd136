"""Multi-GPU plumbing: one process per GPU, the env batch sharded statically across ranks (no data-path collective in
the step itself); the single exchange step is the all-gather of the flat observations, done natively over RCCL/xGMI
by libcc4 (cc4_allgather_obs).

The control plane -- rendezvous, the 128-byte RCCL unique id, barriers, the max-over-ranks timing reduction -- needs no
PyTorch: `control_plane()` returns a FilePlane, a rendezvous through a directory all ranks of the (single-node) job share,
keyed by the launcher's MASTER_PORT and process id.  A GlooPlane (torch.distributed, backend gloo) offers the same five
operations for the launch-compatibility tests and as a fall-back (CC4_CONTROL_PLANE=gloo)."""
import ctypes
import json
import os
import time
import numpy as np
from .vec_env import shard_range


class SoloPlane:
    """world size 1: nothing to exchange."""
    rank, world = 0, 1
    kind = 'solo'

    def exchange(self, payload):
        return [payload]

    def barrier(self):
        pass

    def allreduce(self, values, op):
        return [float(v) for v in values]

    def bcast_bytes(self, payload, src=0):
        return payload

    def gather_obj(self, obj):
        return [obj]

    def close(self):
        pass


class FilePlane(SoloPlane):
    """All ranks of a single-node job share a directory; one primitive -- `exchange`: every rank contributes a small byte string
    and gets everybody's, in rank order -- carries the barrier, the reductions, the broadcast and the gather.  Operation number q
    of rank r is the file `q_r` (written under a temporary name, then renamed: readers never see a partial file); a rank that has
    read all files of operation q knows every rank has finished operation q - 1, and removes its own file of that one."""
    kind = 'file'
    _instances = {}          # directory -> planes created on it by this process: each gets a file-name prefix of its own

    def __init__(self, rank, world, key, root=None, timeout=2000.0):   # longer than bench.py's RCCL setup watchdog (900 s)
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        root = root or os.environ.get('CC4_CONTROL_PLANE_DIR') or ('/dev/shm' if os.path.isdir('/dev/shm') else '/tmp')
        self.dir = os.path.join(root, f'cc4_plane_{key}')
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        # Two planes on one directory (a second handle's communicator, a helper that builds its own plane) must not read each
        # other's operations: plane number k of this process talks to plane number k of every other rank (all ranks run the same
        # program), under the prefix `p<k>_`.  Operation numbers restart at 0 per plane.
        self.inst = FilePlane._instances.get(self.dir, 0)
        FilePlane._instances[self.dir] = self.inst + 1
        self.seq = 0

    def _path(self, q, r):
        return os.path.join(self.dir, f'p{self.inst}_{q}_{r}')

    def exchange(self, payload):
        q = self.seq
        self.seq += 1
        mine = self._path(q, self.rank)
        with open(mine + '.tmp', 'wb') as f:
            f.write(payload)
        os.replace(mine + '.tmp', mine)
        out = []
        t0 = time.monotonic()
        for r in range(self.world):
            p = self._path(q, r)
            spins = 0
            while True:
                try:
                    with open(p, 'rb') as f:
                        out.append(f.read())
                    break
                except FileNotFoundError:
                    spins += 1
                    if spins > 2000:
                        time.sleep(0.0002)
                        if time.monotonic() - t0 > self.timeout:
                            raise TimeoutError(f'control plane: rank {r} did not reach operation {q} within {self.timeout:.0f} s ({self.dir})')
        if q >= 1:
            try:
                os.remove(self._path(q - 1, self.rank))
            except OSError:
                pass
        return out

    def barrier(self):
        self.exchange(b'')

    def allreduce(self, values, op):
        rows = [json.loads(b.decode()) for b in self.exchange(json.dumps([float(v) for v in values]).encode())]
        f = {'max': max, 'min': min, 'sum': sum}[op]
        return [float(f(r[i] for r in rows)) for i in range(len(values))]

    def bcast_bytes(self, payload, src=0):
        return self.exchange(payload if self.rank == src else b'')[src]

    def gather_obj(self, obj):
        return [json.loads(b.decode()) for b in self.exchange(json.dumps(obj).encode())]

    def close(self):
        """Final barrier, then the files go -- but a rank's file of the LAST operation may not be removed by that rank: a peer may
        still be about to read it (that race, lost, left the peer spinning in its last barrier until the timeout).  So every other
        rank reports `done_r` once it is through, and rank 0 -- after all of them did, or leaving everything in place after 10 s --
        removes the directory's files."""
        self.barrier()
        if self.rank != 0:
            mine = os.path.join(self.dir, f'p{self.inst}_done_{self.rank}')
            try:
                with open(mine + '.tmp', 'wb') as f:
                    f.write(b'1')
                os.replace(mine + '.tmp', mine)
            except OSError:
                pass
            return
        t0 = time.monotonic()
        want = [os.path.join(self.dir, f'p{self.inst}_done_{r}') for r in range(1, self.world)]
        while not all(os.path.exists(p) for p in want):
            if time.monotonic() - t0 > 10.0:
                return
            time.sleep(0.001)
        try:
            for name in os.listdir(self.dir):
                if not name.startswith(f'p{self.inst}_'):        # another plane of this job may still be at work in the directory
                    continue
                try:
                    os.remove(os.path.join(self.dir, name))
                except OSError:
                    pass
            os.rmdir(self.dir)                                    # succeeds once the last plane is gone
        except OSError:
            pass


class GlooPlane(SoloPlane):
    kind = 'gloo'

    def __init__(self, rank, world):
        import torch.distributed as dist
        self.rank, self.world = rank, world
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group(backend='gloo', rank=rank, world_size=world)

    def exchange(self, payload):
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, payload)
        return out

    def barrier(self):
        import torch.distributed as dist
        dist.barrier()

    def allreduce(self, values, op):
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        dist.all_reduce(t, op={'max': dist.ReduceOp.MAX, 'min': dist.ReduceOp.MIN, 'sum': dist.ReduceOp.SUM}[op])
        return [float(a) for a in t]

    def bcast_bytes(self, payload, src=0):
        return self.exchange(payload if self.rank == src else b'')[src]

    def gather_obj(self, obj):
        return self.exchange(obj)

    def close(self):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def _launcher_start_ticks():
    """Start time of the parent process (the launcher all workers of one launch are children of), in clock ticks since boot
    (/proc/<pid>/stat field 22): with the pid it names ONE launch -- a later launch that reuses the port and, by pid wrap-around,
    even the pid cannot find the files a crashed earlier one left behind."""
    try:
        with open(f'/proc/{os.getppid()}/stat') as f:
            return f.read().rsplit(')', 1)[1].split()[19]
    except (OSError, IndexError):
        return '0'


_PLANE = None


def control_plane(kind=None, force=False, fresh=False):
    """The job's control plane from the launcher's environment (RANK, WORLD_SIZE, MASTER_PORT; torch.distributed.run sets
    them).  kind: 'file' (default; no PyTorch) or 'gloo' (CC4_CONTROL_PLANE overrides).  force: a one-rank FilePlane instead of
    the SoloPlane (exercises the N>1 code path at world size 1).  ONE plane per process: later calls return the same object
    (a second FilePlane on the same directory would start its operation numbers at 0 again) -- an argument-less call returns
    whatever plane exists; a call that names a kind (or force=True) raises if the existing plane is of another kind; fresh=True
    builds another one, which gets a file-name prefix of its own."""
    global _PLANE
    rank, world, _ = env_rank_world()
    asked = kind is not None or force or 'CC4_CONTROL_PLANE' in os.environ      # an argument-less call asks for "the plane this process runs"
    kind = os.environ.get('CC4_CONTROL_PLANE', kind or 'file')
    want = 'solo' if (world == 1 and not force) else kind
    if _PLANE is not None and not fresh:
        # FilePlane and GlooPlane both derive from SoloPlane: classify by the exact type, most derived first
        have = 'gloo' if isinstance(_PLANE, GlooPlane) else ('file' if isinstance(_PLANE, FilePlane) else 'solo')
        if asked and have != want:      # e.g. a SoloPlane cached by an earlier call, then force=True: never hand back a plane of another kind silently
            raise RuntimeError(f'control_plane(): this process already runs a {have!r} plane, a {want!r} plane was asked for '
                               '(pass fresh=True for a second plane)')
        return _PLANE
    if want == 'solo':
        plane = SoloPlane()
    elif kind == 'gloo':
        plane = GlooPlane(rank, world)
    else:
        # every worker of one launch is a child of the same launcher process: its pid and start time tell launches that reuse a port apart
        key = os.environ.get('CC4_CONTROL_PLANE_KEY') or (f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_"
                                                           f"{os.getppid()}_{_launcher_start_ticks()}")
        plane = FilePlane(rank, world, key)
    if not fresh:
        _PLANE = plane
    return plane


def env_rank_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def init_control_plane(backend='gloo', force=False):
    """Initialise torch.distributed from the torchrun environment (MASTER_ADDR/PORT, RANK, WORLD_SIZE)."""
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_seeds(seed0, num_envs_total, rank, world):
    """Episode e of the global batch is always seeded seed0 + e, whatever the sharding."""
    lo, hi = shard_range(num_envs_total, rank, world)
    return lo, hi, np.uint64(seed0) + np.arange(lo, hi, dtype=np.uint64)


def allgather_host(local, world):
    """Concatenate equally-shaped numpy shards in rank order through the control plane (CPU tests, small metadata)."""
    if world == 1:
        return local.copy()
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(local))
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o.numpy() for o in out], axis=0)


OBS_PACKED_BYTES = 148   # CC4_OBS_PACKED_BYTES: value i of an episode = (row[i >> 2] >> (2 * (i & 3))) & 3


def unpack_obs(rows):
    """[M, OBS_PACKED_BYTES] uint8 packed rows -> [M, 578] uint8 values."""
    rows = np.asarray(rows, np.uint8)
    i = np.arange(578)
    return (rows[:, i >> 2] >> (2 * (i & 3)).astype(np.uint8)) & 3


def init_rccl(vec_env, rank, world, plane=None):
    """Create the RCCL communicator inside libcc4 for this rank's handle (the 128-byte unique id travels over the control plane)."""
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')   # single-node job: bootstrap over loopback, no NIC probing
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (normally exported already; read when the runtime loads)
    ident = b''
    if rank == 0:
        buf = (ctypes.c_uint8 * 128)()
        rc = vec_env.lib.cc4_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError('cc4_comm_unique_id failed')
        ident = bytes(buf)
    if world > 1:
        plane = plane or control_plane()          # the process-wide plane (never a second FilePlane with operation numbers of its own)
        ident = plane.bcast_bytes(ident, src=0)
    assert len(ident) == 128
    raw = (ctypes.c_uint8 * 128).from_buffer_copy(ident)
    vec_env._chk(vec_env.lib.cc4_comm_init(vec_env._h, rank, world, raw), 'cc4_comm_init')


def allgather_obs_device(vec_env):
    """Enqueue the RCCL all-gather of this rank's observations of the latest step -- packed 2 bits per value, OBS_PACKED_BYTES
    per episode (include/cc4.h) -- on the handle's communication stream (it overlaps the next step); returns the device pointer
    of the gathered [world*N, OBS_PACKED_BYTES] rows, valid after allgather_wait().  allgathered_obs_host() unpacks them."""
    p = ctypes.c_void_p()
    vec_env._chk(vec_env.lib.cc4_allgather_obs(vec_env._h, ctypes.byref(p)), 'cc4_allgather_obs')
    return p.value


def allgather_wait(vec_env):
    vec_env._chk(vec_env.lib.cc4_allgather_wait(vec_env._h), 'cc4_allgather_wait')


def allgathered_obs_host(vec_env, world):
    out = np.zeros((world * vec_env.num_envs, 578), np.uint8)
    vec_env._chk(vec_env.lib.cc4_get_allgathered_obs(vec_env._h, out.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_allgathered_obs')
    return out
