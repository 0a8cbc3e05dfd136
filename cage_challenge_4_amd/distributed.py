"""Multi-GPU plumbing: one process per GPU, the env batch sharded statically across ranks (no data-path collective in
the step itself); the single exchange step is the all-gather of the flat observations, done natively over RCCL/xGMI
by libcc4 (cc4_allgather_obs).  torch.distributed (gloo) is only the control plane: rendezvous, the RCCL unique-id
broadcast, barriers and the max-over-ranks timing reduction."""
import ctypes
import os
import numpy as np
from .vec_env import shard_range


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


def init_rccl(vec_env, rank, world):
    """Create the RCCL communicator inside libcc4 for this rank's handle (unique id travels over the control plane)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')   # single-node job: bootstrap over loopback, no NIC probing
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (normally exported already; read when the runtime loads)
    ident = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (ctypes.c_uint8 * 128)()
        rc = vec_env.lib.cc4_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError('cc4_comm_unique_id failed')
        ident = torch.tensor(list(buf), dtype=torch.uint8)
    if world > 1:
        dist.broadcast(ident, src=0)
    raw = (ctypes.c_uint8 * 128)(*ident.tolist())
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
