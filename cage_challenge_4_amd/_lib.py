"""ctypes binding of libcc4.so (include/cc4.h).  No torch, no cffi (cffi is not installed in the image).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).  There is no CPU
fallback anywhere in this package: if the shared object is missing, or no HIP device is visible,
loading / `cc4_create` raises."""
import ctypes
import os

# more hardware queues than the runtime's default four, so that four step launches can run side by side (read when HIP
# initialises; libcc4 sets the same default when it is loaded and measures whether it took effect -- csrc/cc4_api.hip)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CC4_LIB') or os.path.join(_HERE, 'libcc4.so')   # CC4_LIB: another build of the same library (kernel experiments)

OBS_PER_ENV = 578
MASK_PER_ENV = 570
NUM_BLUE = 5
MSG_LEN = 8
TOPOLOGY_BYTES = 27 + 2 * 137
OBS_LEN = (92, 92, 92, 92, 210)
ACT_LEN = (82, 82, 82, 82, 242)
OBS_OFF = (0, 92, 184, 276, 368)
ACT_OFF = (0, 82, 164, 246, 328)


class CC4Config(ctypes.Structure):
    _fields_ = [('num_envs', ctypes.c_int32), ('steps', ctypes.c_int32), ('device_id', ctypes.c_int32),
                ('rng_mode', ctypes.c_int32), ('autoreset', ctypes.c_int32), ('red_policy', ctypes.c_int32),
                ('green_policy', ctypes.c_int32), ('topology_seed', ctypes.c_int32), ('blue_policy', ctypes.c_int32)]


class AgentAction(ctypes.Structure):
    """cc4_agent_action (include/cc4.h): one submitted red / green action of cc4_step_ex."""
    _fields_ = [('type', ctypes.c_int8), ('host', ctypes.c_uint8), ('arg', ctypes.c_uint8), ('ticks', ctypes.c_uint8),
                ('session', ctypes.c_uint16), ('flags', ctypes.c_uint8), ('pad', ctypes.c_uint8),
                ('rate0', ctypes.c_double), ('rate1', ctypes.c_double)]


NUM_RED, MAX_GREEN = 6, 80
AGENT_ACTION_DTYPE = [('type', 'i1'), ('host', 'u1'), ('arg', 'u1'), ('ticks', 'u1'), ('session', 'u2'), ('flags', 'u1'), ('pad', 'u1'),
                      ('rate0', 'f8'), ('rate1', 'f8')]          # numpy view of the same 24-byte record

# every entry point declared in include/cc4.h : (restype, argtypes)
_P = ctypes.c_void_p
SIGNATURES = {
    'cc4_device_count': (ctypes.c_int, []),
    'cc4_create': (ctypes.c_int, [ctypes.POINTER(CC4Config), ctypes.POINTER(_P)]),
    'cc4_destroy': (None, [_P]),
    'cc4_last_error': (ctypes.c_char_p, [_P]),
    'cc4_reset': (ctypes.c_int, [_P, _P, _P]),
    'cc4_step': (ctypes.c_int, [_P, _P, _P]),
    'cc4_step_ex': (ctypes.c_int, [_P, _P, _P, _P, _P]),
    'cc4_edit_state': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'cc4_get_obs': (ctypes.c_int, [_P, _P]),
    'cc4_get_reward_done': (ctypes.c_int, [_P, _P, _P]),
    'cc4_get_action_mask': (ctypes.c_int, [_P, _P]),
    'cc4_get_err': (ctypes.c_int, [_P, _P]),
    'cc4_fetch': (ctypes.c_int, [_P, _P, _P, _P, _P]),
    'cc4_step_fetch': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    'cc4_get_rng_state': (ctypes.c_int, [_P, _P]),
    'cc4_set_seed': (ctypes.c_int, [_P, _P]),
    'cc4_set_rng_state': (ctypes.c_int, [_P, _P]),
    'cc4_step_device': (ctypes.c_int, [_P, _P, _P]),
    'cc4_obs_device': (ctypes.c_int, [_P, ctypes.POINTER(_P)]),
    'cc4_reward_device': (ctypes.c_int, [_P, ctypes.POINTER(_P)]),
    'cc4_done_device': (ctypes.c_int, [_P, ctypes.POINTER(_P)]),
    'cc4_actions_device': (ctypes.c_int, [_P, ctypes.POINTER(_P)]),
    'cc4_get_actions': (ctypes.c_int, [_P, _P]),
    'cc4_random_actions_device': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint32]),
    'cc4_synchronize': (ctypes.c_int, [_P]),
    'cc4_run_random_steps': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32, ctypes.POINTER(ctypes.c_float)]),
    'cc4_launches_per_step': (ctypes.c_int, [_P]),
    'cc4_group_info': (ctypes.c_int, [_P, ctypes.c_int32, _P, _P, _P]),
    'cc4_step_group_device': (ctypes.c_int, [_P, ctypes.c_int32, _P, _P]),
    'cc4_random_actions_group_device': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint32]),
    'cc4_rollout_begin': (ctypes.c_int, [_P, ctypes.c_int32]),
    'cc4_rollout_groups': (ctypes.c_int, [_P, _P, _P]),
    'cc4_rollout_policy_stream': (ctypes.c_int, [_P, _P]),
    'cc4_rollout_obs_packed': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'cc4_rollout_actions': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'cc4_rollout_wait_obs': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, _P]),
    'cc4_rollout_sync': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P]),
    'cc4_rollout_publish': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, _P]),
    'cc4_rollout_random_policy': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint32, _P]),
    'cc4_rollout_hash_policy': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, _P]),
    'cc4_rollout_end': (ctypes.c_int, [_P]),
    'cc4_rollout_standin': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint32]),
    'cc4_debug_comm_delay_us': (ctypes.c_int, [_P, ctypes.c_int]),
    'cc4_host_stats': (ctypes.c_int, [_P, _P]),
    'cc4_verify_stats': (ctypes.c_int, [_P, _P]),
    'cc4_state_bytes': (ctypes.c_size_t, []),
    'cc4_hot_bytes': (ctypes.c_size_t, []),
    'cc4_step_kernel': (ctypes.c_char_p, [_P]),
    'cc4_run_kernel': (ctypes.c_char_p, [_P]),
    'cc4_run_kernel_for': (ctypes.c_char_p, [_P, ctypes.c_int32]),
    'cc4_get_state': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'cc4_get_states': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, _P]),
    'cc4_set_state': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'cc4_cold_bytes': (ctypes.c_size_t, [_P]),
    'cc4_get_cold': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'cc4_set_cold': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'cc4_get_topology': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'cc4_get_true_state': (ctypes.c_int64, [_P, ctypes.c_int32, ctypes.c_char_p, ctypes.c_size_t]),
    'cc4_enable_event_log': (ctypes.c_int, [_P, ctypes.c_int32]),
    'cc4_keep_previous': (ctypes.c_int, [_P, ctypes.c_int32]),
    'cc4_replay_logged': (ctypes.c_int, [_P]),
    'cc4_debug_profile': (ctypes.c_int, [_P, ctypes.c_int, _P]),
    'cc4_debug_policy_probe': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, _P]),
    'cc4_comm_unique_id': (ctypes.c_int, [_P]),
    'cc4_comm_init': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, _P]),
    'cc4_allgather_obs': (ctypes.c_int, [_P, ctypes.POINTER(_P)]),
    'cc4_comm_info': (ctypes.c_int, [_P, _P, ctypes.c_char_p]),
    'cc4_allgather_wait': (ctypes.c_int, [_P]),
    'cc4_exchange_info': (ctypes.c_int, [_P, _P]),
    'cc4_debug_gather_log': (ctypes.c_int, [_P, ctypes.c_int32]),
    'cc4_debug_rollout_state': (ctypes.c_int, [_P, _P]),
    'cc4_debug_stop_phase': (ctypes.c_int, [_P, ctypes.c_int]),
    'cc4_debug_copy_from_device': (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t]),
    'cc4_get_gather_log': (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32]),
    'cc4_get_allgathered_obs': (ctypes.c_int, [_P, _P]),
    'cc4_unpack_obs_device': (ctypes.c_int, [_P, ctypes.POINTER(_P)]),
    'cc4_get_unpacked_obs': (ctypes.c_int, [_P, _P]),
    'cc4_algorithmic_bytes_per_env_step': (ctypes.c_size_t, []),
}

_lib = None


def load():
    """Load libcc4.so and attach signatures.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). cage_challenge_4_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class CC4Error(RuntimeError):
    pass


def check(lib, handle, rc, what):
    if rc != 0:
        msg = lib.cc4_last_error(handle)
        raise CC4Error(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
