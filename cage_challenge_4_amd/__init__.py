"""cage_challenge_4_amd -- MI355X-native vectorised CAGE Challenge 4 (CybORG v4) step engine.

Only what the hot path needs: csrc/ (HIP kernels + C ABI), the ctypes binding, the vectorised env and the
drop-in mirror of the reference's wrapper surface.  Importing the package does not touch the GPU; creating an
environment does, and fails loudly if libcc4.so is missing or no HIP device is visible (no CPU fallback)."""
from .vec_env import (CC4VecEnv, RNG_PCG64, RNG_PHILOX, RED_FSM, RED_SLEEP, RED_DISCOVERY, RED_RANDOM, GREEN_ENTERPRISE,  # noqa: F401
                      GREEN_SLEEP, split_obs, split_mask, shard_range)
from .wrappers import (CybORG, EnterpriseScenarioGenerator, SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent,  # noqa: F401
                       DiscoveryFSRed, RandomSelectRedAgent, cc4BlueRandomAgent,
                       BlueFixedActionWrapper, BlueFlatWrapper, BlueEnterpriseWrapper, EnterpriseMAE)

from .true_state import TrueStateTableWrapper  # noqa: F401

__version__ = '0.1.0'
