// cc4_k_misc.hip -- k_reset and the small kernels of libcc4.so (exchange gate, CU discovery, stand-in policies, seeds, digest, event-log helpers).
#include "cc4_kernel_decls.h"

// The gate of a chunk of steps [k_lo, k_hi] on the communication stream: returns when every group has counted all its episodes in every
// step of the chunk (counts of one episode's consecutive steps may arrive out of order: each wave counts where ITS stores have drained),
// and hands the counters back (zero) for steps k + ring.  P > 0: the groups are the P partitions of the persistent kernel (episodes
// g, g + P, ..), else groups of 32 neighbouring episodes.  Gives up after `ticks` and says so in *fail (the host reports it).
// ONE wave, polling at a growing interval (4 us .. 31 us; the gate of a call's LAST step, behind which the host waits, stays at 4 us): the gate shares a CU with blocks of the step kernel, and in the multi-step kernels
// a block is an episode -- whatever slows one CU's blocks sets the pace of the launch (four busily polling waves cost 1024 episodes 1.2 us per step).
__global__ __launch_bounds__(WAVE) void k_xchg_gate(uint32_t* gcnt, int ring, int groups, int n, int P, int k_lo, int k_hi, long long ticks, uint32_t* fail, int max_naps) {
  const int t = (int)threadIdx.x, steps = k_hi - k_lo + 1;
  const long long t0 = wall_clock64();
  for (int i = t; i < groups * steps; i += (int)blockDim.x) {
    const int g = i / steps, k = k_lo + i % steps;
    const int size = P > 0 ? (n - g + P - 1) / P : (n - (g << 5) < 32 ? n - (g << 5) : 32);
    if (size <= 0) continue;
    uint32_t* c = gcnt + (size_t)g * (size_t)ring + (k % ring);
    int naps = 1;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)size) {
      for (int q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(127);
      if (naps < max_naps) naps <<= 1;
      if (wall_clock64() - t0 > ticks) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// which compute units does this device have?  Many small waves without LDS, each reporting the CU it landed on and idling long enough for
// the grid to spread over the whole chip.  (Not a census of how many waves of the REAL kernel a CU takes: LDS is allocated in 1280-byte
// granules, a proxy with another footprint lands differently -- r05 -- and the schedule does not need to know.)
__global__ __launch_bounds__(WAVE) void k_discover(int32_t* count, long long ticks) {
  if (threadIdx.x == 0) {
    atomicAdd(&count[cu_slot()], 1);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  }
}
__global__ __launch_bounds__(WAVE) void k_reset(ResetArgs a) {
  __shared__ uint8_t obs_lds[OBS_TOTAL + 2];
  __shared__ uint8_t mask_lds[MASK_TOTAL + 2];
  __shared__ StepWork work;
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= a.n) return;
  if (a.env_mask && !a.env_mask[e]) return;
  EnvCold* const cold_e = cold_at(a.cold, (size_t)e, cold_row_bytes(a.steps));
  EnvState* s = a.st + e;
  HostDyn* const hd = s->hd;
  for (int i = lane; i < (int)(sizeof(StepWork) / 4); i += WAVE) reinterpret_cast<uint32_t*>(&work)[i] = 0;
  __syncthreads();
  if (a.rng_mode == 1) {   // counter-based mode: the phases of env_reset_counter_mode, hosts on lanes (the row stays in HBM here)
    __shared__ uint32_t ws[RESET_WS_WORDS];
    Ctx xm{s, cold_e, &s->rng, hd, &work};
    ResetCarry carry; carry.env_key = 0;
    reset_zero(s, hd, cold_e, lane, WAVE);
    __syncthreads();
    if (lane == 0) carry = reset_topology(xm, a.seeds ? a.seeds[e] : 0, a.steps, a.seeds == nullptr, a.policy, a.topo, ws, false);
    __syncthreads();
    Rng rh; rng_fork(&rh, &s->rng, ST_GEN_HOST); rh.mode = 1;
    Ctx xh{s, cold_e, &rh, hd, &work};
    for (int h = lane; h < MAXH; h += WAVE) reset_gen_host(xh, h);
    __syncthreads();
    if (lane == 0) { reset_pid_serial(xm, reset_used_set(s)); reset_agents(xm); }     // pid uniqueness in the reference's order (one lane; once per episode)
    __syncthreads();
    reset_used_clear(s, lane, WAVE);
    __syncthreads();
    for (int h = lane; h < MAXH; h += WAVE) reset_host_sessions(xh, h);
    __syncthreads();
    if (lane == 0) reset_finish(xm, carry, a.steps, a.topo, false);
    __syncthreads();
  } else if (lane == 0) {
    Ctx x{s, cold_e, &s->rng, hd, &work};
    env_reset(x, a.seeds ? a.seeds[e] : 0, 0, a.steps, a.seeds == nullptr, a.policy, a.topo);
  }
  if (lane == 0) {
    env_flat_obs<uint8_t>(s, obs_lds);
    blue_action_mask(s, mask_lds);
    a.reward[e] = 0.f; a.done[e] = s->done; a.err[e] = s->err;
  }
  __syncthreads();
  int32_t* o = a.obs + (size_t)e * OBS_TOTAL;
  for (int i = lane; i < OBS_TOTAL; i += WAVE) o[i] = obs_lds[i];
  uint8_t* m = a.mask + (size_t)e * MASK_TOTAL;
  for (int i = lane; i < MASK_TOTAL; i += WAVE) m[i] = mask_lds[i];
  if (a.obs8) store_packed_row(a.obs8 + (size_t)e * OBS_PACKED, obs_lds, lane, WAVE);
}

// uniform blue action indices over each agent's full range: Philox key (seed0, env),
// counter (t, agent, 0xB10E, 0)
__global__ void k_random_actions(int32_t* actions, int n, uint64_t seed0, uint32_t t, int e0) {      // episodes e0 .. n - 1
  int i = e0 * NBLUE + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * NBLUE) return;
  int e = i / NBLUE, b = i % NBLUE;
  actions[i] = random_blue_action(seed0, t, e, b);
}

// debug: keeps a stream busy for about `cycles` clock ticks (cc4_debug_comm_delay_us: a slow exchange on demand)
__global__ void k_spin(long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}

// one block per gathered row: 148 packed bytes -> 578 byte values (thread j unpacks byte j into values 4j .. 4j+3)
__global__ void k_unpack_obs(const uint8_t* __restrict__ packed, uint8_t* __restrict__ out, int rows) {
  const int r = blockIdx.x, j = threadIdx.x;
  if (r >= rows || j >= OBS_PACKED) return;
  const uint32_t b = packed[(size_t)r * OBS_PACKED + j];
  uint8_t* o = out + (size_t)r * OBS_TOTAL + 4 * j;
#pragma unroll
  for (int k = 0; k < 4; ++k) if (4 * j + k < OBS_TOTAL) o[k] = (uint8_t)((b >> (2 * k)) & 3u);
}

// CybORG.set_seed (env.py:316-325): a fresh generator for the controller, the state and the hosts; the agents' policies keep
// the old one until the next reset (EnvCold.rng2); the episode itself stays as it is
__global__ void k_set_seed(EnvState* st, EnvCold* cold, size_t cold_row, const uint64_t* seeds, int n, int rng_mode) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (rng_mode == 0) {        // numpy stream: the agents' policies stay on the stream they were created with (see EnvCold.rng2)
    if (!st[e].rng_split) cold_at(cold, (size_t)e, cold_row)->rng2 = st[e].rng;
    st[e].rng_split = 1;
  }
  rng_seed(&st[e].rng, seeds[e], (uint32_t)rng_mode);
  if (rng_mode == 1) { rng_begin_episode(&st[e].rng); rng_park(&st[e].rng); }   // counter mode: the words a reset leaves behind
}

// an externally built numpy Generator(PCG64) handed over as CybORG(seed=generator) (env.py:73-76): its bit-generator state
// becomes the episode's stream (words per episode: state high, state low, increment high, increment low, has_uint32, uinteger)
__global__ void k_set_rng_state(EnvState* st, const uint64_t* w, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  Rng r;
  rng_seed(&r, 0, 0);
  r.s_hi = w[6 * e]; r.s_lo = w[6 * e + 1]; r.inc_hi = w[6 * e + 2]; r.inc_lo = w[6 * e + 3];
  r.has32 = (uint32_t)w[6 * e + 4]; r.u32 = (uint32_t)w[6 * e + 5];
  st[e].rng = r;
  st[e].rng_split = 0;
}

__global__ void k_rng_state(const EnvState* st, uint64_t* out, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const Rng& r = st[e].rng;
  uint64_t* o = out + 7 * (size_t)e;
  o[0] = r.s_hi; o[1] = r.s_lo; o[2] = r.inc_hi; o[3] = r.inc_lo; o[4] = r.has32; o[5] = r.u32; o[6] = r.ndraw;
}

// ---------------------------------------------------------------- rollouts: the caller-side kernels (cc4_rollout_*)
// gate of a policy pass: returns when every episode of policy group g has its packed row of the step in slot `slot` in memory (the step kernel
// counts them per partition, RunArgs.act_ready), and hands the counters back zeroed.  One wave, partitions on lanes; gives up after `ticks`.
__global__ __launch_bounds__(WAVE) void k_rollout_gate(uint32_t* cnt, int P, int PG, int ring, int g, int slot, int n, long long ticks, uint32_t* fail) {
  __builtin_amdgcn_s_setprio(3);      // these waves run in what the persistent kernel leaves of a CU, beside 23 of its waves per CU: ahead of them in the SIMDs' issue arbitration
  const long long t0 = wall_clock64();
  for (int p = (int)threadIdx.x; p < P; p += (int)blockDim.x) {
    const int ne = (n - p + P - 1) / P;                       // episodes p, p + P, ..: index i is of group i % PG
    const int want = (ne - g + PG - 1) / PG;
    if (want <= 0) continue;
    uint32_t* c = cnt + ((size_t)p * PG + (size_t)g) * (size_t)ring + slot;
    int naps = 1;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)want) {
      for (int q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(8);
      if (naps < 8) naps <<= 1;
      if (wall_clock64() - t0 > ticks) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// publish and gate in one launch (cc4_rollout_sync): first the publish of a pass whose policy kernels precede this kernel on the stream (their stores
// are in memory: a kernel boundary lies between), then the gate of the next pass.  Half the stream operations of the separate calls.
__global__ __launch_bounds__(WAVE) void k_rollout_sync(uint32_t* ready, int pub_g, uint32_t pub_val, uint32_t* cnt, int P, int PG, int ring, int g, int slot, int n, long long ticks, uint32_t* fail) {
  __builtin_amdgcn_s_setprio(3);      // these waves run in what the persistent kernel leaves of a CU, beside 23 of its waves per CU: ahead of them in the SIMDs' issue arbitration
  // the publish: every CU partition's copy of the group's word (RunArgs.act_ready)
  if (pub_g >= 0) for (int p = (int)threadIdx.x; p < P; p += (int)blockDim.x) __hip_atomic_store(ready + (size_t)p * 32 + pub_g, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (g < 0) return;
  const long long t0 = wall_clock64();
  for (int p = (int)threadIdx.x; p < P; p += (int)blockDim.x) {
    const int ne = (n - p + P - 1) / P;
    const int want = (ne - g + PG - 1) / PG;
    if (want <= 0) continue;
    uint32_t* c = cnt + ((size_t)p * PG + (size_t)g) * (size_t)ring + slot;
    int naps = 1;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)want) {
      for (int q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(8);
      if (naps < 8) naps <<= 1;
      if (wall_clock64() - t0 > ticks) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// stand-in policies for one policy group (bench.py, tests): uniform random indices (the draws of k_random_actions), or indices computed FROM the
// packed observations of the step before (a policy that ignores its input proves nothing about the hand-over)
// (threads over the GROUP's episodes only -- episode i of group g is e = ((i / P) * PG + g) * P + i % P --: a launch is a quarter of the batch's waves,
// one dispatch round into the slots the persistent kernel leaves free)
__device__ __forceinline__ int rollout_group_episode(int i, int P, int PG, int g) { return ((i / P) * PG + g) * P + i % P; }
__global__ __launch_bounds__(WAVE) void k_rollout_random_policy(int32_t* act, int n, int P, int PG, int g, uint64_t seed0, uint32_t t) {
  __builtin_amdgcn_s_setprio(3);      // these waves run in what the persistent kernel leaves of a CU, beside 23 of its waves per CU: ahead of them in the SIMDs' issue arbitration
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = rollout_group_episode(i / NBLUE, P, PG, g), b = i % NBLUE;
  if (e >= n) return;
  act[e * NBLUE + b] = random_blue_action(seed0, t, e, b);
}
__device__ __host__ inline uint32_t rollout_obs_hash(const uint32_t* row) {      // 37 words of a packed observation row
  uint32_t hsh = 2166136261u;
  for (int w = 0; w < OBS_PACKED / 4; ++w) { hsh ^= row[w]; hsh *= 16777619u; }
  return hsh;
}
__global__ __launch_bounds__(WAVE) void k_rollout_hash_policy(int32_t* act, const uint8_t* packed, int n, int P, int PG, int g, uint32_t j) {
  __builtin_amdgcn_s_setprio(3);
  const int e = rollout_group_episode(blockIdx.x * blockDim.x + threadIdx.x, P, PG, g);
  if (e >= n) return;
  const uint32_t hsh = rollout_obs_hash(reinterpret_cast<const uint32_t*>(packed + (size_t)e * OBS_PACKED));
  for (int b = 0; b < NBLUE; ++b) act[e * NBLUE + b] = (int32_t)((hsh + 2654435761u * (uint32_t)(b + 1) + 40503u * j) % (uint32_t)(b == 4 ? ACT_LONG : ACT_SHORT));
}
// the packed rows of the observations as they stand in the int32 buffer (what a rollout's first policy pass reads)
__global__ __launch_bounds__(WAVE) void k_pack_obs_rows(uint8_t* packed, const int32_t* obs, int n) {
  const int e = blockIdx.x;
  if (e < n) pack_row_from_obs(packed + (size_t)e * OBS_PACKED, obs + (size_t)e * OBS_TOTAL, (int)threadIdx.x);
}

// CC4_PERSIST_VERIFY: a digest per episode of everything a call of cc4_run_random_steps leaves behind -- hot row, cold row, observations,
// reward / done / error word, the drawn actions -- in three words (hot, cold, outputs), so that a mismatch says where
__global__ __launch_bounds__(WAVE) void k_digest(const EnvState* st, const EnvCold* cold, size_t cold_row, const int32_t* obs, const float* reward,
                                                 const uint8_t* done, const uint32_t* err, const int32_t* actions, uint64_t* out, int n) {
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= n) return;
  auto mix = [](uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0xD6E8FEB86659FD93ull; };
  auto hash_vecs = [&](const uint4* p, size_t nv) {
    uint64_t h = 0x1234567ull + (uint64_t)lane;
    for (size_t i = lane; i < nv; i += WAVE) { const uint4 v = p[i]; h = mix(h, ((uint64_t)v.x << 32) | v.y); h = mix(h, ((uint64_t)v.z << 32) | v.w); }
    for (int off = 32; off >= 1; off >>= 1) h += __shfl_xor(h, off);     // order-independent across lanes, position-dependent within one
    return h;
  };
  const uint64_t h_hot = hash_vecs(reinterpret_cast<const uint4*>(st + e), sizeof(EnvState) / 16);
  const uint64_t h_cold = hash_vecs(reinterpret_cast<const uint4*>(cold_at(const_cast<EnvCold*>(cold), (size_t)e, cold_row)), cold_row / 16);
  uint64_t h = 0x89ABCDEFull + (uint64_t)lane;
  for (int i = lane; i < OBS_TOTAL; i += WAVE) h = mix(h, (uint64_t)(uint32_t)obs[(size_t)e * OBS_TOTAL + i]);
  if (lane < NBLUE) h = mix(h, (uint64_t)(uint32_t)actions[e * NBLUE + lane]);
  if (lane == 8) { h = mix(h, (uint64_t)__float_as_uint(reward[e])); h = mix(h, ((uint64_t)done[e] << 32) | err[e]); }
  for (int off = 32; off >= 1; off >>= 1) h += __shfl_xor(h, off);
  if (lane == 0) { out[3 * (size_t)e] = h_hot; out[3 * (size_t)e + 1] = h_cold; out[3 * (size_t)e + 2] = h; }
}


// event-log helpers (cc4_enable_event_log, cc4_replay_logged)
extern "C" {
__global__ void k_set_evlog(EnvCold* cold, size_t row_bytes, int n, uint32_t on) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) { EnvCold* c = cold_at(cold, (size_t)e, row_bytes); c->evlog.enabled = on; c->evlog.n = 0; }
}
__global__ void k_copy_evlog(EnvCold* dst, const EnvCold* src, size_t row_bytes, int n) {
  const int e = blockIdx.x;
  if (e >= n) return;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&cold_at(const_cast<EnvCold*>(src), (size_t)e, row_bytes)->evlog);
  uint32_t* d = reinterpret_cast<uint32_t*>(&cold_at(dst, (size_t)e, row_bytes)->evlog);
  for (int i = threadIdx.x; i < (int)(sizeof(EvLog) / 4); i += blockDim.x) d[i] = s[i];
}
}
