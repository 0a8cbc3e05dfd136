// cc4_hip.hip -- gfx950 kernels + the C ABI (include/cc4.h) of libcc4.so.
//
// Execution model of the numpy-stream kernel (k_step): one 64-lane wavefront per episode.  The wave stages the agent part of the
// episode's packed EnvState row (everything in front of the host table: 6.9 KB) HBM -> LDS with coalesced 16-byte loads and
// leaves the host table (8.8 KB, of which a step visits a few dozen rows) in HBM/L2 -- LDS is what bounds the number of
// resident waves of this kernel, and its one working lane hides memory latency only through them.  Lane 0 walks the
// strictly ordered transition (the reference's ~57 agent actions share one RNG stream, so the order is the semantics),
// the wave encodes the 578 flat-observation values, and the agent part goes back LDS -> HBM coalesced.
// The cold part of the episode (process-list overflow, ephemeral-port bitmaps, per-session port knowledge) stays in HBM
// and is touched a handful of times per step.  The counter-mode kernel (k_step_philox, below) stages the whole row and
// runs four wavefronts per episode.
// No MFMA: the path is integer / indexing.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>

#include "../../include/cc4.h"
#include "../../include/cc4_debug.h"
#include "cc4_engine.h"
#include "cc4_export.h"

using namespace cc4;

static_assert(sizeof(EnvState) % 16 == 0 && offsetof(EnvState, hd) % 16 == 0, "EnvState rows are staged with 16-byte accesses");
constexpr int ROW_VEC = (int)(sizeof(EnvState) / 16);
constexpr int HOT_VEC = (int)(offsetof(EnvState, hd) / 16);   // the part in front of the host table
constexpr int WAVE = 64;
constexpr int OBS_PACKED = CC4_OBS_PACKED_BYTES;   // every flat-observation value is 0, 1 or 2: the exchange moves 2 bits per value
static_assert(OBS_PACKED % 4 == 0 && OBS_PACKED * 4 >= OBS_TOTAL, "packed observation row: whole words, four values per byte");
constexpr int PROF_SLOTS = 128;   // 16 phase slots, 8 per red agent (16..63), then (cycles, count) per red action type (64..)

constexpr int cc4_handle_max_groups = 8;   // cc4_handle::MAX_GROUPS

struct StepArgs {
  EnvState* st; EnvCold* cold;
  const int32_t* actions; const uint8_t* msgs;
  int32_t* obs; float* reward; uint8_t* done; uint32_t* err;
  uint8_t* obs8;               // the same observations packed 2 bits per value, OBS_PACKED bytes per episode (what the multi-GPU
                               // all-gather moves), or null
  int32_t* rand_out;           // when non-null: draw the blue actions in-kernel (k_random_actions fused) and record them here
  uint64_t rand_seed0; uint32_t rand_t;
  int n, autoreset, steps, rng_mode, policy;
  int full_obs;               // rewrite every observation value (the output buffer may hold another episode's slowly varying part)
  uint32_t topo;              // cc4_config.topology_seed
  unsigned long long* prof;   // optional [n][PROF_SLOTS] cycle counters (cc4_debug_profile): 16 phase slots + 8 per red agent
  uint32_t* reset_ws;         // k_step_philox1: [n][RESET_WS_WORDS] work area of the in-kernel scenario generation (the other
                              // kernels keep it in LDS; an episode regenerates once in steps-per-episode launches)
  const ExtAct* ext;          // [n][EXT_PER_ENV] externally submitted red / green actions of this step (cc4_step_ex), or null; read by the
                              // full builds of the step kernels only (template parameter LOG)
  int e0;                     // first episode of this launch: block b steps episode e0 + b (a step of a large batch is issued as
                              // several launches on separate streams: see cc4_handle::ngroups); n = one past its last episode
  int act_sys;                // the actions were written by ANOTHER kernel while this one runs (a rollout, RunArgs.act_ready): system-scope loads,
                              // past this XCD's L2, which may still hold the line from two steps ago
};

// uniform blue action index of (episode e, agent b) at step t: Philox key (seed0 + e), counter (t, b, 0xB10E, 0)
__device__ __forceinline__ int32_t random_blue_action(uint64_t seed0, uint32_t t, int e, int b) {
  uint32_t c[4] = {t, (uint32_t)b, 0xB10Eu, 0u};
  uint64_t key = seed0 + (uint64_t)e;
  philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
  uint32_t range = b == 4 ? ACT_LONG : ACT_SHORT;
  return (int32_t)(((uint64_t)c[0] * range) >> 32);
}

// ---------------------------------------------------------------- kernels
// HBM -> LDS row staging with 8 independent 16-byte loads in flight per lane (a plain copy loop serialises on vmcnt)
template <int NVEC>
__device__ __forceinline__ void stage_in(uint4* __restrict__ lds, const uint4* __restrict__ src, int lane) {
  constexpr int U = NVEC / WAVE < 8 ? (NVEC / WAVE > 0 ? NVEC / WAVE : 1) : 8;
  int i = lane;
  for (; i + (U - 1) * WAVE < NVEC; i += U * WAVE) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * WAVE];
#pragma unroll
    for (int u = 0; u < U; ++u) lds[i + u * WAVE] = v[u];
  }
  for (; i < NVEC; i += WAVE) lds[i] = src[i];
}
template <int NVEC>
__device__ __forceinline__ void stage_out(uint4* __restrict__ dst, const uint4* __restrict__ lds, int lane) {
  constexpr int U = NVEC / WAVE < 8 ? (NVEC / WAVE > 0 ? NVEC / WAVE : 1) : 8;
  int i = lane;
  for (; i + (U - 1) * WAVE < NVEC; i += U * WAVE) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = lds[i + u * WAVE];
#pragma unroll
    for (int u = 0; u < U; ++u) dst[i + u * WAVE] = v[u];
  }
  for (; i < NVEC; i += WAVE) dst[i] = lds[i];
}

// LOG: the full build of a step kernel -- it records the HostEvents entries of the step (cc4_enable_event_log) and takes externally
// submitted red / green actions (cc4_step_ex: StepArgs.ext).  A template parameter rather than a run-time flag: even a never-taken
// logging branch at the eleven event sites costs the serial walk 10 %.
// byte j of an episode's packed observation row: values 4j .. 4j+3 (from a byte-per-value row in LDS), 2 bits each, low bits first
__device__ __forceinline__ uint8_t pack_obs_byte(const uint8_t* vals, int j) {
  uint32_t b = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = 4 * j + k; if (i < OBS_TOTAL) b |= (uint32_t)(vals[i] & 3u) << (2 * k); }
  return (uint8_t)b;
}

// An episode's packed observation row (OBS_PACKED bytes = 37 words) to memory, one word per thread, as SYSTEM-scope (write-through) stores:
// the reader is the exchange -- a copy engine, an RCCL kernel on any XCD, a peer GPU -- and, when the writer is a one-launch kernel, there is
// no kernel boundary that would write the XCD's L2 back first (tools/micro/ring_protocol.hip: plain stores arrive stale, these do not).
__device__ __forceinline__ void store_packed_row(uint8_t* o8, const uint8_t* vals, int t, int nt) {
  uint32_t* o32 = reinterpret_cast<uint32_t*>(o8);
  for (int w = t; w < OBS_PACKED / 4; w += nt) {
    const uint32_t v = (uint32_t)pack_obs_byte(vals, 4 * w) | ((uint32_t)pack_obs_byte(vals, 4 * w + 1) << 8) |
                       ((uint32_t)pack_obs_byte(vals, 4 * w + 2) << 16) | ((uint32_t)pack_obs_byte(vals, 4 * w + 3) << 24);
    __hip_atomic_store(o32 + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// The same row packed from the int32 observation row the wave has just (re)written in global memory -- the output buffer persists between
// steps, so it holds every current value although a step only rewrites the ones that changed.  For the one-wave kernels: a byte copy of the
// 578 values in LDS would cost them a seventh 1280-byte LDS granule and with it two of their twenty resident waves per CU
// (profiles/r05_lds_residency.txt).  The wave's own stores are drained first (the vector L1 is write-through: they are in the XCD's L2),
// the loads are agent-scope (served by that L2, never by a stale L1 line).
__device__ __forceinline__ void pack_row_from_obs(uint8_t* o8, const int32_t* o, int lane) {
  // call with the wave's stores drained (s_waitcnt vmcnt(0)): then plain loads see them -- the row was written by this wave, by earlier
  // waves of this CU (same L1; a stolen partition's item starts with an L1 invalidate), or before the launch
  static_assert((OBS_TOTAL * 4) % 8 == 0, "rows of the int32 observation buffer are 8-byte aligned: two values per load");
  if (lane < OBS_PACKED / 4) {
    const uint2* o2 = reinterpret_cast<const uint2*>(o + 16 * lane);
    uint2 w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = (16 * lane + 2 * k < OBS_TOTAL) ? o2[k] : make_uint2(0u, 0u);     // (578 is even: a pair is inside the row or outside)
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v |= ((w[k].x & 3u) << (4 * k)) | ((w[k].y & 3u) << (4 * k + 2));
    __hip_atomic_store(reinterpret_cast<uint32_t*>(o8) + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// The per-step hand-off out of the one-launch kernels (cc4_run_random_steps with a communicator).  Step k of the launch writes
// its packed rows into slab k % ring and, once an episode's row is in memory, counts it in its group's counter of that step (a no-return
// atomic: nothing waits for it); on the communication stream a one-block gate kernel (k_xchg_gate) waits until every group has counted
// every step of a chunk, the chunk's slabs are gathered, and gathered = last + 1 is published (hipStreamWriteValue32); step k + ring of any
// episode waits for gathered > k before it overwrites the slab.  The exchange lags the stepping by up to `ring` steps, with no launch
// boundary in the compute queue.  A wait that lasts longer than wait_ticks gives up, raises *timeout (the host falls back to per-step
// launches and says so) and every later wait of the launch returns at once: a stuck exchange never hangs the kernel.
struct XchgArgs {
  uint8_t* slab;                 // [ring][n][OBS_PACKED], or null: no exchange
  uint32_t* gathered;            // [1]
  uint32_t* timeout;             // [1]
  int ring;
  long long wait_ticks;          // wall_clock64 ticks (100 MHz)
  uint32_t* gcnt;                // [groups][ring]: episodes of a group that finished step k (slot k % ring), see xchg_count
  uint32_t* timeout_host;        // the same flag in pinned host memory, WRITTEN only (the host reads it without a copy; the waits poll the
                                 // device word: a thousand blocks polling a word across PCIe cost a 1024-episode batch 12 us per step)
};
// lane / thread 0 only.  `seen` = the highest value of *gathered this wave has read so far (it only grows): the word is read again --
// an uncached round trip to memory, ~2 us in the middle of the item hand-over -- only when the value at hand does not cover step k.
__device__ __forceinline__ void xchg_wait_slab(const XchgArgs& x, uint32_t k, uint32_t& seen) {
  if (k < (uint32_t)x.ring || !x.gathered) return;      // (no `gathered` word: a rollout -- slab k % ring was consumed by the policy pass of step k - ring + 1, which every episode is long past)
  const uint32_t need = k - (uint32_t)x.ring + 1u;
  if (seen >= need) return;
  seen = __hip_atomic_load(x.gathered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (seen >= need) return;
  if (__hip_atomic_load(x.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
  // Thousands of waves polling one uncached word starve the very write they wait for (tools/micro/ring_protocol.hip: a saturated chip
  // of spinning pollers took 57 us per exchange step instead of < 16): the interval between two polls of a wave doubles from ~3 us to ~50 us.
  const long long w0 = wall_clock64();
  int naps = 1;
  while ((seen = __hip_atomic_load(x.gathered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) < need) {
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
    if (naps < 16) naps <<= 1;
    if (wall_clock64() - w0 > x.wait_ticks || __hip_atomic_load(x.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
      __hip_atomic_store(x.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(x.timeout_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}
// One episode's packed row of step k is in memory (the stores that wrote it have drained): counted in the episode's group (a partition of
// the persistent kernel, 32 neighbouring episodes of the multi-step kernels), slot k % ring.  A no-return agent-scope atomic: the wave
// does not wait for it.  (r05 on the way here: one system-scope counter per step -- 8192 atomics on one word serialise at ~12 ns each,
// twice the step --, then two levels with the group's last episode adding the group to it -- two dependent atomics, ~2 us per item.)
__device__ __forceinline__ void xchg_count(const XchgArgs& x, uint32_t k, int group) {
  (void)__hip_atomic_fetch_add(x.gcnt + (size_t)group * (size_t)x.ring + (k % (uint32_t)x.ring), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
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

// ---------------------------------------------------------------- numpy stream: the two draw-only phases across the wave
// PCG64 is a 128-bit LCG, so the state k steps ahead is A_k * state + B_k * increment (A_k = M^k, B_k = 1 + M + .. + M^(k-1),
// mod 2^128; table filled by cc4_create).  Two phases of a step only CONSUME the stream -- the green agents' policy draws
// (one bounded draw each) and the action-order shuffle (SimulationController.py:418: ~90 masked-rejection draws whose results
// are never used) -- so lane j computes output j+1 directly and the consumption is replayed on the 128 ready words with a
// few wave-wide compares per draw instead of a 128-bit multiply per draw on the walking lane.  Bit-exact with the serial
// walk (rng_below / rng_interval in cc4_rng.h), including has_uint32 / uinteger buffering and the advance counter.
__device__ uint32_t g_obs_fast[OBS_FAST];          // obs_fast_entry(v) for v = 0 .. OBS_FAST-1 (cc4_engine.h), filled by cc4_create
// The observation values that can change with every step, from the table: position, source byte and mask come with one load
// instead of a dozen divisions per value.
template <int nt>
__device__ __forceinline__ void encode_obs_fast(const EnvState* s, int32_t* o, uint8_t* obs_bytes, bool pack, int t) {
  constexpr int NV = (OBS_FAST + nt - 1) / nt;
  uint32_t ent[NV];
#pragma unroll
#ifdef CC4_OBS_TABLE
  for (int k = 0; k < NV; ++k) { const int v = t + k * nt; ent[k] = v < OBS_FAST ? g_obs_fast[v] : 0u; }
#else
  // computed, not loaded: the kernels wait on memory, not on the vector unit (r03 A/B: the table form of this loop -- one L2 load
  // per value instead of a dozen shifts and multiplies -- made the encode phase longer: 5.5k -> 6.9k cycles at 8192 episodes)
  for (int k = 0; k < NV; ++k) { const int v = t + k * nt; ent[k] = v < OBS_FAST ? obs_fast_entry(v) : 0u; }
#endif
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = t + k * nt;
    if (v >= OBS_FAST) continue;
    const int val = obs_fast_value(ent[k], s);
    const int i = (int)(ent[k] & 0x3FF);
    o[i] = val;
    if (pack) obs_bytes[i] = (uint8_t)val;
  }
}
struct PcgJump { uint64_t a_hi, a_lo, b_hi, b_lo; };
__device__ PcgJump g_pcg_jump[WAVE + 1];          // [k]: k = 0 .. 64 steps ahead
__device__ __forceinline__ uint64_t bcast64(uint64_t v) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); }
__device__ __forceinline__ uint64_t lane64(uint64_t v, int src) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src) << 32);
}
__device__ __forceinline__ void pcg_ahead(const PcgJump& J, uint64_t s_hi, uint64_t s_lo, uint64_t i_hi, uint64_t i_lo, uint64_t* o_hi, uint64_t* o_lo) {
  // (a * s + b * inc) mod 2^128
  const uint64_t p_lo = J.a_lo * s_lo, p_hi = __umul64hi(J.a_lo, s_lo) + J.a_hi * s_lo + J.a_lo * s_hi;
  const uint64_t q_lo = J.b_lo * i_lo, q_hi = __umul64hi(J.b_lo, i_lo) + J.b_hi * i_lo + J.b_lo * i_hi;
  const uint64_t lo = p_lo + q_lo;
  *o_lo = lo; *o_hi = p_hi + q_hi + (lo < p_lo ? 1ull : 0ull);
}
__device__ __forceinline__ uint64_t pcg_output(uint64_t hi, uint64_t lo) {   // XSL-RR 128/64
  const uint64_t v = hi ^ lo; const uint32_t rot = (uint32_t)(hi >> 58);
  return (v >> rot) | (v << ((64u - rot) & 63u));
}
// Green policy draws of one step (EnterpriseGreenAgent.get_action: choice of 3 per agent, agent order), all lanes.  `rl` is the
// walking lane's generator (valid on lane 0, updated there).  Returns false without touching anything when a draw would need
// Lemire's re-draw (a zero word: 2^-32 per agent) -- the caller then walks the phase serially.
__device__ __forceinline__ bool wave_green_policy(Rng& rl, int n, uint8_t* green_act, int lane) {
  const uint64_t s_hi = bcast64(rl.s_hi), s_lo = bcast64(rl.s_lo), i_hi = bcast64(rl.inc_hi), i_lo = bcast64(rl.inc_lo);
  const uint32_t has32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.has32), u32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.u32);
  uint64_t h, l;
  pcg_ahead(g_pcg_jump[lane + 1], s_hi, s_lo, i_hi, i_lo, &h, &l);
  const uint64_t out = pcg_output(h, l);
  const uint32_t w0 = (uint32_t)out, w1 = (uint32_t)(out >> 32);
  const int base = (int)has32;                       // agent 0 takes the buffered half word when there is one
  const int g0 = base + 2 * lane, g1 = g0 + 1;
  bool zero = (g0 < n && w0 == 0) || (g1 < n && w1 == 0) || (has32 && u32 == 0);
  if (__ballot(zero)) return false;
  // Lemire, range 3: (word * 3) >> 32 (the leftover test can only fail for word == 0)
  if (g0 < n) green_act[g0] = (uint8_t)(((uint64_t)w0 * 3u) >> 32);
  if (g1 < n) green_act[g1] = (uint8_t)(((uint64_t)w1 * 3u) >> 32);
  if (has32 && lane == 0) green_act[0] = (uint8_t)(((uint64_t)u32 * 3u) >> 32);
  const int fresh = n - base;                        // words taken from new outputs
  const int K = (fresh + 1) >> 1;                    // outputs consumed
  if (K > 0) {
    const uint64_t nh = lane64(h, K - 1), nl = lane64(l, K - 1);
    const uint32_t nu = (uint32_t)__builtin_amdgcn_readlane((int)w1, K - 1);
    if (lane == 0) { rl.s_hi = nh; rl.s_lo = nl; rl.u32 = nu; rl.has32 = (uint32_t)(fresh & 1); rl.ndraw += (uint32_t)K; }
  } else if (lane == 0) rl.has32 = 0;
  return true;
}
// wave64 inclusive scans on the DPP network (row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 and 31 across rows) and
// the shift by one lane that turns them into exclusive ones
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return (uint32_t)x;
}
__device__ __forceinline__ int wave_scan_max(int v) {   // v >= -1
  auto mx = [](int a, int b) { return a > b ? a : b; };
  int x = v;
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x111, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x112, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x114, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x118, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x142, 0xa, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x143, 0xc, 0xf, false));
  return x;
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t lane0) {   // lane k <- lane k - 1, lane 0 <- lane0
  return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138, 0xf, 0xf, false);
}
// Generator.shuffle of an n-item list, consumption only (rng_shuffle_consume): for i = n-1 .. 1 one masked-rejection draw
// (random_interval).  All lanes; `rl` as above.  Which i a word is tested against depends on how many words in front of it
// were accepted: a prefix count that depends on itself, solved by relaxation (every lane tests its two words against its
// current estimate of i, a wave scan of the accepted counts gives the next estimates; a word's verdict only moves when i
// crosses its masked value, so a handful of rounds settle a window of 128 words).
__device__ __forceinline__ void wave_shuffle_consume(Rng& rl, int n, int lane) {
  if (n <= 1) return;
  uint64_t s_hi = bcast64(rl.s_hi), s_lo = bcast64(rl.s_lo);
  const uint64_t i_hi = bcast64(rl.inc_hi), i_lo = bcast64(rl.inc_lo);
  uint32_t has32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.has32), u32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.u32);
  uint32_t adv = 0;
  int i = n - 1;
  auto mask_of = [](uint32_t m) { return 0xFFFFFFFFu >> __builtin_clz(m); };   // m >= 1: the smallest 2^k - 1 >= m (random_interval's mask)
  if (has32) {                                       // the buffered half word is the first candidate
    has32 = 0;
    if ((u32 & mask_of((uint32_t)i)) <= (uint32_t)i) --i;
  }
  while (i >= 1) {
    // a window of 64 outputs = 128 words: word p = half (p & 1) of output (p >> 1) + 1, i.e. lane p >> 1
    uint64_t h, l;
    pcg_ahead(g_pcg_jump[lane + 1], s_hi, s_lo, i_hi, i_lo, &h, &l);
    const uint64_t out = pcg_output(h, l);
    const uint32_t w0 = (uint32_t)out, w1 = (uint32_t)(out >> 32);
    uint32_t f_cur = 3, inc = 0;                     // accepted flags of the lane's two words (first guess: all accepted)
    for (int round = 0; round < WAVE + 2; ++round) {
      inc = wave_scan_add((f_cur & 1u) + (f_cur >> 1));
      const int i0 = i - (int)wave_shr1(inc, 0u);    // the i word 0 of this lane is tested against
      const uint32_t a0 = (i0 >= 1 && (w0 & mask_of((uint32_t)(i0 >= 1 ? i0 : 1))) <= (uint32_t)i0) ? 1u : 0u;
      const int i1 = i0 - (int)a0;
      const uint32_t a1 = (i1 >= 1 && (w1 & mask_of((uint32_t)(i1 >= 1 ? i1 : 1))) <= (uint32_t)i1) ? 1u : 0u;
      const uint32_t f_new = a0 | (a1 << 1);
      const bool moved = f_new != f_cur;
      f_cur = f_new;
      if (!__ballot(moved)) break;                   // lane k is exact after k + 1 rounds at the latest
    }
    const int total = (int)rdlane(inc, WAVE - 1);
    int cur = 2 * WAVE;                              // first unconsumed word of the window
    if (total >= i) {                                // the draw for i = 1 ends inside the window: behind the i-th accepted word
      const uint64_t m = __ballot((int)inc >= i);
      const int L = __ffsll((unsigned long long)m) - 1;
      const int before = L ? (int)rdlane(inc, L - 1) : 0;
      const uint32_t fl = rdlane(f_cur, L);
      cur = 2 * L + ((before + (int)(fl & 1u) >= i) ? 1 : 2);
      i = 0;
    } else i -= total;
    const int K = (cur + 1) >> 1;                    // outputs of this window that were touched
    s_hi = lane64(h, K - 1); s_lo = lane64(l, K - 1);
    u32 = rdlane(w1, K - 1);
    has32 = (uint32_t)(cur & 1);
    adv += (uint32_t)K;
  }
  if (lane == 0) { rl.s_hi = s_hi; rl.s_lo = s_lo; rl.has32 = has32; rl.u32 = u32; rl.ndraw += adv; }
}

// ---------------------------------------------------------------- numpy stream: the green actions across the wave
// GreenAccessService / GreenLocalWork draw from the one shared stream, agent after agent, and every agent's number of draws
// depends on what it drew -- but on nothing it reads from the state that an earlier green action of the same step could have
// changed, with two exceptions: an ephemeral port that is already taken (Host.py:175-187 re-draws once) and a phishing email
// (a red session appears).  So the serial walk (50-odd agents x [a 128-bit multiply per draw + an HBM round trip for the port
// bitmap and one per event byte]) is replaced by:
//  (1) the next 128 outputs of the LCG from the closed form, two per lane, into LDS;
//  (2) where in the stream each agent starts.  That is a prefix sum over the agents' draw counts, which depend on the drawn
//      values, i.e. on the start: solved by relaxation -- every lane (one agent each) replays its action from its current
//      start estimate, a wave scan of the counts gives the next estimates, until nothing moves.  Agent 0's start is given,
//      so agent k is exact after k + 1 rounds at the latest; as an action's draw count rarely depends on the values (a
//      blocked route, a failed reliability roll, the two 1 % events), three rounds are the rule;
//  (3) the agents' effects -- the port bitmap test-and-set (one L2 atomic), the event bits, the reward -- on their lanes.
// A taken port, a phishing email, a Lemire re-draw (n / 2^32 per draw) or the end of the window end a batch: its agents in
// front of that point are committed, the agent at that point is resolved by the serial code on lane 0 (ports set
// speculatively by later agents are cleared first), and the next batch starts behind it.  Bit-exact with the serial walk
// (rng_below / rng_random / has_uint32 buffering in cc4_rng.h).
constexpr int GW_OUT = 2 * WAVE;                           // outputs per window
constexpr double P01_SCALED = 0.01 * 9007199254740992.0;   // Generator.random() < 0.01 on the 53-bit integer: exact scaling
constexpr uint64_t P01_FLOOR = (uint64_t)P01_SCALED;
static_assert((double)P01_FLOOR * (1.0 / 9007199254740992.0) < 0.01 && (double)(P01_FLOOR + 1) * (1.0 / 9007199254740992.0) >= 0.01,
              "integer form of rng_random() < 0.01");
enum : uint32_t { GR_VALID = 1, GR_FAIL = 2, GR_EPH = 4, GR_CONN = 8, GR_PROC = 16, GR_HARD = 32, GR_PHISH = 64 };
__device__ __forceinline__ void wave_green_exec(Ctx x, Rng& rl, uint64_t* win, int lane, unsigned long long* gstat = nullptr) {
  EnvState* s = x.s;
  const int ng = s->n_green;
  const uint64_t i_hi = bcast64(rl.inc_hi), i_lo = bcast64(rl.inc_lo);
  int g0 = 0;
  for (int guard = 0; g0 < ng; ++guard) {
    if (guard > 2 * MAXG + 8) { if (lane == 0) set_err(x, E_UNREACHABLE); break; }   // every batch advances g0
    const unsigned long long t0 = gstat ? clock64() : 0;
    const uint64_t s_hi = bcast64(rl.s_hi), s_lo = bcast64(rl.s_lo);
    const uint32_t has0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.has32), u0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.u32);
    // (1) the window: output p (0-based) is computed on lane p & 63 (set p >> 6) and stored at win[p]
    uint64_t h0, l0, h1, l1;
    pcg_ahead(g_pcg_jump[lane + 1], s_hi, s_lo, i_hi, i_lo, &h0, &l0);
    pcg_ahead(g_pcg_jump[WAVE], h0, l0, i_hi, i_lo, &h1, &l1);
    __syncthreads();           // (one wave per block) the previous batch's window reads are done
    win[lane] = pcg_output(h0, l0);
    win[WAVE + lane] = pcg_output(h1, l1);
    __syncthreads();
    // what the agents of this batch (lane k <-> agent g0 + k) bring along
    const int gi = g0 + lane;
    const bool in = gi < ng;
    const uint32_t my_act = in ? x.w->green_act[gi] : 2u;
    const bool active = my_act < 2;
    // what the action reads from the state (green_prepare: the host's service table -- HBM here --, allowed server counts)
    const uint64_t pre = active ? green_prepare(x, gi, (int)my_act) : 0ull;
    const uint32_t gh = in ? s->green_host[gi] : 0u;
    uint32_t my_blk = 0;       // bit sn: traffic between the agent's subnet and subnet sn is blocked either way
    if (active && my_act == 0) {
      const int own = h_subnet((int)gh);
#pragma unroll
      for (int sn = 0; sn < NSUB - 1; ++sn) if (((s->blocks[own] >> sn) | (s->blocks[sn] >> own)) & 1u) my_blk |= 1u << sn;
    }
    const uint32_t lw_am = green_lw_active(pre);
    // what decides an agent's draw COUNT, reduced to shifts: the leading 32-bit draws (a = 1 or 2; with two, the first one
    // picks -- a server / a service -- out of `npick`), for GreenAccessService the picks that end on a blocked route (no
    // further draw), for GreenLocalWork the reliability (/20) of the pick
    const bool is_as = active && my_act == 0, is_lw = active && my_act == 1 && lw_am != 0;
    uint32_t npick = 0; uint64_t pickinfo = 0;
    if (is_as) {
      npick = (uint32_t)(pre >> 56);
#pragma unroll
      for (int sn = 0; sn < NSUB - 1; ++sn) {
        const uint32_t before = sn ? (uint32_t)((pre >> (8 * (sn - 1))) & 0xFF) : 0u, tot = (uint32_t)((pre >> (8 * sn)) & 0xFF);
        if (((my_blk >> sn) & 1u) && tot > before) pickinfo |= ((tot >= 64 ? ~0ull : (1ull << tot) - 1ull)) & ~((1ull << before) - 1ull);
      }
    } else if (is_lw) {
      npick = (uint32_t)popc32(lw_am);
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < MAXSV; ++i) if ((lw_am >> i) & 1u) { pickinfo |= ((pre >> (8 * i)) & 0x7Full) << (8 * cnt); ++cnt; }
    }
    const uint32_t lead = (is_as || is_lw) ? (npick > 1 ? 2u : 1u) : 0u;
    // (2) relaxation.  Per lane: d = outputs taken | 32-bit draws << 16 ; q = window index of the last output a 32-bit draw
    // fetched (its high half is numpy's buffered `uinteger`), -1: none
    uint32_t d_cur = 0, sp = 0, sh = 0, su = 0;
    int q_cur = -1;
    uint32_t inc = 0; int qinc = -1;    // inclusive scans of the last round
    int rounds = 0;
    bool stuck = false;
    for (;; ++rounds) {
      inc = wave_scan_add(d_cur); qinc = wave_scan_max(q_cur);
      const uint32_t ex = wave_shr1(inc, 0u);
      const int exq = (int)wave_shr1((uint32_t)qinc, 0xFFFFFFFFu);
      const int pos = (int)(ex & 0xFFFFu);
      const uint32_t has = (has0 + (ex >> 16)) & 1u;
      const uint32_t buf = exq < 0 ? u0 : (uint32_t)(win[exq < GW_OUT ? exq : GW_OUT - 1] >> 32);
      sp = (uint32_t)pos; sh = has; su = buf;
      uint32_t d_new = 0; int q_new = -1;
      if (lead) {
        const uint64_t f0 = win[pos < GW_OUT ? pos : GW_OUT - 1], f1 = win[pos + 1 < GW_OUT ? pos + 1 : GW_OUT - 1];
        const uint32_t first = has ? buf : (uint32_t)f0, second = has ? (uint32_t)f0 : (uint32_t)(f0 >> 32);
        uint32_t i = (has && lead == 1) ? 0u : 1u, n32 = lead;
        uint32_t has2 = has ^ (lead & 1u);
        if (i) q_new = pos;
        const uint32_t pick = lead == 2 ? (uint32_t)(((uint64_t)first * npick) >> 32) : 0u;
        if (is_as) {
          if (!((pickinfo >> pick) & 1ull)) ++i;                       // not blocked: the 1 % connection-event roll
        } else {
          const uint32_t roll = (uint32_t)(((uint64_t)(lead == 2 ? second : first) * 100u) >> 32);
          if (roll < (uint32_t)((pickinfo >> (8 * pick)) & 0xFF) * 20u) {
            const uint64_t u1 = i ? f1 : f0;
            ++i;
            if ((u1 >> 11) <= P01_FLOOR) { ++n32; if (!has2) { q_new = pos + (int)i; ++i; } has2 ^= 1u; }   // the false-positive event's port
            ++i;                                                        // the phishing roll
          }
        }
        d_new = i | (n32 << 16);
      }
      const bool moved = d_new != d_cur || q_new != q_cur;
      d_cur = d_new; q_cur = q_new;
      if (!__ballot(moved)) break;          // the estimates the lanes just used were the fixed point
      if (rounds > WAVE + 2) { stuck = true; break; }   // cannot happen (lane k is exact after k + 1 rounds): serial walk
    }
    // the agents' actions in full, from the starts found
    uint32_t rec = 0;
    if (active) {
      const int pos = (int)sp;
      uint32_t has = sh, buf = su;
      int q_new = -1;
      uint64_t f[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) { const int p = pos + j; f[j] = win[p < GW_OUT ? p : GW_OUT - 1]; }
      int i = 0; uint32_t n32 = 0; bool hard = false;
      auto fetch = [&]() { uint64_t v = f[0]; if (i == 1) v = f[1]; if (i == 2) v = f[2]; if (i == 3) v = f[3]; if (i >= 4) v = f[4]; ++i; return v; };
      auto take32 = [&]() { ++n32; if (has) { has = 0; return buf; } const uint64_t o = fetch(); buf = (uint32_t)(o >> 32); q_new = pos + i - 1; has = 1; return (uint32_t)o; };
      auto below = [&](uint32_t n) { if (n <= 1) return 0u; const uint64_t m = (uint64_t)take32() * n; if ((uint32_t)m < n) hard = true; return (uint32_t)(m >> 32); };
      uint32_t r = GR_VALID;
      if (my_act == 0) {       // green_access_service
        const int c = (int)below((uint32_t)(pre >> 56));
        int sn;
        const int dest = green_as_dest(pre, c, &sn);
        const uint32_t p = below(EPH_RANGE);
        r |= GR_EPH | ((uint32_t)dest << 8) | (p << 16);
        if ((my_blk >> sn) & 1u) r |= GR_FAIL | GR_CONN;
        else if ((fetch() >> 11) <= P01_FLOOR) r |= GR_CONN;
      } else if (!lw_am) r |= GR_FAIL;   // green_local_work
      else {
        const int c = nth_bit(lw_am, (int)below((uint32_t)popc32(lw_am)));
        const int rel = (int)((pre >> (8 * c)) & 0x7F) * 20;
        if ((int)below(100) >= rel) r |= GR_FAIL;
        else {
          if ((fetch() >> 11) <= P01_FLOOR) { const uint32_t p = below(EPH_RANGE); r |= GR_EPH | GR_PROC | (gh << 8) | (p << 16); }
          if ((fetch() >> 11) <= P01_FLOOR) r |= GR_PHISH;
        }
      }
      // a Lemire re-draw, a window that may not cover this agent, or (never) a count that differs from the relaxation's
      if (hard || pos + 5 > GW_OUT || ((uint32_t)i | (n32 << 16)) != d_cur || q_new != q_cur) r |= GR_HARD;
      rec = r;
    }
    if (stuck) rec |= GR_HARD;
    // where the batch ends: in front of the first agent the serial code has to resolve, behind the first phishing email
    const int cnt = (ng - g0) < WAVE ? (ng - g0) : WAVE;
    int kend = cnt;
    enum { R_NEXT, R_HARD, R_PHISH } reason = R_NEXT;
    const uint64_t m_hard = __ballot((rec & GR_HARD) != 0), m_phish = __ballot((rec & GR_PHISH) != 0);
    if (m_hard) { const int k = __ffsll((unsigned long long)m_hard) - 1; if (k < kend) { kend = k; reason = R_HARD; } }
    if (m_phish) { const int k = __ffsll((unsigned long long)m_phish) - 1; if (k < kend) { kend = k + 1; reason = R_PHISH; } }
    const unsigned long long t1 = gstat ? clock64() : 0;
    // (3) ports: one atomic test-and-set per agent; the first agent that finds its port taken ends the batch in front of it
    const uint32_t eh = (rec >> 8) & 0xFFu, ep = (rec >> 16) & 0x3FFFu;
    bool coll = false;
    const bool has_port = lane < kend && (rec & GR_EPH);
    if (has_port) coll = eph_test_and_set(x.c, (int)eh, ep);
    const uint64_t cm = __ballot(coll);
    if (cm) {
      const int kc = __ffsll((unsigned long long)cm) - 1;
      if (has_port && !coll && lane >= kc) __hip_atomic_fetch_and(&x.c->eph[eh][ep >> 5], ~(1u << (ep & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      kend = kc; reason = R_HARD;
    }
    if (lane < kend && (rec & GR_VALID)) {
      if (rec & GR_CONN) ev_or(x, (int)eh, EV_CUR_CONN);
      if (rec & GR_PROC) ev_or(x, (int)eh, EV_CUR_PROC);
      if (rec & GR_FAIL) __hip_atomic_fetch_add(&s->brm, reward_table(s->phase, h_subnet((int)gh), my_act == 0 ? RW_ASF : RW_LWF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the stream behind the committed agents = where agent kend starts
    int pos_e = 0; uint32_t has_e = has0, u_e = u0;
    if (kend > 0) {
      const uint32_t ie = rdlane(inc, kend - 1);
      const int qe = (int)rdlane((uint32_t)qinc, kend - 1);
      pos_e = (int)(ie & 0xFFFFu); has_e = (has0 + (ie >> 16)) & 1u;
      if (qe >= 0) u_e = (uint32_t)(win[qe] >> 32);
    }
    uint64_t n_hi = s_hi, n_lo = s_lo;
    if (pos_e > 0) {
      const int j = (pos_e - 1) & 63;
      n_hi = lane64(h0, j); n_lo = lane64(l0, j);
      if (pos_e > WAVE) { n_hi = lane64(h1, j); n_lo = lane64(l1, j); }
    }
    const unsigned long long t2 = gstat ? clock64() : 0;
    const int gend = g0 + kend;
    if (lane == 0) {
      rl.s_hi = n_hi; rl.s_lo = n_lo; rl.has32 = has_e; rl.u32 = u_e; rl.ndraw += (uint32_t)pos_e;
      if (reason == R_PHISH) phishing(x, s->green_host[gend - 1]);
      else if (reason == R_HARD) {
        s->brm += step_green_exec(x, gend);
        if (bit_get(x.w->phish_mask, gend)) { bit_clr(x.w->phish_mask, gend); phishing(x, s->green_host[gend]); }
      }
    }
    if (gstat && lane == 0) {
      gstat[0] += 1; gstat[1] += reason == R_PHISH; gstat[2] += m_hard != 0; gstat[3] += cm != 0;
      gstat[4] += t1 - t0; gstat[5] += t2 - t1; gstat[6] += clock64() - t2; gstat[7] += rounds + 1;
    }
    g0 = reason == R_HARD ? gend + 1 : gend;
  }
}

// One step of one episode of the numpy-stream mode on one wavefront: the body of k_step and of the persistent kernel k_run_pcg
// (there a.rand_t / a.full_obs are the item's: set by the caller).
// first / last: as in philox1_body -- inside a run of steps of one episode on one wave the agent part stays in LDS
template <bool LOG>
__device__ __forceinline__ void pcg_body(StepArgs a, const int e, const int lane, const bool first = true, const bool last = true) {
  // numpy-PCG64 mode: one shared stream => the agent walk is strictly serial (lane 0); only the RNG-free parts
  // (row staging, end-turn Monitor roll-over over the 137 hosts, observation encode) use the other lanes.
  extern __shared__ uint4 lds[];
  // one LDS area, two lives: the LCG window of the green actions (wave_green_exec), then -- from the end-turn roll-over on --
  // the hosts' event bits (what the observation encode reads) and the encoded observation
  __shared__ uint64_t win_lds[GW_OUT];
  constexpr int OBS_LDS = (OBS_TOTAL + 2 + 7) & ~7;
  static_assert(OBS_LDS <= (int)sizeof(uint64_t) * GW_OUT, "the byte copy of the observation fits the window area");
  uint8_t* const obs_lds = reinterpret_cast<uint8_t*>(win_lds);
  __shared__ int ok_lds;
  __shared__ StepWork work;
  EnvCold* const cold_e = cold_at(a.cold, (size_t)e, cold_row_bytes(a.steps));
  unsigned long long t_begin = a.prof ? clock64() : 0;
  const uint4* src = reinterpret_cast<const uint4*>(a.st + e);
  if (first) stage_in<HOT_VEC>(lds, src, lane);
  for (int i = lane; i < (int)(sizeof(StepWork) / 4); i += WAVE) reinterpret_cast<uint32_t*>(&work)[i] = 0;
  __syncthreads();
  EnvState* s = reinterpret_cast<EnvState*>(lds);   // only the part in front of EnvState.hd is valid here
  HostDyn* const hd = a.st[e].hd;                   // the host table stays in HBM / L2
  __shared__ unsigned long long prof_lds[16];   // phase counters accumulate in LDS, flushed once at the end
  unsigned long long* prof = a.prof ? prof_lds : nullptr;
  if (prof && lane < 16) prof_lds[lane] = 0;
  // the shared numpy stream is walked on a register copy (this kernel serves the PCG mode only; mode pinned so the Philox
  // paths fold away) and written back once, before the row leaves LDS
  Rng rl = s->rng;
  rl.mode = 0;
  rl.pad = 0;
  Ctx x{s, cold_e, &rl, hd, &work, lane == 0 ? prof : nullptr};
  x.lg = (LOG && cold_e->evlog.enabled) ? &cold_e->evlog : nullptr;
  const ExtAct* const xt = (LOG && a.ext) ? a.ext + (size_t)e * EXT_PER_ENV : nullptr;   // this episode's submitted red / green actions
  x.ext = xt;
  if (prof && lane == 0) prof[11] += clock64() - t_begin;
  const bool do_reset = a.autoreset && s->done;
  // The ordered walk is lane 0's; between its stretches the whole wave does what needs no order: the two draw-only phases
  // (green policy draws, action-order shuffle) straight from the LCG's closed form, and the green actions' state reads.
  // every lane evaluates the mission-phase check (four words of the row); the accumulators of the step were left initialised by
  // step_end / the reset, so the blue submissions (lanes 1..5: in-kernel action draw, decode, queue) run beside lane 0's step_phase
  const bool step_ok = !do_reset && step_phase_of(s->step_count, s->phase_len[0], s->phase_len[1], s->phase_len[2]) >= 0;
  if (lane == 0) {
    ok_lds = step_ok ? 1 : 0;
    if (do_reset) {
      env_reset(x, 0, 0, a.steps, true, a.policy, a.topo);   // new episode, same stream (CybORG.reset(seed=None)); this kernel serves the numpy-stream mode only
    } else {
      CC4_TICK0(x);
      (void)step_phase(x, false);    // sets E_STEP_PAST_END when !step_ok
      CC4_TICK(x, 0);
      if (step_ok) rng_policy_swap(x, false);     // CybORG.set_seed split: the policies draw from the old stream (EnvCold.rng2)
      if (step_ok && (s->policy & BP_RANDOM_BIT))   // built-in blue policy: its draws are the first of the step, in agent order
        for (int b = 0; b < NBLUE; ++b) {
          int32_t act = a.actions ? a.actions[e * NBLUE + b] : -1;
          if (a.rand_out) { act = random_blue_action(a.rand_seed0, a.rand_t, e, b); a.rand_out[e * NBLUE + b] = act; }
          step_blue_submit(x, b, act);
        }
    }
  } else if (step_ok && lane <= NBLUE && !(s->policy & BP_RANDOM_BIT)) {
    const int b = lane - 1;
    int32_t act = a.actions ? a.actions[e * NBLUE + b] : -1;
    if (a.rand_out) { act = random_blue_action(a.rand_seed0, a.rand_t, e, b); a.rand_out[e * NBLUE + b] = act; }
    Ctx xb{s, cold_e, &rl, hd, &work};
    step_blue_submit(xb, b, act);
  }
  __syncthreads();
  if (ok_lds) {
    bool drawn = false;
    // (with submitted green actions in play the agents that have one do not draw: the walking lane asks them one by one)
    if (!(s->policy & GP_SLEEP_BIT) && !xt) drawn = wave_green_policy(rl, s->n_green, work.green_act, lane);
    // the observation half of the six red policies draws nothing and touches only its own agent: side by side on six lanes
    if (lane < NRED) { Ctx xo{s, cold_e, &rl, hd, &work}; xo.ext = xt; step_red_observe(xo, lane); }
    __syncthreads();
    if (lane == 0) {
      if (!drawn) for (int g = 0; g < s->n_green; ++g) step_green_policy(x, g);   // SleepAgent greens, or the 2^-32 re-draw case
      CC4_TICK(x, 1);
      for (int r = 0; r < NRED; ++r) s->n_actions -= step_red_policy_tick(x, r, (s->policy & 3) != RP_RANDOM);
      rng_policy_swap(x, true);
      CC4_TICK(x, 2);
      for (int b = 0; b < NBLUE; ++b) step_tick_blue(x, b);
      CC4_TICK(x, 3);
    }
    __syncthreads();
    wave_shuffle_consume(rl, s->n_actions, lane);   // sort_action_order's shuffle (SC:398-464) only consumes the stream
    if (lane == 0) { CC4_TICK(x, 4); step_blue_exec(x, true); }
  }
  __syncthreads();
  if (ok_lds) {
    // the green actions: across the wave (wave_green_exec)
    if constexpr (!LOG) wave_green_exec(x, rl, win_lds, lane, a.prof ? a.prof + PROF_SLOTS * (size_t)e + 64 : nullptr);
    else {
      // with the event log on (log entries are ordered): on the walking lane; what the actions read from the state (service
      // tables of their hosts -- HBM here --, allowed server counts) is prepared for all agents at once on the idle lanes
      __shared__ uint64_t gpre_lds[MAXG];
      for (int g = lane; g < s->n_green; g += WAVE) { const int act = work.green_act[g]; if (act < 2) gpre_lds[g] = green_prepare(x, g, act); }
      __syncthreads();
      if (lane == 0) {
        Ctx xg = x; xg.gpre = gpre_lds;
        for (int g = 0; g < s->n_green; ++g) {
          s->brm += step_green_exec(xg, g);
          if (bit_get(work.phish_mask, g)) { bit_clr(work.phish_mask, g); phishing(x, s->green_host[g]); }
        }
      }
    }
    if (lane == 0) {
      CC4_TICK(x, 6);
      step_red_exec(x);
      step_reassign(x, red_foreign_agents(s));
    }
  }
  __syncthreads();
  if (ok_lds) {
    // end-turn Monitor roll-over: the hosts' event bytes are part of the staged row (EnvState.hev)
    for (int h = lane; h < MAXH; h += WAVE) s->hev[h] = monitor_roll(h, s->hev[h]);
    if (lane == 0) step_monitor_pend(x);
    __syncthreads();
    {
      // end-turn RedSessionCheck: it draws only when it has to promote a session to primary; when no agent needs that (the
      // usual case) the six checks run side by side, else in order on the walking lane
      const bool need = lane < NRED && rsc_draws(s, lane);
      const bool serial = __ballot(need) != 0ull;
      if (!serial && lane < NRED) { Ctx xc{s, cold_e, &rl, hd, &work, nullptr, nullptr, x.lg}; step_rsc(xc, lane); }
      if (lane == 0) CC4_TICK(x, 9);
      __syncthreads();
      if (lane == 0) {
        if (serial) for (int r = 0; r < NRED; ++r) step_rsc(x, r);
        CC4_TICK(x, 10);
        step_end(x, a.msgs ? a.msgs + e * NBLUE * MSG_LEN : nullptr);
      }
    }
    __syncthreads();
  }
  if (lane == 0) { s->rng = rl; a.reward[e] = s->reward; a.done[e] = s->done; a.err[e] = s->err; }
  unsigned long long t_obs = a.prof ? clock64() : 0;
  {
    // straight to HBM, kind-sorted (uniform branches); the output buffer persists between steps, so the values that only a
    // Block/Allow or a new mission phase changes are written when that happened (EnvState.obs_dirty), after a reset, or when the
    // caller asks -- as in the counter-mode kernels; the byte copy in LDS only feeds the packed exchange row
    int32_t* o = a.obs + (size_t)e * OBS_TOTAL;
    const bool pack = a.obs8 != nullptr;
    const int nv = (do_reset || a.full_obs || pack || s->obs_dirty) ? OBS_TOTAL : OBS_FAST;
    encode_obs_fast<WAVE>(s, o, obs_lds, pack, lane);
    for (int v = OBS_FAST + lane; v < nv; v += WAVE) { int i; const int val = env_flat_obs_sorted(s, v, &i); o[i] = val; if (pack) obs_lds[i] = (uint8_t)val; }
  }
  __syncthreads();
  unsigned long long t_out = a.prof ? clock64() : 0;
  if (prof && lane == 0) prof[12] += t_out - t_obs;
  uint4* dst = reinterpret_cast<uint4*>(a.st + e);
  if (last) stage_out<HOT_VEC>(dst, lds, lane);
  if (a.obs8) store_packed_row(a.obs8 + (size_t)e * OBS_PACKED, obs_lds, lane, WAVE);
  if (prof && lane == 0) { prof[13] += clock64() - t_out; prof[14] += clock64() - t_begin; }
  if (prof) { __syncthreads(); if (lane < 15) a.prof[PROF_SLOTS * (size_t)e + lane] += prof_lds[lane]; }
}
template <bool LOG>
__global__ __launch_bounds__(WAVE) void k_step(StepArgs a) {
  const int e = a.e0 + (int)blockIdx.x;
  if (e >= a.n) return;
  pcg_body<LOG>(a, e, (int)threadIdx.x);
}

// ---------------------------------------------------------------- Philox mode: wave- and lane-parallel step
// Same phase bodies as the serial walk (cc4_engine.h P0..P9), different schedule.  One block of 4 wavefronts per episode:
// every (agent, phase) owns a Philox counter stream, so heterogeneous agents can run concurrently.  Work that differs in
// control flow goes to different WAVES (a wave executes divergent lanes one after the other): the 6 red FSM policies with
// their queue ticks, the 5 blue actions (disjoint zones), the 6 red actions (those naming the same host are held back and
// run in order on thread 0), the 6 RedSessionChecks (agent r -> wave r % 4, lane r / 4), and the two green action types
// (AccessService / LocalWork lists built with LDS counters).  Work that is uniform goes to LANES: row staging, green agents
// within a type, the 137 Monitor roll-overs, the observation encode (enumerated kind by kind).  Measured on MI355X (r01): 4 waves per
// episode is the build; 5 and 6 (fewer red agents sharing a wave) run 25-30 % slower at 1024 episodes, DESIGN.md 7.
// Cross-thread effects are event-bit ORs and the reward sum (LDS atomics); the rare order-dependent spawns (PhishingEmail,
// cross-subnet session reassignment) are collected and replayed by thread 0 in agent order.
#ifndef CC4_PW
#define CC4_PW 4
#endif
constexpr int PW = CC4_PW;         // waves per episode block; red agent r runs on wave r % PW, lane r / PW
static_assert(PW >= 4 && PW <= 8, "waves 0/1 run the two green action lists, waves PW-2 and PW-1 the green draws, wave PW-1 the blue submissions");
constexpr int PT = PW * WAVE;      // threads per episode block (256)

__device__ __forceinline__ void stage_in_n(uint4* __restrict__ lds, const uint4* __restrict__ src, int tid) {
  constexpr int U = (ROW_VEC / PT) < 6 ? (ROW_VEC / PT) : 6;   // loads in flight per thread (the whole row in one or two rounds)
  int i = tid;
  for (; i + (U - 1) * PT < ROW_VEC; i += U * PT) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * PT];
#pragma unroll
    for (int u = 0; u < U; ++u) lds[i + u * PT] = v[u];
  }
  for (; i < ROW_VEC; i += PT) lds[i] = src[i];
}

// The host table (EnvState.hd, two thirds of the row) is not read before the first action executes.  Its 16-byte vectors
// go HBM -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave-instruction, no staging registers) and stay in flight
// while the rest of the row is staged through registers and the policy phase runs; `dma_wait` drains them before the
// barrier that precedes the first action.  The instruction is issued from inline asm: the compiler would otherwise put a
// vmcnt(0) in front of every LDS read that might alias the DMA destination, i.e. right away.
constexpr int HD_V0 = (int)((offsetof(EnvState, hd) + 15) / 16);                             // first 16-byte vector fully inside hd
constexpr int HD_CHUNKS = (int)(((offsetof(EnvState, hd) + sizeof(HostDyn) * MAXH) / 16 - HD_V0) / 64);   // whole 64-vector chunks
constexpr int HD_V1 = HD_V0 + 64 * HD_CHUNKS;                                                // one past the DMA'd range
__device__ __forceinline__ void dma_chunk(const uint4* gsrc_lane, uint4* lds_chunk_base) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_chunk_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc_lane), "s"(dst) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// MINW = minimum waves per SIMD the register allocation must allow (= resident episode blocks per CU).  1 lets the compiler
// take what it likes (86 VGPRs, 106 SGPRs: 5 blocks per CU) and is the fastest single block: the build for batches that fit
// the chip in one round of <= 5 blocks per CU (the per-GPU share of an 8-GPU job).  7 (72 VGPRs, a dozen spills) and 8 (64
// VGPRs, ~30 spills) keep more blocks resident: the builds for larger, throughput-bound batches.  cc4_create picks per batch
// size.  Measured on MI355X (r02, M agent-env steps/s, MINW 1 / 7 / 8): 1536 episodes 185 / 229 / 215, 2048: 222 / 233 / 261
// (exactly one round of 8), 3072: 253 / 295 / 281, 4096: 275 / 320 / 318, 8192: 320 / 393 / 389, 16384: 332 / 414 / 399;
// 1024 episodes: 167 with MINW 1 vs 156 with 8.
#ifndef CC4_PHILOX_BIG_MINW
#define CC4_PHILOX_BIG_MINW 7
#endif
// one step of one episode on a block of four wavefronts: the body of k_step_philox and of its multi-step form k_run_philox
// RUN (k_run_philox): the row stays in LDS from one step of the episode to the next -- run_flags bit 0: not the first step of the
// launch (nothing is staged in), bit 1: the last one (the whole row goes back; before it, none of it)
template <bool LOG, bool RUN = false>
// obs_row (RUN with the exchange): a byte row of the caller's in LDS that receives all 578 observation values of the step -- the caller packs
// and stores the exchange row from it behind its own end-of-step drain
__device__ __forceinline__ void philox4_body(StepArgs a, const int run_flags = 0, const int tid_in = -1, uint8_t* const obs_row = nullptr) {
  extern __shared__ uint4 lds[];
  __shared__ int conflict_lds;
  __shared__ alignas(16) uint32_t reset_ws[RESET_WS_WORDS];   // pid bitmaps of the scenario generation (autoreset); during a step: the green agents' pre-computed blocks
  static_assert(RESET_WS_WORDS >= 4 * (MAXG + NRED + NBLUE), "one 16-byte block per green agent, red action stream and blue action stream");
  __shared__ int glist_n[2][2];       // [action type][drawing wave]
  __shared__ StepWork work;
  __shared__ uint8_t obs_bytes_own[OBS_TOTAL + 2];   // byte-per-value copy of the observations, only for the packed exchange row
  uint8_t* const obs_bytes = obs_row ? obs_row : obs_bytes_own;
  __shared__ uint8_t glist[2][2][MAXG];  // green agents by action type (0 AccessService, 1 LocalWork) and drawing wave
  __shared__ unsigned long long prof_lds[16];
  const int e = a.e0 + (int)blockIdx.x, tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: kept in an SGPR
  if (e >= a.n) return;
  EnvCold* const cold_e = cold_at(a.cold, (size_t)e, cold_row_bytes(a.steps));
  unsigned long long t_begin = a.prof ? clock64() : 0;
  const uint4* src = reinterpret_cast<const uint4*>(a.st + e);
  // the part outside the host table through registers (3 vectors per thread), then the host-table chunks by DMA
  constexpr int NA = HD_V0 + ROW_VEC - HD_V1;   // indexed 0..NA-1: [0,HD_V0) then [HD_V1,ROW_VEC)
  constexpr int NA_U = (NA + PT - 1) / PT;
  if (!RUN || !(run_flags & 1)) {
    {
      uint4 va[NA_U];
#pragma unroll
      for (int u = 0; u < NA_U; ++u) { int k = tid + u * PT; k = k < NA ? k : NA - 1; int i = k < HD_V0 ? k : k - HD_V0 + HD_V1; va[u] = src[i]; }
#pragma unroll
      for (int u = 0; u < NA_U; ++u) { int k = tid + u * PT; int i = k < HD_V0 ? k : k - HD_V0 + HD_V1; if (k < NA) lds[i] = va[u]; }
    }
    for (int c = wave; c < HD_CHUNKS; c += PW) dma_chunk(src + HD_V0 + 64 * c + lane, lds + HD_V0 + 64 * c);
  }
  unsigned long long* prof = a.prof ? prof_lds : nullptr;
  if (prof && tid < 16) prof_lds[tid] = 0;
  if (tid < 4) (&glist_n[0][0])[tid] = 0;
  if (tid == 0) conflict_lds = 0;
  if (tid >= 64 && tid < 64 + 4 + NRED) (&work.phish_mask[0])[tid - 64] = 0;   // phish_mask[4] and pend_r[NRED] are adjacent
  static_assert(offsetof(StepWork, pend_r) == offsetof(StepWork, phish_mask) + 16, "phish_mask and pend_r are cleared as one run of words");
  if (tid >= 128 && tid < 128 + 5) work.hdirty[tid - 128] = 0;
  __syncthreads();
  EnvState* s = reinterpret_cast<EnvState*>(lds);
  HostDyn* const hd = s->hd;
  if (prof && tid == 0) prof[11] += clock64() - t_begin;
  const bool do_reset = a.autoreset && s->done;
  if (do_reset) {
    dma_wait();
    __syncthreads();
    // new episode on the same key (CybORG.reset(seed=None)): the phases of env_reset_counter_mode, hosts on threads
    reset_zero(s, hd, cold_e, tid, PT);
    __syncthreads();
    Rng rr; ResetCarry carry; carry.env_key = 0;     // thread 0: main reset stream in registers, across the phases
    Ctx xm{s, cold_e, &rr, hd, &work};
    if (tid == 0) { rr = s->rng; rr.mode = 1; carry = reset_topology(xm, 0, a.steps, true, a.policy, a.topo, reset_ws, true); }
    __syncthreads();
    Rng rh; rng_fork(&rh, &s->rng, ST_GEN_HOST); rh.mode = 1;
    Ctx xh{s, cold_e, &rh, hd, &work};
    if (tid < MAXH) reset_gen_host(xh, tid);
    __syncthreads();
    if (tid == 0) { reset_pid_serial(xm, reset_used_set(s)); reset_agents(xm); }      // pid uniqueness in the reference's order (one thread; once per episode)
    __syncthreads();
    reset_used_clear(s, tid, PT);
    __syncthreads();
    if (tid < MAXH) reset_host_sessions(xh, tid);
    __syncthreads();
    if (tid == 0) { reset_finish(xm, carry, a.steps, a.topo, true); a.reward[e] = s->reward; a.done[e] = s->done; }
    __syncthreads();
  } else {
    // the mission phase of this step, evaluated by every thread (four words of the row); thread 0 alone stores what
    // step_phase stores -- nothing the policy phase reads, and the step's accumulators were left initialised by step_end --
    // so no barrier follows
    const int st_now = s->step_count;
    const bool step_ok = step_phase_of(st_now, s->phase_len[0], s->phase_len[1], s->phase_len[2]) >= 0;
    if (tid == 0) {
      Ctx x{s, cold_e, &s->rng, hd, &work, prof};
      x.lg = (LOG && cold_e->evlog.enabled) ? &cold_e->evlog : nullptr;
      x.ext = (LOG && a.ext) ? a.ext + (size_t)e * EXT_PER_ENV : nullptr;
      CC4_TICK0(x);
      (void)step_phase(x, false);
    }
    if (step_ok) {
      Ctx x0p{s, cold_e, nullptr, hd, &work, tid == 0 ? prof : nullptr};
      const int ng = s->n_green;
      // one thread-private generator per thread, in registers: every use starts with rng_set_stream(), which fully
      // determines the stream from (key, step, episode, stream id); mode pinned so the PCG paths fold away
      if (tid == 0) CC4_TICK(x0p, 0);   // slot 0: step_phase
      Rng rl;
      rng_fork(&rl, &s->rng, ST_RESET);
      rl.mode = 1;
      rng_begin_step(&rl, (uint32_t)st_now);   // not read from the row: thread 0 may still be storing it there
      EvLog* const lg = (LOG && cold_e->evlog.enabled) ? &cold_e->evlog : nullptr;
      const ExtAct* const xt = (LOG && a.ext) ? a.ext + (size_t)e * EXT_PER_ENV : nullptr;   // this episode's submitted red / green actions
      Ctx x0{s, cold_e, &rl, hd, &work, tid == 0 ? prof : nullptr};         // thread 0
      x0.lg = lg; x0.ext = xt;
#ifndef CC4_RED_WAVES
#define CC4_RED_WAVES 2
#endif
      constexpr int RW = CC4_RED_WAVES;                                           // red agent r on wave r % RW, lane r / RW
      const int ragent = lane * RW + wave;
      const bool is_red = wave < RW && lane < (NRED + RW - 1) / RW && ragent < NRED;
      unsigned long long* ap = (a.prof && is_red) ? a.prof + PROF_SLOTS * (size_t)e + 16 + 8 * ragent : nullptr;
      // The red actions and the RedSessionChecks run in phases in which all four waves are free: agent r on wave r % 4, lane
      // r / 4 (two waves carry two agents, two carry one).  Measured on MI355X (r03, three launches per step; M agent-env steps/s
      // with the agents on 2 / 3 / 4 waves in these phases): 1024 episodes 177.4 / 179.3 / 180.8, 2048: 297.2 / 309.1 / 312.1,
      // 4096: 440 / 444 / 443.  (The policy phase stays on two waves, three agents side by side: its other two waves carry the
      // blue submissions and the green draws; all four there: 176.8 / 293.3 / 426.7.)
#ifndef CC4_RED_WAVES_EXEC
#define CC4_RED_WAVES_EXEC 4
#endif
      constexpr int RWX = CC4_RED_WAVES_EXEC;
      const int xagent = lane * RWX + wave;
      const bool is_redx = wave < RWX && lane < (NRED + RWX - 1) / RWX && xagent < NRED;
      unsigned long long* apx = (a.prof && is_redx) ? a.prof + PROF_SLOTS * (size_t)e + 16 + 8 * xagent : nullptr;
      Ctx xrx{s, cold_e, &rl, hd, &work, nullptr, apx, lg};
      Ctx xr{s, cold_e, &rl, hd, &work, nullptr, ap, lg};
      xrx.ext = xt; xr.ext = xt;
      // ---- P0-P3a: every agent's policy / submission followed by its own duration-queue tick (SC:236-265), all on the
      // agent's thread: red r on wave r%PW lane r/PW, blue on wave PW-1 lanes 2..6, green draws on lanes >= 8 of the waves
      // that carry a single red agent.  A tick touches only its own agent (queue, observation reset, filter_actions
      // against its own session table, which no other agent edits before the barrier below).

      if (is_red) {
        unsigned long long t0 = ap ? clock64() : 0;
        const int dropped = step_red_policy_tick(xr, ragent);
        if (ap) ap[0] += clock64() - t0;
        if (dropped) atomicSub(&s->n_actions, 1);
      }
      else if (wave == PW - 1 && lane >= 2 && lane < 2 + NBLUE) {
        const int b = lane - 2;
        int32_t act = a.actions ? a.actions[e * NBLUE + b] : -1;
        if (a.rand_out) { act = random_blue_action(a.rand_seed0, a.rand_t, e, b); a.rand_out[e * NBLUE + b] = act; }
        step_blue_submit(x0, b, act);
        step_tick_blue(x0, b);
        step_messages(s, a.msgs ? a.msgs + e * NBLUE * MSG_LEN : nullptr, b);   // read back by this step's observation encode only
        // block 0 of the agent's action stream, for the lane that will resolve the action
        { uint32_t c[4]; rng_block(&rl, ST_BLUE_EXE + (uint32_t)b, 0, c); reinterpret_cast<uint4*>(reset_ws)[MAXG + NRED + b] = make_uint4(c[0], c[1], c[2], c[3]); }
      }
      else if (wave == PW - 2 && lane >= 1 && lane <= NRED) {   // block 0 of the six red action streams, side by side on idle lanes of a wave with slack
        uint32_t c[4]; rng_block(&rl, ST_RED_EXE + (uint32_t)(lane - 1), 0, c); reinterpret_cast<uint4*>(reset_ws)[MAXG + lane - 1] = make_uint4(c[0], c[1], c[2], c[3]);
      }
      else if (lane >= 8 && wave >= PW - 2) {
        static_assert((RW <= PW - 2 || RW >= NRED) && MAXG <= 2 * (WAVE - 8), "every green agent has its own lane (8..63) on one of the last two waves, which carry no red agent or one on lane 0: one pass, one ballot per type");
        const int gw = wave - (PW - 2);
        const int g = gw * (WAVE - 8) + (lane - 8);
        if (g < ng) {
          Ctx xg{s, cold_e, &rl, hd, &work, nullptr, nullptr, lg};
          xg.ext = xt;
          step_green_policy(xg, g);
          const int t = work.green_act[g];
          // compaction by action type with a wavefront ballot + prefix count (agent order, no LDS atomics): each drawing wave
          // fills its own sub-list; the resolving wave walks the two sub-lists one after the other
          const unsigned long long m0 = __ballot(t == 0), m1 = __ballot(t == 1);
          const unsigned long long below = (1ull << lane) - 1ull;
          if (t < 2) {
            const unsigned long long m = t == 0 ? m0 : m1;
            glist[t][gw][__popcll(m & below)] = (uint8_t)g;
            if ((m & below) == 0) glist_n[t][gw] = __popcll(m);      // the first lane of the type publishes the count
            // the first block of the agent's action stream, computed here -- behind the red policies -- and handed to the
            // lane that resolves the action (the generation work area is idle during a step)
            uint32_t c[4];
            rng_block(&rl, ST_GREEN_EXE + (uint32_t)g, 0, c);
            reinterpret_cast<uint4*>(reset_ws)[g] = make_uint4(c[0], c[1], c[2], c[3]);
          }
        }
      }
      if (a.prof && lane == 63) a.prof[PROF_SLOTS * (size_t)e + 100 + wave] += clock64() - t_begin;   // debug: when each wave reaches the end of the policy phase
      dma_wait();          // the host table has landed in LDS behind the policy phase
      __syncthreads();
      CC4_TICK(x0, 2);
      // ---- P3b blue execution
      if (blue_exec_independent(s)) {      // uniform: every thread reads the same five action types
        if (tid == 0) CC4_TICK(x0, 3);
#ifndef CC4_BLUE_WAVES
#define CC4_BLUE_WAVES PW
#endif
        constexpr int BW = CC4_BLUE_WAVES;
        const int bagent = lane * BW + wave;                                      // blue agent b on wave b % BW, lane b / BW
        if (wave < BW && lane < (NBLUE + BW - 1) / BW && bagent < NBLUE) {
          Ctx xb{s, cold_e, &rl, hd, &work, nullptr, nullptr, lg};
          const uint4 blk = reinterpret_cast<const uint4*>(reset_ws)[MAXG + NRED + bagent];
          const uint32_t pre[4] = {blk.x, blk.y, blk.z, blk.w};
          const unsigned long long tb0 = a.prof ? clock64() : 0;
          step_blue_exec_agent(xb, bagent, pre);
          if (a.prof) { unsigned long long* tp = a.prof + PROF_SLOTS * (size_t)e + 108 + 2 * (s->bexec[bagent].type & 7); atomicAdd(tp, (unsigned long long)(clock64() - tb0)); atomicAdd(tp + 1, 1ull); }   // debug: blue action cycles by type
        }
        __syncthreads();
        if (tid == 0) CC4_TICK(x0, 5);
      } else {
        if (tid == 0) step_blue_exec(x0);
        __syncthreads();
      }
      // ---- P4 green actions: wave 0 = AccessService list, wave 1 = LocalWork list (uniform control flow per wave)
      if (wave < 2) {
        unsigned long long tg0 = a.prof ? clock64() : 0;
        int pen = 0;
        const int n0 = glist_n[wave][0], n1 = glist_n[wave][1];
        for (int i = lane; i < n0 + n1; i += WAVE) {
          int g = i < n0 ? glist[wave][0][i] : glist[wave][1][i - n0];
          Ctx xg{s, cold_e, &rl, hd, &work, nullptr, nullptr, lg};
          xg.ext = xt;
          const uint4 blk = reinterpret_cast<const uint4*>(reset_ws)[g];
          const uint32_t pre[4] = {blk.x, blk.y, blk.z, blk.w};
          pen += step_green_exec(xg, g, pre);
        }
        if (pen) atomicAdd(&s->brm, pen);
        if (a.prof && lane == 0) a.prof[PROF_SLOTS * (size_t)e + 96 + wave] += clock64() - tg0;   // debug: per-wave green action time
      }
      __syncthreads();
      CC4_TICK(x0, 6);
      // ---- P5 deferred phishing (ordered), then P6 red actions: one per wave when they name distinct hosts
      if (tid == 0) { step_phishing(x0); CC4_TICK(x0, 1); rs_reserve(x0); conflict_lds = (int)red_conflict_mask(s); if (prof && conflict_lds) prof[4] += 1000000; }
      __syncthreads();
      const uint32_t serial_red = (uint32_t)conflict_lds;
      if (is_redx && !((serial_red >> xagent) & 1u)) {
        unsigned long long t0 = apx ? clock64() : 0;
        const int ty = s->rexec[xagent].type;
        { const uint4 blk = reinterpret_cast<const uint4*>(reset_ws)[MAXG + xagent]; const uint32_t pre[4] = {blk.x, blk.y, blk.z, blk.w}; step_red_exec_agent(xrx, xagent, pre); }
        if (apx) { unsigned long long dt = clock64() - t0; apx[1] += dt; unsigned long long* tp = a.prof + PROF_SLOTS * (size_t)e + 64 + 2 * (ty & 15); atomicAdd(tp, dt); atomicAdd(tp + 1, 1ull); }
      }
      __syncthreads();
      if (serial_red) {   // same-host actions (and everything when some agent withdraws): agent order on thread 0
        if (tid == 0) for (int r = 0; r < NRED; ++r) if ((serial_red >> r) & 1u) step_red_exec_agent(x0, r);
        __syncthreads();
      }
      // ---- pid-event merge and reassignment on thread 0 (the foreign-session test is 5 words per agent); meanwhile P7, the
      // per-host roll-over of the end-turn Monitor, on all threads (host event flags: nothing the reassignment touches)
      if (tid == 0) {
        step_red_merge(x0);
        CC4_TICK(x0, 7);
        step_reassign(x0, red_foreign_agents(s));
      }
      for (int h = tid; h < MAXH; h += PT) step_monitor_host(x0, h);
      __syncthreads();
      CC4_TICK(x0, 9);
      // ---- P8 end-turn RedSessionCheck (one red agent per wave), and on the last thread the Monitor's sus-pid hand-over and
      // the step's bookkeeping: disjoint data (red agent tables / blue lists, counters, reward).  The observation encode below
      // reads none of it, so there is no barrier in between.
      if (is_redx) { unsigned long long t0 = apx ? clock64() : 0; step_rsc(xrx, xagent); if (apx) apx[2] += clock64() - t0; }
      if (tid == PT - 1) {
        Ctx xe{s, cold_e, &rl, hd, &work, nullptr, nullptr, lg};
        xe.ext = xt;
        step_monitor_pend(xe);
        step_end(xe, nullptr, false);
        a.reward[e] = s->reward; a.done[e] = s->done;
      }
      CC4_TICK(x0, 10);
    } else { dma_wait(); if (tid == 0) { a.reward[e] = s->reward; a.done[e] = s->done; } }
  }
  unsigned long long t_obs = a.prof ? clock64() : 0;
  // flat observations: one value per thread straight to HBM (int32 for the host API, bytes for the all-gather)
  {
    int32_t* o = a.obs + (size_t)e * OBS_TOTAL;
    const bool pack = a.obs8 != nullptr || obs_row != nullptr;   // the exchange copy goes through a byte row in LDS and is packed after the barrier below
    // the values that can change with every step (host events, messages) always; blocks, comms policy, subnet one-hots and phase
    // words only when the step changed them (EnvState.obs_dirty), after a reset, or when the caller asks (the buffer persists)
    const int nv = (do_reset || a.full_obs || a.obs8 || s->obs_dirty) ? OBS_TOTAL : OBS_FAST;      // (a caller's obs_row persists from step to step, like the int32 buffer: only what changed is rewritten)
    encode_obs_fast<PT>(s, o, obs_bytes, pack, tid);
    for (int v = OBS_FAST + tid; v < nv; v += PT) { int i; int val = env_flat_obs_sorted(s, v, &i); o[i] = val; if (pack) obs_bytes[i] = (uint8_t)val; }   // kind-sorted enumeration: uniform branches per wave
  }
  __syncthreads();    // the row is final: RedSessionCheck and the step bookkeeping ran beside the encode
  if (tid == 0) a.err[e] = s->err;
  if (a.obs8) store_packed_row(a.obs8 + (size_t)e * OBS_PACKED, obs_bytes, tid, PT);
  unsigned long long t_out = a.prof ? clock64() : 0;
  if (prof && tid == 0) prof[12] += t_out - t_obs;
  // write-back: the agent part always; of the host table (55 % of the row) only the rows this step wrote -- a HostDyn is exactly one
  // 64-byte line, written here by four adjacent lanes, and a step touches a handful of the 137 (hd_touch; everything after a reset)
  uint4* dst = reinterpret_cast<uint4*>(a.st + e);
  static_assert(sizeof(HostDyn) == 64 && offsetof(EnvState, hd) % 64 == 0 && HOT_VEC + 4 * MAXH == ROW_VEC, "one line per host row, the table closes the row");
  if (RUN) {
    if (run_flags & 2) for (int i = tid; i < ROW_VEC; i += PT) dst[i] = lds[i];     // the launch's last step of the episode: the whole row
  } else {
    for (int i = tid; i < HOT_VEC; i += PT) dst[i] = lds[i];
    if (do_reset) { for (int i = HOT_VEC + tid; i < ROW_VEC; i += PT) dst[i] = lds[i]; }
    else for (int k = tid; k < 4 * MAXH; k += PT) if ((work.hdirty[k >> 7] >> ((k >> 2) & 31)) & 1u) dst[HOT_VEC + k] = lds[HOT_VEC + k];
  }
  if (prof && tid == 0) { prof[13] += clock64() - t_out; prof[14] += clock64() - t_begin; }
  if (prof) { __syncthreads(); if (tid < 15) a.prof[PROF_SLOTS * (size_t)e + tid] += prof_lds[tid]; }
}

template <bool LOG, int MINW>
__global__ __launch_bounds__(PT, MINW) void k_step_philox(StepArgs a) { philox4_body<LOG>(a); }

// The multi-step form for batches the chip holds at once (at most five episode blocks per CU: the per-GPU share of an 8-GPU job,
// BASELINE configs[1]): ONE launch runs the K steps of cc4_run_random_steps, every block looping over the steps of ITS episode.
// A launch per step lasts as long as its slowest episode (70.8k cycles against a mean of 41.5k at 1024 episodes,
// profiles/r03_tail_whatif.txt) and the chip idles behind it; here an episode's next step starts the moment its last one ends --
// episodes are independent, so nothing else orders them -- and the batch advances at the MEAN step time.  No ticket, no flag, no
// cache maintenance: a block only ever reads what it wrote itself (its waves drain their stores, s_waitcnt vmcnt(0), and meet at
// the block barrier before the next step stages the row in again; the CU's L1 is coherent for its own waves).  Blocks beyond the
// chip's residency simply start when others have finished all their steps: correct at any batch size, worthwhile below it.
// And the row never leaves the block: it is staged in before the first step and written back after the last (what a step writes
// every time are its outputs: observations, reward, done, error word, the drawn actions).
// The body is a real call: inlined into the step loop its loop-invariant values are hoisted and held across the whole step.
// (register budget of five blocks per CU, stated for the callee as well: left to itself it takes 212 VGPRs)
// (r04, end of round: the body INLINED -- with the thread id made opaque per step, so that nothing derived from it is hoisted out of the loop
// and held across the whole step; ~90 VGPRs spill, and it is still 30 % faster than the call: a kernel that contains a call loses a quarter
// of its rate, profiles/r04_compiler_flags_ab.txt, r04_multistep_inline_ab.txt.)
template <int MINB>
__device__ __forceinline__ void run_philox_loop(StepArgs a, int K, uint32_t t0, const XchgArgs x) {
  a.prof = nullptr; a.obs8 = nullptr; a.ext = nullptr;
  __shared__ uint8_t xrow[OBS_TOTAL + 2];      // the exchange: the step's observation values as bytes (LDS does not bound the four-wave kernels' residency)
  const int full0 = a.full_obs;
  uint32_t seen = 0;
  for (int k = 0; k < K; ++k) {
    a.rand_t = t0 + (uint32_t)k;
    a.full_obs = k == 0 ? (full0 | (x.slab ? 1 : 0)) : 0;       // (the byte row starts empty: the launch's first step writes every value)
    { int tid_i = (int)threadIdx.x; asm volatile("" : "+v"(tid_i));
      philox4_body<false, true>(a, (k > 0 ? 1 : 0) | (k == K - 1 ? 2 : 0), tid_i, x.slab ? xrow : nullptr); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (x.slab && threadIdx.x < WAVE) {
      // As in the one-wave loops: the row of step k - 1 is in memory by now (this step's drain covered its store) and is counted; this step's
      // row goes out from the byte row in LDS -- no global loads, nothing waited for (a system-scope store takes ~1.5 us to land: inside the
      // drain it was 1.4 us of every step; read back from the int32 row, the loads were).  The slab must be free: its previous occupant, step
      // k - ring, gathered -- checked here, by the one wave that writes it, not by the block at the top of the step.
      const int e = a.e0 + (int)blockIdx.x;
      if (threadIdx.x == 0) { if (k > 0) xchg_count(x, (uint32_t)(k - 1), e >> 5); xchg_wait_slab(x, (uint32_t)k, seen); }
      store_packed_row(x.slab + ((size_t)(k % x.ring) * (size_t)a.n + (size_t)e) * OBS_PACKED, xrow, (int)threadIdx.x, WAVE);
    }
  }
  if (x.slab && K > 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) xchg_count(x, (uint32_t)(K - 1), (a.e0 + (int)blockIdx.x) >> 5);
  }
}
__global__ __launch_bounds__(PT, 5) void k_run_philox(StepArgs a, int K, uint32_t t0, XchgArgs x) { run_philox_loop<5>(a, K, t0, x); }
// the same with the register budget of eight blocks per CU: batches of up to 8 x CUs episodes (2048 on MI355X) resident at once
__global__ __launch_bounds__(PT, 8) void k_run_philox8(StepArgs a, int K, uint32_t t0, XchgArgs x) { run_philox_loop<8>(a, K, t0, x); }
// (a build with the budget of four blocks per CU -- 128 registers per lane, 1024 episodes on 256 CUs -- is 0.7 % faster than the one of five: not kept)

// ---------------------------------------------------------------- Philox mode, one wavefront per episode
// The same step as k_step_philox with the agents on the LANES of a single wave instead of on four waves: red agent r on lane
// r, blue agent b on lane 8 + b for its submission and on lane b for its action, green agent g on lane g % 64.  A block of
// four waves spends most of its resident time with three waves parked at a barrier behind the one that carries the red
// agents; with one wave per episode every resident wave works, and as a wave64 instruction occupies its SIMD for four
// cycles whatever the number of active lanes, what counts at large batches is the number of instructions per episode, not
// their spread over waves.  Like the numpy-stream kernel it stages only the agent part of the row (6 992 B) and leaves the
// host table in HBM / L2, so 16 episodes are resident per CU (LDS) instead of 7-8.  The build for throughput-bound batches;
// k_step_philox keeps the shorter single-launch latency of small ones (cc4_create picks; CC4_PHILOX_LEAN overrides).
#ifndef CC4_LEAN_MINW
#define CC4_LEAN_MINW 1
#endif
// ---- the persistent form of the same kernel (PERSIST): K steps of the whole batch in ONE launch.
// A step-per-launch schedule ends every launch with a tail (its last blocks run on a half-empty chip) and starts the next with a
// ramp; cutting the batch into four groups on four streams hides most of that, not all: 8192 episodes x 29.6 us of
// dependent work per episode-step over 5120 resident waves would take 47.4 us per step, four launches take 53.4.  Here the grid is
// one wave per residency slot, and every wave pulls (episode, step) items until the K steps of all episodes are done -- no launch
// boundary inside, no tail but the last one.  Two things make that safe without any cache maintenance:
//  * CU affinity.  A CU's vector L1 is never refreshed by another CU's stores, and the XCDs' L2s are not coherent with each other
//    (MI355X_MICROARCH.md, "inter-workgroup visibility"): an episode's rows must therefore be touched by ONE CU for the whole
//    launch.  The batch is cut into one partition per CU (episode e -> partition e % P, P = the CUs the device showed at first use);
//    a wave reads its CU's identity from the hardware (HW_REG_XCC_ID, HW_REG_HW_ID: shader engine / array / CU), finds the CU's
//    partition in the table of the device's CUs (RunArgs.slot_part) and claims it (owner[p]: compare-and-swap of the CU's slot id); only waves of the owning CU ever
//    work on a partition.  Waves of one CU share its L1, which is coherent for them (what workgroup-scope ordering relies on), so
//    the hand-over between two of them needs ordering only: the writer drains its stores (s_waitcnt vmcnt(0)) before it publishes.
//  * Order per episode.  Items of a partition are handed out by a ticket counter in the order (step 0 of its episodes, step 1, ..):
//    item (e, k) may start once progress[e] == k, which the wave that ran (e, k - 1) stores when its row is back in memory.  With
//    32 episodes and 20 waves per CU the predecessor finished a dozen tickets ago; the wait is a single load, normally.
// A CU that got no wave (never seen in practice: the grid fills every CU) leaves its partition unclaimed; waves that run out of
// work adopt such a partition for THEIR CU (same claim), so every item is executed exactly once whatever the placement.
struct RunArgs {
  uint32_t* ticket;            // [P] next item of partition p
  uint32_t* progress;          // [n] steps of this launch episode e has completed
  int32_t* owner;              // [P] 0 = unclaimed, else 1 + slot id of the owning CU
  const int32_t* slot_part;    // [CC4_SLOTS] CU slot id -> 1 + its partition, 0 = no such CU on this device (k_discover at first use: partitions in
                               // slot order, so the CUs of an XCD own neighbouring partitions and their ticket / progress words share cache lines
                               // only with each other -- handed out in arrival order they interleave the XCDs, and a 20-step call was 6 % slower)
  int P, K;
  int G;                       // the exchange counts episode e in group e % G (the gate kernel's groups: G = the CUs of the device in both schedules)
  uint32_t t0;                 // action time of step 0 (random_blue_action)
  unsigned long long* timeline; // debug (CC4_PERSIST_TIMELINE=1): per wave [entry, first item start, last item end, items] in wall_clock64 ticks, or null
  int order;                   // memory ordering of the hand-over between two items of an episode (CC4_PERSIST_ORDER, persist_loop):
                               // 0 = ordering only (same CU: the waves of a CU share its L1), 1 = every item starts with an agent-scope acquire,
                               // 2 = ... and ends with an agent-scope release, 3 = every item starts with an L1 invalidate (buffer_inv sc0)
  // ---- XCD pools (r06; `pool` != 0): the batch is cut into one partition per XCD (episode e -> pool e % P, P = the XCDs the device showed), every
  // wave of an XCD pulls from its XCD's ticket counter, and every item starts with an invalidate of the CU's vector L1 (buffer_inv sc0: the
  // XCD's L2 is the coherence point of its CUs and the L1 is write-through, so a drained store of ANY CU of the XCD is visible behind it).
  // No owner table, no claim, no stealing: a CU never runs dry while its XCD has an item, so the launch's tail is one item long instead of
  // the lag of the slowest CU's partition.  ticket = this call's counters ([P] words, TK_STRIDE apart), ticket_next = the other parity's
  // (every wave zeroes its pool's word there: the next call needs no memset); progress[] counts steps since the handle's last reset of it
  // (`base` = the count every episode stands at when the call starts).
  // ---- runs of steps (r06).  An item is a RUN of consecutive steps of one episode: nA runs of SA steps, then nB of SB, then single steps
  // (nph runs in all, K steps).  Inside a run the agent part stays in LDS -- no write-back and re-stage between the steps, one ticket, one
  // progress wait and one store drain per run instead of per step; the short runs at the end keep the launch's tail one step long.
  int SA, nA, SB, nB, nph;
  int pool;                    // schedule: 0 = per-CU partitions, a tail shared inside the XCD (r04 / r05); 1 = XCD pools (experiment);
                               // 2 = per-CU partitions BALANCED inside the XCD while the call runs (r06, below)
  uint32_t base;
  uint32_t* ticket_next;
  uint8_t xcc_pool[8];         // XCC id -> pool, 0xFF: no such XCD
  // ---- schedule 2: balanced partitions.  Partitions are per CU as in schedule 0 (an episode normally stays on ONE CU, whose waves share
  // its write-through L1: no cache maintenance), but a wave looks at the ticket counters of its XCD's partitions before every run and, when
  // its own partition is more than `thr` tickets AHEAD of the one that lags most -- or handed out --, takes its run from that one.  The
  // partitions of an XCD so finish within a run of each other, instead of the slowest CU's lag building up to the call's end where
  // helpers can only wait in its episodes' chains.  An episode's progress word carries, beside the steps done, the id of the CU that ran
  // its last run: a run on ANOTHER CU than that one starts with an agent-scope acquire (buffer_inv sc1: tools/micro/l1_inv_scope.hip --
  // nothing less drops a CU's stale L1 lines; profiles/r06_l1_inv_scope.txt), a run on the same CU with none.
  uint8_t xcc_lo[8], xcc_n[8]; // XCC id -> first partition / number of partitions of that XCD (partitions are numbered in slot order)
  int thr;
  // ---- rollouts with the policy in the loop (r06; cc4_rollout_begin): the blue actions of step j are written, while this launch runs, by kernels of
  // the caller's on the caller's stream -- one policy group of episodes at a time: group of e = (e / P) % PG, so every CU holds episodes of every
  // group and works on one group while another waits for its policy.  Step j of an episode of group g starts once act_ready[g] > j (published by
  // the caller behind its policy kernels, cc4_rollout_publish); it reads slot j % 2 of `act` with system-scope loads, writes its packed
  // observation row into slab j % ring with system-scope stores (XchgArgs.slab) and counts itself in cnt[(e % P) * PG + g][j % ring] once that
  // row is in memory -- what the gate of the caller's next policy pass waits for (cc4_rollout_wait_obs).  Every step is an item of its own.
  const uint32_t* act_ready;   // [PG][32 words] (a cache line per group), or null: no rollout
  const int32_t* act;          // [2][n][5]
  int PG;
  long long act_wait_ticks;    // watchdog: a step that waits longer for its actions gives up, raises XchgArgs.timeout, and every later wait returns at once
};
// lane 0: the actions of step j for policy group g are published.  Polls a device word at a growing interval (see xchg_wait_slab).
__device__ __forceinline__ void rollout_wait_actions(const RunArgs& ra, const XchgArgs& x, int g, uint32_t j) {
  const uint32_t* w = ra.act_ready + (size_t)g * 32;
  if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) > j) return;
  if (__hip_atomic_load(x.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
  const long long w0 = wall_clock64();
  int naps = 1;
  while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) <= j) {
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(64);
    if (naps < 8) naps <<= 1;
    if (wall_clock64() - w0 > ra.act_wait_ticks || __hip_atomic_load(x.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
      __hip_atomic_store(x.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(x.timeout_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}
constexpr uint32_t PG_STEPS = 0x7FFFFFu;   // progress word (schedule 2): steps in bits 0..22, 1 + the last runner's partition in bits 23..31 (0: none yet)
__device__ __forceinline__ void run_span(const RunArgs& ra, int j, int& k0, int& len) {
  if (j < ra.nA) { k0 = j * ra.SA; len = ra.SA; }
  else if (j < ra.nA + ra.nB) { k0 = ra.nA * ra.SA + (j - ra.nA) * ra.SB; len = ra.SB; }
  else { k0 = ra.nA * ra.SA + ra.nB * ra.SB + (j - ra.nA - ra.nB); len = 1; }
}
constexpr int TK_STRIDE = 32;  // words between two pools' ticket counters (a cache line of their own each)
constexpr int CC4_SLOTS = 2048;    // (XCC id << 8) | HW_ID[15:8]
__device__ __forceinline__ int cu_slot() {
  const uint32_t hw = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4);     // HW_REG_HW_ID bits 15:0: wave, simd, pipe | cu, sh, se
  const uint32_t xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID bits 3:0
  return (int)(((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu));
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

// the one-wave kernel's in-kernel scenario generation (an episode regenerates once in steps-per-episode launches)
#if defined(CC4_EXP_RESET_CALL)
__device__ __attribute__((noinline))
#else
__device__ __forceinline__
#endif
void philox1_autoreset(const StepArgs& a, const int e, const int lane, EnvState* s, HostDyn* const hd, EnvCold* const cold_e, StepWork& work) {
    // new episode on the same key (CybORG.reset(seed=None)): the phases of env_reset_counter_mode, hosts on lanes; the pid
    // bitmaps of the generation live in HBM here (LDS bounds this kernel's residency, and this path runs once per episode)
    uint32_t* const ws = a.reset_ws + (size_t)e * RESET_WS_WORDS;
    reset_zero(s, hd, cold_e, lane, WAVE);
    __syncthreads();
    Rng rr; ResetCarry carry; carry.env_key = 0;     // lane 0: main reset stream in registers, across the phases
    Ctx xm{s, cold_e, &rr, hd, &work};
    if (lane == 0) { rr = s->rng; rr.mode = 1; carry = reset_topology(xm, 0, a.steps, true, a.policy, a.topo, ws, true); }
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
    if (lane == 0) { reset_finish(xm, carry, a.steps, a.topo, true); a.reward[e] = s->reward; a.done[e] = s->done; }
    __syncthreads();
}

// One step of one episode on one wavefront: the body of k_step_philox1 and of the persistent run kernel.  PERSIST: item_k = the
// step's number within the launch (the first item of an episode rewrites all its observation values when asked to).
// first / last (the one-launch loops): the step is the first / last of a run of consecutive steps of this episode on this wave -- only the
// first stages the agent part in, only the last writes it back; in between the row lives in LDS (the host table, the cold row and the outputs
// are read and written in memory by every step as always).
template <bool LOG, bool PERSIST>
__device__ __forceinline__ void philox1_body(StepArgs a, const int e, const uint32_t rand_t, const uint32_t item_k, const int lane,
                                             const bool first = true, const bool last = true) {
  extern __shared__ uint4 lds[];
  // Static LDS is kept under 512 bytes: agent part (7168 B) + statics then fit SIX 1280-byte LDS granules, 21 waves per CU by LDS and 20
  // by registers; a seventh granule would leave 18 (profiles/r05_lds_residency.txt: the occupancy query, which divides 160 KB by the
  // byte count, says 20 either way).  So: no byte copy of the observations for the packed exchange row (pack_row_from_obs reads the
  // int32 row back), and the debug phase timers exist in the full build only (cc4_debug_profile selects it).
  __shared__ StepWork work;
  __shared__ int conflict_lds;
  __shared__ unsigned long long prof_lds[LOG ? 16 : 1];
  if constexpr (!LOG) a.prof = nullptr;
  (void)item_k;
  EnvCold* const cold_e = cold_at(a.cold, (size_t)e, cold_row_bytes(a.steps));
  unsigned long long t_begin = a.prof ? clock64() : 0;
  const uint4* src = reinterpret_cast<const uint4*>(a.st + e);
  if (first) stage_in<HOT_VEC>(lds, src, lane);
  for (int i = lane; i < (int)(sizeof(StepWork) / 4); i += WAVE) reinterpret_cast<uint32_t*>(&work)[i] = 0;
  unsigned long long* prof = a.prof ? prof_lds : nullptr;
  if (prof && lane < 16) prof_lds[lane] = 0;
  if (lane == 0) conflict_lds = 0;
  __syncthreads();
  EnvState* s = reinterpret_cast<EnvState*>(lds);   // only the part in front of EnvState.hd is valid here
  HostDyn* const hd = a.st[e].hd;                   // the host table stays in HBM / L2
  if (prof && lane == 0) prof[11] += clock64() - t_begin;
  const bool do_reset = a.autoreset && s->done;
  if (do_reset) {
    philox1_autoreset(a, e, lane, s, hd, cold_e, work);
  } else {
    const int st_now = s->step_count;
    const bool step_ok = step_phase_of(st_now, s->phase_len[0], s->phase_len[1], s->phase_len[2]) >= 0;
    if (lane == 0) {
      Ctx x{s, cold_e, &s->rng, hd, &work, prof};
      x.lg = (LOG && cold_e->evlog.enabled) ? &cold_e->evlog : nullptr;
      x.ext = (LOG && a.ext) ? a.ext + (size_t)e * EXT_PER_ENV : nullptr;
      CC4_TICK0(x);
      (void)step_phase(x, false);
    }
    if (step_ok) {
      const int ng = s->n_green;
      // one lane-private generator per lane, in registers: every use starts with rng_set_stream(); mode pinned so the PCG
      // paths fold away
      Rng rl;
      rng_fork(&rl, &s->rng, ST_RESET);
      rl.mode = 1;
      rng_begin_step(&rl, (uint32_t)st_now);   // not read from the row: lane 0 may still be storing it there
      EvLog* const lg = (LOG && cold_e->evlog.enabled) ? &cold_e->evlog : nullptr;
      const ExtAct* const xt = (LOG && a.ext) ? a.ext + (size_t)e * EXT_PER_ENV : nullptr;   // this episode's submitted red / green actions
      Ctx x0{s, cold_e, &rl, hd, &work, lane == 0 ? prof : nullptr};
      x0.lg = lg; x0.ext = xt;
      if (lane == 0) CC4_TICK(x0, 0);
      // ---- the block bank.  A Philox block costs a wave the same ~110 vector instructions whether one lane needs it or
      // sixty-four do, and block 0 of every stream of the step is known from (key, step, episode, stream id) alone.  The streams
      // that have a lane of their own per agent (green policy of agents 0..63, the actions of the compacted green list) are
      // computed where they are used, one pass each; the rest -- the policy draws of green agents 64.., the six red policies and
      // actions, the five blue actions and, in the bench, the five in-kernel blue action draws: 38 requests, five sequential
      // passes when each is computed by the lane that resolves its agent -- share ONE pass here, one request per lane, and reach
      // their agents' lanes through ds_bpermute when their phase comes (bank_fetch; same words as computing them in place:
      // rng_preload).
      enum : int { BK_GPOL = 0, BK_GEXE = 16, BK_RPOL = 32, BK_REXE = 38, BK_BEXE = 44, BK_BRAND = 49, BK_END = 54 };
      uint32_t bank[4];
      {
        uint32_t st = 0;
        if (lane < BK_GEXE) st = ST_GREEN_POL + (uint32_t)(WAVE + lane - BK_GPOL);
        else if (lane < BK_REXE) st = ST_RED_POL + (uint32_t)(lane - BK_RPOL);
        else if (lane < BK_BEXE) st = ST_RED_EXE + (uint32_t)(lane - BK_REXE);
        else if (lane < BK_BRAND) st = ST_BLUE_EXE + (uint32_t)(lane - BK_BEXE);
        bank[0] = 0u; bank[1] = st; bank[2] = (uint32_t)rl.inc_lo; bank[3] = (uint32_t)rl.inc_hi;      // rng_block(&rl, st, 0, .)
        uint32_t k0 = (uint32_t)rl.s_lo, k1 = (uint32_t)(rl.s_lo >> 32);
        if (a.rand_out && lane >= BK_BRAND && lane < BK_END) {   // random_blue_action(seed0, t, e, b): another key and counter layout
          const uint64_t key = a.rand_seed0 + (uint64_t)e;
          bank[0] = rand_t; bank[1] = (uint32_t)(lane - BK_BRAND); bank[2] = 0xB10Eu; bank[3] = 0u; k0 = (uint32_t)key; k1 = (uint32_t)(key >> 32);
        }
        philox4x32_10(bank, k0, k1);
      }
      // the four words lane `lane + shift` holds, on every lane (call with all lanes active: an inactive source lane reads as 0)
      auto bank_fetch = [&](int shift, uint32_t out[4]) {
        const int addr = ((lane + shift) & (WAVE - 1)) << 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)bank[k]);
      };
      const bool is_red = lane < NRED;
      unsigned long long* ap = (a.prof && is_red) ? a.prof + PROF_SLOTS * (size_t)e + 16 + 8 * lane : nullptr;
      Ctx xr{s, cold_e, &rl, hd, &work, nullptr, ap, lg};
      Ctx xg{s, cold_e, &rl, hd, &work, nullptr, nullptr, lg};
      xr.ext = xt; xg.ext = xt;
      // ---- P0-P3a: every agent's policy / submission and its own duration-queue tick (SC:236-265)
      uint32_t pre_rp[4];
      bank_fetch(BK_RPOL, pre_rp);                                               // red r (lane r) <- lane BK_RPOL + r
      const uint32_t brand = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + BK_BRAND - 8) & (WAVE - 1)) << 2, (int)bank[0]);   // blue b (lane 8 + b) <- lane BK_BRAND + b
      if (is_red) {
        unsigned long long t0 = ap ? clock64() : 0;
        const int dropped = step_red_policy_tick(xr, lane, false, pre_rp);
        if (ap) ap[0] += clock64() - t0;
        if (dropped) atomicSub(&s->n_actions, 1);
      } else if (lane >= 8 && lane < 8 + NBLUE) {
        const int b = lane - 8;
        int32_t act = !a.actions ? -1 : a.act_sys ? __hip_atomic_load(a.actions + e * NBLUE + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : a.actions[e * NBLUE + b];
        if (a.rand_out) { act = (int32_t)(((uint64_t)brand * (uint32_t)(b == 4 ? ACT_LONG : ACT_SHORT)) >> 32); a.rand_out[e * NBLUE + b] = act; }   // == random_blue_action
        step_blue_submit(xg, b, act);
        step_tick_blue(xg, b);
        step_messages(s, a.msgs ? a.msgs + e * NBLUE * MSG_LEN : nullptr, b);   // read back by this step's observation encode only
      }
      if (lane < ng) step_green_policy(xg, lane);                               // agents 0..63: their block is computed here, by all of them at once
      if (lane + WAVE < ng) step_green_policy(xg, lane + WAVE, bank);            // agents 64..: from the bank (their lane's own request)
      __syncthreads();
      CC4_TICK(x0, 2);
      // ---- P3b blue actions: side by side when they are independent (no Monitor, no pending pid events)
      if (blue_exec_independent(s)) {
        if (lane == 0) CC4_TICK(x0, 3);
        uint32_t c[4];
        bank_fetch(BK_BEXE, c);                                                   // blue b (lane b) <- lane BK_BEXE + b
        if (lane < NBLUE) step_blue_exec_agent(xg, lane, c);
        __syncthreads();
        if (lane == 0) CC4_TICK(x0, 5);
      } else {
        if (lane == 0) step_blue_exec(x0);
        __syncthreads();
      }
      // ---- P4 green actions, one agent per lane
      {
        // A third of the up to 80 agents sleeps, so the ones with an action nearly always fit the wave's 64 lanes: they are
        // compacted (ballot + prefix count, agent order) into a list and resolved in ONE pass instead of two (the second of
        // which had 16 lanes at most and cost the wave as much as the first).  Per-agent streams make the order immaterial.
        int pen = 0;
        const uint32_t act0 = lane < ng ? work.green_act[lane] : 2u, act1 = lane + WAVE < ng ? work.green_act[lane + WAVE] : 2u;
        const unsigned long long m0 = __ballot(act0 < 2u), m1 = __ballot(act1 < 2u);
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int n0 = __popcll(m0), nact = n0 + __popcll(m1);
        uint8_t* const glist = reinterpret_cast<uint8_t*>(work.scratch);           // the ordered sections' scratch is idle in this phase
        static_assert(sizeof(work.scratch) >= MAXG, "the green list fits the scratch words");
        if (act0 < 2u) glist[__popcll(m0 & lt)] = (uint8_t)lane;
        if (act1 < 2u) glist[n0 + __popcll(m1 & lt)] = (uint8_t)(lane + WAVE);
        __syncthreads();
        for (int i = lane; i < nact; i += WAVE) {
          const int g = glist[i];
          uint32_t c[4]; rng_block(&rl, ST_GREEN_EXE + (uint32_t)g, 0, c);        // ahead of the AccessService / LocalWork split: one block for all
          pen += step_green_exec(xg, g, c);
        }
        if (pen) atomicAdd(&s->brm, pen);
      }
      __syncthreads();
      CC4_TICK(x0, 6);
      // ---- P5 deferred phishing (ordered), then P6 red actions: side by side when they name distinct hosts
      if (lane == 0) { step_phishing(x0); CC4_TICK(x0, 1); rs_reserve(x0); conflict_lds = (int)red_conflict_mask(s); if (prof && conflict_lds) prof[4] += 1000000; }
      __syncthreads();
      const uint32_t serial_red = (uint32_t)conflict_lds;
      uint32_t pre_re[4];
      bank_fetch(BK_REXE, pre_re);                                                // red r (lane r) <- lane BK_REXE + r
      if (is_red && !((serial_red >> lane) & 1u)) {
        unsigned long long t0 = ap ? clock64() : 0;
        step_red_exec_agent(xr, lane, pre_re);
        if (ap) ap[1] += clock64() - t0;
      }
      __syncthreads();
      if (serial_red) {   // same-host actions (and everything when some agent withdraws): agent order on lane 0
        if (lane == 0) for (int r = 0; r < NRED; ++r) if ((serial_red >> r) & 1u) step_red_exec_agent(x0, r);
        __syncthreads();
      }
      if (lane == 0) {
        step_red_merge(x0);
        CC4_TICK(x0, 7);
        step_reassign(x0, red_foreign_agents(s));
        CC4_TICK(x0, 8);
      }
      // P7 end-turn Monitor roll-over: the hosts' event bytes are part of the staged row (EnvState.hev).  (Lane 0's reassignment above
      // moves sessions, not events.)
      for (int h = lane; h < MAXH; h += WAVE) s->hev[h] = monitor_roll(h, s->hev[h]);
      __syncthreads();
      CC4_TICK(x0, 9);
      // ---- P8 end-turn RedSessionCheck on the red lanes; the Monitor's sus-pid hand-over and the step's bookkeeping on the last
      if (is_red) { unsigned long long t0 = ap ? clock64() : 0; step_rsc(xr, lane); if (ap) ap[2] += clock64() - t0; }
      if (lane == WAVE - 1) {
        step_monitor_pend(xg);
        step_end(xg, nullptr, false);
        a.reward[e] = s->reward; a.done[e] = s->done;
      }
      CC4_TICK(x0, 10);
    } else if (lane == 0) { a.reward[e] = s->reward; a.done[e] = s->done; }
  }
  __syncthreads();
  unsigned long long t_obs = a.prof ? clock64() : 0;
  {
    int32_t* o = a.obs + (size_t)e * OBS_TOTAL;
    const int nv = (do_reset || (a.full_obs && (!PERSIST || item_k == 0)) || s->obs_dirty) ? OBS_TOTAL : OBS_FAST;
    encode_obs_fast<WAVE>(s, o, nullptr, false, lane);
    for (int v = OBS_FAST + lane; v < nv; v += WAVE) { int i; const int val = env_flat_obs_sorted(s, v, &i); o[i] = val; }
  }
  __syncthreads();
  if (lane == 0) a.err[e] = s->err;
  unsigned long long t_out = a.prof ? clock64() : 0;
  if (prof && lane == 0) prof[12] += t_out - t_obs;
  if (last) stage_out<HOT_VEC>(reinterpret_cast<uint4*>(a.st + e), lds, lane);
  if (a.obs8) {     // the per-step launches' packed exchange row (the one-launch loops pack behind their own end-of-step drain instead)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pack_row_from_obs(a.obs8 + (size_t)e * OBS_PACKED, a.obs + (size_t)e * OBS_TOTAL, lane);
  }
  if (prof && lane == 0) { prof[13] += clock64() - t_out; prof[14] += clock64() - t_begin; }
  if (prof) { __syncthreads(); if (lane < 15) a.prof[PROF_SLOTS * (size_t)e + lane] += prof_lds[lane]; }
}

template <bool LOG>
__global__ __launch_bounds__(WAVE, CC4_LEAN_MINW) void k_step_philox1(StepArgs a) {
  const int e = a.e0 + (int)blockIdx.x;
  if (e >= a.n) return;
  philox1_body<LOG, false>(a, e, a.rand_t, 0u, (int)threadIdx.x);
}


// Tail of a call: a CU whose own partition is handed out takes items from the partition of another CU OF ITS XCD that has the most left.
// The XCD's L2 is the coherence point of its CUs (vector stores write through to it), but a CU's L1 is not refreshed by another CU's
// stores -- so from the moment a partition is shared (bit 31 of its ticket counter, set by the first thief; every ticket handed out
// afterwards carries it) every item of it starts with an agent-scope acquire (buffer_inv sc1: the CU's L1 dropped), on the owner's waves
// and the thieves' alike.  Items handed out before the bit was set were all the owner's own and read what that CU wrote itself.
// Never across XCDs: their L2s do not agree without a write-back.
// ---- experiment: the red policy phase with the agents of G episodes side by side on ONE wave.  Lane 8 g + r runs
// step_red_policy_tick of agent r of the wave's g-th episode -- what a group schedule would do in the phase that is 31 % of a step -- on the live
// state of the batch (agent parts staged into LDS as in the step kernel, nothing written back).  G = 1 is today's lane layout.  cyc[block] = the
// wave's cycles in the phase; the launch duration (events) / episodes = what the phase costs an episode at that grouping and residency.
#ifdef CC4_POLICY_PROBE      // (a concluded experiment of r05: its four instantiations are not part of the product library)
template <int G>
__global__ __launch_bounds__(WAVE) void k_policy_probe(StepArgs a, unsigned long long* cyc) {
  extern __shared__ uint4 lds[];
  __shared__ StepWork work[G];
  const int lane = (int)threadIdx.x, e0 = (int)blockIdx.x * G;
  for (int g = 0; g < G; ++g) if (e0 + g < a.n) stage_in<HOT_VEC>(lds + g * HOT_VEC, reinterpret_cast<const uint4*>(a.st + e0 + g), lane);
  for (int i = lane; i < (int)(G * sizeof(StepWork) / 4); i += WAVE) reinterpret_cast<uint32_t*>(&work[0])[i] = 0;
  __syncthreads();
  const int g = lane >> 3, r = lane & 7, e = e0 + g;
  const unsigned long long t0 = clock64();
  int dropped = 0;
  if (g < G && r < NRED && e < a.n) {
    EnvState* s = reinterpret_cast<EnvState*>(lds + g * HOT_VEC);
    const int st_now = s->step_count;
    if (!s->done && step_phase_of(st_now, s->phase_len[0], s->phase_len[1], s->phase_len[2]) >= 0) {
      Rng rl;
      rng_fork(&rl, &s->rng, ST_RESET);
      rl.mode = 1;
      rng_begin_step(&rl, (uint32_t)st_now);
      uint32_t pre[4];
      rng_block(&rl, ST_RED_POL + (uint32_t)r, 0, pre);
      Ctx xr{s, cold_at(a.cold, (size_t)e, cold_row_bytes(a.steps)), &rl, a.st[e].hd, &work[g]};
      dropped = step_red_policy_tick(xr, r, false, pre);
    }
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  if (dropped == 12345) cyc[0] = 0;      // (keeps the result alive)
}
#endif
constexpr uint32_t TK_SHARED = 0x80000000u;
// register budget of the persistent counter-mode kernel in waves per SIMD: 6 (80 VGPRs, eight spilled: 24 waves per CU -- by registers and,
// since r06's 5952-byte agent part made a wave FIVE 1280-byte LDS granules, by LDS as well: 25) or 5 (93 VGPRs, nothing spilled: 20 per CU).
// Measured, 8192 episodes, one box (profiles/r06_layout_ab.txt): K = 500: 989-990 vs 914-915 M, K = 20: 831-846 vs 797-807 M.  (r05, when LDS
// capped a CU at 21 waves: 952 vs 939-948 M -- inside the box-to-box spread.)
#ifndef CC4_PERSIST_MINW
#define CC4_PERSIST_MINW 6
#endif
template <bool PCG>
__device__ __forceinline__ void persist_loop(StepArgs a, RunArgs ra, const XchgArgs x) {
  // (the item travels from lane 0 to the wave through v_readfirstlane, not through LDS)
  const int lane = threadIdx.x;
  const int my_slot = cu_slot();
  // The CU's partition, from the table of the compute units this device showed at first use (a CU that is not in it only helps out)
  int part = ra.pool ? -1 : ra.slot_part[my_slot] - 1;  // schedule 0 (lane 0's copy is the one that counts)
  bool mine = false;                     // lane 0: this CU owns `part` (claimed or adopted)
  bool stealing = false;                 // lane 0: `part` belongs to another CU of this XCD; its shared bit is set
  a.prof = nullptr; a.obs8 = nullptr; a.ext = nullptr;
  uint32_t seen_gathered = 0;
  unsigned long long tl_first = 0, tl_last = 0, tl_items = 0;
  const unsigned long long tl_entry = ra.timeline ? wall_clock64() : 0;
  auto tl_flush = [&]() { if (ra.timeline && lane == 0) { unsigned long long* t = ra.timeline + 4 * (size_t)blockIdx.x; t[0] = tl_entry; t[1] = tl_first; t[2] = tl_last; t[3] = tl_items | ((unsigned long long)(my_slot + 1) << 32); } };
  int pend_e = -1; uint32_t pend_k = 0;  // the exchange: the item whose packed row this wave stored last and has not counted yet (its store drains with the next item)
  auto flush_pending = [&]() {
    if (x.slab && pend_e >= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) xchg_count(x, pend_k, pend_e % ra.G); pend_e = -1; }
  };
  const uint32_t my_xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;    // HW_REG_XCC_ID
  const int xlo = ra.xcc_lo[my_xcc], xn = ra.xcc_n[my_xcc];     // schedule 2: this XCD's partitions
  int own = -1; uint32_t my_id = 511u;                           // schedule 2: the CU's own partition (-1: none), its id in the progress words
  if (ra.pool == 2) {
    own = ra.slot_part[my_slot] - 1;
    if (own >= 0) my_id = (uint32_t)own + 1u;
    if (xn <= 0) { tl_flush(); return; }
  }
  if (ra.pool == 1) {
    const uint32_t xcc = my_xcc;
    part = ra.xcc_pool[xcc] == 0xFF ? -1 : (int)ra.xcc_pool[xcc];
    if (part < 0) { tl_flush(); return; }                             // (an XCD the discovery pass did not see: its waves do nothing)
    if (lane == 0) __hip_atomic_store(&ra.ticket_next[part * TK_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (;;) {
    int res_e = -3, res_k = 0, res_sh = 0, res_part = -1;            // -3: nothing from `part`: search
    if (ra.pool == 2) {
      // all lanes: where the XCD's partitions stand
      const int q = xlo + lane;
      uint32_t tk = 0xFFFFFFFFu, tot_q = 0;
      // (every partition's counter on a cache line of its own, TK_STRIDE words apart: 24 waves of one CU on a line, not the 768 of an XCD -- with the
      // XCD's 32 counters on ONE line, its atomics and these loads took the L2 ~50 ns each and the schedule ran at 556 M instead of 884 M)
      if (lane < xn) { tk = __hip_atomic_load(&ra.ticket[q * TK_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tot_q = (uint32_t)(((a.n - q + ra.P - 1) / ra.P) * ra.nph); }
      const bool has = lane < xn && tk < tot_q;
      uint32_t key = has ? ((tk << 6) | (uint32_t)lane) : 0xFFFFFFFFu;          // least tickets handed out = lags most (the partitions' sizes differ by one episode at most)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) { const uint32_t k2 = (uint32_t)__shfl_xor((int)key, off); key = k2 < key ? k2 : key; }
      const uint32_t kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
      if (kmin == 0xFFFFFFFFu) { flush_pending(); tl_flush(); return; }        // every partition of this XCD is handed out
      int target = (int)(kmin & 63u);
      if (own >= 0) {
        const int ol = own - xlo;
        const uint32_t tk_own = (uint32_t)__builtin_amdgcn_readlane((int)tk, ol);
        const uint32_t tot_own = (uint32_t)__builtin_amdgcn_readlane((int)tot_q, ol);
        if (tk_own < tot_own && tk_own <= (kmin >> 6) + (uint32_t)ra.thr) target = ol;
      }
      if (lane == 0) {
        const int tp = xlo + target;
        const int ne = (a.n - tp + ra.P - 1) / ra.P;
        const uint32_t t = __hip_atomic_fetch_add(&ra.ticket[tp * TK_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        res_e = -5;                                                   // handed out meanwhile: look again
        // the partition's last ticket: its counter of the OTHER parity cleared for the next call (exactly one wave per partition and call
        // draws it, whoever runs the partition -- no memset between calls)
        if (t + 1u == (uint32_t)(ne * ra.nph)) __hip_atomic_store(&ra.ticket_next[tp * TK_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < (uint32_t)(ne * ra.nph)) {
          const int j = (int)(t / (uint32_t)ne);
          int i = (int)(t % (uint32_t)ne), pg = 0;
          if (ra.act_ready) {
            // a rollout: the tickets of a step serve one policy group after the other (episode index i of the partition is of group i % PG) --
            // while one group's episodes wait for their policy pass, the CU's waves hold tickets of the other's
            for (; pg < ra.PG; ++pg) { const int c = (ne - pg + ra.PG - 1) / ra.PG; if (i < c) { i = i * ra.PG + pg; break; } i -= c; }
          }
          const int ee = tp + i * ra.P;
          int k, len; run_span(ra, j, k, len);
          uint32_t w;
          while ((((w = __hip_atomic_load(&ra.progress[ee], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & PG_STEPS) - ra.base) < (uint32_t)k) __builtin_amdgcn_s_sleep(8);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (x.slab) xchg_wait_slab(x, (uint32_t)k, seen_gathered);
          if (ra.act_ready) rollout_wait_actions(ra, x, pg, (uint32_t)k);
          const uint32_t last = w >> 23;
          res_e = ee; res_k = j; res_sh = (last != 0u && last != my_id) ? 1 : 0;     // the episode's last run was on another CU: its lines in this CU's L1 may be stale
        }
      }
    } else if (ra.pool) {
      if (lane == 0) {
        const int ne = (a.n - part + ra.P - 1) / ra.P;                // episodes part, part + P, part + 2 P, ..
        const uint32_t t = __hip_atomic_fetch_add(&ra.ticket[part * TK_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        res_e = -4;                                                   // the pool is handed out: leave
        if (t < (uint32_t)(ne * ra.nph)) {
          const int j = (int)(t / (uint32_t)ne), ee = part + (int)(t % (uint32_t)ne) * ra.P;
          int k, len; run_span(ra, j, k, len);
          while (__hip_atomic_load(&ra.progress[ee], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ra.base < (uint32_t)k) __builtin_amdgcn_s_sleep(8);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (x.slab) xchg_wait_slab(x, (uint32_t)k, seen_gathered);
          res_e = ee; res_k = j; res_sh = 2;
        }
      }
    } else
    if (lane == 0) {
      if (part >= 0 && !mine && !stealing) {
        int exp = 0;                                                  // the CU's own partition: claim it (or find it claimed by this CU already)
        mine = __hip_atomic_compare_exchange_strong(&ra.owner[part], &exp, my_slot + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || exp == my_slot + 1;
        if (!mine) part = -1;                                         // somebody else's by now (adopted): search
      }
      if (part >= 0) {
        const bool thief = !mine;                                     // (a partition this wave steals from: `part` was set by the search below)
        const int ne = (a.n - part + ra.P - 1) / ra.P;                // episodes part, part + P, part + 2 P, ..
        const uint32_t tr = __hip_atomic_fetch_add(&ra.ticket[part], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t t = tr & ~TK_SHARED;
        if (t < (uint32_t)(ne * ra.nph)) {
          // (r05, both measured and dropped: asking for the next ticket ahead of the previous item's drain -- the CU's other waves fill that gap
          // already, 813-821 vs 819 M; and shares of the batch per XCD following the XCDs' measured speed -- which XCDs are slow changes from
          // box to box and call to call, the controller chases noise: 20-step calls 812-822 -> 789-796 M.  profiles/r05_xcd_balance.txt)
          // (a ready queue per partition -- a wave never holds an item whose predecessor is still running -- was built and measured in r05:
          // bit-exact, 2-3.5 % slower, and the launch's tail stayed: profiles/r05_ready_queue_ab.txt)
          const int j = (int)(t / (uint32_t)ne), ee = part + (int)(t % (uint32_t)ne) * ra.P;
          int k, len; run_span(ra, j, k, len);
          while (__hip_atomic_load(&ra.progress[ee], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ra.base < (uint32_t)k) __builtin_amdgcn_s_sleep(8);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // nothing the item reads may be read ahead of the flag (compiler and wave)
          if (x.slab) xchg_wait_slab(x, (uint32_t)k, seen_gathered);   // the exchange: slab k % ring must have been gathered (tickets are step-major: normally long ago)
          res_e = ee; res_k = j; res_sh = (thief || (tr & TK_SHARED)) ? 1 : 0;
        }
      }
      res_part = part;
    }
    const int e = __builtin_amdgcn_readfirstlane(res_e);              // (all lanes are active here: the first active lane is lane 0)
    if (e == -4) { flush_pending(); tl_flush(); return; }
    if (e == -5) continue;
    if (e == -3) {
      // search (all lanes): the partition with the most items left among those nobody owns and those owned by a CU of this XCD
      const int cur = __builtin_amdgcn_readfirstlane(res_part);
      int best_rem = 0, best_q = -1, best_ow = 0;
      for (int q0 = 0; q0 < ra.P; q0 += WAVE) {
        const int q = q0 + lane;
        if (q < ra.P && q != cur) {
          const int ow = __hip_atomic_load(&ra.owner[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ow == 0 || (((ow - 1) >> 8) == (my_slot >> 8))) {
            const uint32_t t = __hip_atomic_load(&ra.ticket[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~TK_SHARED;
            const uint32_t tot = (uint32_t)(((a.n - q + ra.P - 1) / ra.P) * ra.nph);
            const int rem = t < tot ? (int)(tot - t) : 0;
            if (rem > best_rem) { best_rem = rem; best_q = q; best_ow = ow; }
          }
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const int r2 = __shfl_xor(best_rem, off), q2 = __shfl_xor(best_q, off), o2 = __shfl_xor(best_ow, off);
        if (r2 > best_rem || (r2 == best_rem && q2 > best_q)) { best_rem = r2; best_q = q2; best_ow = o2; }
      }
      best_rem = __builtin_amdgcn_readfirstlane(best_rem); best_q = __builtin_amdgcn_readfirstlane(best_q); best_ow = __builtin_amdgcn_readfirstlane(best_ow);
      if (best_rem <= 0) { flush_pending(); tl_flush(); return; }     // nothing left anywhere this wave may touch
      part = best_q; mine = false; stealing = false;
      if (lane == 0) {
        if (best_ow == 0) {                                           // nobody's: adopt it (the CAS in the item path), no sharing needed unless that fails
          int exp = 0;
          mine = __hip_atomic_compare_exchange_strong(&ra.owner[part], &exp, my_slot + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || exp == my_slot + 1;
          if (!mine && (((exp - 1) >> 8) != (my_slot >> 8))) part = -1;   // claimed meanwhile by a CU of another XCD: not ours to touch
        }
        if (part >= 0 && !mine) { (void)__hip_atomic_fetch_or(&ra.ticket[part], TK_SHARED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); stealing = true; }
      }
      continue;
    }
    int run_k0, run_len;
    run_span(ra, __builtin_amdgcn_readfirstlane(res_k), run_k0, run_len);
    const int shared = __builtin_amdgcn_readfirstlane(res_sh);
    if (ra.timeline && !tl_items) tl_first = wall_clock64();
    if (shared || ra.order >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (buffer_inv sc1: the CU's L1 dropped.  buffer_inv sc0 does NOT drop it: profiles/r06_l1_inv_scope.txt)
    uint32_t item_k = (uint32_t)run_k0;
    for (int q = 0; q < run_len; ++q, ++item_k) {
    if (q > 0) {
      if (x.slab) {
        // a further step of the run with the exchange on: what the last step stored is drained and counted as at a run's end, and the slab of
        // this step must have been gathered
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (PCG) { if (lane == 0) xchg_count(x, item_k - 1u, e % ra.G); }
        else {
          if (lane == 0 && pend_e >= 0) xchg_count(x, pend_k, pend_e % ra.G);
          pack_row_from_obs(x.slab + ((size_t)((item_k - 1u) % (uint32_t)x.ring) * (size_t)a.n + (size_t)e) * OBS_PACKED, a.obs + (size_t)e * OBS_TOTAL, lane);
          pend_e = e; pend_k = item_k - 1u;
        }
        if (lane == 0) xchg_wait_slab(x, item_k, seen_gathered);
      }
      __syncthreads();
    }
    int lane_i = (int)threadIdx.x;
    asm volatile("" : "+v"(lane_i));
    if (ra.act_ready) { a.actions = ra.act + (size_t)(item_k & 1u) * (size_t)a.n * NBLUE; a.rand_out = nullptr; a.act_sys = 1; }
    if constexpr (PCG) {
      StepArgs b = a;
      b.rand_t = ra.t0 + item_k; b.full_obs = (a.full_obs && item_k == 0) ? 1 : 0;
      if (x.slab) b.obs8 = x.slab + (size_t)(item_k % (uint32_t)x.ring) * (size_t)a.n * OBS_PACKED;
      pcg_body<false>(b, e, lane_i, q == 0, q == run_len - 1);
    } else {
      philox1_body<false, true>(a, e, ra.t0 + item_k, item_k, lane_i, q == 0, q == run_len - 1);      // (a.obs8 is null: the packed row is written below, behind the drain)
    }
    }
    --item_k;        // the run's last step
    // the item is done when everything it wrote has left this wave: then the next step of the episode may start (on this XCD)
    // Release: every lane DRAINS its own stores -- an explicit s_waitcnt vmcnt(0): the vector L1 is write-through, so a drained store is in
    // the XCD's L2 --, the barrier collects the lanes, lane 0 publishes.  The consumer is a wave of the same CU unless the partition is
    // shared, in which case it drops its L1 first (agent-scope acquire above).  The workgroup-scope fence beside it only pins the compiler:
    // without tgsplit the backend emits NO vmcnt wait for it (waves of a work-group share a CU), and the episode's rows and its progress
    // word sit in different L2 channels -- with the fence alone the word can land first.  (r05 ran that way for a day: one disagreement in
    // ~60 self-checked calls, CC4_PERSIST_VERIFY, 5632 episodes, hot row of one episode after a 10-step call.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (ra.order >= 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (x.slab) {
      if constexpr (PCG) {     // (the numpy-stream body stored the row itself, from its LDS byte row: drained by the fence above)
        if (lane == 0) xchg_count(x, item_k, e % ra.G);
      } else {
        // the row this wave stored with its PREVIOUS item is in memory (this item's fence drained it): counted.  Then this episode's row of
        // step item_k, read back from the int32 row before the episode's next step may touch it (the loads feed the store, the store is
        // issued ahead of the progress word) -- not waited for: it drains with the wave's next item, or when the wave leaves.
        if (lane == 0 && pend_e >= 0) xchg_count(x, pend_k, pend_e % ra.G);
        pack_row_from_obs(x.slab + ((size_t)(item_k % (uint32_t)x.ring) * (size_t)a.n + (size_t)e) * OBS_PACKED, a.obs + (size_t)e * OBS_TOTAL, lane);
        pend_e = e; pend_k = item_k;
        if (ra.act_ready) {
          // a rollout: the caller's next policy pass waits for this count -- not deferred to the wave's next item
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) xchg_count(x, item_k, (e % ra.P) * ra.PG + (e / ra.P) % ra.PG);
          pend_e = -1;
        }
      }
    }
    if (lane == 0) __hip_atomic_store(&ra.progress[e], (ra.base + item_k + 1u) | (ra.pool == 2 ? my_id << 23 : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ra.timeline) { tl_last = wall_clock64(); ++tl_items; }
  }
}
__global__ __launch_bounds__(WAVE, CC4_PERSIST_MINW) void k_run_philox1(StepArgs a, RunArgs ra, XchgArgs x) { persist_loop<false>(a, ra, x); }
#ifndef CC4_DEV_FAST
// the same schedule around the numpy-stream step (k_step's body): the bit-exact mode's large batches
__global__ __launch_bounds__(WAVE) void k_run_pcg(StepArgs a, RunArgs ra, XchgArgs x) { persist_loop<true>(a, ra, x); }
#endif

// The plain multi-step form of the one-wave kernel: one wave per episode, every wave loops over the K steps of ITS episode -- no
// tickets, no affinity: a wave only reads what it wrote itself.  For batches one launch holds at once (cc4_create; CC4_RUN1=0/1
// overrides): more waves than residency slots would simply start as slots free up (8192 episodes: 5120 at once, the other 3072
// behind them on a chip that is no longer full).
__global__ __launch_bounds__(WAVE, 5) void k_run_philox1m(StepArgs a, int K, uint32_t t0, XchgArgs x) {
  a.prof = nullptr; a.obs8 = nullptr; a.ext = nullptr;
  const int e = (int)blockIdx.x;
  uint32_t seen = 0;
  for (int k = 0; k < K; ++k) {
    if (x.slab) {
      if (threadIdx.x == 0) xchg_wait_slab(x, (uint32_t)k, seen);
      __syncthreads();
    }
    int lane_i = (int)threadIdx.x;
    asm volatile("" : "+v"(lane_i));
    philox1_body<false, true>(a, e, t0 + (uint32_t)k, (uint32_t)k, lane_i, k == 0, k == K - 1);      // the agent part stays in LDS from the first step to the last
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (x.slab) {
      // the row of step k - 1 is in memory by now (this step's drain covered its store): counted; then this step's row, not waited for
      if (threadIdx.x == 0 && k > 0) xchg_count(x, (uint32_t)(k - 1), e >> 5);
      pack_row_from_obs(x.slab + ((size_t)(k % x.ring) * (size_t)a.n + (size_t)e) * OBS_PACKED, a.obs + (size_t)e * OBS_TOTAL, (int)threadIdx.x);
    }
  }
  if (x.slab && K > 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) xchg_count(x, (uint32_t)(K - 1), e >> 5);
  }
}

struct ResetArgs {
  EnvState* st; EnvCold* cold; const uint64_t* seeds; const uint8_t* env_mask;
  int32_t* obs; float* reward; uint8_t* done; uint32_t* err; uint8_t* mask;
  int n, steps, rng_mode, policy;
  uint32_t topo;
  uint8_t* obs8;               // packed exchange row of the reset observations (multi-GPU), or null
};
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
__global__ void k_random_actions(int32_t* actions, int n, uint64_t seed0, uint32_t t, int e0 = 0) {      // episodes e0 .. n - 1
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
constexpr int RPG = 2;      // policy groups of a rollout: group of episode e = (e / P) % RPG
// gate of a policy pass: returns when every episode of policy group g has its packed row of the step in slot `slot` in memory (the step kernel
// counts them per partition, RunArgs.act_ready), and hands the counters back zeroed.  One wave, partitions on lanes; gives up after `ticks`.
__global__ __launch_bounds__(WAVE) void k_rollout_gate(uint32_t* cnt, int P, int ring, int g, int slot, int n, long long ticks, uint32_t* fail) {
  const long long t0 = wall_clock64();
  for (int p = (int)threadIdx.x; p < P; p += (int)blockDim.x) {
    const int ne = (n - p + P - 1) / P;                       // episodes p, p + P, ..: index i is of group i % RPG
    const int want = (ne - g + RPG - 1) / RPG;
    if (want <= 0) continue;
    uint32_t* c = cnt + ((size_t)p * RPG + (size_t)g) * (size_t)ring + slot;
    int naps = 1;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)want) {
      for (int q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(32);
      if (naps < 8) naps <<= 1;
      if (wall_clock64() - t0 > ticks) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// stand-in policies for one policy group (bench.py, tests): uniform random indices (the draws of k_random_actions), or indices computed FROM the
// packed observations of the step before (a policy that ignores its input proves nothing about the hand-over)
__global__ void k_rollout_random_policy(int32_t* act, int n, int P, int g, uint64_t seed0, uint32_t t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * NBLUE) return;
  const int e = i / NBLUE, b = i % NBLUE;
  if ((e / P) % RPG != g) return;
  act[i] = random_blue_action(seed0, t, e, b);
}
__device__ __host__ inline uint32_t rollout_obs_hash(const uint32_t* row) {      // 37 words of a packed observation row
  uint32_t hsh = 2166136261u;
  for (int w = 0; w < OBS_PACKED / 4; ++w) { hsh ^= row[w]; hsh *= 16777619u; }
  return hsh;
}
__global__ void k_rollout_hash_policy(int32_t* act, const uint8_t* packed, int n, int P, int g, uint32_t j) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || (e / P) % RPG != g) return;
  const uint32_t hsh = rollout_obs_hash(reinterpret_cast<const uint32_t*>(packed + (size_t)e * OBS_PACKED));
  for (int b = 0; b < NBLUE; ++b) act[e * NBLUE + b] = (int32_t)((hsh + 2654435761u * (uint32_t)(b + 1) + 40503u * j) % (uint32_t)(b == 4 ? ACT_LONG : ACT_SHORT));
}
// the packed rows of the observations as they stand in the int32 buffer (what a rollout's first policy pass reads)
__global__ __launch_bounds__(WAVE) void k_pack_obs_rows(uint8_t* packed, const int32_t* obs, int n) {
  const int e = blockIdx.x;
  if (e < n) pack_row_from_obs(packed + (size_t)e * OBS_PACKED, obs + (size_t)e * OBS_TOTAL, (int)threadIdx.x);
}

// ---------------------------------------------------------------- handle
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

struct cc4_handle {
  cc4_config cfg;
  hipStream_t stream = nullptr;
  EnvState* d_state = nullptr; EnvCold* d_cold = nullptr;
  size_t cold_row = 0;             // bytes per cold row: fixed part + the containers sized from cfg.steps (cold_row_bytes)
  int32_t* d_actions = nullptr; uint8_t* d_msgs = nullptr; uint64_t* d_seeds = nullptr; uint8_t* d_envmask = nullptr;
  int32_t* d_obs = nullptr; float* d_reward = nullptr; uint8_t* d_done = nullptr; uint32_t* d_err = nullptr;
  uint8_t* d_mask = nullptr; uint64_t* d_rng = nullptr;
  // d_obs | d_reward | d_err | d_done are ONE allocation (base d_obs), d_actions | d_msgs another (base d_actions): cc4_step_fetch moves
  // a step's inputs and outputs with one copy each; small batches go through pinned staging buffers (a copy to or from pageable
  // memory is staged by the runtime anyway, synchronously and per call)
  size_t out_bytes = 0, in_bytes = 0;
  uint8_t* pin_out = nullptr; uint8_t* pin_in = nullptr;
  // Handles of up to SMALL_IO_ENVS episodes (the single-episode wrapper surface) keep both blocks in pinned HOST memory the device reads
  // and writes directly: the step kernel fetches its five action indices over PCIe and posts its results there, so a step is a launch and
  // one host wait -- no copy engine in either direction (each DMA costs ~10 us of latency for a few hundred bytes).  CC4_SMALL_IO=0: off.
  static constexpr int SMALL_IO_ENVS = 16;
  bool small_io = false;
  // cc4_keep_previous / cc4_replay_logged (small handles): the rows as they stood before the last step, so that the step can be repeated
  // with the event log on when -- and only when -- somebody asks what happened in it (the single-episode wrapper surface: flat
  // observations need no log, and the logging build of the numpy-stream kernel walks its green actions serially: +30 us per step)
  bool keep_prev = false, prev_valid = false;
  EnvState* d_prev_state = nullptr; EnvCold* d_prev_cold = nullptr; uint8_t* d_prev_out = nullptr;
  const int32_t* prev_actions = nullptr; const uint8_t* prev_msgs = nullptr; bool prev_full_obs = false, prev_ext = false;
  // byte observations and gathered observations ([world*N][578]) in a ring of OBS_RING buffers: the all-gather of step t
  // overlaps later steps, and the compute stream waits for the communication stream only once per OBS_WAIT_EVERY steps
  // (a cross-stream wait in front of every launch costs the stream ~10 us)
  static constexpr int OBS_RING = 8, OBS_WAIT_EVERY = 4;
  static constexpr int MAX_GROUPS = cc4_handle_max_groups;          // launches per step (episode groups, below); CC4_GROUPS may ask for up to this many
  uint8_t* d_obs8[OBS_RING] = {};
  uint8_t* d_all_obs8[OBS_RING] = {};
  long long gather_seq[OBS_RING] = {};           // sequence number of the last all-gather that read buffer b (0 = none)
  long long gathers_issued = 0, gathers_waited = 0;
  hipEvent_t tev_start[cc4_handle_max_groups] = {}, tev_stop[cc4_handle_max_groups] = {};   // timing events the NEXT launch of a group carries (cc4_run_random_steps)
  long long comm_delay_ticks = 0;                // debug: spin this long on the communication stream ahead of every all-gather
  long long gather_stalls = 0;                   // a step launch found the all-gather it had to wait for still running
  long long stat_steps = 0; double stat_launch_us = 0, stat_gather_us = 0;   // cc4_host_stats
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_step[OBS_RING][MAX_GROUPS] = {}, ev_comm[OBS_RING] = {};   // ev_step[b][g]: group g's launch that wrote buffer b; ev_comm[q % OBS_RING]: all-gather number q has completed
  int obs_buf = 0;                               // buffer written by the most recent step
  int gather_buf = -1;                           // buffer of the most recent all-gather (-1: none issued)
  bool step_event_attached = false;              // ev_step[obs_buf] was recorded by the launch of that step itself
  // A step of a large batch is issued as `ngroups` launches, one per contiguous group of episodes, each group on its own HIP
  // stream (group 0 on `stream`): episodes are independent, a group's next step depends only on its own previous one, so while
  // one group's launch drains -- its last blocks running on a half-empty chip -- the other group's launch fills the free
  // slots, and the chip stays full across step boundaries.  Measured on MI355X (r03, 8192 episodes, counter mode): one launch
  // per step 507 M agent-env steps/s, two groups of 4096 on two streams 639 M (a single 32768-episode launch per step: 605 M).
  int ngroups = 1;
  int cus = 256;                                 // compute units of the device
  int glo[MAX_GROUPS + 1] = {};                  // group g = episodes [glo[g], glo[g + 1])
  hipStream_t gstream[MAX_GROUPS] = {};          // gstream[0] == stream
  hipEvent_t gev[MAX_GROUPS] = {};               // group stream -> main stream ordering (join_groups)
  hipEvent_t mev = nullptr;                      // main stream -> group streams ordering (fork_groups)
  bool auto_groups = true;                       // the number of groups is the library's choice (no CC4_GROUPS)
  bool groups_busy = false;                      // a group stream other than the main one may hold unfinished step launches
  bool joined_between = false;                   // something ordered the main stream behind all groups (or waited for them) since the last step launches:
                                                 // the caller works on the WHOLE batch between steps (launch_step: one launch then, not one per group)
  bool main_ahead = false;                       // the main stream holds work the group streams have not been ordered behind
  unsigned long long* d_prof = nullptr;
  uint32_t* d_reset_ws = nullptr;    // k_step_philox1's generation work area, [num_envs][RESET_WS_WORDS]
  uint8_t* d_unpacked = nullptr;                 // [world*N][578] bytes: cc4_unpack_obs_device
  int evlog_on = 0;               // cc4_enable_event_log
  // externally submitted red / green actions (cc4_step_ex).  Once a handle has taken any, its steps run the full builds of the
  // kernels (an action queued for several ticks carries its own rates into later steps), with d_ext all XA_NONE for the steps
  // that submit nothing
  // the persistent run kernel (k_run_philox1: K steps of the batch in one launch; RunArgs): per-partition ticket
  // counters, per-episode progress, partition owners in ONE buffer (cleared by one memset per call), the CU table
  uint32_t* d_run = nullptr;      // [P ticket | P owner | n progress]
  int32_t* d_slot_part = nullptr; // [CC4_SLOTS] CU slot id -> 1 + partition (persist_setup)
  unsigned long long* d_timeline = nullptr;   // CC4_PERSIST_TIMELINE: per-wave time stamps of the current persistent launch
  size_t run_words = 0;           // words of d_run
  int run_P = 0, run_grid = 0;    // partitions (= CUs that take waves; XCD pools: = XCDs), waves per launch; 0: the persistent path is off
  int run_G = 0;                  // exchange groups of the persistent kernel (episode e counts in group e % run_G): the device's CUs
  int run_pool = 2;               // the persistent kernel's schedule (RunArgs.pool; CC4_PERSIST_SCHED): 2 = per-CU partitions balanced inside the XCD,
                                  // 0 = the per-CU partitions of r04 / r05 (only a call's tail is shared), 1 = XCD pools (experiment)
  uint8_t xcc_lo[8] = {0}, xcc_n[8] = {0};
  int run_thr = 16;               // schedule 2: a wave helps the partition that lags most once its own is more than this many tickets ahead (CC4_PERSIST_THR)
  uint8_t xcc_pool[8] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
  uint32_t* d_pool = nullptr;     // [2][8][TK_STRIDE] the pools' ticket counters, one set per call parity
  int run_SA = 0, run_SB = 1, run_nB = 0, run_single = 0;   // runs of steps (RunArgs.SA ..; CC4_PERSIST_RUNS="SA,SB,nB,single"; SA = 1: every step an item, as in r05;
                                                            // SA = 0: chosen per call -- 4 steps, 8 in calls of 64 steps and more: profiles/r06_runs_ab.txt, r06_sched_ab2.txt)
  uint32_t pool_base = 0;         // steps every episode's progress word stands at (XCD pools: the words are not cleared between calls)
  int pool_parity = 0;
  int persist_state = -1;         // -1 off / unavailable, 0 not set up yet (persist_setup on first use), 1 on
  bool whole_batch_steps = true;  // CC4_WHOLE_BATCH_STEPS=0: the step entry points always launch per group (A/B)
  int persist_order = 0;          // RunArgs.order (CC4_PERSIST_ORDER)
  int run_margin = 0;             // episode blocks per CU the one-launch forms leave free (choose_run_form)
  // the per-step hand-off out of the one-launch kernels (XchgArgs): with a communicator, cc4_run_random_steps stays ONE launch and the
  // communication stream follows the kernel's per-step counters (xchg_*)
  static constexpr int XRING = 32;
  bool xchg_on = false;           // cc4_comm_init; CC4_EXCHANGE_INKERNEL=0 keeps the per-step launches
  int xchg_chunk = 8;             // steps per gate / publish on the communication stream (CC4_EXCHANGE_CHUNK; their slabs go out in ONE all-gather: the host
                                  // pays ~25 us to enqueue a wait, an all-gather and a publish -- more than a step of a small batch lasts)
  uint8_t* d_xslab = nullptr;     // [XRING][n][OBS_PACKED]
  uint8_t* d_xall = nullptr;      // [XRING][world * n][OBS_PACKED]
  int khz = 0;                    // wall-clock rate (hipDeviceAttributeWallClockRate), asked once
  // ---- rollouts with the policy in the loop (cc4_rollout_begin .. cc4_rollout_end)
  int32_t* d_ract = nullptr;      // [2][n][5] action slots (step j reads slot j % 2)
  uint32_t* d_rready = nullptr;   // [RPG][32] words: actions of steps < value are published for the group
  uint32_t* d_rcnt = nullptr;     // [P][RPG][XRING] episodes of (partition, policy group) whose packed row of step j is in memory (slot j % XRING)
  uint32_t* d_rfail = nullptr;    // [1] a gate gave up
  hipStream_t policy_stream = nullptr;
  hipEvent_t rev = nullptr;       // the rollout's starting observations are packed (slab XRING - 1)
  int rollout_k = 0;              // > 0: a rollout of that many steps is in flight
  int rollout_watchdog_ms = 2000;
  int obs8_from_slab = -1;        // >= 0: the per-step ring's current buffer is to be filled from this slab of the exchange ring (xchg_end), when somebody reads it
  uint32_t* d_xflags = nullptr;   // [0] gathered, [1] timeout (what the waits poll)
  uint32_t* d_xgcnt = nullptr;    // [groups][XRING] group counters (xchg_count)
  uint32_t* h_xtimeout = nullptr; // pinned host word the kernel raises when a wait gives up (read without a copy)
  uint32_t* d_xtimeout = nullptr; // its device address
  int xflags_clean = 0;           // the flags are cleared already (behind the previous call) and xev says when
  hipEvent_t xev = nullptr;
  long long xchg_calls = 0, xchg_timeouts = 0;
  int xchg_watchdog_ms = 2000;
  uint8_t* last_gathered = nullptr;   // gathered rows of the most recent all-gather, whichever path issued it
  uint8_t* d_xlog = nullptr;      // debug (cc4_debug_gather_log): every gathered slab in issue order, [xlog_cap][world * n][OBS_PACKED]
  int xlog_cap = 0, xlog_n = 0;
  bool persist_refused = false;   // persist_setup found an unexpected picture (said so on stderr; cc4_run_kernel reports the per-step kernel)
  // CC4_PERSIST_VERIFY=1: every one-launch call of cc4_run_random_steps is repeated with per-step launches on a shadow handle that starts
  // from a copy of this handle's rows, and the two results are compared episode by episode (verify_*)
  bool verify = false, is_shadow = false;
  int verify_every = 1024;        // without CC4_PERSIST_VERIFY: every verify_every-th persistent call is checked all the same (CC4_PERSIST_VERIFY_EVERY; 0: never)
  uint64_t persist_calls = 0;
  cc4_handle* shadow = nullptr;
  uint64_t* d_digest = nullptr;   // [num_envs] per-episode digest
  long long verify_calls = 0, verify_mismatches = 0;
  int persist_min_k = 10;         // shorter calls keep the per-step launches: a launch's ramp and tail cost a few steps' worth (with the tail's items shared
                                  // among the CUs of an XCD: K = 10: 733 vs 685 M, K = 20: 813 vs 742 M, K = 32: 857 vs 756 M; CC4_PERSIST_MIN_K)
  struct EnqPool* pool = nullptr; // one enqueue thread per group stream beyond the first (cc4_run_random_steps; enq_*)
  bool enq_threads = false;
  bool run1m = false;             // cc4_run_random_steps as ONE launch of k_run_philox1m (batches of the one-wave kernel that one launch holds)
  int multistep_minb = 5;         // which build of it: 5 (k_run_philox) or 8 blocks per CU (k_run_philox8)
  bool multistep = false;         // k_run_philox: cc4_run_random_steps as ONE launch, every block looping over the steps of its episode
  ExtAct* d_ext = nullptr;        // [num_envs][EXT_PER_ENV]
  bool ext_seen = false, ext_dirty = false;   // dirty: d_ext holds the records of an earlier step
  std::vector<ExtAct> h_ext;
  bool full_obs_next = true;      // the next step launch rewrites every observation value (fresh handle, restored state)
  uint32_t full_obs_gmask = 0;    // ... per group, for the group-wise launches of cc4_step_group_device
  bool philox_lean = false;       // k_step_philox1 (one wave per episode) instead of k_step_philox (cc4_create)
  int philox_minw = 1;            // which register budget of k_step_philox this batch size runs (1, 7 or 8 blocks per CU; cc4_create)
  ncclComm_t comm = nullptr; int rank = 0, world = 1;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> evs;                   // timing events of cc4_run_random_steps: [group][2 * timed group of launches + {start, stop}]
  std::string err;
};

static thread_local std::string g_create_err;

// The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a
// queue run their kernels one after the other.  Three launches per step fit the default; a fourth stream needs more queues
// (measured, r03 profiles/r03_hwq_sweep.txt: 8192 episodes, 3 / 4 launches per step: 715 / 445 M with 4 queues, 717 / 742 M with
// 8).  The variable is read when the runtime initialises, so it is set when this library is loaded (never overriding the
// user's choice) -- and whether the four streams of a handle really run side by side is measured on those very streams when the
// handle is created, not assumed (streams_run_concurrently): a process that initialised HIP earlier, or one whose other streams
// already occupy the queues (a second handle next to a busy first one), keeps three launches per step.
__attribute__((constructor)) static void cc4_runtime_env() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

__global__ void k_spin(long long cycles);
// 1 if kernels launched on the n given streams at the same time run side by side, 0 if some of them share a hardware queue and
// run one after the other (or the probe failed).  ~1 ms.
static int streams_run_concurrently(hipStream_t* st, int n) {
  if (n < 2) return 1;
  int khz = 100000, dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  if (khz <= 0) khz = 100000;
  const long long ticks = 300LL * khz / 1000;            // 300 us per kernel
  bool ok = true;
  double one = 0, all = 0;
  for (int pass = 0; pass < 2 && ok; ++pass) {             // pass 0: one stream (also warms the kernel up), pass 1: all of them
    const int m = pass ? n : 1;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < m; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, st[i], ticks);
    for (int i = 0; i < m && ok; ++i) ok = hipStreamSynchronize(st[i]) == hipSuccess;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    (pass ? all : one) = us;
  }
  return ok && all < 1.5 * (one > 300.0 ? one : 300.0);
}

#define HIPCHK(h, call)                                                                        \
  do {                                                                                         \
    hipError_t _e = (call);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(_e);                            \
      return -1;                                                                               \
    }                                                                                          \
  } while (0)

// How a step of this handle is cut into launches, and which build of the counter-mode kernel they run (the kernels are
// chosen from what one LAUNCH puts on a CU and from what the whole batch does).
static void configure_groups(cc4_handle* h, int ng) {
  const int n = h->cfg.num_envs, cus = h->cus;
  if (ng < 1) ng = 1;
  if (ng > cc4_handle::MAX_GROUPS) ng = cc4_handle::MAX_GROUPS;
  if (ng > n) ng = n;
  h->ngroups = ng;
  for (int g = 0; g <= ng; ++g) h->glo[g] = (int)(((long long)n * g) / ng);
  const int gsize = (n + ng - 1) / ng;                                 // episodes per launch
  const int bpc = (gsize + cus - 1) / cus;                             // episode blocks of one launch per CU
  const int bpc_all = (n + cus - 1) / cus;                             // ... of all launches of a step
  // four-wave kernel: one round of <= 5 blocks per CU runs the unconstrained build; else the build whose residency fills whole
  // rounds best (exactly 8 per CU -- 2048 episodes on 256 CUs -- is one round of the 8-block build)
  h->philox_minw = bpc <= 5 ? 1 : (bpc == 8 ? 8 : 7);
  // ... and with several launches per step what counts is what they put on a CU together (r03 profiles: three launches, register
  // budget 1 / 7 / 8: 1024 episodes 176 / 174 / 164 M, 2048: 295 / 302 / 278, 4096: 355 / 442 / 417)
  // (r04 flags, four launches: 1536 episodes 269 / 266 / 257, 2048: 329 / 333 / 323, 3072: 364 / 433 / 423, 4096: 369 / 471 / 490 -- one wave: 507)
  if (ng > 1) h->philox_minw = bpc_all <= 6 ? 1 : 7;
  if (const char* v = getenv("CC4_PHILOX_MINW")) h->philox_minw = atoi(v);   // tuning override: 1, 7 or 8
  // The one-wave-per-episode build when a single launch puts more than eight episodes on a CU, or the launches of a step
  // together more than thirteen.  Measured on MI355X (M agent-env steps/s, four waves / one wave per episode; r02, one launch per
  // step): 1024 episodes 168 / 131, 2048: 262 / 238, 2304: 249 / 260, 3072: 294 / 322, 4096: 319 / 395, 8192: 391 / 510;
  // (r03, three launches per step): 1024: 175 / 137, 2048: 295 / 249, 3072: 347 / 345, 4096: 442 / 422, 6144: 455 / 556,
  // 8192: 466 / 659.  (The same kernel with the host table in LDS as well is no faster anywhere.)
  // (after the r03 changes to the one-wave kernel -- event bytes staged, rows on cache-line boundaries -- with four launches per
  // step: 2048 episodes 312 / 276, 3072: 401 / 388, 4096: 442 / 487, 5120: 453 / 560, 6144: 458 / 627)
  h->philox_lean = bpc > 8 || bpc_all > 13;
  if (const char* v = getenv("CC4_PHILOX_LEAN")) h->philox_lean = atoi(v) != 0;   // tuning / test override
}

// Every API call other than the step launches works on the main stream: order it behind whatever the group streams still
// hold (device-side waits, no host synchronisation), and remember that the next step launches must be ordered behind it.
extern "C" __global__ void k_set_evlog(EnvCold* cold, size_t row_bytes, int n, uint32_t on);
static int join_groups(cc4_handle* h) {
  if (h->ngroups > 1) {
    if (h->groups_busy) {
      for (int g = 1; g < h->ngroups; ++g) {
        HIPCHK(h, hipEventRecord(h->gev[g], h->gstream[g]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->gev[g], 0));
      }
      h->groups_busy = false;
    }
    h->main_ahead = true;
    h->joined_between = true;
  }
  return 0;
}
static int sync_all(cc4_handle* h) {
  // (!groups_busy: whatever the group streams were given, the main stream already waits for -- join_groups, or the joined end of
  // cc4_run_random_steps -- and a host wait on an idle stream is not free: ~8 us each inside a short timed region)
  for (int g = h->ngroups - 1; g >= 1; --g) if (h->groups_busy) HIPCHK(h, hipStreamSynchronize(h->gstream[g]));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->groups_busy = false;
  h->joined_between = true;
  return 0;
}

// rand: draw the blue actions inside the step kernel from (seed0, t) and record them in the handle's action buffer
// one group's launch of a step: the kernel cc4_create picked for this handle, on the group's stream, carrying `start` / `stop` as the
// launch's own timing events (or null)
static void launch_range(cc4_handle* h, StepArgs a, int e0, int e1, hipStream_t st, bool full, hipEvent_t start, hipEvent_t stop);
static void launch_group(cc4_handle* h, StepArgs a, int g, bool full, hipEvent_t start, hipEvent_t stop) {
  launch_range(h, a, h->glo[g], h->glo[g + 1], h->gstream[g], full, start, stop);
}
static void launch_range(cc4_handle* h, StepArgs a, int e0, int e1, hipStream_t st, bool full, hipEvent_t start, hipEvent_t stop) {
  a.e0 = e0; a.n = e1;
  const size_t lds1 = offsetof(EnvState, hd);     // one-wave kernels: the agent part
  const dim3 grid(a.n - a.e0);
#ifdef CC4_DEV_FAST     // kernel experiments (tools/ab/ab.sh): only the one-wave counter-mode kernel is instantiated -- a quarter of the compile time
  if (h->philox_lean) hipExtLaunchKernelGGL(k_step_philox1<false>, grid, dim3(WAVE), lds1, st, start, stop, 0, a);
  else hipExtLaunchKernelGGL((k_step_philox<false, 1>), grid, dim3(PT), sizeof(EnvState), st, start, stop, 0, a);
#else
  if (h->cfg.rng_mode == 1) {
    if (h->philox_lean) {
      if (full || h->d_prof) hipExtLaunchKernelGGL(k_step_philox1<true>, grid, dim3(WAVE), lds1, st, start, stop, 0, a);
      else hipExtLaunchKernelGGL(k_step_philox1<false>, grid, dim3(WAVE), lds1, st, start, stop, 0, a);
    }
    else if (full) hipExtLaunchKernelGGL((k_step_philox<true, 1>), grid, dim3(PT), sizeof(EnvState), st, start, stop, 0, a);
    else if (h->philox_minw == 8) hipExtLaunchKernelGGL((k_step_philox<false, 8>), grid, dim3(PT), sizeof(EnvState), st, start, stop, 0, a);
    else if (h->philox_minw == 7) hipExtLaunchKernelGGL((k_step_philox<false, 7>), grid, dim3(PT), sizeof(EnvState), st, start, stop, 0, a);
#ifndef CC4_SMALL_MINW
#define CC4_SMALL_MINW 1
#endif
    else hipExtLaunchKernelGGL((k_step_philox<false, CC4_SMALL_MINW>), grid, dim3(PT), sizeof(EnvState), st, start, stop, 0, a);
  } else {
    if (full) hipExtLaunchKernelGGL(k_step<true>, grid, dim3(WAVE), lds1, st, start, stop, 0, a);
    else hipExtLaunchKernelGGL(k_step<false>, grid, dim3(WAVE), lds1, st, start, stop, 0, a);
  }
#endif
}

// ---- one enqueue thread per group stream (cc4_run_random_steps without a communicator).  The groups of a batch never wait for each
// other, so their launches need not come from one thread: the first launch on a stream that has been synchronised costs the calling
// thread ~10 us (3.5 us in the steady state), four in a row delay the last group's first kernel by 30-45 us in every timed region;
// issued side by side they cost one.  A worker spins for 200 us after a call (CC4_ENQ_SPIN_US; a loop of calls keeps it hot), then sleeps.
struct EnqPool {
  std::vector<std::thread> th;
  std::mutex mu; std::condition_variable cv;
  std::atomic<uint64_t> gen{0};
  std::atomic<int> pending{0}, failed{0};
  std::atomic<bool> quit{false};
  int spin_us = 200;            // how long a worker spins for the next call before it parks on the condition variable (CC4_ENQ_SPIN_US): a loop of
                                // calls with nothing in between keeps it hot, a caller that does host work between bursts gets its cores back
  StepArgs a{}; int k = 0; uint32_t t0 = 0; bool full = false, first_full_obs = false, join = false;
  hipEvent_t start[cc4_handle_max_groups] = {}, stop[cc4_handle_max_groups] = {};
};
static void enq_run_group(cc4_handle* h, EnqPool* P, int g) {
  StepArgs a = P->a;
  for (int i = 0; i < P->k; ++i) {
    a.rand_t = P->t0 + (uint32_t)i;
    a.full_obs = (i == 0 && P->first_full_obs) ? 1 : 0;
    launch_group(h, a, g, P->full, i == 0 ? P->start[g] : nullptr, i == P->k - 1 ? P->stop[g] : nullptr);
  }
  if (g > 0 && P->join && hipEventRecord(h->gev[g], h->gstream[g]) != hipSuccess) P->failed.fetch_add(1);   // the main stream waits for it: one host wait per call
  if (hipGetLastError() != hipSuccess) P->failed.fetch_add(1);
}
static void enq_worker(cc4_handle* h, EnqPool* P, int g) {
  (void)hipSetDevice(h->cfg.device_id);
  uint64_t seen = 0;
  for (;;) {
    auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (P->gen.load(std::memory_order_acquire) == seen && !P->quit.load(std::memory_order_relaxed)) {
      __builtin_ia32_pause();
      if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(P->spin_us)) {
        std::unique_lock<std::mutex> lk(P->mu);
        P->cv.wait(lk, [&] { return P->gen.load(std::memory_order_acquire) != seen || P->quit.load(); });
      }
    }
    if (P->quit.load()) return;
    seen = P->gen.load(std::memory_order_acquire);
    enq_run_group(h, P, g);
    P->pending.fetch_sub(1, std::memory_order_release);
  }
}
static void enq_pool_start(cc4_handle* h) {
  if (h->pool || h->ngroups < 2) return;
  h->pool = new EnqPool;
  if (const char* v = getenv("CC4_ENQ_SPIN_US")) h->pool->spin_us = atoi(v) > 0 ? atoi(v) : 0;
  for (int g = 1; g < h->ngroups; ++g) h->pool->th.emplace_back(enq_worker, h, h->pool, g);
}
static void enq_pool_stop(cc4_handle* h) {
  if (!h->pool) return;
  { std::lock_guard<std::mutex> lk(h->pool->mu); h->pool->quit.store(true); }
  h->pool->cv.notify_all();
  for (auto& t : h->pool->th) t.join();
  delete h->pool; h->pool = nullptr;
}

// api_step: one of the step entry points (cc4_step / _ex / _fetch / _device), as opposed to the loop of cc4_run_random_steps.  When the caller did
// something with the WHOLE batch since the last step (an upload, a fetch, a policy kernel over all observations: anything that went through
// join_groups or waited for the streams), the groups cannot run ahead of each other anyway -- a launch per group then pays a fork and a join
// across streams per step for nothing: ONE launch on the main stream (8192 episodes, k_random_actions + cc4_step_device per step: 334 -> 583 M;
// a loop of cc4_step_device with nothing in between keeps the groups and their overlap across steps).
static int launch_step(cc4_handle* h, const int32_t* d_actions, const uint8_t* d_msgs, bool rand = false, uint64_t seed0 = 0,
                       uint32_t t = 0, bool ext_uploaded = false, bool api_step = false) {
  if (h->ext_seen && h->ext_dirty && !ext_uploaded) {     // this step submits no red / green action: every record says so
    if (join_groups(h)) return -1;
    HIPCHK(h, hipMemsetAsync(h->d_ext, 0xFF, (size_t)h->cfg.num_envs * EXT_PER_ENV * sizeof(ExtAct), h->stream));
    h->ext_dirty = false;
  }
  const bool full = h->evlog_on || h->ext_seen;
  // the byte-observation buffer about to be overwritten may still be read by an overlapped all-gather
  int buf = h->comm ? (h->obs_buf + 1) % cc4_handle::OBS_RING : 0;
  if (h->comm && h->gather_seq[buf] > h->gathers_waited) {
    // the last all-gather that read this buffer must be complete; wait for a slightly newer one (the communication stream is
    // in order), so the next OBS_WAIT_EVERY-1 launches need no wait of their own -- but never for the newest one, which is
    // the one meant to overlap this step
    long long q = h->gather_seq[buf] + cc4_handle::OBS_WAIT_EVERY - 1;
    if (q > h->gathers_issued - 1) q = h->gathers_issued - 1;
    if (q < h->gather_seq[buf]) q = h->gather_seq[buf];
    // waited for by the host, not by the stream: the all-gather in question is several steps old and normally complete, and
    // a wait packet in the compute queue costs stream time whether or not it has to wait
    hipError_t qs = hipEventQuery(h->ev_comm[q % cc4_handle::OBS_RING]);
    if (qs == hipErrorNotReady) { h->gather_stalls++; HIPCHK(h, hipEventSynchronize(h->ev_comm[q % cc4_handle::OBS_RING])); }
    else HIPCHK(h, qs);
    h->gathers_waited = q;
  }
  const bool whole = api_step && h->whole_batch_steps && h->ngroups > 1 && h->joined_between && !h->groups_busy && !h->comm;
  h->joined_between = false;
  if (!whole && h->ngroups > 1 && h->main_ahead) {   // e.g. an action upload or a reset on the main stream: the group streams start behind it
    HIPCHK(h, hipEventRecord(h->mev, h->stream));
    for (int g = 1; g < h->ngroups; ++g) HIPCHK(h, hipStreamWaitEvent(h->gstream[g], h->mev, 0));
    h->main_ahead = false;
  }
  if (h->keep_prev) {      // the rows as they stand before this step (cc4_replay_logged); a small handle: one launch per step, main stream
    const size_t n = (size_t)h->cfg.num_envs;
    HIPCHK(h, hipMemcpyAsync(h->d_prev_state, h->d_state, n * sizeof(EnvState), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_prev_cold, h->d_cold, n * h->cold_row, hipMemcpyDeviceToDevice, h->stream));
    h->prev_actions = d_actions; h->prev_msgs = d_msgs; h->prev_full_obs = h->full_obs_next; h->prev_ext = h->ext_seen;
    h->prev_valid = !rand;
  }
  StepArgs a{h->d_state, h->d_cold, d_actions, d_msgs, h->d_obs, h->d_reward, h->d_done, h->d_err,
             h->comm ? h->d_obs8[buf] : nullptr, rand ? h->d_actions : nullptr, seed0, t,
             h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), h->full_obs_next ? 1 : 0,
             (uint32_t)h->cfg.topology_seed, h->d_prof, h->d_reset_ws, h->ext_seen ? h->d_ext : nullptr, 0};
  h->full_obs_next = false;
#ifdef CC4_DEV_FAST
  if (h->cfg.rng_mode != 1 || full) { h->err = "CC4_DEV_FAST build: only k_step_philox1<false> and k_step_philox<false, 1> exist"; return -1; }
#endif
  if (whole) {
    hipEvent_t stop = h->tev_stop[0], start = h->tev_start[0];
    for (int g = 0; g < h->ngroups; ++g) h->tev_start[g] = h->tev_stop[g] = nullptr;
    // (fewer waves per CU so that the batch runs in whole rounds -- 8192 episodes: 16 per CU, two even rounds instead of 1.6 at 20 -- is slower at every
    // residency tried: 578 M at 20, 567 at 18, 538 at 16, 503 at 14; tools/ab/ab_whole_residency.sh)
    launch_range(h, a, 0, h->cfg.num_envs, h->stream, full, start, stop);
    HIPCHK(h, hipGetLastError());
    h->step_event_attached = false;
    h->main_ahead = true;            // (the group streams have not been ordered behind this launch)
    h->obs_buf = buf;
    return 0;
  }
  for (int g = 0; g < h->ngroups; ++g) {
    // with a communicator, the launch carries ev_step[buf][g] as its stop event: the event rides on the kernel's own completion
    // signal, where a separate hipEventRecord would put a marker packet between two step kernels (~5 us of idle stream time)
    hipEvent_t stop = h->comm ? h->ev_step[buf][g] : h->tev_stop[g];
    hipEvent_t start = h->comm ? nullptr : h->tev_start[g];        // timing rides on the kernels' own signals too: no marker packets
    h->tev_start[g] = h->tev_stop[g] = nullptr;
    launch_group(h, a, g, full, start, stop);
    HIPCHK(h, hipGetLastError());
  }
  h->step_event_attached = h->comm != nullptr;
  if (h->ngroups > 1) h->groups_busy = true;
  h->obs_buf = buf;
  h->obs8_from_slab = -1;           // (this step's packed rows are in the ring buffer it wrote)
  return 0;
}

// the persistent kernel of a handle's mode
static const void* persist_kernel(const cc4_handle* h) {
#ifndef CC4_DEV_FAST
  if (h->cfg.rng_mode == 0) return reinterpret_cast<const void*>(k_run_pcg);
#endif
  return reinterpret_cast<const void*>(k_run_philox1);
}


// Which form cc4_run_random_steps takes on this handle (decided at cc4_create, again at cc4_comm_init): the multi-step form of the four-wave
// kernel (k_run_philox / k_run_philox8) for batches the chip holds at once, the plain multi-step form of the one-wave kernel (k_run_philox1m)
// up to 20 episodes per CU, the persistent kernel beyond.  `margin` = episode blocks per CU the multi-step kernels leave free,
// `persist_margin` = waves per CU the persistent kernel's grid leaves free (see cc4_comm_init).
static int choose_run_form(cc4_handle* h, int margin, int persist_margin = -1) {
  const cc4_config* cfg = &h->cfg;
  if (persist_margin < 0) persist_margin = margin;
  h->multistep = false; h->run1m = false;
  if (cfg->rng_mode == 1 && !h->philox_lean) {
    // the multi-step form of the four-wave kernel (k_run_philox): for batches the chip holds at once
    int per_cu = 0, per_cu8 = 0;
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_run_philox, PT, sizeof(EnvState)));
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu8, k_run_philox8, PT, sizeof(EnvState)));
    h->multistep = per_cu - margin > 0 && cfg->num_envs <= (per_cu - margin) * h->cus;
    h->multistep_minb = 5;
    if (!h->multistep && per_cu8 > per_cu && cfg->num_envs <= (per_cu8 - margin) * h->cus) { h->multistep = true; h->multistep_minb = 8; }
    if (const char* v = getenv("CC4_MULTISTEP")) {            // 0: off; 1: on (the build that holds the batch); 5 / 8: that build
      const int m = atoi(v);
      h->multistep = m != 0;
      if (m == 5 || m == 8) h->multistep_minb = m;
    }
    if (getenv("CC4_PERSIST_DEBUG")) fprintf(stderr, "[cc4] k_run_philox: %d / %d blocks per CU resident (margin %d), multistep %d (build %d)\n", per_cu, per_cu8, margin, (int)h->multistep, h->multistep_minb);
  }
  if (cfg->rng_mode == 1 && !h->multistep) {
    // (whichever per-step kernel the handle runs: a batch of 2049-5120 episodes that cc4_step serves with the four-wave kernel is served here by the one-wave loop)
    // the plain multi-step form of the one-wave kernel (k_run_philox1m) where one launch holds the whole batch: 20 waves per CU
    // (4096 episodes 507 -> 709 M, 5120: 586 -> 811 M; beyond the residency the second round runs on a half-empty chip and four
    // streams of per-step launches win: 8192: 740 vs 789 M, 16384: 812 vs 864 M -- profiles/r04_run1m_ab.txt)
    int per_cu = 0;
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_run_philox1m, WAVE, offsetof(EnvState, hd)));
    h->run1m = per_cu - margin > 0 && cfg->num_envs <= (per_cu - margin) * h->cus;
    if (const char* v = getenv("CC4_RUN1")) h->run1m = atoi(v) != 0;
  }
  // one enqueue thread per group stream in cc4_run_random_steps (EnqPool): on where the host has cores to spare; CC4_ENQ_THREADS=0/1 decides otherwise
  h->enq_threads = std::thread::hardware_concurrency() >= 8;
  if (const char* v = getenv("CC4_ENQ_THREADS")) h->enq_threads = atoi(v) != 0;
  // the persistent run kernel of large batches (k_run_philox1): set up on first use (persist_setup); CC4_PERSIST=0 keeps it off
  h->persist_state = -1;
  h->run_margin = persist_margin;
  bool persist_mode = cfg->rng_mode == 1 && !h->multistep && !h->run1m;
#ifndef CC4_DEV_FAST
  persist_mode = persist_mode || cfg->rng_mode == 0;
#endif
  if (persist_mode) {
    int per_cu = 0;
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persist_kernel(h), WAVE, offsetof(EnvState, hd)));
    const int grid = (per_cu - persist_margin) * h->cus;
    // batches of more than the chip holds at once (with the tail shared, also just more).  The numpy-stream mode has no other one-launch form: there
    // the persistent kernel also serves batches from half the residency up (a partition of fewer episodes than the CU has waves just leaves waves idle)
    if (per_cu - persist_margin > 0 && (cfg->num_envs > grid || (cfg->rng_mode == 0 && 2 * cfg->num_envs > grid))) h->persist_state = 0;
  }
  if (const char* v = getenv("CC4_PERSIST")) { if (atoi(v) == 0) h->persist_state = -1; }
  return 0;
}

extern "C" {

const char* cc4_last_error(cc4_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }
size_t cc4_state_bytes(void) { return sizeof(EnvState); }
int cc4_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
size_t cc4_algorithmic_bytes_per_env_step(void) {
  // state row in + out, flat obs out (int32), actions in, reward + done + err out (DESIGN.md "algorithmic bytes")
  return 2 * sizeof(EnvState) + 4 * OBS_TOTAL + 4 * NBLUE + 4 + 1 + 4;
}
size_t cc4_hot_bytes(void) { return offsetof(EnvState, hd); }
const char* cc4_step_kernel(cc4_handle* h) {
  if (!h) return "";
  if (h->cfg.rng_mode != 1) return "k_step";
  return h->philox_lean ? "k_step_philox1" : "k_step_philox";
}

// the kernel cc4_run_random_steps launches on this handle as it stands (no communicator, no event log): the step kernel, once per
// step and group -- or one of the one-launch forms
const char* cc4_run_kernel(cc4_handle* h) {
  if (!h) return "";
  const bool plain = (!h->comm || h->xchg_on) && !h->evlog_on && !h->ext_seen && !h->d_prof;
  if (plain && h->multistep) return h->multistep_minb == 8 ? "k_run_philox8" : "k_run_philox";
  if (plain && h->run1m) return "k_run_philox1m";
  if (plain && h->persist_state >= 0) return h->cfg.rng_mode == 0 ? "k_run_pcg" : "k_run_philox1";      // (calls of fewer than persist_min_k steps: the per-step launches)
  return cc4_step_kernel(h);
}
static int persist_setup(cc4_handle* h);
const char* cc4_run_kernel_for(cc4_handle* h, int32_t k) {
  if (!h) return "";
  // (the persistent kernel's discovery pass runs on first use: asking which kernel a call of k steps will launch is such a use -- the answer depends on it,
  // and a caller that asks before its timed region keeps it out of that region)
  if (h->persist_state == 0 && !h->run1m && !h->multistep && (!h->comm || h->xchg_on) && !h->evlog_on && !h->ext_seen && !h->d_prof && k >= h->persist_min_k && hipSetDevice(h->cfg.device_id) == hipSuccess) (void)persist_setup(h);
  const char* r = cc4_run_kernel(h);
  if (k < 2) return cc4_step_kernel(h);
  if (!h->multistep && !h->run1m && h->persist_state >= 0 && k < h->persist_min_k) return cc4_step_kernel(h);
  return r;
}

int cc4_create(const cc4_config* cfg, cc4_handle** out) {
  if (!cfg || !out || cfg->num_envs <= 0 || cfg->steps <= 0 || cfg->red_policy < 0 || cfg->red_policy > 3 ||
      cfg->green_policy < 0 || cfg->green_policy > 2 || cfg->blue_policy < 0 || cfg->blue_policy > 1 || cfg->rng_mode < 0 || cfg->rng_mode > 1) { g_create_err = "cc4_create: bad config"; return -2; }
  if (cfg->topology_seed != 0 && cfg->rng_mode != 1) { g_create_err = "cc4_create: topology_seed needs rng_mode 1 (the numpy stream draws scenario and dynamics from one generator)"; return -2; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = "cc4_create: no HIP device available (libcc4 has no CPU fallback)";
    return -3;
  }
  if (cfg->device_id < 0 || cfg->device_id >= ndev) { g_create_err = "cc4_create: device_id out of range"; return -2; }
  cc4_handle* h = new cc4_handle();
  h->cfg = *cfg;
  *out = h;
  HIPCHK(h, hipSetDevice(cfg->device_id));
  {   // PCG64 jump table of the numpy-stream kernel (see wave_green_policy): A_k = M^k, B_k = 1 + M + .. + M^(k-1) mod 2^128
    PcgJump tab[WAVE + 1];
    const unsigned __int128 M = ((unsigned __int128)CC4_PCG_MULT_HI << 64) | CC4_PCG_MULT_LO;
    unsigned __int128 A = 1, B = 0;
    for (int k = 0; k <= WAVE; ++k) {
      tab[k].a_hi = (uint64_t)(A >> 64); tab[k].a_lo = (uint64_t)A; tab[k].b_hi = (uint64_t)(B >> 64); tab[k].b_lo = (uint64_t)B;
      B = B * M + 1; A = A * M;
    }
    HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(g_pcg_jump), tab, sizeof(tab)));
    uint32_t ot[OBS_FAST];
    for (int v = 0; v < OBS_FAST; ++v) ot[v] = obs_fast_entry(v);
    HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(g_obs_fast), ot, sizeof(ot)));
  }
  {
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, cfg->device_id));
    h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  {
    // Episode groups (see cc4_handle::ngroups).  Measured on MI355X (r03, profiles/r03_groups_sweep_philox.txt; M agent-env steps/s,
    // counter mode, 1 / 2 / 3 launches per step): 1024 episodes 165 / 173 / 175, 2048: 257 / 289 / 295, 4096: 392 / 413 / 442,
    // 8192: 503 / 629 / 659, 16384: 570 / 710 / 701; numpy stream, 8192 episodes: 272 / 352 / 363.  A fourth stream halves the
    // rate (the runtime's hardware queues), so three it is.  CC4_GROUPS overrides (1 .. 4).
    int ng = cfg->num_envs >= 1024 ? 3 : (cfg->num_envs >= 512 ? 2 : 1);
    // (a fourth launch where four streams really overlap: decided below, once the streams exist)
    h->auto_groups = getenv("CC4_GROUPS") == nullptr;
    if (const char* v = getenv("CC4_GROUPS")) ng = atoi(v);
    configure_groups(h, ng);
  }
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  h->gstream[0] = h->stream;
  for (int g = 1; g < h->ngroups; ++g) {   // only the streams that are used: the runtime spreads streams over few hardware queues
    HIPCHK(h, hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->gev[g], hipEventDisableTiming));
  }
  HIPCHK(h, hipEventCreateWithFlags(&h->mev, hipEventDisableTiming));
  if (h->auto_groups && h->ngroups == 3) {
    // a fourth launch per step where this handle's four streams really run side by side (8192 episodes 717 -> 742 M, 2048: 310 ->
    // 314 M, 1024: 179 -> 183 M; with two of them on one hardware queue: 445 M)
    HIPCHK(h, hipStreamCreateWithFlags(&h->gstream[3], hipStreamNonBlocking));
    if (streams_run_concurrently(h->gstream, 4)) {
      HIPCHK(h, hipEventCreateWithFlags(&h->gev[3], hipEventDisableTiming));
      configure_groups(h, 4);
    } else {
      (void)hipStreamDestroy(h->gstream[3]);
      h->gstream[3] = nullptr;
    }
  }
  if (!h->auto_groups && getenv("CC4_EXP_PROBE")) {   // experiment: explicit groups, but the streams / probe of the automatic path exist as well
    hipStream_t tmp[4];
    for (int g = 0; g < 4; ++g) { if (g < h->ngroups) tmp[g] = h->gstream[g]; else HIPCHK(h, hipStreamCreateWithFlags(&tmp[g], hipStreamNonBlocking)); }
    if (atoi(getenv("CC4_EXP_PROBE")) > 1) (void)streams_run_concurrently(tmp, 4);
  }
  size_t n = (size_t)cfg->num_envs;
  // experiment (CC4_EXP_MEM=1 fine-grained, 2 uncached): the episodes' rows in memory whose lines the vector L1 does not keep
  static const int exp_mem = getenv("CC4_EXP_MEM") ? atoi(getenv("CC4_EXP_MEM")) : 0;
  auto row_alloc = [&](void** p, size_t bytes) -> hipError_t {
    if (exp_mem == 1) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    if (exp_mem == 2) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
    return hipMalloc(p, bytes);
  };
  HIPCHK(h, row_alloc((void**)&h->d_state, n * sizeof(EnvState)));
  h->cold_row = cold_row_bytes(cfg->steps);
  HIPCHK(h, row_alloc((void**)&h->d_cold, n * h->cold_row));
  if (cfg->rng_mode == 1) HIPCHK(h, hipMalloc(&h->d_reset_ws, n * RESET_WS_WORDS * sizeof(uint32_t)));   // the one-wave kernel's generation work area
  h->in_bytes = n * NBLUE * sizeof(int32_t) + n * NBLUE * MSG_LEN;
  h->small_io = cfg->num_envs <= cc4_handle::SMALL_IO_ENVS;
  if (const char* v = getenv("CC4_SMALL_IO")) h->small_io = h->small_io && atoi(v) != 0;
  if (h->small_io) {
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), h->in_bytes, hipHostMallocDefault));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_actions), h->pin_in, 0));
    memset(h->pin_in, 0, h->in_bytes);
  } else
  HIPCHK(h, hipMalloc(&h->d_actions, h->in_bytes));
  h->d_msgs = reinterpret_cast<uint8_t*>(h->d_actions) + n * NBLUE * sizeof(int32_t);
  HIPCHK(h, hipMalloc(&h->d_seeds, n * sizeof(uint64_t)));
  HIPCHK(h, hipMalloc(&h->d_envmask, n));
  h->out_bytes = n * OBS_TOTAL * sizeof(int32_t) + n * sizeof(float) + n * sizeof(uint32_t) + n;
  if (h->small_io) {
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), h->out_bytes, hipHostMallocDefault));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_obs), h->pin_out, 0));
  } else
  HIPCHK(h, row_alloc((void**)&h->d_obs, h->out_bytes));
  h->d_reward = reinterpret_cast<float*>(h->d_obs + n * OBS_TOTAL);
  h->d_err = reinterpret_cast<uint32_t*>(h->d_reward + n);
  h->d_done = reinterpret_cast<uint8_t*>(h->d_err + n);
  HIPCHK(h, hipMalloc(&h->d_mask, n * MASK_TOTAL));
  HIPCHK(h, hipMalloc(&h->d_rng, n * 7 * sizeof(uint64_t)));
  HIPCHK(h, hipMemsetAsync(h->d_state, 0, n * sizeof(EnvState), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_cold, 0, n * h->cold_row, h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_obs, 0, n * OBS_TOTAL * sizeof(int32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_done, 0, n, h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_err, 0, n * sizeof(uint32_t), h->stream));
  HIPCHK(h, hipEventCreate(&h->ev0));
  HIPCHK(h, hipEventCreate(&h->ev1));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (choose_run_form(h, 0)) return -1;
  if (const char* v = getenv("CC4_PERSIST_MIN_K")) h->persist_min_k = atoi(v);
  if (const char* v = getenv("CC4_PERSIST_ORDER")) h->persist_order = atoi(v);
  if (const char* v = getenv("CC4_WHOLE_BATCH_STEPS")) h->whole_batch_steps = atoi(v) != 0;
  if (const char* v = getenv("CC4_PERSIST_VERIFY")) h->verify = atoi(v) != 0;
  if (const char* v = getenv("CC4_PERSIST_VERIFY_EVERY")) h->verify_every = atoi(v) > 0 ? atoi(v) : 0;
  return 0;
}

void cc4_destroy(cc4_handle* h) {
  if (!h) return;
  enq_pool_stop(h);
  (void)hipSetDevice(h->cfg.device_id);
  for (int g = cc4_handle::MAX_GROUPS - 1; g >= 0; --g) if (h->gstream[g]) (void)hipStreamSynchronize(h->gstream[g]);
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  if (h->comm) ncclCommDestroy(h->comm);
  for (int b = 0; b < cc4_handle::OBS_RING; ++b) { for (int g = 0; g < cc4_handle::MAX_GROUPS; ++g) if (h->ev_step[b][g]) (void)hipEventDestroy(h->ev_step[b][g]); if (h->ev_comm[b]) (void)hipEventDestroy(h->ev_comm[b]); }
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  if (h->policy_stream) (void)hipStreamDestroy(h->policy_stream);
  if (h->rev) (void)hipEventDestroy(h->rev);
  for (void* p : {(void*)h->d_ract, (void*)h->d_rready, (void*)h->d_rcnt, (void*)h->d_rfail}) if (p) (void)hipFree(p);
  void* ptrs[] = {h->d_state, h->d_cold, h->small_io ? nullptr : (void*)h->d_actions, h->d_seeds, h->d_envmask, h->small_io ? nullptr : (void*)h->d_obs,
                  h->d_mask, h->d_rng, h->d_reset_ws, h->d_ext, h->d_run, h->d_slot_part, h->d_pool};     // (d_msgs, d_reward, d_err, d_done live inside d_actions / d_obs; small handles: pinned host memory, freed below)
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->shadow) { cc4_destroy(h->shadow); h->shadow = nullptr; (void)hipSetDevice(h->cfg.device_id); }
  for (void* p : {(void*)h->d_prev_state, (void*)h->d_prev_cold, (void*)h->d_prev_out}) if (p) (void)hipFree(p);
  if (h->d_digest) (void)hipFree(h->d_digest);
  if (h->pin_in) (void)hipHostFree(h->pin_in);
  if (h->pin_out) (void)hipHostFree(h->pin_out);
  for (int b = 0; b < cc4_handle::OBS_RING; ++b) { if (h->d_obs8[b]) (void)hipFree(h->d_obs8[b]); if (h->d_all_obs8[b]) (void)hipFree(h->d_all_obs8[b]); }
  if (h->d_unpacked) (void)hipFree(h->d_unpacked);
  for (void* p : {(void*)h->d_xslab, (void*)h->d_xall, (void*)h->d_xflags, (void*)h->d_xlog, (void*)h->d_xgcnt}) if (p) (void)hipFree(p);
  if (h->h_xtimeout) (void)hipHostFree(h->h_xtimeout);
  if (h->xev) (void)hipEventDestroy(h->xev);
  for (hipEvent_t e : h->evs) if (e) (void)hipEventDestroy(e);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (int g = 1; g < cc4_handle::MAX_GROUPS; ++g) { if (h->gev[g]) (void)hipEventDestroy(h->gev[g]); if (h->gstream[g]) (void)hipStreamDestroy(h->gstream[g]); }
  if (h->mev) (void)hipEventDestroy(h->mev);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int cc4_reset(cc4_handle* h, const uint64_t* seeds, const uint8_t* env_mask) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  size_t n = (size_t)h->cfg.num_envs;
  if (seeds) HIPCHK(h, hipMemcpyAsync(h->d_seeds, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  if (env_mask) HIPCHK(h, hipMemcpyAsync(h->d_envmask, env_mask, n, hipMemcpyHostToDevice, h->stream));
  ResetArgs a{h->d_state, h->d_cold, seeds ? h->d_seeds : nullptr, env_mask ? h->d_envmask : nullptr, h->d_obs, h->d_reward,
              h->d_done, h->d_err, h->d_mask, h->cfg.num_envs, h->cfg.steps, h->cfg.rng_mode,
              (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), (uint32_t)h->cfg.topology_seed,
              h->comm ? h->d_obs8[h->obs_buf] : nullptr};
  // with a communicator the reset also writes the packed exchange row of its observations into the current ring buffer; an
  // overlapped all-gather may still be reading that buffer
  if (h->comm && h->gathers_issued > h->gathers_waited) { HIPCHK(h, hipStreamSynchronize(h->comm_stream)); h->gathers_waited = h->gathers_issued; }
  if (h->comm && h->obs8_from_slab >= 0) {      // the reset writes the current ring buffer's packed rows itself -- all of them, unless it is masked
    const size_t row = (size_t)h->cfg.num_envs * OBS_PACKED;
    if (env_mask) HIPCHK(h, hipMemcpyAsync(h->d_obs8[h->obs_buf], h->d_xslab + (size_t)h->obs8_from_slab * row, row, hipMemcpyDeviceToDevice, h->stream));
    h->obs8_from_slab = -1;
  }
  hipLaunchKernelGGL(k_reset, dim3(h->cfg.num_envs), dim3(WAVE), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  h->prev_valid = false;            // (cc4_replay_logged: no step to repeat)
  h->step_event_attached = false;   // the buffer's event must be recorded again before the next all-gather
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int cc4_step(cc4_handle* h, const int32_t* actions, const uint8_t* messages) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  size_t n = (size_t)h->cfg.num_envs;
  if (actions) HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, n * NBLUE * sizeof(int32_t), hipMemcpyDefault, h->stream));
  if (messages) HIPCHK(h, hipMemcpyAsync(h->d_msgs, messages, n * NBLUE * MSG_LEN, hipMemcpyDefault, h->stream));
  if (launch_step(h, actions ? h->d_actions : nullptr, messages ? h->d_msgs : nullptr, false, 0, 0, false, true)) return -1;
  return sync_all(h);
}

// The outputs of the last step (or reset) in one copy and one host synchronisation: observations, reward, done and error flags live in
// one device allocation.  Batches of up to PIN_MAX_ENVS episodes come through a pinned staging buffer (the copy is a real asynchronous
// DMA; a copy into pageable memory is staged by the runtime, call by call).  Any of the four pointers may be null.
constexpr int PIN_MAX_ENVS = 4096;
static int fetch_outputs(cc4_handle* h, int32_t* obs, float* reward, uint8_t* done, uint32_t* err) {
  if (join_groups(h)) return -1;
  const size_t n = (size_t)h->cfg.num_envs;
  const size_t b_obs = n * OBS_TOTAL * sizeof(int32_t), b_rew = n * sizeof(float), b_err = n * sizeof(uint32_t);
  if (h->small_io) {                                          // the kernels wrote into pinned host memory: wait, then read it
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (obs) memcpy(obs, h->pin_out, b_obs);
    if (reward) memcpy(reward, h->pin_out + b_obs, b_rew);
    if (err) memcpy(err, h->pin_out + b_obs + b_rew, b_err);
    if (done) memcpy(done, h->pin_out + b_obs + b_rew + b_err, n);
    return 0;
  }
  if (h->cfg.num_envs <= PIN_MAX_ENVS) {
    if (!h->pin_out) HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), h->out_bytes, hipHostMallocDefault));
    const size_t lo = obs ? 0 : b_obs;                         // (a caller that wants no observations does not pay for them)
    HIPCHK(h, hipMemcpyAsync(h->pin_out + lo, reinterpret_cast<const uint8_t*>(h->d_obs) + lo, h->out_bytes - lo, hipMemcpyDefault, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (obs) memcpy(obs, h->pin_out, b_obs);
    if (reward) memcpy(reward, h->pin_out + b_obs, b_rew);
    if (err) memcpy(err, h->pin_out + b_obs + b_rew, b_err);
    if (done) memcpy(done, h->pin_out + b_obs + b_rew + b_err, n);
    return 0;
  }
  if (obs) HIPCHK(h, hipMemcpyAsync(obs, h->d_obs, b_obs, hipMemcpyDefault, h->stream));
  if (reward) HIPCHK(h, hipMemcpyAsync(reward, h->d_reward, b_rew, hipMemcpyDefault, h->stream));
  if (err) HIPCHK(h, hipMemcpyAsync(err, h->d_err, b_err, hipMemcpyDefault, h->stream));
  if (done) HIPCHK(h, hipMemcpyAsync(done, h->d_done, n, hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_fetch(cc4_handle* h, int32_t* obs, float* reward, uint8_t* done, uint32_t* err) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  return fetch_outputs(h, obs, reward, done, err);
}
// cc4_step + cc4_fetch with one host synchronisation in all: inputs up in one copy, the step's launches, outputs down in one copy
int cc4_step_fetch(cc4_handle* h, const int32_t* actions, const uint8_t* messages, int32_t* obs, float* reward, uint8_t* done, uint32_t* err) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  const size_t n = (size_t)h->cfg.num_envs;
  const size_t b_act = n * NBLUE * sizeof(int32_t), b_msg = n * NBLUE * MSG_LEN;
  if (h->small_io) {                                          // (every earlier launch has completed: each call of this surface ends with a host wait)
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (actions) memcpy(h->pin_in, actions, b_act);
    if (messages) memcpy(h->pin_in + b_act, messages, b_msg);
  } else if (h->cfg.num_envs <= PIN_MAX_ENVS && (actions || messages)) {
    if (!h->pin_in) HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), h->in_bytes, hipHostMallocDefault));
    if (actions) memcpy(h->pin_in, actions, b_act);
    if (messages) memcpy(h->pin_in + b_act, messages, b_msg);
    const size_t lo = actions ? 0 : b_act, hi = messages ? b_act + b_msg : b_act;
    HIPCHK(h, hipMemcpyAsync(reinterpret_cast<uint8_t*>(h->d_actions) + lo, h->pin_in + lo, hi - lo, hipMemcpyDefault, h->stream));
  } else {
    if (actions) HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, b_act, hipMemcpyDefault, h->stream));
    if (messages) HIPCHK(h, hipMemcpyAsync(h->d_msgs, messages, b_msg, hipMemcpyDefault, h->stream));
  }
  if (launch_step(h, actions ? h->d_actions : nullptr, messages ? h->d_msgs : nullptr, false, 0, 0, false, true)) return -1;
  return fetch_outputs(h, obs, reward, done, err);
}

// cc4_step plus the red / green entries of the step's `actions` dict (SimulationController.py:236-240)
int cc4_step_ex(cc4_handle* h, const int32_t* actions, const uint8_t* messages, const cc4_agent_action* red, const cc4_agent_action* green) {
  static_assert(sizeof(cc4_agent_action) == sizeof(ExtAct) && offsetof(cc4_agent_action, session) == offsetof(ExtAct, sid) &&
                offsetof(cc4_agent_action, rate0) == offsetof(ExtAct, rate0) && offsetof(cc4_agent_action, flags) == offsetof(ExtAct, flags),
                "cc4_agent_action (include/cc4.h) is ExtAct (csrc/cc4_state.h)");
  if (!red && !green) return cc4_step(h, actions, messages);
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  const size_t n = (size_t)h->cfg.num_envs;
  if (!h->d_ext) HIPCHK(h, hipMalloc(&h->d_ext, n * EXT_PER_ENV * sizeof(ExtAct)));
  h->h_ext.resize(n * EXT_PER_ENV);
  memset(h->h_ext.data(), 0xFF, h->h_ext.size() * sizeof(ExtAct));      // type -1 everywhere: nothing submitted
  for (size_t e = 0; e < n; ++e) {
    if (red) memcpy(&h->h_ext[e * EXT_PER_ENV], red + e * NRED, NRED * sizeof(ExtAct));
    if (green) memcpy(&h->h_ext[e * EXT_PER_ENV + NRED], green + e * MAXG, MAXG * sizeof(ExtAct));
  }
  for (size_t i = 0; i < h->h_ext.size(); ++i) {     // what the kernels index with must be in range; everything else is the engine's validity check
    ExtAct& a = h->h_ext[i];
    const bool is_red = (i % EXT_PER_ENV) < (size_t)NRED;
    if (a.type == XA_NONE) continue;
    if (a.type < 0 || (is_red ? a.type > RA_INVALID : a.type > XG_INVALID) || a.host >= MAXH || (is_red && a.type == RA_DRS && a.arg >= NSUB) ||
        (is_red && a.type == RA_WITHDRAW && a.arg >= MAXH) || (!is_red && a.ticks > 1)) {
      h->err = "cc4_step_ex: action record " + std::to_string(i % EXT_PER_ENV) + " of episode " + std::to_string(i / EXT_PER_ENV) + " is out of range (type / host / subnet; a green action takes one tick)";
      return -2;
    }
  }
  HIPCHK(h, hipMemcpyAsync(h->d_ext, h->h_ext.data(), h->h_ext.size() * sizeof(ExtAct), hipMemcpyHostToDevice, h->stream));
  h->ext_seen = true; h->ext_dirty = true;
  if (actions) HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, n * NBLUE * sizeof(int32_t), hipMemcpyDefault, h->stream));
  if (messages) HIPCHK(h, hipMemcpyAsync(h->d_msgs, messages, n * NBLUE * MSG_LEN, hipMemcpyDefault, h->stream));
  if (launch_step(h, actions ? h->d_actions : nullptr, messages ? h->d_msgs : nullptr, false, 0, 0, true, true)) return -1;
  return sync_all(h);
}

// direct edits of one episode between steps (state_edit, csrc/cc4_engine.h): the row and its cold part come to the host, the
// engine's own host build edits them, they go back
int cc4_edit_state(cc4_handle* h, int32_t env, int32_t op, int32_t a0, int32_t a1, int32_t a2) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_edit_state: env out of range"; return -2; }
  EnvState* st = (EnvState*)malloc(sizeof(EnvState));
  EnvCold* cold = (EnvCold*)malloc(h->cold_row);
  int rc = cc4_get_state(h, env, st);
  if (rc == 0) rc = cc4_get_cold(h, env, cold);
  if (rc == 0) {
    StepWork w; memset(&w, 0, sizeof(w));
    Ctx x{st, cold, &st->rng, st->hd, &w};
    rc = state_edit(x, op, a0, a1, a2);
    if (rc < 0) { h->err = "cc4_edit_state: unknown op or bad argument"; rc = -2; }
    else { int r2 = cc4_set_state(h, env, st); if (r2 == 0) r2 = cc4_set_cold(h, env, cold); if (r2) rc = r2; }
  }
  free(st); free(cold);
  return rc;
}

int cc4_step_device(cc4_handle* h, const int32_t* d_actions, const uint8_t* d_messages) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  return launch_step(h, d_actions, d_messages, false, 0, 0, false, true);
}

// ---- group-wise stepping for a policy that lives on the GPU.  A step of a large batch is one launch per episode group, each group on its
// own stream, and the groups never wait for each other -- unless the caller's policy makes them: a policy kernel over the WHOLE batch
// between two steps is a barrier across the groups (cc4_step_device behind it: bench.py `policy_in_loop`).  A policy is batch-independent,
// though: applied per group, on the group's own stream, it keeps the groups' pipelines apart.  cc4_group_info says which episodes a group
// holds and which stream its launches run on (the caller enqueues its policy kernel for those episodes there); cc4_step_group_device
// launches that group's step behind it.  Without a communicator, event log or submitted red / green actions.
int cc4_group_info(cc4_handle* h, int32_t g, int32_t* lo, int32_t* hi, void** hip_stream) {
  if (g < 0 || g >= h->ngroups) { h->err = "cc4_group_info: no such group"; return -2; }
  if (lo) *lo = h->glo[g];
  if (hi) *hi = h->glo[g + 1];
  if (hip_stream) *hip_stream = reinterpret_cast<void*>(h->gstream[g]);
  return 0;
}
static int group_prologue(cc4_handle* h, int32_t g, const char* who) {
  if (g < 0 || g >= h->ngroups) { h->err = std::string(who) + ": no such group"; return -2; }
  if (h->comm || h->evlog_on || h->ext_seen) { h->err = std::string(who) + ": group-wise stepping serves handles without a communicator, event log or submitted red / green actions"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (h->ngroups > 1 && h->main_ahead) {     // e.g. a reset or an upload on the main stream: the group streams start behind it
    HIPCHK(h, hipEventRecord(h->mev, h->stream));
    for (int q = 1; q < h->ngroups; ++q) HIPCHK(h, hipStreamWaitEvent(h->gstream[q], h->mev, 0));
    h->main_ahead = false;
  }
  return 0;
}
int cc4_step_group_device(cc4_handle* h, int32_t g, const int32_t* d_actions, const uint8_t* d_messages) {
  if (int rc = group_prologue(h, g, "cc4_step_group_device")) return rc;
  if (h->full_obs_next) { h->full_obs_gmask = (1u << h->ngroups) - 1u; h->full_obs_next = false; }
  const bool full_obs = (h->full_obs_gmask >> g) & 1u;
  h->full_obs_gmask &= ~(1u << g);
  StepArgs a{h->d_state, h->d_cold, d_actions, d_messages, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, nullptr, 0, 0,
             h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), full_obs ? 1 : 0,
             (uint32_t)h->cfg.topology_seed, h->d_prof, h->d_reset_ws, nullptr, 0};
  launch_group(h, a, g, false, nullptr, nullptr);
  HIPCHK(h, hipGetLastError());
  if (h->ngroups > 1) h->groups_busy = true;
  return 0;
}
// the stand-in policy of bench.py for one group: uniform random action indices for the group's episodes into the handle's device action
// buffer, on the group's stream (what a policy network's kernel would do there)
int cc4_random_actions_group_device(cc4_handle* h, int32_t g, uint64_t seed0, uint32_t t) {
  if (int rc = group_prologue(h, g, "cc4_random_actions_group_device")) return rc;
  const int tot = (h->glo[g + 1] - h->glo[g]) * NBLUE;
  hipLaunchKernelGGL(k_random_actions, dim3((tot + 255) / 256), dim3(256), 0, h->gstream[g], h->d_actions, h->glo[g + 1], seed0, t, h->glo[g]);
  HIPCHK(h, hipGetLastError());
  if (h->ngroups > 1) h->groups_busy = true;
  return 0;
}

int cc4_get_obs(cc4_handle* h, int32_t* obs) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(obs, h->d_obs, (size_t)h->cfg.num_envs * OBS_TOTAL * sizeof(int32_t), hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_get_reward_done(cc4_handle* h, float* reward, uint8_t* done) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  size_t n = (size_t)h->cfg.num_envs;
  if (reward) HIPCHK(h, hipMemcpyAsync(reward, h->d_reward, n * sizeof(float), hipMemcpyDefault, h->stream));
  if (done) HIPCHK(h, hipMemcpyAsync(done, h->d_done, n, hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_get_action_mask(cc4_handle* h, uint8_t* mask) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(mask, h->d_mask, (size_t)h->cfg.num_envs * MASK_TOTAL, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_get_err(cc4_handle* h, uint32_t* err) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(err, h->d_err, (size_t)h->cfg.num_envs * sizeof(uint32_t), hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_get_rng_state(cc4_handle* h, uint64_t* out) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  int n = h->cfg.num_envs;
  hipLaunchKernelGGL(k_rng_state, dim3((n + 127) / 128), dim3(128), 0, h->stream, h->d_state, h->d_rng, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out, h->d_rng, (size_t)n * 7 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_set_seed(cc4_handle* h, const uint64_t* seeds) {
  h->prev_valid = false;        // (cc4_replay_logged would repeat a step from rows that have moved on)
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  int n = h->cfg.num_envs;
  HIPCHK(h, hipMemcpyAsync(h->d_seeds, seeds, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_set_seed, dim3((n + 127) / 128), dim3(128), 0, h->stream, h->d_state, h->d_cold, h->cold_row, h->d_seeds, n, h->cfg.rng_mode);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_set_rng_state(cc4_handle* h, const uint64_t* words) {
  h->prev_valid = false;
  if (h->cfg.rng_mode != 0) { h->err = "cc4_set_rng_state: a numpy PCG64 state needs rng_mode 0"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  int n = h->cfg.num_envs;
  HIPCHK(h, hipMemcpyAsync(h->d_rng, words, (size_t)n * 6 * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));   // d_rng holds 7 words per episode
  hipLaunchKernelGGL(k_set_rng_state, dim3((n + 127) / 128), dim3(128), 0, h->stream, h->d_state, h->d_rng, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_obs_device(cc4_handle* h, int32_t** p) { *p = h->d_obs; return 0; }
int cc4_reward_device(cc4_handle* h, float** p) { *p = h->d_reward; return 0; }
int cc4_done_device(cc4_handle* h, uint8_t** p) { *p = h->d_done; return 0; }
int cc4_actions_device(cc4_handle* h, int32_t** p) { *p = h->d_actions; return 0; }
// host copy of the handle's device action buffer: the indices cc4_step uploaded, or the ones the last step of
// cc4_run_random_steps / cc4_random_actions_device drew on the device
int cc4_get_actions(cc4_handle* h, int32_t* out) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(out, h->d_actions, (size_t)h->cfg.num_envs * NBLUE * sizeof(int32_t), hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int cc4_random_actions_device(cc4_handle* h, uint64_t seed0, uint32_t t) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  int tot = h->cfg.num_envs * NBLUE;
  hipLaunchKernelGGL(k_random_actions, dim3((tot + 255) / 256), dim3(256), 0, h->stream, h->d_actions, h->cfg.num_envs, seed0, t, 0);
  HIPCHK(h, hipGetLastError());
  return 0;
}
int cc4_synchronize(cc4_handle* h) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  return sync_all(h);
}
// The persistent run kernel (cc4_run_random_steps without a communicator, batches beyond what one launch holds): one wave per
// residency slot, the batch cut into one partition per CU.  Which CUs the device has is found once per handle (k_discover: many small
// waves reporting HW_REG_XCC_ID / HW_REG_HW_ID; the path stays off unless exactly as many CUs show up as the device properties
// promise -- a mis-decoded id would merge CUs and show here); how many waves of the run kernel a CU takes is the dispatcher's business.
// History: r04 built it with the step body as a call and measured it 18-38 % slower than four streams of per-step launches; the call
// was the brake (a kernel that contains one loses a quarter of its rate).  Inlined (lane id opaque per item) and compiled without
// machine LICM (which hoisted ~200 registers' worth of loop-invariant values across the item loop and spilled them) it is the faster
// schedule from ~20 steps per call on: 8192 episodes 795 -> 917 M at K = 500 (profiles/r04_persistent_kernel_ab.txt).
static int persist_setup(cc4_handle* h) {
  h->persist_state = -1;
  const size_t n = (size_t)h->cfg.num_envs;
  int per_cu = 0;
  HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persist_kernel(h), WAVE, offsetof(EnvState, hd)));
  // (the occupancy query divides 160 KB by the kernel's LDS bytes; the hardware allocates 1280-byte granules -- profiles/r05_lds_residency.txt)
  hipFuncAttributes fa{};
  HIPCHK(h, hipFuncGetAttributes(&fa, persist_kernel(h)));
  const int granules = (int)((offsetof(EnvState, hd) + fa.sharedSizeBytes + 1279) / 1280);
  if (granules > 0 && 128 / granules < per_cu) per_cu = 128 / granules;
  per_cu -= h->run_margin;            // (with ranks to talk to: a slot per CU stays free for RCCL's kernels)
  if (per_cu <= 0) return 0;
  // The hand-over between two items of an episode relies on what gfx942 / gfx950 do in their default (non-tgsplit) mode: the waves of a
  // CU share one write-through vector L1 (validated on MI355X in SPX mode, the only partition mode of this pool).  Any other
  // architecture keeps the per-step launches -- and says so.
  hipDeviceProp_t prop;
  HIPCHK(h, hipGetDeviceProperties(&prop, h->cfg.device_id));
  if (!(strncmp(prop.gcnArchName, "gfx942", 6) == 0 || strncmp(prop.gcnArchName, "gfx950", 6) == 0)) {
    fprintf(stderr, "[cc4] the persistent run kernel stays OFF for this handle (per-step launches instead): architecture %s is neither gfx942 nor gfx950\n", prop.gcnArchName);
    h->persist_refused = true;
    return 0;
  }
  if (join_groups(h)) return -1;
  int32_t* d_count = nullptr;
  HIPCHK(h, hipMalloc(&d_count, CC4_SLOTS * sizeof(int32_t)));
  HIPCHK(h, hipMemsetAsync(d_count, 0, CC4_SLOTS * sizeof(int32_t), h->stream));
  int khz = 100000;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id);
  hipLaunchKernelGGL(k_discover, dim3(24 * h->cus), dim3(WAVE), 0, h->stream, d_count, 100LL * (khz > 0 ? khz : 100000) / 1000);   // ~100 us each
  std::vector<int32_t> count(CC4_SLOTS);
  HIPCHK(h, hipMemcpyAsync(count.data(), d_count, CC4_SLOTS * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  (void)hipFree(d_count);
  std::vector<int32_t> table(CC4_SLOTS, 0);
  int P = 0;
  for (int sl = 0; sl < CC4_SLOTS; ++sl) if (count[sl] > 0) table[sl] = ++P;        // 1 + partition, in slot order: an XCD's CUs own neighbouring partitions
  if (getenv("CC4_PERSIST_DEBUG")) fprintf(stderr, "[cc4] persistent kernel: %d compute units seen (device: %d), %d LDS granules per wave, %d waves per CU\n", P, h->cus, granules, per_cu);
  {
    // the partition mode the hand-over was validated in: SPX -- one device, all eight XCDs, every CU of each (MI355X: 8 x 32).  Another
    // picture (CPX / DPX / QPX partitions, a part with CUs fused off differently per XCD) may well work -- an XCD's L2 is still the
    // coherence point of its CUs -- but nobody has run the self-check there: CC4_PERSIST_ANY_PARTITION=1 takes the responsibility.
    int nx = 0, per_x[8] = {0};
    for (int sl = 0; sl < 8 << 8; ++sl) if (count[sl] > 0) ++per_x[sl >> 8];
    bool even = true;
    for (int xc = 0; xc < 8; ++xc) { if (per_x[xc]) ++nx; if (per_x[xc] && per_x[xc] != per_x[0]) even = false; }
    const bool spx = nx == 8 && even && per_x[0] > 0;
    if (!spx && !(getenv("CC4_PERSIST_ANY_PARTITION") && atoi(getenv("CC4_PERSIST_ANY_PARTITION")) != 0)) {
      fprintf(stderr, "[cc4] the persistent run kernel stays OFF for this handle (per-step launches instead): the device shows %d XCD(s) with %d..CUs each -- not the SPX "
                      "picture (8 XCDs, equal CU counts) the hand-over between waves was validated in; CC4_PERSIST_ANY_PARTITION=1 overrides\n", nx, per_x[0]);
      h->persist_refused = true;
      return 0;
    }
  }
  if (P != h->cus) {       // a CU id that does not tell CUs apart would put two CUs on one partition: never run on a guess
    fprintf(stderr, "[cc4] the persistent run kernel stays OFF for this handle (per-step launches instead): its discovery pass saw %d compute units, the device has %d\n", P, h->cus);
    h->persist_refused = true;
    return 0;
  }
  h->run_P = P; h->run_G = P; h->run_grid = per_cu * h->cus;
  if (const char* v = getenv("CC4_PERSIST_SCHED")) h->run_pool = atoi(v);
  if (const char* v = getenv("CC4_PERSIST_THR")) h->run_thr = atoi(v);
  if (h->run_pool < 0 || h->run_pool > 2) h->run_pool = 2;
  if (const char* v = getenv("CC4_PERSIST_RUNS")) {
    int q[4] = {h->run_SA, h->run_SB, h->run_nB, h->run_single};
    (void)sscanf(v, "%d,%d,%d,%d", &q[0], &q[1], &q[2], &q[3]);
    h->run_SA = q[0] < 0 ? 0 : q[0]; h->run_SB = q[1] < 1 ? 1 : q[1]; h->run_nB = q[2] < 0 ? 0 : q[2]; h->run_single = q[3] < 0 ? 0 : q[3];
  }
  if (h->run_pool == 2) {
    // the partitions of an XCD: a contiguous range (they are numbered in slot order, slot id = XCC id << 8 | CU)
    bool ok = true;
    for (int xc = 0; xc < 8; ++xc) {
      int lo = -1, cnt = 0;
      for (int sl = xc << 8; sl < (xc + 1) << 8; ++sl) if (table[sl] > 0) { if (lo < 0) lo = table[sl] - 1; ++cnt; }
      if (cnt > WAVE || lo > 255) ok = false;          // (one lane per partition of the XCD; the range's start travels as a byte)
      h->xcc_lo[xc] = (uint8_t)(lo < 0 ? 0 : lo); h->xcc_n[xc] = (uint8_t)(cnt > WAVE ? 0 : cnt);
    }
    for (int sl = 8 << 8; sl < CC4_SLOTS; ++sl) if (count[sl] > 0) ok = false;          // an XCC id beyond 7: not a device this schedule knows
    if (P > 510) ok = false;                            // the runner's id in the progress words: 9 bits
    if (!ok) h->run_pool = 0;
    else {
      if (!h->d_pool) HIPCHK(h, hipMalloc(&h->d_pool, 2 * (size_t)CC4_SLOTS * TK_STRIDE * sizeof(uint32_t)));
      HIPCHK(h, hipMemset(h->d_pool, 0, 2 * (size_t)CC4_SLOTS * TK_STRIDE * sizeof(uint32_t)));
      h->pool_base = 0; h->pool_parity = 0;
    }
  }
  if (h->run_pool == 1) {
    // one pool per XCD the discovery pass saw waves on (slot id = XCC id << 8 | CU)
    int nx = 0;
    for (int xc = 0; xc < 8; ++xc) {
      bool seen = false;
      for (int sl = xc << 8; sl < (xc + 1) << 8; ++sl) seen = seen || count[sl] > 0;
      h->xcc_pool[xc] = seen ? (uint8_t)nx++ : (uint8_t)0xFF;
    }
    for (int sl = 8 << 8; sl < CC4_SLOTS; ++sl) if (count[sl] > 0) nx = 0;          // an XCC id beyond 7: not a device this schedule knows
    if (nx <= 0) h->run_pool = 0;
    else {
      h->run_P = nx;
      if (!h->d_pool) HIPCHK(h, hipMalloc(&h->d_pool, 2 * (size_t)CC4_SLOTS * TK_STRIDE * sizeof(uint32_t)));
      HIPCHK(h, hipMemset(h->d_pool, 0, 2 * (size_t)CC4_SLOTS * TK_STRIDE * sizeof(uint32_t)));
      h->pool_base = 0; h->pool_parity = 0;
      if (getenv("CC4_PERSIST_DEBUG")) fprintf(stderr, "[cc4] persistent kernel: %d XCD pools\n", nx);
    }
  }
  if (!h->d_slot_part) HIPCHK(h, hipMalloc(&h->d_slot_part, CC4_SLOTS * sizeof(int32_t)));
  HIPCHK(h, hipMemcpy(h->d_slot_part, table.data(), CC4_SLOTS * sizeof(int32_t), hipMemcpyHostToDevice));
  if (h->d_run) { (void)hipFree(h->d_run); h->d_run = nullptr; }
  h->run_words = 2 * (size_t)h->run_G + n;                                            // [P ticket | P owner | n progress]: one memset per call
  HIPCHK(h, hipMalloc(&h->d_run, h->run_words * sizeof(uint32_t)));
  HIPCHK(h, hipMemset(h->d_run, 0, h->run_words * sizeof(uint32_t)));
  h->persist_state = 1;
  return 0;
}
// ---- the exchange around a one-launch kernel (XchgArgs).  Before the launch: the call's flags cleared on the main stream, the
// communication stream ordered behind that.  After the launch: per chunk of steps, on the communication stream, wait for the chunk's last
// step to be complete (done[k] == episodes: the kernel counts an episode once its packed row is in memory), all-gather the chunk's
// slabs, publish gathered = k + 1.  After the main stream's synchronisation: the communication stream drained, the watchdog flag read.
static int xchg_begin(cc4_handle* h, int k, XchgArgs* x) {
  (void)k;
  const size_t groups = (size_t)h->cfg.num_envs / 32 + 1 > (size_t)h->cus ? (size_t)h->cfg.num_envs / 32 + 1 : (size_t)h->cus;
  {
    const size_t nb = (size_t)h->cfg.num_envs * OBS_PACKED;
    if (!h->d_xslab) HIPCHK(h, hipMalloc(&h->d_xslab, nb * cc4_handle::XRING));
    if (!h->d_xall) HIPCHK(h, hipMalloc(&h->d_xall, nb * (size_t)h->world * cc4_handle::XRING));
  }
  if (!h->d_xflags) { HIPCHK(h, hipMalloc(&h->d_xflags, 2 * sizeof(uint32_t))); h->xflags_clean = 0; }
  if (!h->d_xgcnt) { HIPCHK(h, hipMalloc(&h->d_xgcnt, groups * cc4_handle::XRING * sizeof(uint32_t))); h->xflags_clean = 0; }
  if (!h->h_xtimeout) {
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_xtimeout), sizeof(uint32_t), hipHostMallocDefault));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_xtimeout), h->h_xtimeout, 0));
  }
  *h->h_xtimeout = 0;
  if (!h->xflags_clean) {       // normally cleared behind the previous call already (xchg_end): nothing of it in front of this call's launch
    HIPCHK(h, hipMemsetAsync(h->d_xflags, 0, 2 * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_xgcnt, 0, groups * cc4_handle::XRING * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipEventRecord(h->xev, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->xev, 0));
  }
  // (clean -- the usual case: the previous call's communication stream zeroed both words behind its last publish, xchg_enqueue, and the host
  // has waited for that stream since -- nothing of this call's is ordered behind anything: no memset, no event, no cross-stream wait)
  h->xflags_clean = 0;
  if (h->khz <= 0) { int khz = 100000; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id); h->khz = khz > 0 ? khz : 100000; }
  const int khz = h->khz;
  *x = XchgArgs{h->d_xslab, h->d_xflags, h->d_xflags + 1, cc4_handle::XRING, (long long)h->xchg_watchdog_ms * (khz > 0 ? khz : 100000), h->d_xgcnt, h->d_xtimeout};
  return 0;
}
// form: 3 = the persistent kernel (groups = its partitions), else groups of 32 neighbouring episodes
static int xchg_enqueue(cc4_handle* h, int k, const XchgArgs& x, int form) {
  const size_t row = (size_t)h->cfg.num_envs * OBS_PACKED;
  const int C = h->xchg_chunk, n = h->cfg.num_envs;
  const int P = form == 3 ? h->run_G : 0, groups = form == 3 ? h->run_G : (n + 31) / 32;
  const long long gate_ticks = 30000LL * (h->khz > 0 ? h->khz : 100000);         // 30 s: a step kernel that never gets there (the host would wait for it forever anyway)
  for (int c0 = 0, hi = 0; c0 < k; c0 = hi + 1) {
    hi = (c0 + C < k ? c0 + C : k) - 1;
    if (hi == k - 1 && hi > c0) --hi;       // the call's last step is a chunk of its own: behind the kernel's end only ONE all-gather is left
    if (c0 % cc4_handle::XRING + (hi - c0) >= cc4_handle::XRING) hi = c0 + cc4_handle::XRING - 1 - c0 % cc4_handle::XRING;     // a chunk's slabs are neighbours in the ring
    hipLaunchKernelGGL(k_xchg_gate, dim3(1), dim3(WAVE), 0, h->comm_stream, x.gcnt, x.ring, groups, n, P, c0, hi, gate_ticks, x.timeout_host, hi == k - 1 ? 1 : 8);
    HIPCHK(h, hipGetLastError());
    if (h->comm_delay_ticks > 0) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, h->comm_stream, h->comm_delay_ticks); HIPCHK(h, hipGetLastError()); }
    // ONE all-gather for the chunk's m neighbouring slabs (an ncclAllGather costs the host ~10 us to enqueue, grouped or not: eight of them
    // per chunk were as much as the eight steps of a 1024-episode batch last).  The gathered block of a chunk is rank-major: rank r's rows of
    // the chunk's step j at ((r * m + j - c0) * N) -- for a chunk of one step, the call's last among them, plain [world * N] rows.
    const int s0 = c0 % cc4_handle::XRING, m = hi - c0 + 1;
    uint8_t* const block = h->d_xall + (size_t)s0 * row * (size_t)h->world;
    ncclResult_t r = ncclAllGather(h->d_xslab + (size_t)s0 * row, block, (size_t)m * row, ncclUint8, h->comm, h->comm_stream);
    if (r != ncclSuccess) { h->err = std::string("ncclAllGather: ") + ncclGetErrorString(r); return -1; }
    if (h->d_xlog) for (int j = c0; j <= hi && h->xlog_n < h->xlog_cap; ++j, ++h->xlog_n)     // debug: keep every step's gathered rows as [world * N] (cc4_debug_gather_log)
      for (int rk = 0; rk < h->world; ++rk)
        HIPCHK(h, hipMemcpyAsync(h->d_xlog + ((size_t)h->xlog_n * h->world + rk) * row, block + ((size_t)rk * m + (size_t)(j - c0)) * row, row, hipMemcpyDeviceToDevice, h->comm_stream));
    HIPCHK(h, hipStreamWriteValue32(h->comm_stream, x.gathered, (uint32_t)(hi + 1), 0));
  }
  // behind the call's last publish (every episode has counted its last step: nobody reads the two words any more) the communication stream
  // itself hands them back zeroed for the next call -- the host waits for this stream in xchg_end, so the next launch finds them clean
  HIPCHK(h, hipStreamWriteValue32(h->comm_stream, x.gathered, 0u, 0));
  HIPCHK(h, hipStreamWriteValue32(h->comm_stream, x.timeout, 0u, 0));
  h->gathers_issued += k;
  return 0;
}
static int xchg_end(cc4_handle* h, int k) {
  const size_t row = (size_t)h->cfg.num_envs * OBS_PACKED;
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  h->gathers_waited = h->gathers_issued;
  const uint32_t flag = *reinterpret_cast<volatile uint32_t*>(h->h_xtimeout);      // (both streams are drained: the kernel's system-scope store has landed)
  h->xchg_calls++;
  const int last = (k - 1) % cc4_handle::XRING;
  h->last_gathered = h->d_xall + last * row * (size_t)h->world;      // (the call's last step is a chunk of its own: plain [world * N] rows)
  h->gather_buf = -2;                               // (not one of the per-step ring's buffers: last_gathered says where)
  // the per-step path's current buffer is to hold the observations of the last step as well -- filled when an explicit cc4_allgather_obs asks
  // for it (a per-step launch or a reset that follows writes a buffer of its own)
  h->obs8_from_slab = last;
  h->step_event_attached = false;
  if (flag) {   // a watchdog fired: some counts may never have been collected -- everything cleared the long way before the next call
    h->xflags_clean = 0;
  } else h->xflags_clean = 1;   // (both words zeroed by the communication stream behind its last publish, the group counters by the gates)
  if (flag) {
    // an item waited longer than the watchdog for its slab: the exchange did not keep up at all (e.g. its kernels found no room beside the
    // one-launch kernel).  The episodes are intact -- a wait that gives up only stops protecting slabs, so gathers of this call may have
    // carried a later step's rows -- and the handle goes back to per-step launches, loudly.
    h->xchg_timeouts++;
    h->xchg_on = false;
    // what this call gathered is not published as valid: the gather log forgets the call's steps, and the observations of the call's last
    // step -- whose slab nothing overwrote -- are gathered again through the per-step path when somebody asks (obs8_from_slab stays)
    h->last_gathered = nullptr; h->gather_buf = -1;
    if (h->d_xlog) h->xlog_n = h->xlog_n >= k ? h->xlog_n - k : 0;
    h->err = "the in-kernel exchange timed out in the last cc4_run_random_steps call (episodes intact; its all-gathers are void; per-step launches from now on)";
    (void)hipFree(h->d_xall); h->d_xall = nullptr;      // (the gathered twin of the ring: world times the ring; the ring itself still holds the last step's rows)
    fprintf(stderr, "[cc4] the in-kernel exchange timed out (a step waited > %d ms for the all-gather of %d steps earlier): this handle returns to per-step launches with the exchange\n",
            h->xchg_watchdog_ms, cc4_handle::XRING);
  }
  return 0;
}
static int run_random_steps_impl(cc4_handle* h, uint64_t seed0, uint32_t t0, int32_t k, float* ms_step_kernels);
// CC4_PERSIST_VERIFY=1 (a self-check mode, not a fast one): a call that takes a one-launch form -- the persistent kernels, whose hand-over
// between the steps of an episode leans on how a CU's L1 behaves, and the plain multi-step kernels -- is run a second time
// from the same starting rows with per-step launches on a shadow handle, and the two outcomes are compared episode by episode.
static int verify_digest(cc4_handle* h, std::vector<uint64_t>& out) {
  const int n = h->cfg.num_envs;
  if (!h->d_digest) HIPCHK(h, hipMalloc(&h->d_digest, 3 * (size_t)n * sizeof(uint64_t)));
  if (join_groups(h)) return -1;
  hipLaunchKernelGGL(k_digest, dim3(n), dim3(WAVE), 0, h->stream, h->d_state, h->d_cold, h->cold_row, h->d_obs, h->d_reward, h->d_done, h->d_err, h->d_actions, h->d_digest, n);
  HIPCHK(h, hipGetLastError());
  out.resize(3 * (size_t)n);
  HIPCHK(h, hipMemcpyAsync(out.data(), h->d_digest, out.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_run_random_steps(cc4_handle* h, uint64_t seed0, uint32_t t0, int32_t k, float* ms_step_kernels) {
  if (h->is_shadow || k < 2) return run_random_steps_impl(h, seed0, t0, k, ms_step_kernels);
  // The self-check: with CC4_PERSIST_VERIFY=1 every one-launch call is repeated on a shadow handle and compared; WITHOUT it every
  // verify_every-th call that takes the PERSISTENT form is (CC4_PERSIST_VERIFY_EVERY, default 1024, 0 = never; the communicator-less handles
  // only: a shadow handle cannot join the exchange) -- the hand-over between the waves of a CU rests on behaviour the memory model does not
  // promise, so the path keeps checking itself in production at < 1 % of its time (a checked call costs ~10 x a plain one; the shadow
  // handle -- a second copy of the batch's rows -- is allocated by the first checked call).
  bool check = h->verify;
  if (!check && h->verify_every > 0 && !h->comm && h->persist_state >= 0 && k >= h->persist_min_k && !h->run1m && !h->multistep) {
    if (++h->persist_calls % (uint64_t)h->verify_every == 0) check = true;
  }
  if (!check) return run_random_steps_impl(h, seed0, t0, k, ms_step_kernels);
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (strncmp(cc4_run_kernel_for(h, k), "k_run_", 6) != 0) return run_random_steps_impl(h, seed0, t0, k, ms_step_kernels);
  if (!h->shadow) {
    cc4_handle* sh = nullptr;
    if (cc4_create(&h->cfg, &sh) != 0) { h->err = std::string("CC4_PERSIST_VERIFY: the shadow handle could not be created: ") + cc4_last_error(sh); if (sh) cc4_destroy(sh); return -1; }
    sh->is_shadow = true; sh->verify = false; sh->persist_state = -1; sh->multistep = false; sh->run1m = false;
    h->shadow = sh;
  }
  cc4_handle* sh = h->shadow;
  const size_t n = (size_t)h->cfg.num_envs;
  if (join_groups(h) || join_groups(sh)) return -1;
  HIPCHK(h, hipStreamSynchronize(sh->stream));
  HIPCHK(h, hipMemcpyAsync(sh->d_state, h->d_state, n * sizeof(EnvState), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(sh->d_cold, h->d_cold, n * h->cold_row, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(sh->d_obs, h->d_obs, h->out_bytes, hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  sh->full_obs_next = h->full_obs_next; sh->main_ahead = sh->ngroups > 1;
  int rc = run_random_steps_impl(h, seed0, t0, k, ms_step_kernels);
  if (rc) return rc;
  rc = run_random_steps_impl(sh, seed0, t0, k, nullptr);
  if (rc) { h->err = "CC4_PERSIST_VERIFY: the shadow run failed: " + sh->err; return rc; }
  std::vector<uint64_t> a, b;
  if (verify_digest(h, a)) return -1;
  if (verify_digest(sh, b)) { h->err = "CC4_PERSIST_VERIFY: " + sh->err; return -1; }
  h->verify_calls++;
  for (size_t e = 0; e < n; ++e) {
    const bool hot = a[3 * e] != b[3 * e], cold = a[3 * e + 1] != b[3 * e + 1], outp = a[3 * e + 2] != b[3 * e + 2];
    if (hot || cold || outp) {
      h->verify_mismatches++;
      h->err = "CC4_PERSIST_VERIFY: " + std::string(cc4_run_kernel_for(h, k)) + " and the per-step launches disagree after " + std::to_string(k) + " steps: first episode " +
               std::to_string(e) + " (" + (hot ? "hot row " : "") + (cold ? "cold row " : "") + (outp ? "outputs" : "") + ")";
      fprintf(stderr, "[cc4] %s\n", h->err.c_str());
      return -5;
    }
  }
  return 0;
}
// out[0] calls checked, out[1] calls that disagreed (CC4_PERSIST_VERIFY)
int cc4_verify_stats(cc4_handle* h, int64_t* out /* [2] */) { out[0] = h->verify_calls; out[1] = h->verify_mismatches; return 0; }
// One launch of the persistent kernel for k steps of the whole batch (cc4_run_random_steps form 3; cc4_rollout_begin with rollout = true: every step
// an item of its own, the actions from the rollout's slots behind the caller's publishes).
static int persist_launch(cc4_handle* h, StepArgs a, int k, uint32_t t0, const XchgArgs& x, hipEvent_t e0, hipEvent_t e1, bool rollout) {
  if (!h->run_pool) HIPCHK(h, hipMemsetAsync(h->d_run, 0, h->run_words * sizeof(uint32_t), h->stream));
  else if (h->pool_base + (uint32_t)k > (h->run_pool == 2 ? 0x700000u : 0x7F000000u)) {      // (the progress words count steps since they were last cleared)
    HIPCHK(h, hipMemsetAsync(h->d_run, 0, h->run_words * sizeof(uint32_t), h->stream));
    h->pool_base = 0;
  }
  unsigned long long* d_tl = nullptr;
  if (getenv("CC4_PERSIST_TIMELINE")) { HIPCHK(h, hipMalloc(&d_tl, 4 * sizeof(unsigned long long) * (size_t)h->run_grid)); HIPCHK(h, hipMemsetAsync(d_tl, 0, 4 * sizeof(unsigned long long) * (size_t)h->run_grid, h->stream)); }
  h->d_timeline = d_tl;
  RunArgs ra{h->d_run, h->d_run + 2 * h->run_G, reinterpret_cast<int32_t*>(h->d_run + h->run_G), h->d_slot_part, h->run_P, k, h->run_G, t0, d_tl, h->persist_order,
             1, k, 1, 0, k, 0, 0u, nullptr, {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF}, {0}, {0}, h->run_thr};
  const int SA = rollout ? 1 : (h->run_SA > 0 ? h->run_SA : (k >= 64 ? 8 : 4));
  if (SA > 1) {
    // runs of steps: nB runs of SB and tail_single single steps close the call, runs of SA fill the rest (what is left over goes to the single steps)
    int single = h->run_single < k ? h->run_single : k;
    int nB = h->run_SB > 1 ? h->run_nB : 0;
    while (nB > 0 && single + nB * h->run_SB > k) --nB;
    const int nA = (k - single - nB * h->run_SB) / SA;
    single = k - nA * SA - nB * h->run_SB;
    ra.SA = SA; ra.nA = nA; ra.SB = h->run_SB > 1 ? h->run_SB : 1; ra.nB = nB; ra.nph = nA + nB + single;
  }
  if (h->run_pool) {
    ra.pool = h->run_pool; ra.base = h->pool_base;
    ra.ticket = h->d_pool + (size_t)h->pool_parity * CC4_SLOTS * TK_STRIDE;
    ra.ticket_next = h->d_pool + (size_t)(h->pool_parity ^ 1) * CC4_SLOTS * TK_STRIDE;
    memcpy(ra.xcc_pool, h->xcc_pool, 8); memcpy(ra.xcc_lo, h->xcc_lo, 8); memcpy(ra.xcc_n, h->xcc_n, 8);
    h->pool_parity ^= 1; h->pool_base += (uint32_t)k;
  }
  if (rollout) {
    ra.act_ready = h->d_rready; ra.act = h->d_ract; ra.PG = RPG;
    ra.act_wait_ticks = (long long)h->rollout_watchdog_ms * (h->khz > 0 ? h->khz : 100000);
  }
#ifndef CC4_DEV_FAST
  if (h->cfg.rng_mode == 0) hipExtLaunchKernelGGL(k_run_pcg, dim3(h->run_grid), dim3(WAVE), offsetof(EnvState, hd), h->stream, e0, e1, 0, a, ra, x);
  else
#endif
  hipExtLaunchKernelGGL(k_run_philox1, dim3(h->run_grid), dim3(WAVE), offsetof(EnvState, hd), h->stream, e0, e1, 0, a, ra, x);
  return 0;
}
static int run_random_steps_impl(cc4_handle* h, uint64_t seed0, uint32_t t0, int32_t k, float* ms_step_kernels) {
  h->prev_valid = false;        // (every form of this call moves the rows without refreshing the kept copy of cc4_keep_previous)
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (k <= 0) { if (ms_step_kernels) *ms_step_kernels = 0.f; return 0; }   // nothing to launch, no timing event to read
  const bool plain = (!h->comm || h->xchg_on) && !h->evlog_on && !h->ext_seen && !h->d_prof;
  if (plain && h->persist_state == 0 && !h->run1m && !h->multistep && k >= h->persist_min_k) { if (persist_setup(h)) return -1; }
  const int form = !plain ? 0 : (h->multistep && k >= 2) ? 1 : (h->run1m && k >= 2) ? 2 : (h->persist_state == 1 && h->run_P > 0 && k >= h->persist_min_k) ? 3 : 0;
  if (form) {
    // ONE launch for the k steps: 1 = every block loops over the steps of its episode (k_run_philox / k_run_philox8), 2 = the same on one wave
    // per episode (k_run_philox1m), 3 = the persistent form (k_run_philox1 / k_run_pcg: one wave per residency slot pulling (episode, step) items)
    if (join_groups(h)) return -1;
    StepArgs a{h->d_state, h->d_cold, nullptr, nullptr, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, h->d_actions, seed0, t0,
               h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
               (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0),
               h->full_obs_next ? 1 : 0, (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, nullptr, 0};
    XchgArgs x{};
    const bool exchange = h->comm != nullptr;
    static const bool xprof = getenv("CC4_EXCHANGE_PROF") != nullptr;      // debug: where the host's time goes around a one-launch call with the exchange
    static double xp[6] = {0}; static long xpn = 0;
    const auto xp0 = std::chrono::steady_clock::now();
    if (exchange && xchg_begin(h, k, &x)) return -1;
    const auto xp1 = std::chrono::steady_clock::now();
    if (ms_step_kernels && h->evs.size() < 2) { h->evs.resize(2, nullptr); for (auto& e : h->evs) if (!e) HIPCHK(h, hipEventCreate(&e)); }
    hipEvent_t e0 = ms_step_kernels ? h->evs[0] : nullptr, e1 = ms_step_kernels ? h->evs[1] : nullptr;
    auto c0 = std::chrono::steady_clock::now();
    if (form == 1) {
      if (h->multistep_minb == 8) hipExtLaunchKernelGGL(k_run_philox8, dim3(h->cfg.num_envs), dim3(PT), sizeof(EnvState), h->stream, e0, e1, 0, a, (int)k, t0, x);
      else hipExtLaunchKernelGGL(k_run_philox, dim3(h->cfg.num_envs), dim3(PT), sizeof(EnvState), h->stream, e0, e1, 0, a, (int)k, t0, x);
    } else if (form == 2) {
      hipExtLaunchKernelGGL(k_run_philox1m, dim3(h->cfg.num_envs), dim3(WAVE), offsetof(EnvState, hd), h->stream, e0, e1, 0, a, (int)k, t0, x);
    } else {
      if (persist_launch(h, a, k, t0, x, e0, e1, false)) return -1;
    }
    HIPCHK(h, hipGetLastError());
    h->stat_launch_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
    h->stat_steps += k;
    h->full_obs_next = false;     // (asked for, the first step of every episode rewrote all its observation values)
    h->main_ahead = h->ngroups > 1;
    const auto xp2 = std::chrono::steady_clock::now();
    if (exchange) {
      auto g0 = std::chrono::steady_clock::now();
      if (xchg_enqueue(h, k, x, form)) return -1;
      h->stat_gather_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g0).count();
    }
    const auto xp3 = std::chrono::steady_clock::now();
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const auto xp4 = std::chrono::steady_clock::now();
    if (h->d_timeline) {      // debug: where a call's time goes between the kernel's entry and its last item (ticks of the 100 MHz wall clock)
      std::vector<unsigned long long> tl(4 * (size_t)h->run_grid);
      HIPCHK(h, hipMemcpy(tl.data(), h->d_timeline, tl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      (void)hipFree(h->d_timeline); h->d_timeline = nullptr;
      unsigned long long e0 = ~0ull, e1 = 0, f1 = 0, l0 = ~0ull, l1 = 0; double fs = 0, ls = 0, items = 0; int nw = 0, idle = 0;
      for (int w = 0; w < h->run_grid; ++w) { const unsigned long long* t = &tl[4 * (size_t)w]; if (!t[0]) continue; ++nw; e0 = t[0] < e0 ? t[0] : e0; e1 = t[0] > e1 ? t[0] : e1; if (!(uint32_t)t[3]) { ++idle; continue; } f1 = t[1] > f1 ? t[1] : f1; l0 = t[2] < l0 ? t[2] : l0; l1 = t[2] > l1 ? t[2] : l1; fs += (double)t[1]; ls += (double)t[2]; items += (double)(uint32_t)t[3]; }
      const int busy = nw - idle;
      float ms = 0.f; if (ms_step_kernels) (void)hipEventElapsedTime(&ms, h->evs[0], h->evs[1]);
      fprintf(stderr, "[cc4 timeline] k=%d: %d waves reported (%d without an item); entry spread %.1f us; first item starts: mean +%.1f us, last +%.1f us after the first entry; "
                      "last item ends: earliest +%.1f us, mean +%.1f us, latest +%.1f us; items per busy wave %.1f; kernel (events) %.1f us\n",
              k, nw, idle, (e1 - e0) / 100.0, busy ? (fs / busy - (double)e0) / 100.0 : 0.0, (f1 - e0) / 100.0, (l0 - e0) / 100.0, busy ? (ls / busy - (double)e0) / 100.0 : 0.0, (l1 - e0) / 100.0, busy ? items / busy : 0.0, ms * 1000.0);
      // per CU: when its LAST wave ran dry, and how many items its waves executed (more than its own partition's = it helped out)
      { std::map<int, std::pair<unsigned long long, double>> cu;
        for (int w = 0; w < h->run_grid; ++w) { const unsigned long long* t = &tl[4 * (size_t)w]; if (!t[0] || !(uint32_t)t[3]) continue; auto& c = cu[(int)((t[3] >> 32) - 1)]; if (t[2] > c.first) c.first = t[2]; c.second += (double)(uint32_t)t[3]; }
        std::vector<double> last, its; for (auto& kv : cu) { last.push_back((kv.second.first - e0) / 100.0); its.push_back(kv.second.second); }
        std::sort(last.begin(), last.end()); std::sort(its.begin(), its.end());
        if (!last.empty()) { const size_t m = last.size(); fprintf(stderr, "[cc4 timeline]   per CU (%zu): last wave dry at min %.1f / 10%% %.1f / median %.1f / 90%% %.1f / max %.1f us; items executed min %.0f / median %.0f / max %.0f\n", m,
                                   last[0], last[m / 10], last[m / 2], last[m * 9 / 10], last[m - 1], its[0], its[m / 2], its[m - 1]); } }
      // per XCD: when its waves ran dry (intra-XCD sharing evens a tail out inside an XCD; what is left between XCDs is not shareable)
      { double xs[8] = {0}, xi[8] = {0}; unsigned long long xl[8] = {0}, xf[8]; int xn[8] = {0}; for (int i = 0; i < 8; ++i) xf[i] = ~0ull;
        for (int w = 0; w < h->run_grid; ++w) { const unsigned long long* t = &tl[4 * (size_t)w]; if (!t[0] || !(uint32_t)t[3]) continue; const int xc = (int)(((t[3] >> 32) - 1) >> 8) & 7;
          xs[xc] += (double)t[2]; xi[xc] += (double)(uint32_t)t[3]; ++xn[xc]; if (t[2] > xl[xc]) xl[xc] = t[2]; if (t[2] < xf[xc]) xf[xc] = t[2]; }
        for (int i = 0; i < 8; ++i) if (xn[i]) fprintf(stderr, "[cc4 timeline]   XCD %d: %d busy waves, %.0f items; waves ran dry: earliest +%.1f, mean +%.1f, latest +%.1f us\n", i, xn[i], xi[i],
                                                      (xf[i] - e0) / 100.0, (xs[i] / xn[i] - (double)e0) / 100.0, (xl[i] - e0) / 100.0); }
    }
    if (exchange && xchg_end(h, k)) return -1;
    if (xprof && exchange) {
      const auto xp5 = std::chrono::steady_clock::now();
      auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      xp[0] += us(xp0, xp1); xp[1] += us(xp1, xp2); xp[2] += us(xp2, xp3); xp[3] += us(xp3, xp4); xp[4] += us(xp4, xp5); xp[5] += us(xp0, xp5);
      if (++xpn % 200 == 0) {
        fprintf(stderr, "[cc4 exchange prof] k=%d, mean of 200 calls (us): begin %.1f, launch %.1f, enqueue of the chunks %.1f, wait for the kernel %.1f, then for the communication stream %.1f; call %.1f\n",
                k, xp[0] / 200, xp[1] / 200, xp[2] / 200, xp[3] / 200, xp[4] / 200, xp[5] / 200);
        for (double& v : xp) v = 0;
      }
    }
    if (ms_step_kernels) HIPCHK(h, hipEventElapsedTime(ms_step_kernels, h->evs[0], h->evs[1]));
    return 0;
  }
  if (h->enq_threads && !h->pool && h->ngroups > 1 && !h->comm) enq_pool_start(h);     // on first use: most handles never come here
  if (h->pool && (int)h->pool->th.size() == h->ngroups - 1 && !h->comm && !h->evlog_on && !h->ext_seen && !h->d_prof && k >= 1) {
    // every group's k launches from its own thread (EnqPool); this thread takes group 0
    const int G = h->ngroups;
    EnqPool* P = h->pool;
    if (ms_step_kernels && (int)h->evs.size() < 2 * G) {
      size_t old = h->evs.size();
      h->evs.resize(2 * (size_t)G, nullptr);
      for (size_t i = old; i < h->evs.size(); ++i) HIPCHK(h, hipEventCreate(&h->evs[i]));
    }
    if (h->main_ahead) {
      HIPCHK(h, hipEventRecord(h->mev, h->stream));
      for (int g = 1; g < G; ++g) HIPCHK(h, hipStreamWaitEvent(h->gstream[g], h->mev, 0));
      h->main_ahead = false;
    }
    P->a = StepArgs{h->d_state, h->d_cold, nullptr, nullptr, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, h->d_actions, seed0, t0,
                    h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
                    (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0),
                    0, (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, nullptr, 0};
    P->k = k; P->t0 = t0; P->full = false; P->first_full_obs = h->full_obs_next; P->join = !getenv("CC4_ENQ_NOJOIN");
    for (int g = 0; g < G; ++g) { P->start[g] = ms_step_kernels ? h->evs[2 * g] : nullptr; P->stop[g] = ms_step_kernels ? h->evs[2 * g + 1] : nullptr; }
    P->failed.store(0);
    P->pending.store(G - 1, std::memory_order_relaxed);
    auto c0 = std::chrono::steady_clock::now();
    { std::lock_guard<std::mutex> lk(P->mu); P->gen.fetch_add(1, std::memory_order_release); }
    P->cv.notify_all();
    enq_run_group(h, P, 0);
    while (P->pending.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    h->stat_launch_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
    h->stat_steps += k;
    h->full_obs_next = false;
    h->groups_busy = true;
    if (P->failed.load()) { h->err = "cc4_run_random_steps: a step launch failed"; return -1; }
    if (P->join) {
      for (int g = 1; g < G; ++g) HIPCHK(h, hipStreamWaitEvent(h->stream, h->gev[g], 0));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      h->groups_busy = false;
    } else if (sync_all(h)) return -1;
    if (ms_step_kernels) {
      float worst = 0.f;
      for (int g = 0; g < G; ++g) { float ms = 0.f; HIPCHK(h, hipEventElapsedTime(&ms, h->evs[2 * g], h->evs[2 * g + 1])); if (ms > worst) worst = ms; }
      *ms_step_kernels = worst;
    }
    return 0;
  }
  // Timing: HIP events on the launch streams around chunks of TIMED_CHUNK consecutive steps (an event pair around every single
  // launch costs the stream ~5 us of idle time per step); per stream, the sum over the chunks is the on-stream time of its k
  // launches, read back after the loop -- no host synchronisation inside the timed region.  With several episode groups
  // (cc4_handle::ngroups) every group's stream is timed; the slowest stream is reported: the on-stream time of the k STEPS.
  constexpr int TIMED_CHUNK = 25;
  const int G = h->ngroups;
  const int nchunks = ms_step_kernels ? (k + TIMED_CHUNK - 1) / TIMED_CHUNK : 0;
  if ((int)h->evs.size() < 2 * nchunks * G) {
    size_t old = h->evs.size();
    h->evs.resize(2 * (size_t)nchunks * G, nullptr);
    for (size_t i = old; i < h->evs.size(); ++i) HIPCHK(h, hipEventCreate(&h->evs[i]));
  }
  auto ev = [&](int chunk, int g, int which) { return h->evs[(size_t)(2 * (chunk * G + g) + which)]; };
  const bool hp = getenv("CC4_HOST_PROF") != nullptr;
  double t_launch = 0, t_ag = 0, t_first = 0;
  const long long stalls0 = h->gather_stalls;
  // Without a communicator the two timing events of a stream ride on its first and its last launch of the call (start / stop
  // event of hipExtLaunchKernelGGL: the kernels' own start and completion timestamps) -- marker packets from hipEventRecord cost
  // the streams 0.5 us per step at k = 500 and 1.2 us per step at k = 20 (tools/short_region_probe.py).  CC4_TIMING_MARKERS=1
  // keeps the marker form; with a communicator the launches' stop events belong to the exchange and the markers stay.
  const bool attach = ms_step_kernels && !h->comm && !getenv("CC4_TIMING_MARKERS");
  for (int i = 0; i < k; ++i) {
    if (attach) {
      if (i == 0) for (int g = 0; g < G; ++g) h->tev_start[g] = ev(0, g, 0);
      if (i == k - 1) for (int g = 0; g < G; ++g) h->tev_stop[g] = ev(0, g, 1);
    }
    if (ms_step_kernels && !attach && i % TIMED_CHUNK == 0) {
      if (G > 1 && h->main_ahead) {     // the group streams' first event must not be recorded ahead of what their first launch waits for
        HIPCHK(h, hipEventRecord(h->mev, h->stream));
        for (int g = 1; g < G; ++g) HIPCHK(h, hipStreamWaitEvent(h->gstream[g], h->mev, 0));
        h->main_ahead = false;
      }
      for (int g = 0; g < G; ++g) HIPCHK(h, hipEventRecord(ev(i / TIMED_CHUNK, g, 0), h->gstream[g]));
    }
    auto c0 = std::chrono::steady_clock::now();
    if (launch_step(h, nullptr, nullptr, true, seed0, t0 + (uint32_t)i)) return -1;   // actions drawn in-kernel
    auto c1 = std::chrono::steady_clock::now();
    if (ms_step_kernels && !attach && (i % TIMED_CHUNK == TIMED_CHUNK - 1 || i == k - 1))
      for (int g = 0; g < G; ++g) HIPCHK(h, hipEventRecord(ev(i / TIMED_CHUNK, g, 1), h->gstream[g]));
    auto c2 = std::chrono::steady_clock::now();
    if (h->comm) { if (cc4_allgather_obs(h, nullptr)) return -1; }                       // overlaps the next step
    auto c3 = std::chrono::steady_clock::now();
    t_launch += std::chrono::duration<double, std::micro>(c1 - c0).count();
    if (i == 0) t_first = std::chrono::duration<double, std::micro>(c1 - c0).count();
    t_ag += std::chrono::duration<double, std::micro>(c3 - c2).count();
  }
  h->stat_steps += k; h->stat_launch_us += t_launch; h->stat_gather_us += t_ag;
  if (hp) fprintf(stderr, "[cc4 host prof] k=%d launch_step %.2f us/step (%d launches per step), allgather enqueue %.2f us/step, %lld buffer-reuse stalls; the call's first launch_step %.1f us\n", k, t_launch / k, G, t_ag / k, h->gather_stalls - stalls0, t_first);
  auto p0 = std::chrono::steady_clock::now();
  if (h->comm) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  if (sync_all(h)) return -1;
  auto p1 = std::chrono::steady_clock::now();
  if (hp) {
    auto q0 = std::chrono::steady_clock::now();
    if (sync_all(h)) return -1;
    auto q1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[cc4 host prof] enqueue loop done -> all streams synchronised: %.1f us; a second sync_all on idle streams: %.1f us\n",
            std::chrono::duration<double, std::micro>(p1 - p0).count(), std::chrono::duration<double, std::micro>(q1 - q0).count());
  }
  if (ms_step_kernels) {
    float worst = 0.f;
    for (int g = 0; g < G; ++g) {
      float total = 0.f;
      for (int c = 0; c < (attach ? 1 : nchunks); ++c) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, ev(c, g, 0), ev(c, g, 1)));
        total += ms;
      }
      if (total > worst) worst = total;
    }
    *ms_step_kernels = worst;
    if (hp) fprintf(stderr, "[cc4 host prof] reading the timing events: %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - p1).count());
  }
  return 0;
}
// ---- rollouts with the policy in the loop (include/cc4.h).  ONE launch of the persistent kernel per k-step rollout; the caller's policy
// runs between the steps on the caller's stream, one policy group of episodes at a time, ordered against the stepping through device words only.
static int rollout_ready(cc4_handle* h, const char* who) {
  if (h->rollout_k <= 0) { h->err = std::string(who) + ": no rollout is in flight (cc4_rollout_begin)"; return -2; }
  return 0;
}
int cc4_rollout_begin(cc4_handle* h, int32_t k) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (h->rollout_k > 0) { h->err = "cc4_rollout_begin: a rollout is in flight (cc4_rollout_end)"; return -2; }
  if (k <= 0 || k > 0x100000) { h->err = "cc4_rollout_begin: 1 .. 2^20 steps"; return -2; }
  if (h->cfg.rng_mode != 1 || h->comm || h->evlog_on || h->ext_seen || h->d_prof) { h->err = "cc4_rollout_begin: for counter-mode handles without a communicator, event log or submitted red / green actions"; return -2; }
  if (h->persist_state == 0) { if (persist_setup(h)) return -1; }
  if (h->persist_state != 1 || h->run_pool != 2) { h->err = "cc4_rollout_begin: this handle has no persistent kernel (a batch the chip holds at once, or a device picture the schedule refuses): step it with cc4_step_device"; return -2; }
  if (join_groups(h)) return -1;
  h->prev_valid = false;
  const size_t n = (size_t)h->cfg.num_envs, row = n * OBS_PACKED;
  if (!h->d_ract) {
    HIPCHK(h, hipMalloc(&h->d_ract, 2 * n * NBLUE * sizeof(int32_t)));
    HIPCHK(h, hipMalloc(&h->d_rready, RPG * 32 * sizeof(uint32_t)));
    HIPCHK(h, hipMalloc(&h->d_rcnt, (size_t)h->run_P * RPG * cc4_handle::XRING * sizeof(uint32_t)));
    HIPCHK(h, hipMalloc(&h->d_rfail, sizeof(uint32_t)));
    HIPCHK(h, hipStreamCreateWithFlags(&h->policy_stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->rev, hipEventDisableTiming));
    if (const char* v = getenv("CC4_ROLLOUT_WATCHDOG_MS")) h->rollout_watchdog_ms = atoi(v) > 0 ? atoi(v) : 2000;
  }
  if (!h->d_xslab) HIPCHK(h, hipMalloc(&h->d_xslab, row * cc4_handle::XRING));
  if (!h->d_xflags) { HIPCHK(h, hipMalloc(&h->d_xflags, 2 * sizeof(uint32_t))); }
  if (!h->h_xtimeout) {
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_xtimeout), sizeof(uint32_t), hipHostMallocDefault));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_xtimeout), h->h_xtimeout, 0));
  }
  if (h->khz <= 0) { int khz = 100000; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id); h->khz = khz > 0 ? khz : 100000; }
  *h->h_xtimeout = 0;
  HIPCHK(h, hipMemsetAsync(h->d_xflags, 0, 2 * sizeof(uint32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_rready, 0, RPG * 32 * sizeof(uint32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_rcnt, 0, (size_t)h->run_P * RPG * cc4_handle::XRING * sizeof(uint32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_rfail, 0, sizeof(uint32_t), h->stream));
  // what the first policy pass reads: the observations as they stand, packed into the slab in front of step 0's
  hipLaunchKernelGGL(k_pack_obs_rows, dim3((unsigned)n), dim3(WAVE), 0, h->stream, h->d_xslab + (size_t)(cc4_handle::XRING - 1) * row, h->d_obs, (int)n);
  HIPCHK(h, hipEventRecord(h->rev, h->stream));
  StepArgs a{h->d_state, h->d_cold, nullptr, nullptr, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, nullptr, 0, 0,
             h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0),
             h->full_obs_next ? 1 : 0, (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, nullptr, 0};
  XchgArgs x{h->d_xslab, nullptr, h->d_xflags + 1, cc4_handle::XRING, 0, h->d_rcnt, h->d_xtimeout};
  if (persist_launch(h, a, k, 0u, x, nullptr, nullptr, true)) return -1;
  HIPCHK(h, hipGetLastError());
  h->stat_steps += k;
  h->full_obs_next = false;
  h->main_ahead = h->ngroups > 1;
  h->rollout_k = k;
  return 0;
}
int cc4_rollout_groups(cc4_handle* h, int32_t* groups, int32_t* block) {
  if (h->persist_state == 0) { HIPCHK(h, hipSetDevice(h->cfg.device_id)); if (persist_setup(h)) return -1; }
  *groups = RPG; *block = h->run_P > 0 ? h->run_P : h->cus;
  return 0;
}
int cc4_rollout_obs_packed(cc4_handle* h, int32_t j, const uint8_t** d_rows) {
  if (!h->d_xslab || !h->d_ract) { h->err = "cc4_rollout_obs_packed: no rollout was begun on this handle"; return -2; }      // (also behind cc4_rollout_end: the ring keeps the last 32 steps)
  *d_rows = h->d_xslab + (size_t)((j + cc4_handle::XRING - 1) % cc4_handle::XRING) * (size_t)h->cfg.num_envs * OBS_PACKED;
  return 0;
}
int cc4_rollout_actions(cc4_handle* h, int32_t j, int32_t** d_actions) {
  if (!h->d_ract) { h->err = "cc4_rollout_actions: no rollout was begun on this handle"; return -2; }
  *d_actions = h->d_ract + (size_t)(j & 1) * (size_t)h->cfg.num_envs * NBLUE;
  return 0;
}
int cc4_rollout_policy_stream(cc4_handle* h, void** hip_stream) {
  if (!h->policy_stream) { h->err = "cc4_rollout_policy_stream: no rollout was begun on this handle"; return -2; }
  *hip_stream = h->policy_stream;
  return 0;
}
int cc4_rollout_wait_obs(cc4_handle* h, int32_t g, int32_t j, void* hip_stream) {
  if (rollout_ready(h, "cc4_rollout_wait_obs")) return -2;
  if (g < 0 || g >= RPG || j < 0 || j >= h->rollout_k) { h->err = "cc4_rollout_wait_obs: group or step out of range"; return -2; }
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->policy_stream;
  if (j == 0) { HIPCHK(h, hipStreamWaitEvent(st, h->rev, 0)); return 0; }
  hipLaunchKernelGGL(k_rollout_gate, dim3(1), dim3(WAVE), 0, st, h->d_rcnt, h->run_P, (int)cc4_handle::XRING, (int)g, (int)((j - 1) % cc4_handle::XRING), h->cfg.num_envs,
                     (long long)h->rollout_watchdog_ms * h->khz, h->d_rfail);
  HIPCHK(h, hipGetLastError());
  return 0;
}
int cc4_rollout_publish(cc4_handle* h, int32_t g, int32_t j, void* hip_stream) {
  if (rollout_ready(h, "cc4_rollout_publish")) return -2;
  if (g < 0 || g >= RPG || j < 0 || j >= h->rollout_k) { h->err = "cc4_rollout_publish: group or step out of range"; return -2; }
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->policy_stream;
  HIPCHK(h, hipStreamWriteValue32(st, h->d_rready + (size_t)g * 32, (uint32_t)(j + 1), 0));
  return 0;
}
int cc4_rollout_random_policy(cc4_handle* h, int32_t g, int32_t j, uint64_t seed0, uint32_t t, void* hip_stream) {
  if (rollout_ready(h, "cc4_rollout_random_policy")) return -2;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->policy_stream;
  const int tot = h->cfg.num_envs * NBLUE;
  hipLaunchKernelGGL(k_rollout_random_policy, dim3((tot + 255) / 256), dim3(256), 0, st, h->d_ract + (size_t)(j & 1) * (size_t)tot, h->cfg.num_envs, h->run_P, (int)g, seed0, t);
  HIPCHK(h, hipGetLastError());
  return 0;
}
int cc4_rollout_hash_policy(cc4_handle* h, int32_t g, int32_t j, void* hip_stream) {
  if (rollout_ready(h, "cc4_rollout_hash_policy")) return -2;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->policy_stream;
  const int n = h->cfg.num_envs;
  const uint8_t* rows = h->d_xslab + (size_t)((j + cc4_handle::XRING - 1) % cc4_handle::XRING) * (size_t)n * OBS_PACKED;
  hipLaunchKernelGGL(k_rollout_hash_policy, dim3((n + 255) / 256), dim3(256), 0, st, h->d_ract + (size_t)(j & 1) * (size_t)n * NBLUE, rows, n, h->run_P, (int)g, (uint32_t)j);
  HIPCHK(h, hipGetLastError());
  return 0;
}
int cc4_rollout_end(cc4_handle* h) {
  if (rollout_ready(h, "cc4_rollout_end")) return -2;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipStreamSynchronize(h->policy_stream));
  const int k = h->rollout_k;
  h->rollout_k = 0;
  uint32_t gate_failed = 0;
  HIPCHK(h, hipMemcpy(&gate_failed, h->d_rfail, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (*reinterpret_cast<volatile uint32_t*>(h->h_xtimeout) || gate_failed) {
    h->err = "cc4_rollout_end: a step of the " + std::to_string(k) + "-step rollout waited longer than " + std::to_string(h->rollout_watchdog_ms) +
             " ms for its actions (or a policy gate for its observations): not every group's policy pass of every step was published -- the episodes were "
             "stepped with whatever the action slots held (CC4_ROLLOUT_WATCHDOG_MS)";
    return -6;
  }
  return 0;
}
int cc4_launches_per_step(cc4_handle* h) { return h ? h->ngroups : 0; }
// host-side counters since cc4_create: steps issued by cc4_run_random_steps, microseconds the host spent enqueueing their step
// launches and their all-gathers, all-gathers issued, and how many times a step had to WAIT for an old all-gather before it
// could reuse that observation buffer (0 = the exchange never held the compute stream up)
int cc4_host_stats(cc4_handle* h, double* out /* [5] */) {
  out[0] = (double)h->stat_steps; out[1] = h->stat_launch_us; out[2] = h->stat_gather_us; out[3] = (double)h->gathers_issued; out[4] = (double)h->gather_stalls;
  return 0;
}

int cc4_get_state(cc4_handle* h, int32_t env, void* buf) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_get_state: env out of range"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(buf, h->d_state + env, sizeof(EnvState), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_set_state(cc4_handle* h, int32_t env, const void* buf) {
  h->prev_valid = false;        // (also every cc4_edit_state, which writes the rows back through here)
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_set_state: env out of range"; return -2; }
  {   // the cold containers of this handle were sized from cfg.steps; the row says how long ITS episode is (EnvState.steps)
    const int st_steps = static_cast<const EnvState*>(buf)->steps;
    // (0 = a never-reset, all-zero row; a negative length would turn into negative container capacities on the device)
    if (st_steps < 0 || (st_steps > 0 && cold_row_bytes(st_steps) != h->cold_row)) {
      h->err = "cc4_set_state: the row belongs to an episode of " + std::to_string(st_steps) + " steps, whose cold containers differ from this handle's (steps=" + std::to_string(h->cfg.steps) + ")";
      return -2;
    }
  }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(h->d_state + env, buf, sizeof(EnvState), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->full_obs_next = true;      // the observation buffer still holds the previous occupant's slowly varying values
  return 0;
}

size_t cc4_cold_bytes(cc4_handle* h) { return h ? h->cold_row : 0; }
int cc4_get_cold(cc4_handle* h, int32_t env, void* buf) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_get_cold: env out of range"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(buf, cold_at(h->d_cold, (size_t)env, h->cold_row), h->cold_row, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_set_cold(cc4_handle* h, int32_t env, const void* buf) {
  h->prev_valid = false;
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_set_cold: env out of range"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(cold_at(h->d_cold, (size_t)env, h->cold_row), buf, h->cold_row, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int cc4_get_topology(cc4_handle* h, int32_t env, uint8_t* out) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_get_topology: env out of range"; return -2; }
  EnvState* tmp = (EnvState*)malloc(sizeof(EnvState));
  HostStatic* hs = (HostStatic*)malloc(sizeof(HostStatic) * MAXH);
  int rc = cc4_get_state(h, env, tmp);
  if (rc == 0) {
    hipError_t e = hipMemcpy(hs, cold_at(h->d_cold, (size_t)env, h->cold_row)->hs, sizeof(HostStatic) * MAXH, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { h->err = std::string("cc4_get_topology: ") + hipGetErrorString(e); rc = -1; }
  }
  if (rc == 0) {
    for (int i = 0; i < NSUB; ++i) { out[i] = tmp->cidr_octet[i]; out[9 + i] = tmp->n_users[i]; out[18 + i] = tmp->n_servers[i]; }
    for (int i = 0; i < MAXH; ++i) { out[27 + 2 * i] = bit_get(tmp->exists, i) ? 1 : 0; out[28 + 2 * i] = hs[i].ip_octet; }
  }
  free(tmp); free(hs);
  return rc;
}

__global__ void k_set_evlog(EnvCold* cold, size_t row_bytes, int n, uint32_t on) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) { EnvCold* c = cold_at(cold, (size_t)e, row_bytes); c->evlog.enabled = on; c->evlog.n = 0; }
}
int cc4_enable_event_log(cc4_handle* h, int32_t enable) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  hipLaunchKernelGGL(k_set_evlog, dim3((h->cfg.num_envs + 255) / 256), dim3(256), 0, h->stream, h->d_cold, h->cold_row, h->cfg.num_envs, enable ? 1u : 0u);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->evlog_on = enable ? 1 : 0;
  return 0;
}
__global__ void k_copy_evlog(EnvCold* dst, const EnvCold* src, size_t row_bytes, int n) {
  const int e = blockIdx.x;
  if (e >= n) return;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&cold_at(const_cast<EnvCold*>(src), (size_t)e, row_bytes)->evlog);
  uint32_t* d = reinterpret_cast<uint32_t*>(&cold_at(dst, (size_t)e, row_bytes)->evlog);
  for (int i = threadIdx.x; i < (int)(sizeof(EvLog) / 4); i += blockDim.x) d[i] = s[i];
}
// The event log on demand (handles of up to 16 episodes without a communicator).  cc4_keep_previous(1): every step launch is preceded by a
// device-side copy of the episodes' rows.  cc4_replay_logged: the LAST step again, on that copy, with the logging build of the step kernel
// and the same inputs (the action / message / submitted-action buffers still hold them) -- the copy ends where the live rows are, and its
// event log is copied into the live cold rows: cc4_get_true_state then reports the HostEvents entries of the last step although the step
// itself ran the fast build.  With no step since the reset the log is simply empty.  Same generator positions: logging draws nothing in
// the numpy-stream mode and only side streams in the counter mode.
int cc4_keep_previous(cc4_handle* h, int32_t on) {
  if (on && (h->comm || h->cfg.num_envs > 16 || h->cfg.autoreset)) { h->err = "cc4_keep_previous: for handles of up to 16 episodes without a communicator or autoreset"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t n = (size_t)h->cfg.num_envs;
  if (on && !h->d_prev_state) {
    HIPCHK(h, hipMalloc(&h->d_prev_state, n * sizeof(EnvState)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->d_prev_cold), n * h->cold_row));
    HIPCHK(h, hipMalloc(&h->d_prev_out, h->out_bytes));
  }
  h->keep_prev = on != 0;
  h->prev_valid = false;
  return 0;
}
int cc4_replay_logged(cc4_handle* h) {
  if (!h->keep_prev) { h->err = "cc4_replay_logged: cc4_keep_previous is off"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  const int n = h->cfg.num_envs;
  if (!h->prev_valid) {      // no step since the reset (or steps whose inputs are gone): an enabled, empty log
    hipLaunchKernelGGL(k_set_evlog, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_cold, h->cold_row, n, 1u);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
  }
  hipLaunchKernelGGL(k_set_evlog, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_prev_cold, h->cold_row, n, 1u);
  HIPCHK(h, hipGetLastError());
  int32_t* o = reinterpret_cast<int32_t*>(h->d_prev_out);
  float* rw = reinterpret_cast<float*>(o + (size_t)n * OBS_TOTAL);
  uint32_t* er = reinterpret_cast<uint32_t*>(rw + n);
  uint8_t* dn = reinterpret_cast<uint8_t*>(er + n);
  StepArgs a{h->d_prev_state, h->d_prev_cold, h->prev_actions, h->prev_msgs, o, rw, dn, er, nullptr, nullptr, 0, 0,
             n, 0, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), 1,
             (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, h->prev_ext ? h->d_ext : nullptr, 0};
  for (int g = 0; g < h->ngroups; ++g) launch_group(h, a, g, true, nullptr, nullptr);
  HIPCHK(h, hipGetLastError());
  if (h->ngroups > 1) { h->groups_busy = true; if (join_groups(h)) return -1; }
  hipLaunchKernelGGL(k_copy_evlog, dim3(n), dim3(64), 0, h->stream, h->d_cold, h->d_prev_cold, h->cold_row, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->prev_valid = false;          // (the copy has moved on: a second replay would repeat the step from the wrong rows)
  return 0;
}
int64_t cc4_get_true_state(cc4_handle* h, int32_t env, char* json, size_t cap) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_get_true_state: env out of range"; return -2; }
  EnvState* st = (EnvState*)malloc(sizeof(EnvState));
  EnvCold* cold = (EnvCold*)malloc(h->cold_row);
  int64_t rc = cc4_get_state(h, env, st);
  if (rc == 0) rc = cc4_get_cold(h, env, cold);
  if (rc == 0) {
    std::string doc = export_true_state(*st, *cold);
    rc = (int64_t)doc.size() + 1;
    if (json && cap >= doc.size() + 1) memcpy(json, doc.c_str(), doc.size() + 1);
  }
  free(st); free(cold);
  return rc;
}

// debug: enable (buf != NULL first call allocates) / read per-episode cycle counters [N][64] (16 phase slots, then 8 per red agent)
int cc4_debug_profile(cc4_handle* h, int enable, unsigned long long* out) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  size_t bytes = (size_t)h->cfg.num_envs * PROF_SLOTS * sizeof(unsigned long long);
  if (enable && !h->d_prof) { HIPCHK(h, hipMalloc(&h->d_prof, bytes)); HIPCHK(h, hipMemsetAsync(h->d_prof, 0, bytes, h->stream)); }
  if (out && h->d_prof) { HIPCHK(h, hipMemcpyAsync(out, h->d_prof, bytes, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream)); }
  if (!enable && h->d_prof) { (void)hipFree(h->d_prof); h->d_prof = nullptr; }
  return 0;
}

// debug: the red policy phase of every episode with G episodes' agents per wave; out[0] = mean launch duration in us, out[1] = mean cycles
// of a wave in the phase, out[2] = waves per launch.  Reads the batch as it stands, writes nothing back.
int cc4_debug_policy_probe(cc4_handle* h, int32_t G, int32_t reps, double* out) {
#ifndef CC4_POLICY_PROBE
  (void)G; (void)reps; (void)out;
  h->err = "cc4_debug_policy_probe: this library was built without -DCC4_POLICY_PROBE (the experiment of DESIGN 3.4 is concluded; tools/policy_group_probe.py says how to build it)";
  return -2;
#else
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (h->cfg.rng_mode != 1) { h->err = "cc4_debug_policy_probe: counter mode only"; return -2; }
  if (join_groups(h)) return -1;
  const int n = h->cfg.num_envs, waves = (n + G - 1) / G;
  unsigned long long* d_cyc = nullptr;
  HIPCHK(h, hipMalloc(&d_cyc, (size_t)waves * sizeof(unsigned long long)));
  StepArgs a{h->d_state, h->d_cold, nullptr, nullptr, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, nullptr, 0, 0,
             n, 0, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), 0,
             (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, nullptr, 0};
  hipEvent_t e0, e1;
  HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
  const size_t dyn = (size_t)G * offsetof(EnvState, hd);
  auto launch = [&]() {
    switch (G) {
      case 1: hipLaunchKernelGGL(k_policy_probe<1>, dim3(waves), dim3(WAVE), dyn, h->stream, a, d_cyc); break;
      case 2: hipLaunchKernelGGL(k_policy_probe<2>, dim3(waves), dim3(WAVE), dyn, h->stream, a, d_cyc); break;
      case 4: hipLaunchKernelGGL(k_policy_probe<4>, dim3(waves), dim3(WAVE), dyn, h->stream, a, d_cyc); break;
      default: hipLaunchKernelGGL(k_policy_probe<8>, dim3(waves), dim3(WAVE), dyn, h->stream, a, d_cyc); break;
    }
  };
  if (G != 1 && G != 2 && G != 4 && G != 8) { h->err = "cc4_debug_policy_probe: G is 1, 2, 4 or 8"; (void)hipFree(d_cyc); return -2; }
  if (G == 8) HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_policy_probe<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  launch();                                                     // warm-up
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(e0, h->stream));
  for (int i = 0; i < reps; ++i) launch();
  HIPCHK(h, hipEventRecord(e1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f; HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc((size_t)waves);
  HIPCHK(h, hipMemcpy(cyc.data(), d_cyc, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double sum = 0; for (auto c : cyc) sum += (double)c;
  out[0] = (double)ms * 1000.0 / (reps > 0 ? reps : 1); out[1] = sum / waves; out[2] = waves;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d_cyc);
  return 0;
#endif
}
int cc4_debug_copy_from_device(cc4_handle* h, void* host_dst, const void* device_src, size_t bytes) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->policy_stream) HIPCHK(h, hipStreamSynchronize(h->policy_stream));
  HIPCHK(h, hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int cc4_comm_unique_id(void* id128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, 128);
  return 0;
}
int cc4_comm_init(cc4_handle* h, int32_t rank, int32_t world, const void* id128) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclResult_t r = ncclCommInitRank(&h->comm, world, id, rank);
  if (r != ncclSuccess) { h->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); return -1; }
  h->rank = rank; h->world = world;
  if (!getenv("CC4_GROUPS")) {
    // With the exchange every launch carries a completion event and the host guards the observation ring, so a launch costs the
    // host several times what it costs without; small shards then run into the host.  Measured on MI355X with the exchange on a
    // one-rank communicator (r03, profiles/r03_exchange_groups_world1.txt; M agent-env steps/s, 1 / 2 / 3 launches per step):
    // 1024 episodes 159 / 104 / 111, 2048: 242 / 179 / 220, 4096: 380 / 282 / 420, 8192: 478 / 498 / 607.
    int ng = h->cfg.num_envs >= 4096 ? 3 : 1;      // (8192 episodes with the exchange, 3 / 4 launches per step: 563 / 509 M)
    if (const char* v = getenv("CC4_EXCHANGE_GROUPS")) { ng = atoi(v); if (ng <= 0 || ng > h->ngroups) ng = h->ngroups; }   // tuning override: 0 = keep the handle's groups
    if (ng != h->ngroups) {
      if (sync_all(h)) return -1;
      const int old = h->ngroups;
      configure_groups(h, ng);
      for (int g = old; g < h->ngroups; ++g) {
        if (!h->gstream[g]) HIPCHK(h, hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking));
        if (!h->gev[g]) HIPCHK(h, hipEventCreateWithFlags(&h->gev[g], hipEventDisableTiming));
      }
      h->main_ahead = true;
    }
  }
  size_t nb = (size_t)h->cfg.num_envs * OBS_PACKED;
  HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  for (int b = 0; b < cc4_handle::OBS_RING; ++b) {
    HIPCHK(h, hipMalloc(&h->d_obs8[b], nb));
    HIPCHK(h, hipMalloc(&h->d_all_obs8[b], nb * (size_t)world));
    HIPCHK(h, hipMemsetAsync(h->d_obs8[b], 0, nb, h->stream));
    for (int g = 0; g < h->ngroups; ++g) HIPCHK(h, hipEventCreateWithFlags(&h->ev_step[b][g], hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_comm[b], hipEventDisableTiming));
  }
  // (the ring of step slabs the one-launch kernels write, XchgArgs, and its gathered twin -- 32 x (1 + world) x N x 148 B -- are allocated by
  // the first call that takes a one-launch form: xchg_begin)
  HIPCHK(h, hipEventCreateWithFlags(&h->xev, hipEventDisableTiming));
  int can_wait = 0;
  (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, h->cfg.device_id);
  h->xchg_on = can_wait != 0;
  if (const char* v = getenv("CC4_EXCHANGE_INKERNEL")) h->xchg_on = h->xchg_on && atoi(v) != 0;
  if (const char* v = getenv("CC4_EXCHANGE_CHUNK")) { h->xchg_chunk = atoi(v); if (h->xchg_chunk < 1) h->xchg_chunk = 1; if (h->xchg_chunk > cc4_handle::XRING / 2) h->xchg_chunk = cc4_handle::XRING / 2; }
  if (const char* v = getenv("CC4_EXCHANGE_WATCHDOG_MS")) { h->xchg_watchdog_ms = atoi(v) > 0 ? atoi(v) : 2000; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  // the one-launch forms again, now that a step of one episode waits for the slowest episode of sixteen steps earlier: the multi-step kernels
  // must hold the whole batch with a block per CU to spare (at exactly full residency one block that is placed late stalls everybody until
  // the watchdog: tools/micro/ring_protocol.hip), and with peers RCCL's kernels need that room on every form (the persistent kernel's waves
  // pull items, so on one rank it keeps every slot)
  if (h->xchg_on) { if (choose_run_form(h, 1, world > 1 ? 1 : 0)) return -1; }
  return 0;
}
// the in-kernel exchange of this handle: out[0] on (1) / off (0), out[1] ring depth in steps, out[2] steps per publish (CC4_EXCHANGE_CHUNK),
// out[3] calls of cc4_run_random_steps it served, out[4] calls whose watchdog fired (the handle then returns to per-step launches)
int cc4_exchange_info(cc4_handle* h, int32_t* out /* [5] */) {
  out[0] = h->xchg_on ? 1 : 0; out[1] = cc4_handle::XRING; out[2] = h->xchg_chunk; out[3] = (int32_t)h->xchg_calls; out[4] = (int32_t)h->xchg_timeouts;
  return 0;
}
// debug / test hook: keep the gathered rows of the next `steps` steps cc4_run_random_steps exchanges from inside a one-launch kernel
// ([steps][world * N] packed rows, in step order), so that a test can check EVERY step's all-gather, not only the last of a burst.
// steps = 0 frees the log.
int cc4_debug_gather_log(cc4_handle* h, int32_t steps) {
  if (!h->comm) { h->err = "cc4_debug_gather_log: cc4_comm_init was not called"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  if (h->d_xlog) { (void)hipFree(h->d_xlog); h->d_xlog = nullptr; }
  h->xlog_cap = 0; h->xlog_n = 0;
  if (steps > 0) {
    HIPCHK(h, hipMalloc(&h->d_xlog, (size_t)steps * h->world * h->cfg.num_envs * OBS_PACKED));
    h->xlog_cap = steps;
  }
  return 0;
}
// host copy of the log: out [count][world * N][CC4_OBS_PACKED_BYTES]; returns the number of steps logged so far (< 0: error)
int cc4_get_gather_log(cc4_handle* h, uint8_t* out, int32_t first, int32_t count) {
  if (!h->d_xlog || first < 0 || count < 0 || first + count > h->xlog_n) { h->err = "cc4_get_gather_log: no log, or the range was not logged"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  const size_t row = (size_t)h->world * h->cfg.num_envs * OBS_PACKED;
  if (count) HIPCHK(h, hipMemcpy(out, h->d_xlog + (size_t)first * row, (size_t)count * row, hipMemcpyDeviceToHost));
  return h->xlog_n;
}
// What a multi-GPU run needs to PROVE its scaling line: RCCL's own view of the communicator (how many ranks it spans, which one this
// is, which device it is bound to) and the identity of the device this handle runs on.  out[0] ncclCommCount (1 without a
// communicator), out[1] ncclCommUserRank (0), out[2] ncclCommCuDevice (-1), out[3] the handle's HIP device ordinal, out[4] PCI
// domain, out[5] PCI bus, out[6] PCI device, out[7] compute units; uuid_hex: 32 hex digits + NUL of hipDeviceProp_t::uuid.
int cc4_comm_info(cc4_handle* h, int32_t* out, char* uuid_hex) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int count = 1, urank = 0, cudev = -1;
  if (h->comm) {
    if (ncclCommCount(h->comm, &count) != ncclSuccess || ncclCommUserRank(h->comm, &urank) != ncclSuccess || ncclCommCuDevice(h->comm, &cudev) != ncclSuccess) {
      h->err = "cc4_comm_info: RCCL did not answer"; return -1;
    }
  }
  hipDeviceProp_t prop;
  HIPCHK(h, hipGetDeviceProperties(&prop, h->cfg.device_id));
  out[0] = count; out[1] = urank; out[2] = cudev; out[3] = h->cfg.device_id;
  out[4] = prop.pciDomainID; out[5] = prop.pciBusID; out[6] = prop.pciDeviceID; out[7] = prop.multiProcessorCount;
  if (uuid_hex) { for (int i = 0; i < 16; ++i) snprintf(uuid_hex + 2 * i, 3, "%02x", (unsigned)(unsigned char)prop.uuid.bytes[i]); }
  return 0;
}
// All-gather of the observations written by the most recent step (as bytes, [world*N][578]) over RCCL/xGMI on the
// handle's communication stream: it waits for that step's kernel, runs concurrently with whatever is enqueued next on
// the compute stream (later steps write other buffers of the ring), and is awaited by cc4_allgather_wait / the step that
// reuses its buffer.  *d_all_obs8 is valid after cc4_allgather_wait().
int cc4_allgather_obs(cc4_handle* h, uint8_t** d_all_obs8) {
  if (!h->comm) { h->err = "cc4_allgather_obs: cc4_comm_init was not called"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int buf = h->obs_buf;
  if (h->obs8_from_slab >= 0) {    // the last step ran inside a one-launch kernel with the exchange: its packed rows are in the exchange ring
    const size_t row = (size_t)h->cfg.num_envs * OBS_PACKED;
    HIPCHK(h, hipMemcpyAsync(h->d_obs8[buf], h->d_xslab + (size_t)h->obs8_from_slab * row, row, hipMemcpyDeviceToDevice, h->stream));
    h->obs8_from_slab = -1;
  }
  if (!h->step_event_attached) {   // e.g. the observations of a reset: main-stream work, behind which the group streams' work was joined
    if (join_groups(h)) return -1;
    HIPCHK(h, hipEventRecord(h->ev_step[buf][0], h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_step[buf][0], 0));
  } else {
    for (int g = 0; g < h->ngroups; ++g) HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_step[buf][g], 0));
  }
  if (h->comm_delay_ticks > 0) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, h->comm_stream, h->comm_delay_ticks); HIPCHK(h, hipGetLastError()); }
  size_t cnt = (size_t)h->cfg.num_envs * OBS_PACKED;
  ncclResult_t r = ncclAllGather(h->d_obs8[buf], h->d_all_obs8[buf], cnt, ncclUint8, h->comm, h->comm_stream);
  if (r != ncclSuccess) { h->err = std::string("ncclAllGather: ") + ncclGetErrorString(r); return -1; }
  const long long q = ++h->gathers_issued;
  h->gather_seq[buf] = q;
  h->gather_buf = buf;
  h->last_gathered = h->d_all_obs8[buf];
  HIPCHK(h, hipEventRecord(h->ev_comm[q % cc4_handle::OBS_RING], h->comm_stream));
  if (d_all_obs8) *d_all_obs8 = h->d_all_obs8[buf];
  return 0;
}
// debug / test hook: every all-gather is preceded by a kernel that keeps the communication stream busy for about `us`
// microseconds -- an exchange slower than the step, which is what makes the observation ring's reuse guard work for its living
int cc4_debug_comm_delay_us(cc4_handle* h, int us) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int khz = 100000;   // wall_clock64 ticks at the constant 100 MHz reference clock
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id);
  if (khz <= 0) khz = 100000;
  h->comm_delay_ticks = (long long)us * khz / 1000;
  return 0;
}
int cc4_allgather_wait(cc4_handle* h) {
  if (!h->comm) { h->err = "cc4_allgather_wait: cc4_comm_init was not called"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  return 0;
}
// host copy of the gathered observations of the most recent cc4_allgather_obs (tests / debugging)
int cc4_get_allgathered_obs(cc4_handle* h, uint8_t* out /* [world*N][578] */) {
  if (!h->comm) { h->err = "cc4_get_allgathered_obs: cc4_comm_init was not called"; return -2; }
  if (!h->last_gathered) { h->err = "cc4_get_allgathered_obs: no all-gather has been issued"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  const size_t rows = (size_t)h->world * h->cfg.num_envs;
  std::vector<uint8_t> packed(rows * OBS_PACKED);
  HIPCHK(h, hipMemcpy(packed.data(), h->last_gathered, packed.size(), hipMemcpyDeviceToHost));
  for (size_t r = 0; r < rows; ++r)      // unpack to one byte per value for the host caller
    for (int i = 0; i < OBS_TOTAL; ++i) out[r * OBS_TOTAL + i] = (uint8_t)((packed[r * OBS_PACKED + (i >> 2)] >> (2 * (i & 3))) & 3u);
  return 0;
}
// Device-side consumer of the exchange format: the gathered rows of the most recent cc4_allgather_obs ([world*N] rows of
// CC4_OBS_PACKED_BYTES, 2 bits per value) unpacked to [world*N][578] bytes in a buffer owned by the handle -- what a shared
// on-GPU policy reads.  Enqueued on the communication stream behind the all-gather; *d_obs_u8 is valid after
// cc4_allgather_wait() (or after any later operation ordered behind ev_comm of that gather).
int cc4_unpack_obs_device(cc4_handle* h, uint8_t** d_obs_u8) {
  if (!h->comm) { h->err = "cc4_unpack_obs_device: cc4_comm_init was not called"; return -2; }
  if (!h->last_gathered) { h->err = "cc4_unpack_obs_device: no all-gather has been issued"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t rows = (size_t)h->world * h->cfg.num_envs;
  if (!h->d_unpacked) HIPCHK(h, hipMalloc(&h->d_unpacked, rows * OBS_TOTAL));
  hipLaunchKernelGGL(k_unpack_obs, dim3((unsigned)rows), dim3(192), 0, h->comm_stream, h->last_gathered, h->d_unpacked, (int)rows);
  HIPCHK(h, hipGetLastError());
  if (d_obs_u8) *d_obs_u8 = h->d_unpacked;
  return 0;
}
// host copy of that buffer (tests)
int cc4_get_unpacked_obs(cc4_handle* h, uint8_t* out /* [world*N][578] */) {
  if (!h->d_unpacked) { h->err = "cc4_get_unpacked_obs: cc4_unpack_obs_device was not called"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  HIPCHK(h, hipMemcpy(out, h->d_unpacked, (size_t)h->world * h->cfg.num_envs * OBS_TOTAL, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
