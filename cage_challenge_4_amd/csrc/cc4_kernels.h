// cc4_kernels.h -- what the translation units of libcc4.so share: the argument blocks of the kernels (StepArgs, XchgArgs, RunArgs, ResetArgs),
// the device helpers every step kernel uses (row staging, packed observation rows, the exchange's slab protocol, the schedule of the persistent
// kernel), and the declarations of the kernels for the host side (csrc/cc4_api.hip).  The kernels themselves:
//   cc4_k_pcg.hip      numpy-stream mode: k_step<LOG>, k_run_pcg
//   cc4_k_philox4.hip  counter mode, four wavefronts per episode: k_step_philox<LOG, MINW>, k_run_philox, k_run_philox8
//   cc4_k_philox1.hip  counter mode, one wavefront per episode: k_step_philox1<LOG>, k_run_philox1m (cc4_philox1_body.h: the step's body)
//   cc4_k_run1.hip     the persistent kernel of large batches: k_run_philox1 (cc4_persist.h: its schedule, shared with k_run_pcg)
//   cc4_k_run1x.hip    its other builds: k_run_philox1x (beside RCCL), k_run_philox1r (rollouts with the policy in the loop)
//   cc4_k_misc.hip     k_reset and the small helpers (exchange gate, CU discovery, stand-in policies, digest, ...)
// No MFMA anywhere: the path is integer / indexing.
#pragma once
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
  int dbg_stop;               // measurement (cc4_debug_stop_phase, full build of k_step_philox1 only): the step ends after its phase number dbg_stop and
                              // writes no row back -- the instruction counters of such launches, differenced, are the instructions of each phase
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
// The observation values that can change with every step: position, source byte and mask of value v = obs_fast_entry(v) (cc4_engine.h), as a table the
// compiler fills (one copy per translation unit, 1.5 KB of constant memory that the vector L1 keeps).  Until r06 the entries were computed per value and
// step (r03 A/B: a dozen shifts and multiplies beat one load while the kernels waited on memory, not on the vector unit); at 24 waves per CU the step is
// bound by vector issue slots and those instructions were a sixth of it (profiles/r06_valu_phases.txt: 453 of 2 689 per episode-step).
// Monitor's end-turn roll-over of the hosts' event bytes, four hosts per lane (monitor_roll4): the watched-hosts byte masks of the 35 words
struct MonitorWatchTab { uint32_t v[(MAXH + 3) / 4]; };
constexpr MonitorWatchTab make_monitor_watch_tab() { MonitorWatchTab t{}; for (int w = 0; w < (MAXH + 3) / 4; ++w) t.v[w] = monitor_watch_mask(w); return t; }
static __device__ const MonitorWatchTab monitor_watch_tab = make_monitor_watch_tab();
__device__ __forceinline__ void monitor_roll_all(EnvState* s, int lane) {
  static_assert((MAXH + 3) / 4 <= WAVE && offsetof(EnvState, hev) % 4 == 0, "one pass of the wave over aligned words");
  if (lane < (MAXH + 3) / 4) {
    uint32_t* const w = reinterpret_cast<uint32_t*>(s->hev) + lane;
    *w = monitor_roll4(*w, monitor_watch_tab.v[lane]);
  }
}
// Device entry: byte position in the vector (4 x index, 12 bits) | byte offset of the source byte in the staged row << 12 (hev[] and msg[][] both live in
// the agent part, below 8 KB) | bit mask << 25 (the event masks are nibbles), so that a value is one LDS byte read, an and, a compare and a store.
struct ObsFastTab { uint32_t v[OBS_FAST]; };
constexpr ObsFastTab make_obs_fast_tab() {
  ObsFastTab t{};
  for (int v = 0; v < OBS_FAST; ++v) {
    const uint32_t e = obs_fast_entry(v), src = (e >> 10) & 0xFFu;
    const uint32_t off = src < (uint32_t)MAXH ? (uint32_t)offsetof(EnvState, hev) + src : (uint32_t)offsetof(EnvState, msg) + (src - (uint32_t)MAXH);
    t.v[v] = ((e & 0x3FFu) << 2) | (off << 12) | ((e >> 18) << 25);
  }
  return t;
}
static_assert(offsetof(EnvState, hev) + MAXH <= 8192 && offsetof(EnvState, msg) + NBLUE * MSG_LEN <= 8192, "source offsets fit 13 bits");
static_assert((EV_CUR_PROC | EV_OLD_PROC | EV_CUR_CONN | EV_OLD_CONN) < 128 && OBS_TOTAL * 4 <= 4096, "mask and byte position fit their fields");
static __device__ const ObsFastTab obs_fast_tab = make_obs_fast_tab();
// msgs_clean: the message values ([224, 384) of the enumeration) need no rewrite -- no agent sent a message with this step nor with the step before, whose
// encode wrote their zeros (the persistent kernels know that of every step of a launch but the first: cc4_run_random_steps carries no messages)
template <int nt>
__device__ __forceinline__ void encode_obs_fast(const EnvState* s, int32_t* o, uint8_t* obs_bytes, bool pack, int t, bool msgs_clean = false) {
  constexpr int NV = (OBS_FAST + nt - 1) / nt;
  constexpr int K_MSG = (224 + nt - 1) / nt;             // rounds from this one on hold message values only
  constexpr bool EXACT = OBS_FAST % nt == 0;             // one wave: 6 x 64 values, no lane is ever out of range
  const uint32_t tt = (uint32_t)t;
  __builtin_assume(tt < (uint32_t)nt);
  uint32_t ent[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) { const uint32_t v = tt + (uint32_t)(k * nt); ent[k] = (EXACT || v < (uint32_t)OBS_FAST) ? obs_fast_tab.v[v] : 0u; }   // NV independent loads, one wait
  const uint8_t* const row = reinterpret_cast<const uint8_t*>(s);
  char* const ob = reinterpret_cast<char*>(o);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const uint32_t v = tt + (uint32_t)(k * nt);
    if (!EXACT && v >= (uint32_t)OBS_FAST) continue;
    if (k >= K_MSG && msgs_clean) continue;              // (wave-uniform)
    const uint32_t byte = row[(ent[k] >> 12) & 0x1FFFu];
    const int val = (byte & (ent[k] >> 25)) != 0 ? 1 : 0;
    const uint32_t pos = ent[k] & 0xFFFu;
    *reinterpret_cast<int32_t*>(ob + pos) = val;
    if (pack) obs_bytes[pos >> 2] = (uint8_t)val;
  }
}
// The slowly varying observation values by kind, one wave (env_flat_obs_sorted's enumeration: 63 blocked bits, 63 comms-policy bits, 63 subnet one-hots,
// 5 phase words -- one pass of the wave each): `dirty` = what the step changed (EnvState.obs_dirty: OD_BLOCKS, OD_PHASE), OD_ALL after a reset or when the
// caller's buffer is new (the one-hots never change otherwise)
__device__ __forceinline__ void encode_obs_slow(const EnvState* s, int32_t* o, uint32_t dirty, int lane) {
  if (!dirty) return;
  auto put = [&](int v) { int i; const int val = env_flat_obs_sorted(s, v, &i); o[i] = val; };
  if ((dirty & OD_BLOCKS) && lane < 63) put(OBS_FAST + lane);
  if (dirty & OD_PHASE) { if (lane < 63) put(OBS_FAST + 63 + lane); if (lane < 5) put(OBS_FAST + 189 + lane); }
  if ((dirty & ~(uint32_t)(OD_BLOCKS | OD_PHASE)) && lane < 63) put(OBS_FAST + 126 + lane);
}
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
  int pool;                    // (2: the balanced schedule below -- the only one since r06; r05's tail-only sharing and the XCD pools experiment are in docs/HISTORY.md)
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
  const uint32_t* act_ready;   // [P][32 words]: one cache line per CU partition, word g = the steps of policy group g whose actions are published -- the
                               // publisher writes all P copies, a wave polls its own CU's (thousands of waves polling ONE uncached line starve the very
                               // store they wait for: ~50 us per pass, profiles/r06_rollout.txt); null: no rollout
  const int32_t* act;          // [2][n][5]
  int PG;
  long long act_wait_ticks;    // watchdog: a step that waits longer for its actions gives up, raises XchgArgs.timeout, and every later wait returns at once
};
// lane 0: the actions of step j for policy group g are published.  Polls a device word at a growing interval (see xchg_wait_slab).
__device__ __forceinline__ void rollout_wait_actions(const RunArgs& ra, const XchgArgs& x, int line, int g, uint32_t j) {
  const uint32_t* w = ra.act_ready + (size_t)line * 32 + g;
  if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) > j) return;
  if (__hip_atomic_load(x.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
  const long long w0 = wall_clock64();
  int naps = 1;
  while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) <= j) {
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(32);
    if (naps < 4) naps <<= 1;
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
constexpr uint32_t TK_SHARED = 0x80000000u;
constexpr int RPG_MAX = 4;  // policy groups of a rollout (cc4_handle::rpg of them, CC4_ROLLOUT_GROUPS): group of episode e = (e / P) % groups -- the others step while one group's policy pass is under way
struct ResetArgs {
  EnvState* st; EnvCold* cold; const uint64_t* seeds; const uint8_t* env_mask;
  int32_t* obs; float* reward; uint8_t* done; uint32_t* err; uint8_t* mask;
  int n, steps, rng_mode, policy;
  uint32_t topo;
  uint8_t* obs8;               // packed exchange row of the reset observations (multi-GPU), or null
};
// the step bodies (defined in cc4_k_pcg.hip / cc4_philox1_body.h; declared here for the persistent schedule, cc4_persist.h)
template <bool LOG> __device__ __forceinline__ void pcg_body(StepArgs a, const int e, const int lane, const bool first = true, const bool last = true);
template <bool LOG, bool PERSIST> __device__ __forceinline__ void philox1_body(StepArgs a, const int e, const uint32_t rand_t, const uint32_t item_k, const int lane,
                                                                const bool first = true, const bool last = true);

