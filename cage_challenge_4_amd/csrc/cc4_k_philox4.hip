// cc4_k_philox4.hip -- counter mode, four wavefronts per episode: k_step_philox<LOG, MINW>, k_run_philox, k_run_philox8.  See cc4_kernels.h.
#include "cc4_kernels.h"

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
      rl.mode = 3;                             // philox, thread-private (cc4_rng.h)
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


// the kernels the host side launches (cc4_kernel_decls.h)
template __global__ void k_step_philox<false, 1>(StepArgs);
template __global__ void k_step_philox<false, 7>(StepArgs);
template __global__ void k_step_philox<false, 8>(StepArgs);
template __global__ void k_step_philox<true, 1>(StepArgs);
