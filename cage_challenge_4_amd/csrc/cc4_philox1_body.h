// cc4_philox1_body.h -- one step of one episode on one wavefront (counter mode): the body of k_step_philox1, k_run_philox1m and the persistent kernel.
#pragma once
#include "cc4_kernels.h"

// ---------------------------------------------------------------- Philox mode, one wavefront per episode
// The same step as k_step_philox with the agents on the LANES of a single wave instead of on four waves: red agent r on lane
// r, blue agent b on lane 8 + b for its submission and on lane b for its action, green agent g on lane g % 64.  A block of
// four waves spends most of its resident time with three waves parked at a barrier behind the one that carries the red
// agents; with one wave per episode every resident wave works, and as a wave64 instruction occupies its SIMD for four
// cycles whatever the number of active lanes, what counts at large batches is the number of instructions per episode, not
// their spread over waves.  Like the numpy-stream kernel it stages only the agent part of the row (6 992 B) and leaves the
// host table in HBM / L2, so 16 episodes are resident per CU (LDS) instead of 7-8.  The build for throughput-bound batches;
// k_step_philox keeps the shorter single-launch latency of small ones (cc4_create picks; CC4_PHILOX_LEAN overrides).
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
// cc4_debug_stop_phase (full build only; wave-uniform): leave the step behind phase n.  A library built with -DCC4_STOP_IN_FAST=1 has the stops in the
// fast build too (tools/valu_phases.sh measures that variant: on the fast path rates and flags are compile-time constants the full build carries at run time)
#ifndef CC4_STOP_IN_FAST
#define CC4_STOP_IN_FAST 0
#endif
#define CC4_STOP_ON (LOG || CC4_STOP_IN_FAST)
#define CC4_STOP(n) do { if (CC4_STOP_ON && a.dbg_stop == (n)) return; } while (0)
template <bool LOG, bool PERSIST>
__device__ __forceinline__ void philox1_body(StepArgs a, const int e, const uint32_t rand_t, const uint32_t item_k, const int lane,
                                             const bool first, const bool last) {
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
  // a later step of a one-launch call that carries no messages: the step before (of this very launch) stored the agents' zero message bytes and encoded
  // their zeros -- neither needs redoing (a regeneration in between rewrites everything, and leaves zeros)
  const bool msgs_clean = PERSIST && item_k > 0 && !a.msgs;
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
      CC4_STOP(1);                                                               // staged in, work area zeroed, step_phase
      const int ng = s->n_green;
      // one lane-private generator per lane, in registers: every use starts with rng_set_stream(); mode pinned so the PCG
      // paths fold away
      Rng rl;
      rng_fork(&rl, &s->rng, ST_RESET);
      rl.mode = 3;                             // philox, lane-private (cc4_rng.h)
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
        philox4x32_10(bank, k0, k1, true);
      }
      // the four words lane `lane + shift` holds, on every lane (call with all lanes active: an inactive source lane reads as 0)
      auto bank_fetch = [&](int shift, uint32_t out[4]) {
        const int addr = ((lane + shift) & (WAVE - 1)) << 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)bank[k]);
      };
      CC4_STOP(2);                                                               // + the block bank
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
      } else if (lane >= 8 && lane < 8 + NBLUE && !(CC4_STOP_ON && a.dbg_stop == 3)) {
        const int b = lane - 8;
        int32_t act = !a.actions ? -1 : a.act_sys ? __hip_atomic_load(a.actions + e * NBLUE + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : a.actions[e * NBLUE + b];
        if (a.rand_out) { act = (int32_t)(((uint64_t)brand * (uint32_t)(b == 4 ? ACT_LONG : ACT_SHORT)) >> 32); a.rand_out[e * NBLUE + b] = act; }   // == random_blue_action
        step_blue_submit(xg, b, act);
        step_tick_blue(xg, b);
        if (!msgs_clean) step_messages(s, a.msgs ? a.msgs + e * NBLUE * MSG_LEN : nullptr, b);   // read back by this step's observation encode only
      }
      if (!(CC4_STOP_ON && (a.dbg_stop == 3 || a.dbg_stop == 4))) {
        if (lane < ng) step_green_policy(xg, lane);                             // agents 0..63: their block is computed here, by all of them at once
        if (lane + WAVE < ng) step_green_policy(xg, lane + WAVE, bank);          // agents 64..: from the bank (their lane's own request)
      }
      __syncthreads();
      CC4_TICK(x0, 2);
      CC4_STOP(3); CC4_STOP(4); CC4_STOP(5);                                     // + red policies and queue ticks / + blue submissions / + green policies
      // ---- P3b blue actions: side by side when they are independent (no Monitor, no pending pid events), else in the reference's order on
      // lane 0 (ControlTraffic first, then agent order: step_blue_exec).  ONE call site of blue_execute for both forms -- the action bodies are
      // a quarter of the kernel's code, and a second and third inlined copy of them is what the instruction cache holds least well.
      {
        const bool indep = blue_exec_independent(s);
        if (lane == 0) CC4_TICK(x0, 3);
        uint32_t c[4];
        bank_fetch(BK_BEXE, c);                                                   // blue b (lane b) <- lane BK_BEXE + b
        Ctx xb = xg;
        if (!indep) xb.prof = x0.prof;
        const int rounds = indep ? 1 : 2 * NBLUE;
#pragma nounroll
        for (int i = 0; i < rounds; ++i) {
          int b = lane;
          bool go = lane < NBLUE;
          if (!indep) {
            b = i < NBLUE ? i : i - NBLUE;
            const int ty = s->bexec[b].type;
            go = lane == 0 && ((ty == BA_BLOCK || ty == BA_ALLOW) == (i < NBLUE));
          }
          if (go) {
            rng_set_stream(&rl, ST_BLUE_EXE + (uint32_t)b);
            if (indep) rng_preload(&rl, c);
            blue_execute(xb, b, s->bexec[b]);
          }
        }
        __syncthreads();
        if (lane == 0) CC4_TICK(x0, 5);
      }
      CC4_STOP(6);                                                               // + blue actions
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
      CC4_STOP(7);                                                               // + green actions
      // ---- P5 deferred phishing (ordered), then P6 red actions: side by side when they name distinct hosts
      if (lane == 0) { step_phishing(x0); CC4_TICK(x0, 1); rs_reserve(x0); conflict_lds = (int)red_conflict_mask(s); if (prof && conflict_lds) prof[4] += 1000000; }
      __syncthreads();
      const uint32_t serial_red = (uint32_t)conflict_lds;
      CC4_STOP(8);                                                               // + phishing, slot reservation, conflict mask (lane 0)
      uint32_t pre_re[4];
      bank_fetch(BK_REXE, pre_re);                                                // red r (lane r) <- lane BK_REXE + r
      // round 0: the agents whose actions commute, each on its lane; then the same-host actions (everything when some agent withdraws) in agent
      // order on lane 0 -- through the same call site (see P3b)
      {
        uint32_t todo = serial_red;
        bool side = true;
#pragma nounroll
        for (;;) {
          int r = lane;
          bool go = is_red && !((serial_red >> lane) & 1u);
          Ctx xe = xr;
          if (!side) { r = __ffs((int)todo) - 1; todo &= todo - 1u; go = lane == 0; xe.prof = x0.prof; xe.aprof = nullptr; }
          if (go) {
            unsigned long long t0 = xe.aprof ? clock64() : 0;
            if (s->rexec[r].type != RA_NONE) {                                      // == step_red_exec_agent
              rng_set_stream(&rl, ST_RED_EXE + (uint32_t)r);
              if (side) rng_preload(&rl, pre_re);
              red_execute(xe, r, s->rexec[r]);
            }
            if (xe.aprof) xe.aprof[1] += clock64() - t0;
          }
          __syncthreads();
          side = false;
          if (!todo) break;
        }
      }
      CC4_STOP(9);                                                               // + red actions
      if (lane == 0) {
        step_red_merge(x0);
        CC4_TICK(x0, 7);
        step_reassign(x0, red_foreign_agents(s));
        CC4_TICK(x0, 8);
      }
      // P7 end-turn Monitor roll-over: the hosts' event bytes are part of the staged row (EnvState.hev).  (Lane 0's reassignment above
      // moves sessions, not events.)
      CC4_STOP(10);                                                              // + merge, reassignment (lane 0)
      monitor_roll_all(s, lane);
      __syncthreads();
      CC4_TICK(x0, 9);
      CC4_STOP(11);                                                              // + Monitor roll-over
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
  CC4_STOP(12);                                                                  // + RedSessionCheck, Monitor hand-over, end of step
  unsigned long long t_obs = a.prof ? clock64() : 0;
  {
    int32_t* o = a.obs + (size_t)e * OBS_TOTAL;
    // the slowly varying values by kind (env_flat_obs_sorted's enumeration: 63 blocked bits, 63 comms-policy bits, 63 subnet one-hots, 5 phase words --
    // one pass of the wave each): what the step changed (EnvState.obs_dirty), everything after a reset or when the caller's buffer is new
    const uint32_t dirty = (do_reset || (a.full_obs && (!PERSIST || item_k == 0))) ? (uint32_t)OD_ALL : (uint32_t)s->obs_dirty;
    encode_obs_fast<WAVE>(s, o, nullptr, false, lane, msgs_clean && !do_reset && dirty != (uint32_t)OD_ALL);
    encode_obs_slow(s, o, dirty, lane);
  }
  __syncthreads();
  CC4_STOP(13);                                                                  // + observation encode; 0 = the whole step (row written back)
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

