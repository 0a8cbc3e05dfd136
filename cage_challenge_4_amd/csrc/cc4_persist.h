// cc4_persist.h -- the schedule of the persistent kernels (k_run_philox1 / k_run_philox1x / k_run_pcg): one wave per residency slot pulling runs of
// steps of episodes from its CU's partition.  See cc4_kernels.h (RunArgs) and DESIGN 3.3.
#pragma once
#include "cc4_kernels.h"

// The schedule (RunArgs).  The batch is cut into one partition per CU (episode e -> partition e % P); a partition's tickets hand out RUNS of
// consecutive steps of its episodes in step-major order; a run of episode e may start once progress[e] says the steps before it are done.  A wave
// normally serves its own CU's partition -- an episode then stays on one CU, whose waves share a write-through L1: no cache maintenance -- but it
// looks at the ticket counters of its XCD's partitions before every run and takes the run from the partition that lags most when its own is more
// than `thr` tickets ahead of it, or handed out.  The progress word carries the CU that ran the episode's last run: a run on ANOTHER CU starts with
// an agent-scope acquire (buffer_inv sc1; nothing less drops stale L1 lines: tools/micro/l1_inv_scope.hip).  Never across XCDs: their L2s do not
// agree without a write-back.  The hand-over between two waves: the writer drains its stores (s_waitcnt vmcnt(0): the L1 is write-through, a drained
// store is in the XCD's L2), then publishes the progress word; the reader reads the word, then the rows.
// ROLLOUT: the build that serves cc4_rollout_begin (k_run_philox1r) -- the actions-in protocol is compiled into that kernel only (in the others its code
// cost the headline kernel 30 more spilled registers)
// XCHG: the build serves the exchange (the packed rows of every step into the slab ring, counted for the communication stream's gates).  The
// headline kernel k_run_philox1 is built without it: handles with a communicator launch k_run_philox1x.
template <bool PCG, bool ROLLOUT = false, bool XCHG = true>
__device__ __forceinline__ void persist_loop(StepArgs a, RunArgs ra, const XchgArgs x) {
  // (the item travels from lane 0 to the wave through v_readfirstlane, not through LDS)
  const int lane = threadIdx.x;
  const int my_slot = cu_slot();
  a.prof = nullptr; a.obs8 = nullptr; a.ext = nullptr;
  uint32_t seen_gathered = 0;
  unsigned long long tl_first = 0, tl_last = 0, tl_items = 0;
  const unsigned long long tl_entry = ra.timeline ? wall_clock64() : 0;
  auto tl_flush = [&]() { if (ra.timeline && lane == 0) { unsigned long long* t = ra.timeline + 4 * (size_t)blockIdx.x; t[0] = tl_entry; t[1] = tl_first; t[2] = tl_last; t[3] = tl_items | ((unsigned long long)(my_slot + 1) << 32); } };
  int pend_e = -1; uint32_t pend_k = 0;  // the exchange: the item whose packed row this wave stored last and has not counted yet (its store drains with the next item)
  auto flush_pending = [&]() {
    if (XCHG && x.slab && pend_e >= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) xchg_count(x, pend_k, pend_e % ra.G); pend_e = -1; }
  };
  const uint32_t my_xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;    // HW_REG_XCC_ID
  const int xlo = ra.xcc_lo[my_xcc], xn = ra.xcc_n[my_xcc];     // this XCD's partitions
  // the CU's own partition, from the table of the compute units this device showed at first use (-1: a CU that is not in it only helps out), and
  // its id in the progress words
  const int own = ra.slot_part[my_slot] - 1;
  const uint32_t my_id = own >= 0 ? (uint32_t)own + 1u : 511u;
  if (xn <= 0) { tl_flush(); return; }
  for (;;) {
    int res_e = -5, res_k = 0, res_sh = 0;                            // -5: look again, -4: leave
    if (ROLLOUT && ra.act_ready) {
      // ---- a rollout: every (partition, policy group) has a ticket counter of its own (words 0 .. PG-1 of the partition's ticket line), and a wave
      // only ever draws a ticket of a group whose NEXT step is published -- it never holds a ticket it cannot run.  (With ONE step-major sequence over
      // all groups the waves piled up on tickets of unpublished passes while the published group's next tickets lay further down the sequence: the
      // groups advanced in lock step, 114 us per step whatever their number -- profiles/r06_rollout.txt.)  A CU serves its own partition only.
      if (own < 0) { flush_pending(); tl_flush(); return; }
      if (lane == 0) {
        const int ne = (a.n - own + ra.P - 1) / ra.P;
        const uint32_t* rdy = ra.act_ready + (size_t)own * 32;
        uint32_t* tkl = ra.ticket + (size_t)own * TK_STRIDE;
        int naps = 1;
        const long long w0 = wall_clock64();
        res_e = -4;
        for (;;) {
          int best_g = -1; uint32_t best_j = 0xFFFFFFFFu; bool left = false;
          for (int g = 0; g < ra.PG; ++g) {
            const int ng = (ne - g + ra.PG - 1) / ra.PG;
            if (ng <= 0) continue;
            const uint32_t t = __hip_atomic_load(tkl + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t >= (uint32_t)(ng * ra.K)) continue;
            left = true;
            const uint32_t j = t / (uint32_t)ng;
            if (j < best_j && __hip_atomic_load(rdy + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) > j) { best_j = j; best_g = g; }
          }
          if (!left) break;                                           // every group of this partition is handed out: leave
          if (best_g >= 0) {
            const int ng = (ne - best_g + ra.PG - 1) / ra.PG;
            const uint32_t t = __hip_atomic_fetch_add(tkl + best_g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1u == (uint32_t)(ng * ra.K)) __hip_atomic_store(ra.ticket_next + (size_t)own * TK_STRIDE + best_g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < (uint32_t)(ng * ra.K)) {
              const int j = (int)(t / (uint32_t)ng), ee = own + ((int)(t % (uint32_t)ng) * ra.PG + best_g) * ra.P;
              uint32_t w;
              while ((((w = __hip_atomic_load(&ra.progress[ee], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & PG_STEPS) - ra.base) < (uint32_t)j) __builtin_amdgcn_s_sleep(8);
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
              rollout_wait_actions(ra, x, own, best_g, (uint32_t)j);   // (another wave may have drawn the last published ticket in between: then this one is of the next step)
              const uint32_t last = w >> 23;
              res_e = ee; res_k = j; res_sh = (last != 0u && last != my_id) ? 1 : 0;
              break;
            }
            continue;
          }
          // nothing is published that this partition has not handed out: wait (a growing nap, the watchdog of the action waits)
          for (int q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(32);
          if (naps < 4) naps <<= 1;
          if (__hip_atomic_load(x.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) || wall_clock64() - w0 > ra.act_wait_ticks) {
            __hip_atomic_store(x.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(x.timeout_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // give up on the policy: run what is left with whatever the slots hold (cc4_rollout_end reports it)
            for (int g = 0; g < ra.PG && res_e == -4; ++g) {
              const int ng = (ne - g + ra.PG - 1) / ra.PG;
              if (ng <= 0) continue;
              const uint32_t t = __hip_atomic_fetch_add(tkl + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (t + 1u == (uint32_t)(ng * ra.K)) __hip_atomic_store(ra.ticket_next + (size_t)own * TK_STRIDE + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (t < (uint32_t)(ng * ra.K)) {
                const int j = (int)(t / (uint32_t)ng), ee = own + ((int)(t % (uint32_t)ng) * ra.PG + g) * ra.P;
                while (((__hip_atomic_load(&ra.progress[ee], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & PG_STEPS) - ra.base) < (uint32_t)j) __builtin_amdgcn_s_sleep(8);
                res_e = ee; res_k = j; res_sh = 1;
              }
            }
            if (res_e != -4) break;
          }
        }
      }
    } else {
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
          const int ee = tp + (int)(t % (uint32_t)ne) * ra.P;
          int k, len; run_span(ra, j, k, len);
          uint32_t w;
          while ((((w = __hip_atomic_load(&ra.progress[ee], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & PG_STEPS) - ra.base) < (uint32_t)k) __builtin_amdgcn_s_sleep(8);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (XCHG && x.slab) xchg_wait_slab(x, (uint32_t)k, seen_gathered);
          const uint32_t last = w >> 23;
          res_e = ee; res_k = j; res_sh = (last != 0u && last != my_id) ? 1 : 0;     // the episode's last run was on another CU: its lines in this CU's L1 may be stale
        }
      }
    }
    const int e = __builtin_amdgcn_readfirstlane(res_e);              // (all lanes are active here: the first active lane is lane 0)
    if (e == -4) { flush_pending(); tl_flush(); return; }
    if (e == -5) continue;
    int run_k0, run_len;
    run_span(ra, __builtin_amdgcn_readfirstlane(res_k), run_k0, run_len);
    const int shared = __builtin_amdgcn_readfirstlane(res_sh);
    if (ra.timeline && !tl_items) tl_first = wall_clock64();
    if (shared || ra.order >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (buffer_inv sc1: the CU's L1 dropped.  buffer_inv sc0 does NOT drop it: profiles/r06_l1_inv_scope.txt)
    uint32_t item_k = (uint32_t)run_k0;
    for (int q = 0; q < run_len; ++q, ++item_k) {
    if (q > 0) {
      if (XCHG && x.slab) {
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
    if (ROLLOUT && ra.act_ready) { a.actions = ra.act + (size_t)(item_k & 1u) * (size_t)a.n * NBLUE; a.rand_out = nullptr; a.act_sys = 1; }
    if constexpr (PCG) {
      StepArgs b = a;
      b.rand_t = ra.t0 + item_k; b.full_obs = (a.full_obs && item_k == 0) ? 1 : 0;
      if (XCHG && x.slab) b.obs8 = x.slab + (size_t)(item_k % (uint32_t)x.ring) * (size_t)a.n * OBS_PACKED;
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
    if (XCHG && x.slab) {
      if constexpr (PCG) {     // (the numpy-stream body stored the row itself, from its LDS byte row: drained by the fence above)
        if (lane == 0) xchg_count(x, item_k, e % ra.G);
      } else {
        // the row this wave stored with its PREVIOUS item is in memory (this item's fence drained it): counted.  Then this episode's row of
        // step item_k, read back from the int32 row before the episode's next step may touch it (the loads feed the store, the store is
        // issued ahead of the progress word) -- not waited for: it drains with the wave's next item, or when the wave leaves.
        if (lane == 0 && pend_e >= 0) xchg_count(x, pend_k, pend_e % ra.G);
        pack_row_from_obs(x.slab + ((size_t)(item_k % (uint32_t)x.ring) * (size_t)a.n + (size_t)e) * OBS_PACKED, a.obs + (size_t)e * OBS_TOTAL, lane);
        pend_e = e; pend_k = item_k;
        if (ROLLOUT && ra.act_ready) {
          // a rollout: the caller's next policy pass waits for this count -- not deferred to the wave's next item
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) xchg_count(x, item_k, (e % ra.P) * ra.PG + (e / ra.P) % ra.PG);
          pend_e = -1;
        }
      }
    }
    if (lane == 0) __hip_atomic_store(&ra.progress[e], (ra.base + item_k + 1u) | (my_id << 23), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ra.timeline) { tl_last = wall_clock64(); ++tl_items; }
  }
}
