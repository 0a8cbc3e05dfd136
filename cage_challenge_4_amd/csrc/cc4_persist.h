// cc4_persist.h -- the schedule of the persistent kernels (k_run_philox1 / k_run_philox1x / k_run_pcg): one wave per residency slot pulling runs of
// steps of episodes from its CU's partition.  See cc4_kernels.h (RunArgs) and DESIGN 3.3.
#pragma once
#include "cc4_kernels.h"

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
