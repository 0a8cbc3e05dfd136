// cc4_k_philox1.hip -- counter mode, one wavefront per episode: k_step_philox1<LOG> and the multi-step k_run_philox1m.  See cc4_kernels.h.
#include "cc4_philox1_body.h"

#ifndef CC4_LEAN_MINW
#define CC4_LEAN_MINW 1
#endif
template <bool LOG>
__global__ __launch_bounds__(WAVE, CC4_LEAN_MINW) void k_step_philox1(StepArgs a) {
  const int e = a.e0 + (int)blockIdx.x;
  if (e >= a.n) return;
  philox1_body<LOG, false>(a, e, a.rand_t, 0u, (int)threadIdx.x);
}
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

// the kernels the host side launches (cc4_kernel_decls.h)
template __global__ void k_step_philox1<false>(StepArgs);
template __global__ void k_step_philox1<true>(StepArgs);
