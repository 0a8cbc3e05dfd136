// cc4_kernel_decls.h -- the kernels of libcc4.so as the host side (cc4_api.hip) sees them: declarations only; each is defined -- and, where it is a
// template, explicitly instantiated -- in the translation unit cc4_kernels.h names.
#pragma once
#include "cc4_kernels.h"

constexpr int PT = 4 * WAVE;       // threads per episode block of the four-wave kernels (cc4_k_philox4.hip: PW = 4)
#ifndef CC4_SMALL_MINW
#define CC4_SMALL_MINW 1
#endif
// per-step kernels
template <bool LOG> __global__ void k_step(StepArgs a);
template <bool LOG, int MINW> __global__ void k_step_philox(StepArgs a);
template <bool LOG> __global__ void k_step_philox1(StepArgs a);
extern template __global__ void k_step<false>(StepArgs);
extern template __global__ void k_step<true>(StepArgs);
extern template __global__ void k_step_philox<false, 1>(StepArgs);
extern template __global__ void k_step_philox<false, 7>(StepArgs);
extern template __global__ void k_step_philox<false, 8>(StepArgs);
extern template __global__ void k_step_philox<true, 1>(StepArgs);
extern template __global__ void k_step_philox1<false>(StepArgs);
extern template __global__ void k_step_philox1<true>(StepArgs);
// one-launch kernels
__global__ void k_run_philox(StepArgs a, int K, uint32_t t0, XchgArgs x);
__global__ void k_run_philox8(StepArgs a, int K, uint32_t t0, XchgArgs x);
__global__ void k_run_philox1m(StepArgs a, int K, uint32_t t0, XchgArgs x);
__global__ void k_run_philox1(StepArgs a, RunArgs ra, XchgArgs x);
__global__ void k_run_philox1x(StepArgs a, RunArgs ra, XchgArgs x);
__global__ void k_run_philox1r(StepArgs a, RunArgs ra, XchgArgs x);
__global__ void k_run_pcg(StepArgs a, RunArgs ra, XchgArgs x);
// reset and helpers (cc4_k_misc.hip)
__global__ void k_reset(ResetArgs a);
__global__ void k_xchg_gate(uint32_t* gcnt, int ring, int groups, int n, int P, int k_lo, int k_hi, long long ticks, uint32_t* fail, int max_naps);
__global__ void k_discover(int32_t* count, long long ticks);
__global__ void k_random_actions(int32_t* actions, int n, uint64_t seed0, uint32_t t, int e0 = 0);
__global__ void k_spin(long long cycles);
__global__ void k_unpack_obs(const uint8_t* __restrict__ packed, uint8_t* __restrict__ out, int rows);
__global__ void k_set_seed(EnvState* st, EnvCold* cold, size_t cold_row, const uint64_t* seeds, int n, int rng_mode);
__global__ void k_set_rng_state(EnvState* st, const uint64_t* w, int n);
__global__ void k_rng_state(const EnvState* st, uint64_t* out, int n);
__global__ void k_rollout_gate(uint32_t* cnt, int P, int PG, int ring, int g, int slot, int n, long long ticks, uint32_t* fail);
__global__ void k_rollout_sync(uint32_t* ready, int pub_g, uint32_t pub_val, uint32_t* cnt, int P, int PG, int ring, int g, int slot, int n, long long ticks, uint32_t* fail);
__global__ void k_rollout_random_policy(int32_t* act, int n, int P, int PG, int g, uint64_t seed0, uint32_t t);
__global__ void k_rollout_hash_policy(int32_t* act, const uint8_t* packed, int n, int P, int PG, int g, uint32_t j);
__global__ void k_pack_obs_rows(uint8_t* packed, const int32_t* obs, int n);
__global__ void k_digest(const EnvState* st, const EnvCold* cold, size_t cold_row, const int32_t* obs, const float* reward,
                         const uint8_t* done, const uint32_t* err, const int32_t* actions, uint64_t* out, int n);
extern "C" __global__ void k_set_evlog(EnvCold* cold, size_t row_bytes, int n, uint32_t on);
extern "C" __global__ void k_copy_evlog(EnvCold* dst, const EnvCold* src, size_t row_bytes, int n);
#ifdef CC4_POLICY_PROBE
template <int G> __global__ void k_policy_probe(StepArgs a, unsigned long long* cyc);
#endif
// the numpy-stream kernels' jump table (cc4_k_pcg.hip)
hipError_t cc4_upload_pcg_tables();
