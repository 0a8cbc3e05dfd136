// cc4_k_run1x.hip -- the persistent kernel's other builds: k_run_philox1x (five waves per SIMD: beside RCCL's kernels) and k_run_philox1r (rollouts with the
// policy in the loop: the actions-in protocol, cc4_rollout_begin).  Same schedule (cc4_persist.h) and step body (cc4_philox1_body.h) as k_run_philox1.
#include "cc4_philox1_body.h"
#include "cc4_persist.h"

// The same kernel at five waves per SIMD, for handles with a communicator: the all-gathers of the exchange run BESIDE this kernel, and RCCL's kernels
// need more registers than the 32 per SIMD six 80-register waves leave over (r06: with the six-wave build a 40-step call sat in its slab waits until
// the watchdog fired -- the all-gather never found a SIMD to run on).  One wave per CU less (cc4_comm_init) then leaves a SIMD at four waves.
__global__ __launch_bounds__(WAVE, 5) void k_run_philox1x(StepArgs a, RunArgs ra, XchgArgs x) { persist_loop<false>(a, ra, x); }
// the rollout build: waits per step and policy group for the caller's publishes, reads the action slots with system-scope loads, counts every step's
// packed observation row for the caller's gates (RunArgs.act_ready ..)
__global__ __launch_bounds__(WAVE, 6) void k_run_philox1r(StepArgs a, RunArgs ra, XchgArgs x) { persist_loop<false, true>(a, ra, x); }
