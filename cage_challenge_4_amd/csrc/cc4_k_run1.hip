// cc4_k_run1.hip -- the persistent kernel of large counter-mode batches: k_run_philox1 (six waves per SIMD) and k_run_philox1x (five: beside RCCL).
#include "cc4_philox1_body.h"
#include "cc4_persist.h"

// register budget of the persistent counter-mode kernel in waves per SIMD: 6 (80 VGPRs, eight spilled: 24 waves per CU -- by registers and,
// since r06's 5952-byte agent part made a wave FIVE 1280-byte LDS granules, by LDS as well: 25) or 5 (93 VGPRs, nothing spilled: 20 per CU).
// Measured, 8192 episodes, one box (profiles/r06_layout_ab.txt): K = 500: 989-990 vs 914-915 M, K = 20: 831-846 vs 797-807 M.  (r05, when LDS
// capped a CU at 21 waves: 952 vs 939-948 M -- inside the box-to-box spread.)
#ifndef CC4_PERSIST_MINW
#define CC4_PERSIST_MINW 6
#endif
__global__ __launch_bounds__(WAVE, CC4_PERSIST_MINW) void k_run_philox1(StepArgs a, RunArgs ra, XchgArgs x) { persist_loop<false, false, false>(a, ra, x); }      // (no exchange, no rollout protocol in this build)
