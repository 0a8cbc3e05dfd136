// l1_inv_scope.hip -- r06: what does a reader on ANOTHER CU of the same XCD need before it reads rows a writer CU has drained to the XCD's L2?
// A writer wave and a reader wave on two different CUs of one XCD (found through HW_REG_XCC_ID / HW_REG_HW_ID), per XCD.  Round i:
//   reader: plain loads of a 4 KB region (value i - 1; the lines now sit in the reader CU's vector L1), then flag1 = i
//   writer: waits for flag1 == i, stores value i to the region (plain stores), s_waitcnt vmcnt(0), flag2 = i (agent-scope store)
//   reader: waits for flag2 == i (agent-scope loads), then ONE of: nothing | buffer_inv sc0 | buffer_inv sc1 | loads with sc1 | loads with sc0 sc1
//           plain loads of the region again; every lane that still reads i - 1 is a stale read
// Prints stale reads per variant and the cycles one invalidate costs the wave.   hipcc --offload-arch=gfx950 -O2 -o l1_inv_scope l1_inv_scope.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int REGION_WORDS = 1024;     // 4 KB = 32 lines of 128 B
__device__ __forceinline__ int cu_slot() {
  const uint32_t hw = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4);
  const uint32_t xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
  return (int)(((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu));
}
__device__ __forceinline__ uint32_t ld_plain(const uint32_t* p) { uint32_t v; asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) { uint32_t v; asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ uint32_t ld_sc0(const uint32_t* p) { uint32_t v; asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ uint32_t ld_sc01(const uint32_t* p) { uint32_t v; asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_plain(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
// roles[xcd] : [0] writer slot + 1, [1] reader slot + 1; flags[xcd] : [0] flag1, [1] flag2
__global__ __launch_bounds__(64) void k_test(int32_t* roles, uint32_t* flags, uint32_t* data, unsigned long long* stale, unsigned long long* cyc, int variant, int rounds, long long spin_ticks) {
  const int lane = threadIdx.x, slot = cu_slot(), xcd = slot >> 8;
  __shared__ int role_s;
  if (lane == 0) {
    int role = -1, exp = 0;
    if (atomicCAS(&roles[xcd * 2], 0, slot + 1) == 0) role = 0;
    else if (roles[xcd * 2] != slot + 1 && atomicCAS(&roles[xcd * 2 + 1], exp, slot + 1) == 0) role = 1;
    role_s = role;
    if (role < 0) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(64); }   // keep the CU occupied: later blocks land elsewhere
  }
  __syncthreads();
  const int role = role_s;
  if (role < 0) return;
  uint32_t* reg = data + (size_t)xcd * REGION_WORDS;
  uint32_t* f1 = flags + xcd * 64, *f2 = flags + xcd * 64 + 32;
  // wait until both roles of this XCD are taken (or give up)
  { const long long t0 = wall_clock64(); while (__hip_atomic_load(&roles[xcd * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) { if (wall_clock64() - t0 > spin_ticks) return; __builtin_amdgcn_s_sleep(32); } }
  unsigned long long bad = 0, cycles = 0;
  for (int i = 1; i <= rounds; ++i) {
    if (role == 1) {
      uint32_t acc = 0;
      for (int w = lane; w < REGION_WORDS; w += 64) acc += ld_plain(reg + w);          // the region is in this CU's L1 now (value i - 1)
      if (acc == 0xFFFFFFFFu) bad += 1000000;
      __syncthreads();
      if (lane == 0) __hip_atomic_store(f1, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) while (__hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) __builtin_amdgcn_s_sleep(2);
      __syncthreads();
      const unsigned long long c0 = clock64();
      if (variant == 1) asm volatile("buffer_inv sc0" ::: "memory");
      else if (variant == 2) asm volatile("buffer_inv sc1" ::: "memory");
      else if (variant == 6) asm volatile("buffer_inv sc0 sc1" ::: "memory");
      const unsigned long long c1 = clock64();
      cycles += c1 - c0;
      for (int w = lane; w < REGION_WORDS; w += 64) {
        const uint32_t v = variant == 3 ? ld_sc1(reg + w) : variant == 4 ? ld_sc01(reg + w) : variant == 5 ? ld_sc0(reg + w) : ld_plain(reg + w);
        if (v != (uint32_t)i) ++bad;
      }
    } else {
      if (lane == 0) while (__hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) __builtin_amdgcn_s_sleep(2);
      __syncthreads();
      for (int w = lane; w < REGION_WORDS; w += 64) st_plain(reg + w, (uint32_t)i);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (lane == 0) __hip_atomic_store(f2, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (role == 1) { atomicAdd(&stale[xcd], bad); if (lane == 0) cyc[xcd] = cycles; }
}
int main(int argc, char** argv) {
  int khz = 100000; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  int32_t* roles; uint32_t* flags; uint32_t* data; unsigned long long *stale, *cyc;
  CHK(hipMalloc(&roles, 16 * 4)); CHK(hipMalloc(&flags, 8 * 64 * 4)); CHK(hipMalloc(&stale, 8 * 8)); CHK(hipMalloc(&cyc, 8 * 8));
  const int mem = argc > 1 ? atoi(argv[1]) : 0;      // 0 hipMalloc, 1 fine-grained, 2 uncached
  if (mem == 1) CHK(hipExtMallocWithFlags((void**)&data, 8 * REGION_WORDS * 4, hipDeviceMallocFinegrained));
  else if (mem == 2) CHK(hipExtMallocWithFlags((void**)&data, 8 * REGION_WORDS * 4, hipDeviceMallocUncached));
  else CHK(hipMalloc(&data, 8 * REGION_WORDS * 4));
  printf("## data region: %s\n", mem == 1 ? "hipExtMallocWithFlags(hipDeviceMallocFinegrained)" : mem == 2 ? "hipExtMallocWithFlags(hipDeviceMallocUncached)" : "hipMalloc");
  const char* names[] = {"nothing", "buffer_inv sc0", "buffer_inv sc1", "loads with sc1", "loads with sc0 sc1", "loads with sc0", "buffer_inv sc0 sc1"};
  const int rounds = 2000;
  for (int variant = 0; variant < 7; ++variant) {
    CHK(hipMemset(roles, 0, 64)); CHK(hipMemset(flags, 0, 8 * 64 * 4)); CHK(hipMemset(data, 0, 8 * REGION_WORDS * 4)); CHK(hipMemset(stale, 0, 64)); CHK(hipMemset(cyc, 0, 64));
    hipLaunchKernelGGL(k_test, dim3(1024), dim3(64), 0, 0, roles, flags, data, stale, cyc, variant, rounds, 20LL * khz);    // bystanders idle 20 ms
    CHK(hipDeviceSynchronize());
    unsigned long long hs[8], hc[8]; int32_t hr[16];
    CHK(hipMemcpy(hs, stale, 64, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hr, roles, 64, hipMemcpyDeviceToHost));
    unsigned long long tot = 0, tc = 0; int pairs = 0;
    for (int x = 0; x < 8; ++x) if (hr[2 * x] && hr[2 * x + 1]) { ++pairs; tot += hs[x]; tc += hc[x]; }
    printf("%-20s: %d writer/reader pairs (one per XCD, two CUs each), %d rounds x %d words: stale reads %llu (%.4f %%), invalidate %.0f cycles per round\n",
           names[variant], pairs, rounds, REGION_WORDS, tot, pairs ? 100.0 * tot / ((double)pairs * rounds * REGION_WORDS) : 0.0, pairs ? (double)tc / pairs / rounds : 0.0);
  }
  return 0;
}
