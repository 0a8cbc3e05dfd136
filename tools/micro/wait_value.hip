// Does a stream wait on a value a RUNNING kernel writes (hipStreamWaitValue32), and can a running kernel see a value a stream writes
// (hipStreamWriteValue32)?  What a per-step exchange around a one-launch kernel would stand on.  build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_producer(volatile uint32_t* flag, volatile uint32_t* back, uint32_t* seen, long long ticks, int steps) {
  for (int k = 0; k < steps; ++k) {
    long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {}
    __threadfence_system();
    __hip_atomic_store((uint32_t*)&flag[k], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (k >= 2) { long long w0 = wall_clock64(); while (__hip_atomic_load((uint32_t*)&back[k - 2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - w0 < 100000000LL) {} seen[k - 2] = back[k - 2]; }
  }
}
int main() {
  int can = 0; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if (!can) return 0;
  const int S = 8;
  uint32_t *flag, *back, *seen, *marks;
  hipMalloc(&flag, S * 4); hipMalloc(&back, S * 4); hipMalloc(&seen, S * 4); hipMalloc(&marks, S * 4);
  hipMemset(flag, 0, S * 4); hipMemset(back, 0, S * 4); hipMemset(seen, 0xFF, S * 4); hipMemset(marks, 0, S * 4);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k_producer, dim3(1), dim3(1), 0, a, flag, back, seen, 200LL * khz / 1000, S);   // a flag every 200 us
  for (int k = 0; k < S; ++k) {
    hipError_t e1 = hipStreamWaitValue32(b, flag + k, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
    hipMemsetAsync(marks + k, 1, 4, b);
    hipError_t e2 = hipStreamWriteValue32(b, back + k, 7 + k, 0);
    if (e1 != hipSuccess || e2 != hipSuccess) { printf("wait/write value failed: %d %d\n", (int)e1, (int)e2); return 0; }
  }
  hipStreamSynchronize(b); hipStreamSynchronize(a);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  uint32_t hs[S]; hipMemcpy(hs, seen, S * 4, hipMemcpyDeviceToHost);
  printf("%d steps of 200 us in %.0f us; values the running kernel saw come back (steps 0..%d):", S, us, S - 3);
  for (int k = 0; k < S - 2; ++k) printf(" %u", hs[k]);
  printf("\n");
  return 0;
}
