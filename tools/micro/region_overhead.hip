// What a short multi-stream region costs outside its kernels (MI355X): S streams x K spin kernels of D us each, from the first launch call to
// the host seeing everything complete.  Variants: how the host waits, whether timing events ride on the launches.
// build: hipcc --offload-arch=gfx950 -O2 -o region_overhead region_overhead.hip ; run through gpurun
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__global__ void k_spin(long long ticks) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipStream_t st[4]; for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t ev[8]; for (auto& e : ev) hipEventCreate(&e);
  hipEvent_t join[4]; for (auto& e : join) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int S : {1, 2, 4}) for (int K : {1, 20}) for (int D : {5, 50}) for (int variant = 0; variant < 5; ++variant) {
    // 0: hipStreamSynchronize each; 1: same with start/stop events on first/last launch; 2: device-side join into stream 0, sync stream 0;
    // 3: hipDeviceSynchronize; 4: poll hipStreamQuery
    const long long ticks = (long long)D * khz / 1000;
    std::vector<double> t;
    for (int rep = 0; rep < 60; ++rep) {
      hipDeviceSynchronize();
      double t0 = now_us();
      for (int k = 0; k < K; ++k) for (int s = 0; s < S; ++s) {
        if (variant == 1) hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, st[s], k == 0 ? ev[2 * s] : nullptr, k == K - 1 ? ev[2 * s + 1] : nullptr, 0, ticks);
        else hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, st[s], ticks);
      }
      if (variant == 2) { for (int s = 1; s < S; ++s) { hipEventRecord(join[s], st[s]); hipStreamWaitEvent(st[0], join[s], 0); } hipStreamSynchronize(st[0]); }
      else if (variant == 3) hipDeviceSynchronize();
      else if (variant == 4) { for (int s = 0; s < S; ++s) while (hipStreamQuery(st[s]) == hipErrorNotReady) {} }
      else for (int s = 0; s < S; ++s) hipStreamSynchronize(st[s]);
      t.push_back(now_us() - t0);
    }
    std::sort(t.begin(), t.end());
    printf("S=%d K=%2d D=%2dus variant %d: median %.1f us (kernels alone %d us) -> outside %.1f us\n", S, K, D, variant, t[t.size() / 2], K * D, t[t.size() / 2] - K * D);
  }
  return 0;
}
