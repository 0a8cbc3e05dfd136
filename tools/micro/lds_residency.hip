// How many single-wave workgroups with X bytes of LDS are REALLY resident on a CU at once (MI355X: 160 KB per CU)?  The occupancy query
// divides 163 840 by X; the hardware allocates in granules and may keep some for itself.  Every wave bumps a per-CU counter on arrival,
// records the highest value it sees, idles, and drops the counter when it leaves.  build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int SLOTS = 2048;
__device__ __forceinline__ int cu_slot() {
  const uint32_t hw = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4);
  const uint32_t xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
  return (int)(((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu));
}
__global__ __launch_bounds__(64) void k_probe(int* resident, int* peak, int* total, long long ticks) {
  extern __shared__ uint4 lds[];
  if (threadIdx.x == 1) reinterpret_cast<volatile uint32_t*>(lds)[0] = 0;
  if (threadIdx.x == 0) {
    const int s = cu_slot();
    const int now = atomicAdd(&resident[s], 1) + 1;
    atomicMax(&peak[s], now);
    atomicAdd(&total[s], 1);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    atomicSub(&resident[s], 1);
  }
}
int main(int argc, char** argv) {
  int cus = 256; hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); cus = p.multiProcessorCount;
  int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  int *res, *peak, *tot;
  hipMalloc(&res, SLOTS * 4); hipMalloc(&peak, SLOTS * 4); hipMalloc(&tot, SLOTS * 4);
  std::vector<int> hp(SLOTS), ht(SLOTS);
  printf("%d CUs, LDS per CU by the device properties: %zu B (maxSharedMemoryPerMultiProcessor), per block %zu B\n", cus, (size_t)p.maxSharedMemoryPerMultiProcessor, (size_t)p.sharedMemPerBlock);
  const int sizes[] = {6144, 7168, 7616, 7680, 7700, 7936, 8000, 8064, 8128, 8176, 8192, 8208, 8320, 8448, 8608, 8624, 8640, 8704, 8752, 8960, 9216, 10240, 16384, 19616, 19744};
  for (int X : sizes) {
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_probe, 64, (size_t)X);
    for (int grid_per_cu : {occ, 32}) {
      hipMemset(res, 0, SLOTS * 4); hipMemset(peak, 0, SLOTS * 4); hipMemset(tot, 0, SLOTS * 4);
      hipLaunchKernelGGL(k_probe, dim3(grid_per_cu * cus), dim3(64), (size_t)X, 0, res, peak, tot, 300LL * khz / 1000);
      hipDeviceSynchronize();
      hipMemcpy(hp.data(), peak, SLOTS * 4, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), tot, SLOTS * 4, hipMemcpyDeviceToHost);
      int seen = 0, pmax = 0, pmin = 1 << 30, tmax = 0;
      for (int i = 0; i < SLOTS; ++i) if (ht[i]) { ++seen; pmax = hp[i] > pmax ? hp[i] : pmax; pmin = hp[i] < pmin ? hp[i] : pmin; tmax = ht[i] > tmax ? ht[i] : tmax; }
      printf("LDS %5d B: occupancy query %2d per CU; grid %2d per CU -> %d CUs seen, resident at once per CU: %d .. %d, most waves one CU took over the launch: %d\n", X, occ, grid_per_cu, seen, pmin, pmax, tmax);
    }
  }
  return 0;
}
