// Prototype of the per-step hand-off out of a one-launch kernel: a running kernel writes, per step k, a slab of packed rows
// into a ring and counts the rows it finished (done[k]); a second stream waits for done[k] == rows with hipStreamWaitValue32, copies the
// slab (what ncclAllGather of a one-rank communicator does), and publishes gathered = k + 1 with hipStreamWriteValue32; the kernel's step
// k + RING waits for gathered >= k + 1 before it overwrites the slab.  What is measured: (1) is every copied slab the one the kernel
// wrote -- with plain stores, and with system-scope (sc0 sc1, write-through) stores -- although the writing XCD's L2 is not the
// reader's; (2) what the hand-off costs a step of S microseconds; (3) the latency flag -> value written back.
// build: hipcc --offload-arch=gfx950 -O2 -o ring_protocol ring_protocol.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int WORDS = 37;          // packed observation row: 148 bytes
__global__ __launch_bounds__(64) void k_producer(uint32_t* slab, uint32_t* done, uint32_t* gathered, uint32_t* timeouts, long long* lat,
                                                 int rows_per_block, int rows, int K, int ring, long long ticks, int sys_stores, int exchange) {
  const int lane = threadIdx.x;
  for (int k = 0; k < K; ++k) {
    for (int r = 0; r < rows_per_block; ++r) {
      const int row = blockIdx.x * rows_per_block + r;
      long long t0 = wall_clock64();
      while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);        // the step itself
      if (exchange && k >= ring) {                                           // the slab's previous occupant must have been gathered
        long long w0 = wall_clock64();
        while (__hip_atomic_load(gathered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (uint32_t)(k - ring + 1)) {
          __builtin_amdgcn_s_sleep(16);
          if (wall_clock64() - w0 > 200000000LL) { if (lane == 0) atomicAdd(timeouts, 1u); break; }     // 2 s: give up, never hang
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      uint32_t* dst = slab + ((size_t)(k % ring) * rows + row) * WORDS;
      if (lane < WORDS) {
        const uint32_t v = (uint32_t)k * 2654435761u ^ (uint32_t)(row * WORDS + lane);
        if (sys_stores) __hip_atomic_store(dst + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else dst[lane] = v;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0 && exchange) {
        const uint32_t before = __hip_atomic_fetch_add(done + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (before + 1 == (uint32_t)rows && lat) lat[k] = wall_clock64();     // the moment step k was complete
      }
    }
  }
}
__global__ void k_stamp(long long* out, int k) { out[k] = wall_clock64(); }
__global__ void k_copy(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t nv) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024, rpb = argc > 2 ? atoi(argv[2]) : 1, K = argc > 3 ? atoi(argv[3]) : 200;
  const double step_us = argc > 4 ? atof(argv[4]) : 15.0;
  const int ring = argc > 5 ? atoi(argv[5]) : 8;
  const int copy_mode = argc > 6 ? atoi(argv[6]) : 1;     // 0: no copy (wait + write only), 1: hipMemcpyAsync, 2: a copy kernel of 32 x 256 threads
  const int leave = argc > 7 ? atoi(argv[7]) : 0;          // LDS bytes per producer block (bounds how many are resident per CU: 8192 -> 20)
  int can = 0; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  if (!can) { printf("hipDeviceAttributeCanUseStreamWaitValue = 0\n"); return 0; }
  int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  const int rows = blocks * rpb;
  const size_t slab_words = (size_t)rows * WORDS;
  uint32_t *slab, *done, *gathered, *timeouts, *dst;
  long long *lat, *stamp;
  hipMalloc(&slab, ring * slab_words * 4); hipMalloc(&done, K * 4); hipMalloc(&gathered, 4); hipMalloc(&timeouts, 4);
  hipMalloc(&dst, (size_t)K * slab_words * 4); hipMalloc(&lat, K * 8); hipMalloc(&stamp, K * 8);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  std::vector<uint32_t> host((size_t)K * slab_words);
  std::vector<long long> hl(K), hs(K);
  printf("%d rows (%d blocks x %d), %d steps of %.1f us, ring %d, copy mode %d, %d B of LDS per block, wall clock %d kHz\n", rows, blocks, rpb, K, step_us, ring, copy_mode, leave, khz);
  for (int mode = 0; mode < 4; ++mode) {      // 0: no exchange (the kernel alone), 1: plain stores, 2: system-scope stores, 3: system-scope stores + a stamp kernel per step
    const int exchange = mode > 0, sys = mode >= 2;
    hipMemset(done, 0, K * 4); hipMemset(gathered, 0, 4); hipMemset(timeouts, 0, 4); hipMemset(dst, 0xEE, (size_t)K * slab_words * 4);
    hipMemset(slab, 0xDD, ring * slab_words * 4); hipMemset(lat, 0, K * 8); hipMemset(stamp, 0, K * 8);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_producer, dim3(blocks), dim3(64), (size_t)leave, a, slab, done, gathered, timeouts, lat, rpb, rows, K, ring,
                       (long long)(step_us * khz / 1000.0), sys, exchange);
    if (exchange) for (int k = 0; k < K; ++k) {
      if (hipStreamWaitValue32(b, done + k, (uint32_t)rows, hipStreamWaitValueGte, 0xFFFFFFFFu) != hipSuccess) { printf("wait value failed\n"); return 1; }
      if (copy_mode == 1) hipMemcpyAsync(dst + (size_t)k * slab_words, slab + (size_t)(k % ring) * slab_words, slab_words * 4, hipMemcpyDeviceToDevice, b);
      else if (copy_mode == 2) hipLaunchKernelGGL(k_copy, dim3(32), dim3(256), 0, b, reinterpret_cast<uint4*>(dst + (size_t)k * slab_words),
                                                  reinterpret_cast<const uint4*>(slab + (size_t)(k % ring) * slab_words), slab_words / 4);
      if (mode == 3) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, b, stamp, k);
      if (hipStreamWriteValue32(b, gathered, (uint32_t)(k + 1), 0) != hipSuccess) { printf("write value failed\n"); return 1; }
    }
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(a); hipStreamSynchronize(b);
    auto t2 = std::chrono::steady_clock::now();
    uint32_t to = 0; hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost);
    size_t bad = 0; int first_bad = -1;
    if (exchange && copy_mode) {
      hipMemcpy(host.data(), dst, host.size() * 4, hipMemcpyDeviceToHost);
      for (int k = 0; k < K; ++k) for (size_t i = 0; i < slab_words; ++i)
        if (host[(size_t)k * slab_words + i] != ((uint32_t)k * 2654435761u ^ (uint32_t)i)) { ++bad; if (first_bad < 0) first_bad = k; }
    }
    double lat_us = 0; int nl = 0;
    if (mode == 3) {
      hipMemcpy(hl.data(), lat, K * 8, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), stamp, K * 8, hipMemcpyDeviceToHost);
      for (int k = 0; k < K; ++k) if (hl[k] && hs[k]) { lat_us += (double)(hs[k] - hl[k]) * 1000.0 / khz; ++nl; }
    }
    const double enq = std::chrono::duration<double, std::micro>(t1 - t0).count(), all = std::chrono::duration<double, std::micro>(t2 - t0).count();
    printf("mode %d (%s): %.1f us per step (%.0f us in all; host enqueue %.0f us), wrong words %zu (first bad step %d), watchdog timeouts %u",
           mode, mode == 0 ? "kernel alone" : mode == 1 ? "exchange, plain stores" : mode == 2 ? "exchange, system-scope stores" : "exchange, system-scope stores, stamp kernel",
           all / K, all, enq, bad, first_bad, to);
    if (nl) printf(", step complete -> stamp kernel behind the copy: %.1f us mean over %d steps", lat_us / nl, nl);
    printf("\n");
  }
  return 0;
}
