// Does the latency of a memory-side access depend on WHICH XCD asks for WHICH page?  (r05: with equal shares the waves of XCDs 0-3 ran dry 45-70 us later
// than those of XCDs 4-7 in every 20-step persistent launch, and the picture flipped when the episodes were dealt to the CUs differently:
// profiles/r05_xcd_balance.txt.)  One wave per XCD at a time walks a chain of dependent agent-scope atomic loads (they bypass the XCD's L2, like the ticket /
// progress words of the persistent kernel) over the words of ONE 4 KB page; repeated for a sample of pages of a 256 MB buffer.  Output: per XCD the
// mean / min / max latency over the pages, and the pages' latency pattern (is there a near and a far half?).
// build: hipcc --offload-arch=gfx950 -O2 -o xcd_page_latency xcd_page_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(64) void k_probe(const uint32_t* buf, int pages, size_t page_stride_words, int want_xcd, int iters, unsigned long long* out, int* done) {
  const uint32_t xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;
  if ((int)xcc != want_xcd) return;
  if (threadIdx.x != 0) return;
  if (atomicAdd(done, 1) != 0) return;            // one wave of that XCD does the walk
  for (int p = 0; p < pages; ++p) {
    const uint32_t* page = buf + (size_t)p * page_stride_words;
    uint32_t idx = 0;
    // warm the TLB for this page
    idx = __hip_atomic_load(page + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) idx = __hip_atomic_load(page + (idx & 1023u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // dependent chain inside the page
    const long long t1 = wall_clock64();
    out[p] = (unsigned long long)(t1 - t0) + (idx == 0xFFFFFFFFu ? 1 : 0);
  }
}

int main(int argc, char** argv) {
  const int pages = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 256;
  const size_t total = 256u << 20, stride_words = total / 4 / pages;
  uint32_t* buf; hipMalloc(&buf, total);
  std::vector<uint32_t> h(total / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)((i * 97u + 13u) & 1023u);       // the chain's next index, within a 4 KB page
  hipMemcpy(buf, h.data(), total, hipMemcpyHostToDevice);
  unsigned long long* out; hipMalloc(&out, pages * 8);
  int* done; hipMalloc(&done, 4);
  int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  std::vector<std::vector<double>> lat(8, std::vector<double>(pages, 0.0));
  for (int x = 0; x < 8; ++x) {
    hipMemset(done, 0, 4); hipMemset(out, 0, pages * 8);
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, buf, pages, stride_words, x, iters, out, done);
    hipDeviceSynchronize();
    std::vector<unsigned long long> o(pages);
    hipMemcpy(o.data(), out, pages * 8, hipMemcpyDeviceToHost);
    for (int p = 0; p < pages; ++p) lat[x][p] = (double)o[p] * 1e6 / khz / iters;      // ns per dependent load
  }
  printf("%d pages of a 256 MB buffer (every %zu KB), %d dependent agent-scope loads per page; ns per load\n", pages, stride_words * 4 / 1024, iters);
  for (int x = 0; x < 8; ++x) {
    std::vector<double> s = lat[x]; std::sort(s.begin(), s.end());
    double m = 0; for (double v : s) m += v; m /= pages;
    printf("XCD %d: mean %.0f  min %.0f  10%% %.0f  median %.0f  90%% %.0f  max %.0f\n", x, m, s[0], s[pages / 10], s[pages / 2], s[pages * 9 / 10], s[pages - 1]);
  }
  // do the XCDs agree on which pages are near?  correlation of the per-page latency between XCD 0 and the others
  auto corr = [&](int a, int b) { double ma = 0, mb = 0; for (int p = 0; p < pages; ++p) { ma += lat[a][p]; mb += lat[b][p]; } ma /= pages; mb /= pages;
    double sab = 0, saa = 0, sbb = 0; for (int p = 0; p < pages; ++p) { sab += (lat[a][p] - ma) * (lat[b][p] - mb); saa += (lat[a][p] - ma) * (lat[a][p] - ma); sbb += (lat[b][p] - mb) * (lat[b][p] - mb); }
    return sab / (sqrt(saa * sbb) + 1e-30); };
  printf("correlation of the per-page latency, XCD 0 against XCD 1..7:");
  for (int x = 1; x < 8; ++x) printf(" %.2f", corr(0, x));
  printf("\nfirst 16 pages, ns per load by XCD:\n");
  for (int p = 0; p < 16 && p < pages; ++p) { printf("  page %3d:", p); for (int x = 0; x < 8; ++x) printf(" %5.0f", lat[x][p]); printf("\n"); }
  return 0;
}
