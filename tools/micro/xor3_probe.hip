// r06 micro: does gfx950 execute a three-input xor: v_bitop3_b32 with truth table 0x96 (v_xor3_b32 is not in this ISA; the compiler emits two v_xor_b32 for a ^ b ^ c)?  And what does a
// dependent chain of v_mad_u64_u32 cost against v_mul_lo_u32 + v_mul_hi_u32 (one Philox round needs both halves of two products)?
//   hipcc --offload-arch=gfx950 -O2 -o xor3_probe xor3_probe.hip && ./xor3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_xor3(const unsigned* in, unsigned* out) {
  const unsigned a = in[threadIdx.x], b = in[threadIdx.x + 64], c = in[threadIdx.x + 128];
  unsigned d;
  asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  out[threadIdx.x] = d;
}
template <int MODE> __global__ void k_chain(unsigned* out, int n, unsigned seed) {
  unsigned c0 = threadIdx.x + seed, c1 = c0 * 3u, c2 = c0 * 5u, c3 = c0 * 7u, k0 = seed, k1 = seed * 9u;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned n0, n2;
    if (MODE == 0) { n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0; n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1; }
    else {
      asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"((unsigned)(p1 >> 32)), "v"(c1), "s"(k0));
      asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"((unsigned)(p0 >> 32)), "v"(c3), "s"(k1));
    }
    c0 = n0; c1 = (unsigned)p1; c2 = n2; c3 = (unsigned)p0; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  long long t1 = clock64();
  out[threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;
  if (threadIdx.x == 0) out[64] = (unsigned)(t1 - t0);
}
int main() {
  unsigned *in, *out;
  hipMalloc(&in, 192 * 4); hipMalloc(&out, 65 * 4);
  std::vector<unsigned> h(192), r(65);
  for (int i = 0; i < 192; ++i) h[i] = 0x9E3779B9u * (i + 1);
  hipMemcpy(in, h.data(), 192 * 4, hipMemcpyHostToDevice);
  k_xor3<<<1, 64>>>(in, out);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(r.data(), out, 64 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) bad += r[i] != (h[i] ^ h[i + 64] ^ h[i + 128]);
  printf("v_bitop3_b32 0x96: %s, %d of 64 lanes wrong\n", hipGetErrorString(e), bad);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) { if (mode == 0) k_chain<0><<<1, 64>>>(out, 1000, 12345u); else k_chain<1><<<1, 64>>>(out, 1000, 12345u); hipDeviceSynchronize(); }
    hipMemcpy(r.data(), out, 65 * 4, hipMemcpyDeviceToHost);
    printf("1000 Philox rounds, lone wave, %s: %u clock64 ticks, check %08x\n", mode ? "v_bitop3_b32" : "two v_xor_b32", r[64], r[0]);
  }
  return 0;
}
