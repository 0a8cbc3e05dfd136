// cc4_rng.h -- random streams of the CC4 step engine (host + gfx950 device, single source).
//
// Mode 0 ("pcg"): bit-exact restatement of the stream the reference consumes,
//   gym.utils.seeding.np_random(seed) == numpy.random.Generator(PCG64(SeedSequence(seed)))
//   (reference: CybORG/env.py:73-77,236-237; numpy 1.26 / 2.x `_pcg64.pyx`, `pcg64.h`,
//   `bit_generator.pyx` SeedSequence, `distributions.c` bounded integers / random_interval).
// Mode 1 ("philox"): Philox4x32-10 keyed by (seed) with counter (draw index, step): the
//   (counter words: draw index, stream id, step, episode) --
//   counter-based stream BASELINE.json's north_star asks for.  Same draw *sites*, different bits.
//
// Every distribution helper below states the numpy routine it restates.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#if defined(CC4_EXP_NO_FORCEINLINE)       // experiment (profiles/r04_compiler_flags_ab.txt): the inliner's own choice
#define CC4_HD __host__ __device__ inline
#else
#define CC4_HD __host__ __device__ __forceinline__
#endif
#define CC4_UNROLL _Pragma("unroll")     // small fixed-trip loops over register arrays must be fully unrolled on the device:
#else                                     // a dynamically indexed local array lives in scratch (global) memory there
#define CC4_HD inline
#define CC4_UNROLL
#endif

namespace cc4 {

CC4_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

struct Rng {
  uint64_t s_hi, s_lo;      // PCG64 128-bit LCG state   | philox: key (s_lo), draw counter (s_hi)
  uint64_t inc_hi, inc_lo;  // PCG64 increment (odd)     | philox: inc_lo = step word
  uint32_t has32;           // numpy pcg64_state.has_uint32
  uint32_t u32;             // numpy pcg64_state.uinteger (buffered high half)
  uint32_t mode;            // 0 pcg, 1 philox, 3 philox on a lane-private generator of a wave kernel (philox4x32_10's vec form; never stored in a row)
  uint32_t ndraw;           // pcg: number of 64-bit advances (diagnostics) | philox: current stream id
  uint64_t buf64;           // philox: third and fourth word of the current block
  uint32_t has64, pad;      // philox: unread words of the current block (philox_next32)
};

// Philox stream ids: every (agent, phase) of a step draws from its own counter stream, so agents can be resolved on
// separate lanes and still reproduce the serial oracle bit for bit.  Counter = (draw#, stream, step, episode).
enum : uint32_t { ST_RESET = 0, ST_BLUE_EXE = 0x100, ST_GREEN_POL = 0x200, ST_GREEN_EXE = 0x300, ST_GREEN_PHISH = 0x400,
                  ST_RED_POL = 0x500, ST_RED_EXE = 0x600, ST_RED_RSC = 0x700,
                  // scenario generation (reset counter words): per-host streams, so hosts can be generated on separate lanes
                  ST_GEN_HOST = 0x800, ST_GEN_REDRAW = 0x900, ST_GEN_SESS = 0xA00,
                  ST_BLUE_POL = 0xB00 /* built-in blue policy (cc4BlueRandomAgent) */ };

// ---- SeedSequence (numpy/random/bit_generator.pyx: SeedSequence.mix_entropy / generate_state) ----
CC4_HD uint32_t ss_hashmix(uint32_t value, uint32_t* hash_const) {
  value ^= *hash_const;
  *hash_const *= 0x931e8875u;  // MULT_A
  value *= *hash_const;
  value ^= value >> 16;
  return value;
}
CC4_HD uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;  // MIX_MULT_L, MIX_MULT_R
  r ^= r >> 16;
  return r;
}
// entropy = non-negative integer seed < 2^64 as little-endian uint32 words (one word minimum).
CC4_HD void seed_sequence_state(uint64_t seed, uint64_t out[4]) {
  uint32_t ent[2];
  int n_ent = 1;
  ent[0] = (uint32_t)seed;
  ent[1] = (uint32_t)(seed >> 32);
  if (ent[1] != 0) n_ent = 2;
  uint32_t pool[4];
  uint32_t hc = 0x43b0d7e5u;  // INIT_A
  for (int i = 0; i < 4; ++i) pool[i] = ss_hashmix(i < n_ent ? ent[i] : 0u, &hc);
  for (int i_src = 0; i_src < 4; ++i_src)
    for (int i_dst = 0; i_dst < 4; ++i_dst)
      if (i_src != i_dst) pool[i_dst] = ss_mix(pool[i_dst], ss_hashmix(pool[i_src], &hc));
  // generate_state(4, uint64) == 8 uint32 words viewed little-endian
  uint32_t hb = 0x8b51f9ddu;  // INIT_B
  uint32_t w[8];
  for (int i = 0; i < 8; ++i) {
    uint32_t d = pool[i & 3];
    d ^= hb;
    hb *= 0x58f38dedu;  // MULT_B
    d *= hb;
    d ^= d >> 16;
    w[i] = d;
  }
  for (int i = 0; i < 4; ++i) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

// ---- PCG64 (numpy/random/src/pcg64/pcg64.h: pcg_setseq_128_*; XSL-RR 128/64) ----
#define CC4_PCG_MULT_HI 0x2360ED051FC65DA4ull
#define CC4_PCG_MULT_LO 0x4385DF649FCCF645ull

CC4_HD void pcg_step(Rng* r) {
  // state = state * MULT + inc  (mod 2^128)
  uint64_t lo = r->s_lo * CC4_PCG_MULT_LO;
  uint64_t hi = mulhi64(r->s_lo, CC4_PCG_MULT_LO) + r->s_hi * CC4_PCG_MULT_LO + r->s_lo * CC4_PCG_MULT_HI;
  uint64_t nlo = lo + r->inc_lo;
  hi += r->inc_hi + (nlo < lo ? 1ull : 0ull);
  r->s_lo = nlo;
  r->s_hi = hi;
}

// ---- Philox4x32-10 (Salmon et al. 2011; constants as in Random123) ----
// vec (device): the words live in vector registers (a lane-private generator, Rng.mode 3).  The step kernels at 24 waves per CU are bound by VALU issue
// slots (profiles/r06_pmc_valu_busy.json: 91 % busy), and a round is two 32x32->64 products and two THREE-input xors: gfx950's v_bitop3_b32 (truth table
// 0x96) does each in one slot, the compiler emits two v_xor_b32 (tools/micro/xor3_probe.hip: same words).  Not for wave-uniform generators: there the
// plain form is scalar code, which an instruction with vector operands would drag onto the VALU.
CC4_HD uint32_t philox_xor3(uint32_t a, uint32_t b, uint32_t c, bool vec) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (vec) return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#endif
  (void)vec;
  return a ^ b ^ c;
}
CC4_HD void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1, bool vec = false) {
  for (int i = 0; i < 10; ++i) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = philox_xor3((uint32_t)(p1 >> 32), c[1], k0, vec);
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = philox_xor3((uint32_t)(p0 >> 32), c[3], k1, vec);
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

CC4_HD void rng_seed(Rng* r, uint64_t seed, uint32_t mode) {
  r->has32 = 0; r->u32 = 0; r->mode = mode; r->ndraw = 0; r->buf64 = 0; r->has64 = 0; r->pad = 0;
  if (mode == 0) {
    uint64_t v[4];
    seed_sequence_state(seed, v);
    // pcg64_set_seed(state, seed = {v[0] (high), v[1] (low)}, inc = {v[2], v[3]})
    // pcg_setseq_128_srandom_r: state = 0; inc = (initseq << 1) | 1; step; state += initstate; step
    r->inc_hi = (v[2] << 1) | (v[3] >> 63);
    r->inc_lo = (v[3] << 1) | 1ull;
    r->s_hi = 0; r->s_lo = 0;
    pcg_step(r);
    uint64_t lo = r->s_lo + v[1];
    r->s_hi = r->s_hi + v[0] + (lo < r->s_lo ? 1ull : 0ull);
    r->s_lo = lo;
    pcg_step(r);
  } else {
    r->s_lo = seed;  // key
    r->s_hi = 0;     // draw counter
    r->inc_hi = 0;
    r->inc_lo = 0;   // step word
  }
}

// philox mode: called at the start of every env step -> counter = (draw#, step)
CC4_HD void rng_begin_step(Rng* r, uint32_t step) {
  if (r->mode & 1u) { r->inc_lo = (uint64_t)step; r->s_hi = 0; r->has32 = 0; r->has64 = 0; r->ndraw = 0; }
}
// philox: switch to stream `id` at draw 0 (each stream is used once per step). pcg: no-op (one shared stream).
CC4_HD void rng_set_stream(Rng* r, uint32_t id) {
  if (r->mode & 1u) { r->ndraw = id; r->s_hi = 0; r->has32 = 0; r->has64 = 0; }
}
// philox: between steps only (key, episode) matter; park the scratch words so that the serial walk (which switches
// streams in place) and the lane-parallel kernel (which forks lane-local generators) leave identical bytes behind
CC4_HD void rng_park(Rng* r) {
  if (r->mode & 1u) { r->s_hi = 0; r->has32 = 0; r->u32 = 0; r->ndraw = 0; r->has64 = 0; r->buf64 = 0; r->pad = 0; }
}
// philox: lane-local generator for stream `id` of the current (step, episode) of `parent`
CC4_HD void rng_fork(Rng* r, const Rng* parent, uint32_t id) {
  *r = *parent;
  rng_set_stream(r, id);
}
// philox mode: a new episode (reset) bumps the 4th counter word so successive episodes differ
CC4_HD void rng_begin_episode(Rng* r) {
  if (r->mode & 1u) { r->inc_hi++; r->inc_lo = 0xFFFFFFFFull; r->s_hi = 0; r->has32 = 0; r->has64 = 0; r->ndraw = ST_RESET; }
}

// Counter mode: the unread words of the stream's current Philox block wait in (u32, pad, buf64 low, buf64 high) in that order, has64 = how many of them;
// a draw takes u32 and moves the others up.  (Until r06 the block was handed out as two buffered 64-bit halves with the numpy stream's 32-bit buffer on
// top: three state words to test per draw.  Where the compiler knows the state -- the first draws behind rng_set_stream / rng_preload -- either form folds
// away; behind a branch or a loop it does not, and the step kernels are bound by vector issue slots: this form is a test of one count and four moves.)
CC4_HD uint32_t philox_next32(Rng* r) {
  if (r->has64 == 0) {
    uint32_t c[4] = {(uint32_t)r->s_hi, r->ndraw, (uint32_t)r->inc_lo, (uint32_t)r->inc_hi};
    philox4x32_10(c, (uint32_t)r->s_lo, (uint32_t)(r->s_lo >> 32), r->mode == 3u);
    r->s_hi++;
    r->u32 = c[0]; r->pad = c[1]; r->buf64 = (uint64_t)c[2] | ((uint64_t)c[3] << 32); r->has64 = 4;
  }
  const uint32_t out = r->u32;
  r->u32 = r->pad; r->pad = (uint32_t)r->buf64; r->buf64 >>= 32; r->has64--;
  return out;
}
CC4_HD uint64_t rng_next64(Rng* r) {
  if (r->mode & 1u) { const uint64_t lo = philox_next32(r); return lo | ((uint64_t)philox_next32(r) << 32); }   // (no caller in the counter mode: it draws 32-bit words)
  r->ndraw++;
  pcg_step(r);
  uint64_t x = r->s_hi ^ r->s_lo;
  uint32_t rot = (uint32_t)(r->s_hi >> 58);
  return (x >> rot) | (x << ((64u - rot) & 63u));
}

// philox: block `ctr` of stream `stream` of r's (key, step, episode) -- what the generator would compute there
CC4_HD void rng_block(const Rng* r, uint32_t stream, uint32_t ctr, uint32_t c[4]) {
  c[0] = ctr; c[1] = stream; c[2] = (uint32_t)r->inc_lo; c[3] = (uint32_t)r->inc_hi;
  philox4x32_10(c, (uint32_t)r->s_lo, (uint32_t)(r->s_lo >> 32), r->mode == 3u);
}
// philox, right after rng_set_stream: block 0 of the stream was computed elsewhere (rng_block; the lane-parallel kernel
// computes it where a thread has slack); the generator hands out these words instead of computing them
CC4_HD void rng_preload(Rng* r, const uint32_t c[4]) {
  r->u32 = c[0]; r->pad = c[1]; r->buf64 = (uint64_t)c[2] | ((uint64_t)c[3] << 32); r->has64 = 4; r->has32 = 0; r->s_hi = 1;
}
// pcg64_next32: low half first, high half buffered (numpy pcg64.h)
CC4_HD uint32_t rng_next32(Rng* r) {
  if (r->mode & 1u) return philox_next32(r);
  if (r->has32) { r->has32 = 0; return r->u32; }
  uint64_t n = rng_next64(r);
  r->has32 = 1;
  r->u32 = (uint32_t)(n >> 32);
  return (uint32_t)n;
}

// Generator.random(): next_double = (next_uint64 >> 11) * 2^-53
CC4_HD uint32_t rng_next32(Rng* r);
CC4_HD double rng_random(Rng* r) {
  // counter-based mode: one 32-bit word per uniform (every threshold on the path is a multiple of 1/100 or 1/4), so that a
  // typical agent-phase (a bounded int, a uniform, two more bounded ints) fits one Philox block instead of two
  if (r->mode & 1u) return (double)rng_next32(r) * (1.0 / 4294967296.0);
  return (double)(rng_next64(r) >> 11) * (1.0 / 9007199254740992.0);
}

// rng_random(r) < t and rng_random(r) <= t as the caller writes them, one draw.  Counter mode: the uniform is k * 2^-32 for the 32-bit word k, and both that
// product and t * 2^32 are exact in double, so  k * 2^-32 < t  <=>  k < ceil(t * 2^32)  and  k * 2^-32 <= t  <=>  k <= floor(t * 2^32)  -- one integer
// compare against a constant wherever t is one (every rate on the fast path), instead of a convert, a multiply and a double compare on the vector unit.
// NaN thresholds compare false, as a double compare does.  The numpy stream keeps its 53-bit doubles.
CC4_HD bool rng_random_lt(Rng* r, double t) {
  if (r->mode & 1u) {
    const uint32_t k = rng_next32(r);
    const double s = t * 4294967296.0;
    if (!(s > 0.0)) return false;
    if (s >= 4294967296.0) return true;
    const uint64_t f = (uint64_t)s;                                  // floor (s > 0)
    return (uint64_t)k < f + ((double)f < s ? 1u : 0u);              // ceil
  }
  return rng_random(r) < t;
}
CC4_HD bool rng_random_le(Rng* r, double t) {
  if (r->mode & 1u) {
    const uint32_t k = rng_next32(r);
    const double s = t * 4294967296.0;
    if (!(s >= 0.0)) return false;
    if (s >= 4294967296.0) return true;
    return (uint64_t)k <= (uint64_t)s;
  }
  return rng_random(r) <= t;
}
// (int)(rng_random(r) * 4.0): the quarter the uniform falls in
CC4_HD int rng_random_quarter(Rng* r) {
  if (r->mode & 1u) return (int)(rng_next32(r) >> 30);
  return (int)(rng_random(r) * 4.0);    // exact: scaling by a power of two
}

// Generator.integers(0, n) / Generator.choice(n) for 1 <= n <= 2^32:
// distributions.c random_bounded_uint64_fill -> buffered_bounded_lemire_uint32 (rng = n-1); rng==0 draws nothing.
CC4_HD uint32_t rng_below(Rng* r, uint32_t n) {
  if (n <= 1) return 0;
  uint32_t rng_excl = n;  // rng + 1
  uint64_t m = (uint64_t)rng_next32(r) * rng_excl;
  uint32_t leftover = (uint32_t)m;
  if (leftover < rng_excl) {
    uint32_t threshold = (uint32_t)((0xFFFFFFFFu - (n - 1)) % rng_excl);
    while (leftover < threshold) {
      m = (uint64_t)rng_next32(r) * rng_excl;
      leftover = (uint32_t)m;
    }
  }
  return (uint32_t)(m >> 32);
}
// Generator.integers(lo, hi) (hi exclusive)
CC4_HD int32_t rng_range(Rng* r, int32_t lo, int32_t hi) { return lo + (int32_t)rng_below(r, (uint32_t)(hi - lo)); }

// distributions.c random_interval(max): masked rejection on next_uint32 (max < 2^32)
CC4_HD uint32_t rng_interval(Rng* r, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint32_t v;
  while ((v = (rng_next32(r) & mask)) > max) {}
  return v;
}
// Generator.shuffle(python list of n items): for i in reversed(range(1, n)): j = random_interval(i); swap.
// Only the stream consumption matters to the caller (SimulationController.py:418 shuffles an index list
// whose order is never used for anything with bandwidth_usage == 0).
CC4_HD void rng_shuffle_consume(Rng* r, int n) {
  for (int i = n - 1; i >= 1; --i) (void)rng_interval(r, (uint32_t)i);
}

}  // namespace cc4
