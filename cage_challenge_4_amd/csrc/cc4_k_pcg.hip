// cc4_k_pcg.hip -- the numpy-stream (PCG64) kernels of libcc4.so: k_step<LOG> and the persistent k_run_pcg.  See cc4_kernels.h.
#include "cc4_kernels.h"
#include "cc4_persist.h"

// ---------------------------------------------------------------- numpy stream: the two draw-only phases across the wave
// PCG64 is a 128-bit LCG, so the state k steps ahead is A_k * state + B_k * increment (A_k = M^k, B_k = 1 + M + .. + M^(k-1),
// mod 2^128; table filled by cc4_create).  Two phases of a step only CONSUME the stream -- the green agents' policy draws
// (one bounded draw each) and the action-order shuffle (SimulationController.py:418: ~90 masked-rejection draws whose results
// are never used) -- so lane j computes output j+1 directly and the consumption is replayed on the 128 ready words with a
// few wave-wide compares per draw instead of a 128-bit multiply per draw on the walking lane.  Bit-exact with the serial
// walk (rng_below / rng_interval in cc4_rng.h), including has_uint32 / uinteger buffering and the advance counter.
struct PcgJump { uint64_t a_hi, a_lo, b_hi, b_lo; };
__device__ PcgJump g_pcg_jump[WAVE + 1];          // [k]: k = 0 .. 64 steps ahead
__device__ __forceinline__ uint64_t bcast64(uint64_t v) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); }
__device__ __forceinline__ uint64_t lane64(uint64_t v, int src) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src) << 32);
}
__device__ __forceinline__ void pcg_ahead(const PcgJump& J, uint64_t s_hi, uint64_t s_lo, uint64_t i_hi, uint64_t i_lo, uint64_t* o_hi, uint64_t* o_lo) {
  // (a * s + b * inc) mod 2^128
  const uint64_t p_lo = J.a_lo * s_lo, p_hi = __umul64hi(J.a_lo, s_lo) + J.a_hi * s_lo + J.a_lo * s_hi;
  const uint64_t q_lo = J.b_lo * i_lo, q_hi = __umul64hi(J.b_lo, i_lo) + J.b_hi * i_lo + J.b_lo * i_hi;
  const uint64_t lo = p_lo + q_lo;
  *o_lo = lo; *o_hi = p_hi + q_hi + (lo < p_lo ? 1ull : 0ull);
}
__device__ __forceinline__ uint64_t pcg_output(uint64_t hi, uint64_t lo) {   // XSL-RR 128/64
  const uint64_t v = hi ^ lo; const uint32_t rot = (uint32_t)(hi >> 58);
  return (v >> rot) | (v << ((64u - rot) & 63u));
}
// Green policy draws of one step (EnterpriseGreenAgent.get_action: choice of 3 per agent, agent order), all lanes.  `rl` is the
// walking lane's generator (valid on lane 0, updated there).  Returns false without touching anything when a draw would need
// Lemire's re-draw (a zero word: 2^-32 per agent) -- the caller then walks the phase serially.
__device__ __forceinline__ bool wave_green_policy(Rng& rl, int n, uint8_t* green_act, int lane) {
  const uint64_t s_hi = bcast64(rl.s_hi), s_lo = bcast64(rl.s_lo), i_hi = bcast64(rl.inc_hi), i_lo = bcast64(rl.inc_lo);
  const uint32_t has32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.has32), u32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.u32);
  uint64_t h, l;
  pcg_ahead(g_pcg_jump[lane + 1], s_hi, s_lo, i_hi, i_lo, &h, &l);
  const uint64_t out = pcg_output(h, l);
  const uint32_t w0 = (uint32_t)out, w1 = (uint32_t)(out >> 32);
  const int base = (int)has32;                       // agent 0 takes the buffered half word when there is one
  const int g0 = base + 2 * lane, g1 = g0 + 1;
  bool zero = (g0 < n && w0 == 0) || (g1 < n && w1 == 0) || (has32 && u32 == 0);
  if (__ballot(zero)) return false;
  // Lemire, range 3: (word * 3) >> 32 (the leftover test can only fail for word == 0)
  if (g0 < n) green_act[g0] = (uint8_t)(((uint64_t)w0 * 3u) >> 32);
  if (g1 < n) green_act[g1] = (uint8_t)(((uint64_t)w1 * 3u) >> 32);
  if (has32 && lane == 0) green_act[0] = (uint8_t)(((uint64_t)u32 * 3u) >> 32);
  const int fresh = n - base;                        // words taken from new outputs
  const int K = (fresh + 1) >> 1;                    // outputs consumed
  if (K > 0) {
    const uint64_t nh = lane64(h, K - 1), nl = lane64(l, K - 1);
    const uint32_t nu = (uint32_t)__builtin_amdgcn_readlane((int)w1, K - 1);
    if (lane == 0) { rl.s_hi = nh; rl.s_lo = nl; rl.u32 = nu; rl.has32 = (uint32_t)(fresh & 1); rl.ndraw += (uint32_t)K; }
  } else if (lane == 0) rl.has32 = 0;
  return true;
}
// wave64 inclusive scans on the DPP network (row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 and 31 across rows) and
// the shift by one lane that turns them into exclusive ones
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return (uint32_t)x;
}
__device__ __forceinline__ int wave_scan_max(int v) {   // v >= -1
  auto mx = [](int a, int b) { return a > b ? a : b; };
  int x = v;
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x111, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x112, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x114, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x118, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x142, 0xa, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(-1, x, 0x143, 0xc, 0xf, false));
  return x;
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t lane0) {   // lane k <- lane k - 1, lane 0 <- lane0
  return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138, 0xf, 0xf, false);
}
// Generator.shuffle of an n-item list, consumption only (rng_shuffle_consume): for i = n-1 .. 1 one masked-rejection draw
// (random_interval).  All lanes; `rl` as above.  Which i a word is tested against depends on how many words in front of it
// were accepted: a prefix count that depends on itself, solved by relaxation (every lane tests its two words against its
// current estimate of i, a wave scan of the accepted counts gives the next estimates; a word's verdict only moves when i
// crosses its masked value, so a handful of rounds settle a window of 128 words).
__device__ __forceinline__ void wave_shuffle_consume(Rng& rl, int n, int lane) {
  if (n <= 1) return;
  uint64_t s_hi = bcast64(rl.s_hi), s_lo = bcast64(rl.s_lo);
  const uint64_t i_hi = bcast64(rl.inc_hi), i_lo = bcast64(rl.inc_lo);
  uint32_t has32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.has32), u32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.u32);
  uint32_t adv = 0;
  int i = n - 1;
  auto mask_of = [](uint32_t m) { return 0xFFFFFFFFu >> __builtin_clz(m); };   // m >= 1: the smallest 2^k - 1 >= m (random_interval's mask)
  if (has32) {                                       // the buffered half word is the first candidate
    has32 = 0;
    if ((u32 & mask_of((uint32_t)i)) <= (uint32_t)i) --i;
  }
  while (i >= 1) {
    // a window of 64 outputs = 128 words: word p = half (p & 1) of output (p >> 1) + 1, i.e. lane p >> 1
    uint64_t h, l;
    pcg_ahead(g_pcg_jump[lane + 1], s_hi, s_lo, i_hi, i_lo, &h, &l);
    const uint64_t out = pcg_output(h, l);
    const uint32_t w0 = (uint32_t)out, w1 = (uint32_t)(out >> 32);
    uint32_t f_cur = 3, inc = 0;                     // accepted flags of the lane's two words (first guess: all accepted)
    for (int round = 0; round < WAVE + 2; ++round) {
      inc = wave_scan_add((f_cur & 1u) + (f_cur >> 1));
      const int i0 = i - (int)wave_shr1(inc, 0u);    // the i word 0 of this lane is tested against
      const uint32_t a0 = (i0 >= 1 && (w0 & mask_of((uint32_t)(i0 >= 1 ? i0 : 1))) <= (uint32_t)i0) ? 1u : 0u;
      const int i1 = i0 - (int)a0;
      const uint32_t a1 = (i1 >= 1 && (w1 & mask_of((uint32_t)(i1 >= 1 ? i1 : 1))) <= (uint32_t)i1) ? 1u : 0u;
      const uint32_t f_new = a0 | (a1 << 1);
      const bool moved = f_new != f_cur;
      f_cur = f_new;
      if (!__ballot(moved)) break;                   // lane k is exact after k + 1 rounds at the latest
    }
    const int total = (int)rdlane(inc, WAVE - 1);
    int cur = 2 * WAVE;                              // first unconsumed word of the window
    if (total >= i) {                                // the draw for i = 1 ends inside the window: behind the i-th accepted word
      const uint64_t m = __ballot((int)inc >= i);
      const int L = __ffsll((unsigned long long)m) - 1;
      const int before = L ? (int)rdlane(inc, L - 1) : 0;
      const uint32_t fl = rdlane(f_cur, L);
      cur = 2 * L + ((before + (int)(fl & 1u) >= i) ? 1 : 2);
      i = 0;
    } else i -= total;
    const int K = (cur + 1) >> 1;                    // outputs of this window that were touched
    s_hi = lane64(h, K - 1); s_lo = lane64(l, K - 1);
    u32 = rdlane(w1, K - 1);
    has32 = (uint32_t)(cur & 1);
    adv += (uint32_t)K;
  }
  if (lane == 0) { rl.s_hi = s_hi; rl.s_lo = s_lo; rl.has32 = has32; rl.u32 = u32; rl.ndraw += adv; }
}

// ---------------------------------------------------------------- numpy stream: the green actions across the wave
// GreenAccessService / GreenLocalWork draw from the one shared stream, agent after agent, and every agent's number of draws
// depends on what it drew -- but on nothing it reads from the state that an earlier green action of the same step could have
// changed, with two exceptions: an ephemeral port that is already taken (Host.py:175-187 re-draws once) and a phishing email
// (a red session appears).  So the serial walk (50-odd agents x [a 128-bit multiply per draw + an HBM round trip for the port
// bitmap and one per event byte]) is replaced by:
//  (1) the next 128 outputs of the LCG from the closed form, two per lane, into LDS;
//  (2) where in the stream each agent starts.  That is a prefix sum over the agents' draw counts, which depend on the drawn
//      values, i.e. on the start: solved by relaxation -- every lane (one agent each) replays its action from its current
//      start estimate, a wave scan of the counts gives the next estimates, until nothing moves.  Agent 0's start is given,
//      so agent k is exact after k + 1 rounds at the latest; as an action's draw count rarely depends on the values (a
//      blocked route, a failed reliability roll, the two 1 % events), three rounds are the rule;
//  (3) the agents' effects -- the port bitmap test-and-set (one L2 atomic), the event bits, the reward -- on their lanes.
// A taken port, a phishing email, a Lemire re-draw (n / 2^32 per draw) or the end of the window end a batch: its agents in
// front of that point are committed, the agent at that point is resolved by the serial code on lane 0 (ports set
// speculatively by later agents are cleared first), and the next batch starts behind it.  Bit-exact with the serial walk
// (rng_below / rng_random / has_uint32 buffering in cc4_rng.h).
constexpr int GW_OUT = 2 * WAVE;                           // outputs per window
constexpr double P01_SCALED = 0.01 * 9007199254740992.0;   // Generator.random() < 0.01 on the 53-bit integer: exact scaling
constexpr uint64_t P01_FLOOR = (uint64_t)P01_SCALED;
static_assert((double)P01_FLOOR * (1.0 / 9007199254740992.0) < 0.01 && (double)(P01_FLOOR + 1) * (1.0 / 9007199254740992.0) >= 0.01,
              "integer form of rng_random() < 0.01");
enum : uint32_t { GR_VALID = 1, GR_FAIL = 2, GR_EPH = 4, GR_CONN = 8, GR_PROC = 16, GR_HARD = 32, GR_PHISH = 64 };
__device__ __forceinline__ void wave_green_exec(Ctx x, Rng& rl, uint64_t* win, int lane, unsigned long long* gstat = nullptr) {
  EnvState* s = x.s;
  const int ng = s->n_green;
  const uint64_t i_hi = bcast64(rl.inc_hi), i_lo = bcast64(rl.inc_lo);
  int g0 = 0;
  for (int guard = 0; g0 < ng; ++guard) {
    if (guard > 2 * MAXG + 8) { if (lane == 0) set_err(x, E_UNREACHABLE); break; }   // every batch advances g0
    const unsigned long long t0 = gstat ? clock64() : 0;
    const uint64_t s_hi = bcast64(rl.s_hi), s_lo = bcast64(rl.s_lo);
    const uint32_t has0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.has32), u0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl.u32);
    // (1) the window: output p (0-based) is computed on lane p & 63 (set p >> 6) and stored at win[p]
    uint64_t h0, l0, h1, l1;
    pcg_ahead(g_pcg_jump[lane + 1], s_hi, s_lo, i_hi, i_lo, &h0, &l0);
    pcg_ahead(g_pcg_jump[WAVE], h0, l0, i_hi, i_lo, &h1, &l1);
    __syncthreads();           // (one wave per block) the previous batch's window reads are done
    win[lane] = pcg_output(h0, l0);
    win[WAVE + lane] = pcg_output(h1, l1);
    __syncthreads();
    // what the agents of this batch (lane k <-> agent g0 + k) bring along
    const int gi = g0 + lane;
    const bool in = gi < ng;
    const uint32_t my_act = in ? x.w->green_act[gi] : 2u;
    const bool active = my_act < 2;
    // what the action reads from the state (green_prepare: the host's service table -- HBM here --, allowed server counts)
    const uint64_t pre = active ? green_prepare(x, gi, (int)my_act) : 0ull;
    const uint32_t gh = in ? s->green_host[gi] : 0u;
    uint32_t my_blk = 0;       // bit sn: traffic between the agent's subnet and subnet sn is blocked either way
    if (active && my_act == 0) {
      const int own = h_subnet((int)gh);
#pragma unroll
      for (int sn = 0; sn < NSUB - 1; ++sn) if (((s->blocks[own] >> sn) | (s->blocks[sn] >> own)) & 1u) my_blk |= 1u << sn;
    }
    const uint32_t lw_am = green_lw_active(pre);
    // what decides an agent's draw COUNT, reduced to shifts: the leading 32-bit draws (a = 1 or 2; with two, the first one
    // picks -- a server / a service -- out of `npick`), for GreenAccessService the picks that end on a blocked route (no
    // further draw), for GreenLocalWork the reliability (/20) of the pick
    const bool is_as = active && my_act == 0, is_lw = active && my_act == 1 && lw_am != 0;
    uint32_t npick = 0; uint64_t pickinfo = 0;
    if (is_as) {
      npick = (uint32_t)(pre >> 56);
#pragma unroll
      for (int sn = 0; sn < NSUB - 1; ++sn) {
        const uint32_t before = sn ? (uint32_t)((pre >> (8 * (sn - 1))) & 0xFF) : 0u, tot = (uint32_t)((pre >> (8 * sn)) & 0xFF);
        if (((my_blk >> sn) & 1u) && tot > before) pickinfo |= ((tot >= 64 ? ~0ull : (1ull << tot) - 1ull)) & ~((1ull << before) - 1ull);
      }
    } else if (is_lw) {
      npick = (uint32_t)popc32(lw_am);
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < MAXSV; ++i) if ((lw_am >> i) & 1u) { pickinfo |= ((pre >> (8 * i)) & 0x7Full) << (8 * cnt); ++cnt; }
    }
    const uint32_t lead = (is_as || is_lw) ? (npick > 1 ? 2u : 1u) : 0u;
    // (2) relaxation.  Per lane: d = outputs taken | 32-bit draws << 16 ; q = window index of the last output a 32-bit draw
    // fetched (its high half is numpy's buffered `uinteger`), -1: none
    uint32_t d_cur = 0, sp = 0, sh = 0, su = 0;
    int q_cur = -1;
    uint32_t inc = 0; int qinc = -1;    // inclusive scans of the last round
    int rounds = 0;
    bool stuck = false;
    for (;; ++rounds) {
      inc = wave_scan_add(d_cur); qinc = wave_scan_max(q_cur);
      const uint32_t ex = wave_shr1(inc, 0u);
      const int exq = (int)wave_shr1((uint32_t)qinc, 0xFFFFFFFFu);
      const int pos = (int)(ex & 0xFFFFu);
      const uint32_t has = (has0 + (ex >> 16)) & 1u;
      const uint32_t buf = exq < 0 ? u0 : (uint32_t)(win[exq < GW_OUT ? exq : GW_OUT - 1] >> 32);
      sp = (uint32_t)pos; sh = has; su = buf;
      uint32_t d_new = 0; int q_new = -1;
      if (lead) {
        const uint64_t f0 = win[pos < GW_OUT ? pos : GW_OUT - 1], f1 = win[pos + 1 < GW_OUT ? pos + 1 : GW_OUT - 1];
        const uint32_t first = has ? buf : (uint32_t)f0, second = has ? (uint32_t)f0 : (uint32_t)(f0 >> 32);
        uint32_t i = (has && lead == 1) ? 0u : 1u, n32 = lead;
        uint32_t has2 = has ^ (lead & 1u);
        if (i) q_new = pos;
        const uint32_t pick = lead == 2 ? (uint32_t)(((uint64_t)first * npick) >> 32) : 0u;
        if (is_as) {
          if (!((pickinfo >> pick) & 1ull)) ++i;                       // not blocked: the 1 % connection-event roll
        } else {
          const uint32_t roll = (uint32_t)(((uint64_t)(lead == 2 ? second : first) * 100u) >> 32);
          if (roll < (uint32_t)((pickinfo >> (8 * pick)) & 0xFF) * 20u) {
            const uint64_t u1 = i ? f1 : f0;
            ++i;
            if ((u1 >> 11) <= P01_FLOOR) { ++n32; if (!has2) { q_new = pos + (int)i; ++i; } has2 ^= 1u; }   // the false-positive event's port
            ++i;                                                        // the phishing roll
          }
        }
        d_new = i | (n32 << 16);
      }
      const bool moved = d_new != d_cur || q_new != q_cur;
      d_cur = d_new; q_cur = q_new;
      if (!__ballot(moved)) break;          // the estimates the lanes just used were the fixed point
      if (rounds > WAVE + 2) { stuck = true; break; }   // cannot happen (lane k is exact after k + 1 rounds): serial walk
    }
    // the agents' actions in full, from the starts found
    uint32_t rec = 0;
    if (active) {
      const int pos = (int)sp;
      uint32_t has = sh, buf = su;
      int q_new = -1;
      uint64_t f[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) { const int p = pos + j; f[j] = win[p < GW_OUT ? p : GW_OUT - 1]; }
      int i = 0; uint32_t n32 = 0; bool hard = false;
      auto fetch = [&]() { uint64_t v = f[0]; if (i == 1) v = f[1]; if (i == 2) v = f[2]; if (i == 3) v = f[3]; if (i >= 4) v = f[4]; ++i; return v; };
      auto take32 = [&]() { ++n32; if (has) { has = 0; return buf; } const uint64_t o = fetch(); buf = (uint32_t)(o >> 32); q_new = pos + i - 1; has = 1; return (uint32_t)o; };
      auto below = [&](uint32_t n) { if (n <= 1) return 0u; const uint64_t m = (uint64_t)take32() * n; if ((uint32_t)m < n) hard = true; return (uint32_t)(m >> 32); };
      uint32_t r = GR_VALID;
      if (my_act == 0) {       // green_access_service
        const int c = (int)below((uint32_t)(pre >> 56));
        int sn;
        const int dest = green_as_dest(pre, c, &sn);
        const uint32_t p = below(EPH_RANGE);
        r |= GR_EPH | ((uint32_t)dest << 8) | (p << 16);
        if ((my_blk >> sn) & 1u) r |= GR_FAIL | GR_CONN;
        else if ((fetch() >> 11) <= P01_FLOOR) r |= GR_CONN;
      } else if (!lw_am) r |= GR_FAIL;   // green_local_work
      else {
        const int c = nth_bit(lw_am, (int)below((uint32_t)popc32(lw_am)));
        const int rel = (int)((pre >> (8 * c)) & 0x7F) * 20;
        if ((int)below(100) >= rel) r |= GR_FAIL;
        else {
          if ((fetch() >> 11) <= P01_FLOOR) { const uint32_t p = below(EPH_RANGE); r |= GR_EPH | GR_PROC | (gh << 8) | (p << 16); }
          if ((fetch() >> 11) <= P01_FLOOR) r |= GR_PHISH;
        }
      }
      // a Lemire re-draw, a window that may not cover this agent, or (never) a count that differs from the relaxation's
      if (hard || pos + 5 > GW_OUT || ((uint32_t)i | (n32 << 16)) != d_cur || q_new != q_cur) r |= GR_HARD;
      rec = r;
    }
    if (stuck) rec |= GR_HARD;
    // where the batch ends: in front of the first agent the serial code has to resolve, behind the first phishing email
    const int cnt = (ng - g0) < WAVE ? (ng - g0) : WAVE;
    int kend = cnt;
    enum { R_NEXT, R_HARD, R_PHISH } reason = R_NEXT;
    const uint64_t m_hard = __ballot((rec & GR_HARD) != 0), m_phish = __ballot((rec & GR_PHISH) != 0);
    if (m_hard) { const int k = __ffsll((unsigned long long)m_hard) - 1; if (k < kend) { kend = k; reason = R_HARD; } }
    if (m_phish) { const int k = __ffsll((unsigned long long)m_phish) - 1; if (k < kend) { kend = k + 1; reason = R_PHISH; } }
    const unsigned long long t1 = gstat ? clock64() : 0;
    // (3) ports: one atomic test-and-set per agent; the first agent that finds its port taken ends the batch in front of it
    const uint32_t eh = (rec >> 8) & 0xFFu, ep = (rec >> 16) & 0x3FFFu;
    bool coll = false;
    const bool has_port = lane < kend && (rec & GR_EPH);
    if (has_port) coll = eph_test_and_set(x.c, (int)eh, ep);
    const uint64_t cm = __ballot(coll);
    if (cm) {
      const int kc = __ffsll((unsigned long long)cm) - 1;
      if (has_port && !coll && lane >= kc) __hip_atomic_fetch_and(&x.c->eph[eh][ep >> 5], ~(1u << (ep & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      kend = kc; reason = R_HARD;
    }
    if (lane < kend && (rec & GR_VALID)) {
      if (rec & GR_CONN) ev_or(x, (int)eh, EV_CUR_CONN);
      if (rec & GR_PROC) ev_or(x, (int)eh, EV_CUR_PROC);
      if (rec & GR_FAIL) __hip_atomic_fetch_add(&s->brm, reward_table(s->phase, h_subnet((int)gh), my_act == 0 ? RW_ASF : RW_LWF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the stream behind the committed agents = where agent kend starts
    int pos_e = 0; uint32_t has_e = has0, u_e = u0;
    if (kend > 0) {
      const uint32_t ie = rdlane(inc, kend - 1);
      const int qe = (int)rdlane((uint32_t)qinc, kend - 1);
      pos_e = (int)(ie & 0xFFFFu); has_e = (has0 + (ie >> 16)) & 1u;
      if (qe >= 0) u_e = (uint32_t)(win[qe] >> 32);
    }
    uint64_t n_hi = s_hi, n_lo = s_lo;
    if (pos_e > 0) {
      const int j = (pos_e - 1) & 63;
      n_hi = lane64(h0, j); n_lo = lane64(l0, j);
      if (pos_e > WAVE) { n_hi = lane64(h1, j); n_lo = lane64(l1, j); }
    }
    const unsigned long long t2 = gstat ? clock64() : 0;
    const int gend = g0 + kend;
    if (lane == 0) {
      rl.s_hi = n_hi; rl.s_lo = n_lo; rl.has32 = has_e; rl.u32 = u_e; rl.ndraw += (uint32_t)pos_e;
      if (reason == R_PHISH) phishing(x, s->green_host[gend - 1]);
      else if (reason == R_HARD) {
        s->brm += step_green_exec(x, gend);
        if (bit_get(x.w->phish_mask, gend)) { bit_clr(x.w->phish_mask, gend); phishing(x, s->green_host[gend]); }
      }
    }
    if (gstat && lane == 0) {
      gstat[0] += 1; gstat[1] += reason == R_PHISH; gstat[2] += m_hard != 0; gstat[3] += cm != 0;
      gstat[4] += t1 - t0; gstat[5] += t2 - t1; gstat[6] += clock64() - t2; gstat[7] += rounds + 1;
    }
    g0 = reason == R_HARD ? gend + 1 : gend;
  }
}

// One step of one episode of the numpy-stream mode on one wavefront: the body of k_step and of the persistent kernel k_run_pcg
// (there a.rand_t / a.full_obs are the item's: set by the caller).
// first / last: as in philox1_body -- inside a run of steps of one episode on one wave the agent part stays in LDS
template <bool LOG>
__device__ __forceinline__ void pcg_body(StepArgs a, const int e, const int lane, const bool first, const bool last) {
  // numpy-PCG64 mode: one shared stream => the agent walk is strictly serial (lane 0); only the RNG-free parts
  // (row staging, end-turn Monitor roll-over over the 137 hosts, observation encode) use the other lanes.
  extern __shared__ uint4 lds[];
  // r06: the 1 KB LCG window of the green actions (wave_green_exec: 128 outputs, written once per batch, read ~9 times per lane at computed
  // positions) lives in MEMORY now (StepArgs.reset_ws, 1 KB per episode; the CU's L1 / the XCD's L2 serve it): with it, the byte copy of the
  // observations (the packed exchange row is read back from the int32 row, pack_row_from_obs) and the phase timers (full build only) out of LDS,
  // agent part + statics are 6 368 B = FIVE 1280-byte granules -- 24 waves per CU at 80 VGPRs instead of 20 at 6 granules / 91 VGPRs.
  uint64_t* const win_mem = reinterpret_cast<uint64_t*>(a.reset_ws) + (size_t)e * GW_OUT;
  if constexpr (!LOG) a.prof = nullptr;
  __shared__ int ok_lds;
  __shared__ StepWork work;
  EnvCold* const cold_e = cold_at(a.cold, (size_t)e, cold_row_bytes(a.steps));
  unsigned long long t_begin = a.prof ? clock64() : 0;
  const uint4* src = reinterpret_cast<const uint4*>(a.st + e);
  if (first) stage_in<HOT_VEC>(lds, src, lane);
  for (int i = lane; i < (int)(sizeof(StepWork) / 4); i += WAVE) reinterpret_cast<uint32_t*>(&work)[i] = 0;
  __syncthreads();
  EnvState* s = reinterpret_cast<EnvState*>(lds);   // only the part in front of EnvState.hd is valid here
  HostDyn* const hd = a.st[e].hd;                   // the host table stays in HBM / L2
  __shared__ unsigned long long prof_lds[LOG ? 16 : 1];   // phase counters accumulate in LDS, flushed once at the end (full build: cc4_debug_profile selects it)
  unsigned long long* prof = a.prof ? prof_lds : nullptr;
  if (prof && lane < 16) prof_lds[lane] = 0;
  // the shared numpy stream is walked on a register copy (this kernel serves the PCG mode only; mode pinned so the Philox
  // paths fold away) and written back once, before the row leaves LDS
  Rng rl = s->rng;
  rl.mode = 0;
  rl.pad = 0;
  Ctx x{s, cold_e, &rl, hd, &work, lane == 0 ? prof : nullptr};
  x.lg = (LOG && cold_e->evlog.enabled) ? &cold_e->evlog : nullptr;
  const ExtAct* const xt = (LOG && a.ext) ? a.ext + (size_t)e * EXT_PER_ENV : nullptr;   // this episode's submitted red / green actions
  x.ext = xt;
  if (prof && lane == 0) prof[11] += clock64() - t_begin;
  const bool do_reset = a.autoreset && s->done;
  // The ordered walk is lane 0's; between its stretches the whole wave does what needs no order: the two draw-only phases
  // (green policy draws, action-order shuffle) straight from the LCG's closed form, and the green actions' state reads.
  // every lane evaluates the mission-phase check (four words of the row); the accumulators of the step were left initialised by
  // step_end / the reset, so the blue submissions (lanes 1..5: in-kernel action draw, decode, queue) run beside lane 0's step_phase
  const bool step_ok = !do_reset && step_phase_of(s->step_count, s->phase_len[0], s->phase_len[1], s->phase_len[2]) >= 0;
  if (lane == 0) {
    ok_lds = step_ok ? 1 : 0;
    if (do_reset) {
      env_reset(x, 0, 0, a.steps, true, a.policy, a.topo);   // new episode, same stream (CybORG.reset(seed=None)); this kernel serves the numpy-stream mode only
    } else {
      CC4_TICK0(x);
      (void)step_phase(x, false);    // sets E_STEP_PAST_END when !step_ok
      CC4_TICK(x, 0);
      if (step_ok) rng_policy_swap(x, false);     // CybORG.set_seed split: the policies draw from the old stream (EnvCold.rng2)
      if (step_ok && (s->policy & BP_RANDOM_BIT))   // built-in blue policy: its draws are the first of the step, in agent order
        for (int b = 0; b < NBLUE; ++b) {
          int32_t act = a.actions ? a.actions[e * NBLUE + b] : -1;
          if (a.rand_out) { act = random_blue_action(a.rand_seed0, a.rand_t, e, b); a.rand_out[e * NBLUE + b] = act; }
          step_blue_submit(x, b, act);
        }
    }
  } else if (step_ok && lane <= NBLUE && !(s->policy & BP_RANDOM_BIT)) {
    const int b = lane - 1;
    int32_t act = a.actions ? a.actions[e * NBLUE + b] : -1;
    if (a.rand_out) { act = random_blue_action(a.rand_seed0, a.rand_t, e, b); a.rand_out[e * NBLUE + b] = act; }
    Ctx xb{s, cold_e, &rl, hd, &work};
    step_blue_submit(xb, b, act);
  }
  __syncthreads();
  if (ok_lds) {
    bool drawn = false;
    // (with submitted green actions in play the agents that have one do not draw: the walking lane asks them one by one)
    if (!(s->policy & GP_SLEEP_BIT) && !xt) drawn = wave_green_policy(rl, s->n_green, work.green_act, lane);
    // the observation half of the six red policies draws nothing and touches only its own agent: side by side on six lanes
    if (lane < NRED) { Ctx xo{s, cold_e, &rl, hd, &work}; xo.ext = xt; step_red_observe(xo, lane); }
    __syncthreads();
    if (lane == 0) {
      if (!drawn) for (int g = 0; g < s->n_green; ++g) step_green_policy(x, g);   // SleepAgent greens, or the 2^-32 re-draw case
      CC4_TICK(x, 1);
      for (int r = 0; r < NRED; ++r) s->n_actions -= step_red_policy_tick(x, r, (s->policy & 3) != RP_RANDOM);
      rng_policy_swap(x, true);
      CC4_TICK(x, 2);
      for (int b = 0; b < NBLUE; ++b) step_tick_blue(x, b);
      CC4_TICK(x, 3);
    }
    __syncthreads();
    wave_shuffle_consume(rl, s->n_actions, lane);   // sort_action_order's shuffle (SC:398-464) only consumes the stream
    if (lane == 0) { CC4_TICK(x, 4); step_blue_exec(x, true); }
  }
  __syncthreads();
  if (ok_lds) {
    // the green actions: across the wave (wave_green_exec)
    if constexpr (!LOG) wave_green_exec(x, rl, win_mem, lane, nullptr);
    else {
      // with the event log on (log entries are ordered): on the walking lane; what the actions read from the state (service
      // tables of their hosts -- HBM here --, allowed server counts) is prepared for all agents at once on the idle lanes
      __shared__ uint64_t gpre_lds[MAXG];
      for (int g = lane; g < s->n_green; g += WAVE) { const int act = work.green_act[g]; if (act < 2) gpre_lds[g] = green_prepare(x, g, act); }
      __syncthreads();
      if (lane == 0) {
        Ctx xg = x; xg.gpre = gpre_lds;
        for (int g = 0; g < s->n_green; ++g) {
          s->brm += step_green_exec(xg, g);
          if (bit_get(work.phish_mask, g)) { bit_clr(work.phish_mask, g); phishing(x, s->green_host[g]); }
        }
      }
    }
    if (lane == 0) {
      CC4_TICK(x, 6);
      step_red_exec(x);
      step_reassign(x, red_foreign_agents(s));
    }
  }
  __syncthreads();
  if (ok_lds) {
    // end-turn Monitor roll-over: the hosts' event bytes are part of the staged row (EnvState.hev)
    monitor_roll_all(s, lane);
    if (lane == 0) step_monitor_pend(x);
    __syncthreads();
    {
      // end-turn RedSessionCheck: it draws only when it has to promote a session to primary; when no agent needs that (the
      // usual case) the six checks run side by side, else in order on the walking lane
      const bool need = lane < NRED && rsc_draws(s, lane);
      const bool serial = __ballot(need) != 0ull;
      if (!serial && lane < NRED) { Ctx xc{s, cold_e, &rl, hd, &work, nullptr, nullptr, x.lg}; step_rsc(xc, lane); }
      if (lane == 0) CC4_TICK(x, 9);
      __syncthreads();
      if (lane == 0) {
        if (serial) for (int r = 0; r < NRED; ++r) step_rsc(x, r);
        CC4_TICK(x, 10);
        step_end(x, a.msgs ? a.msgs + e * NBLUE * MSG_LEN : nullptr);
      }
    }
    __syncthreads();
  }
  if (lane == 0) { s->rng = rl; a.reward[e] = s->reward; a.done[e] = s->done; a.err[e] = s->err; }
  unsigned long long t_obs = a.prof ? clock64() : 0;
  {
    // straight to HBM, kind-sorted (uniform branches); the output buffer persists between steps, so the values that only a
    // Block/Allow or a new mission phase changes are written when that happened (EnvState.obs_dirty), after a reset, or when the
    // caller asks -- as in the counter-mode kernels; the byte copy in LDS only feeds the packed exchange row
    int32_t* o = a.obs + (size_t)e * OBS_TOTAL;
    const uint32_t dirty = (do_reset || a.full_obs) ? (uint32_t)OD_ALL : (uint32_t)s->obs_dirty;
    encode_obs_fast<WAVE>(s, o, nullptr, false, lane);
    encode_obs_slow(s, o, dirty, lane);
  }
  __syncthreads();
  unsigned long long t_out = a.prof ? clock64() : 0;
  if (prof && lane == 0) prof[12] += t_out - t_obs;
  uint4* dst = reinterpret_cast<uint4*>(a.st + e);
  if (last) stage_out<HOT_VEC>(dst, lds, lane);
  if (a.obs8) {     // the packed exchange row, read back from the int32 row this wave has just (re)written (the buffer persists between steps: it holds every current value)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pack_row_from_obs(a.obs8 + (size_t)e * OBS_PACKED, a.obs + (size_t)e * OBS_TOTAL, lane);
  }
  if (prof && lane == 0) { prof[13] += clock64() - t_out; prof[14] += clock64() - t_begin; }
  if (prof) { __syncthreads(); if (lane < 15) a.prof[PROF_SLOTS * (size_t)e + lane] += prof_lds[lane]; }
}
template <bool LOG>
__global__ __launch_bounds__(WAVE) void k_step(StepArgs a) {
  const int e = a.e0 + (int)blockIdx.x;
  if (e >= a.n) return;
  pcg_body<LOG>(a, e, (int)threadIdx.x);
}

// the persistent schedule (cc4_persist.h) around the numpy-stream step: the bit-exact mode's large batches
__global__ __launch_bounds__(WAVE, 6) void k_run_pcg(StepArgs a, RunArgs ra, XchgArgs x) { persist_loop<true>(a, ra, x); }


// the kernels the host side launches (cc4_kernel_decls.h)
template __global__ void k_step<false>(StepArgs);
template __global__ void k_step<true>(StepArgs);

// PCG64 jump table of the wave-wide phases (see wave_green_policy): A_k = M^k, B_k = 1 + M + .. + M^(k-1) mod 2^128.  Called by cc4_create.
hipError_t cc4_upload_pcg_tables() {
  PcgJump tab[WAVE + 1];
  const unsigned __int128 M = ((unsigned __int128)CC4_PCG_MULT_HI << 64) | CC4_PCG_MULT_LO;
  unsigned __int128 A = 1, B = 0;
  for (int k = 0; k <= WAVE; ++k) {
    tab[k].a_hi = (uint64_t)(A >> 64); tab[k].a_lo = (uint64_t)A; tab[k].b_hi = (uint64_t)(B >> 64); tab[k].b_lo = (uint64_t)B;
    B = B * M + 1; A = A * M;
  }
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pcg_jump), tab, sizeof(tab));
}
