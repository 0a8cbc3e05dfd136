// cc4_engine.h -- the CC4 episode transition: reset (scenario generation), step, flat observation.
//
// Single source for the gfx950 device build (the cc4_k_*.hip translation units) and for the host build used as the CPU
// oracle (oracle/cc4_oracle.cpp).  Each function cites the reference lines it restates
// (paths under /root/reference/CybORG).  See docs/REFERENCE_NOTES.md for the condensed semantics.
#pragma once
#include "cc4_state.h"
#include "cc4_tables.h"

namespace cc4 {

// s: the episode's row (the device kernels pass their LDS copy); hd: its host table -- &s->hd[0] when the whole row is staged,
// the HBM row's table when only the part in front of it is (numpy-stream kernel); w: the step's work area;
// lg: the episode's event log when it is enabled (the callers know that without a memory read)
// gpre: per green agent, what its action needs from the state (green_prepare), computed ahead by other lanes; null: computed
// where it is needed
struct Ctx { EnvState* s; EnvCold* c; Rng* r; HostDyn* hd; StepWork* w; unsigned long long* prof = nullptr;
             unsigned long long* aprof = nullptr; EvLog* lg = nullptr; const uint64_t* gpre = nullptr;
             // this episode's externally submitted red / green actions (ExtAct[EXT_PER_ENV], cc4_state.h), or null: the builds of the
             // step that serve them are the "full" ones (the same template parameter as the event log), so that on the fast path
             // every test of it folds away
             const ExtAct* ext = nullptr; };

// optional phase timing (device only, debug builds of the kernel pass a buffer): prof[i] += cycles since last tick
#if defined(__HIP_DEVICE_COMPILE__)
#define CC4_TICK(x, i) do { if ((x).prof) { unsigned long long _t = clock64(); (x).prof[(i)] += _t - (x).prof[15]; (x).prof[15] = _t; } } while (0)
#define CC4_TICK0(x) do { if ((x).prof) (x).prof[15] = clock64(); } while (0)
// per-red-agent section timers (Ctx.aprof = that agent's 8 slots of the debug buffer)
#define CC4_AT0(x) unsigned long long _at = (x).aprof ? clock64() : 0
#define CC4_AT(x, k) do { if ((x).aprof) { unsigned long long _n = clock64(); (x).aprof[(k)] += _n - _at; _at = _n; } } while (0)
#if defined(CC4_FINE)
// finer sections of red agent 0's policy / tick (debug build -DCC4_FINE): slots 104 + k of the episode's row (aprof of agent 0 = slot 16)
#define CC4_FT0(x, r) unsigned long long _ft = ((x).aprof && (r) == 0) ? clock64() : 0
#define CC4_FT(x, r, k) do { if ((x).aprof && (r) == 0) { unsigned long long _n = clock64(); (x).aprof[88 + (k)] += _n - _ft; _ft = _n; } } while (0)
#else
#define CC4_FT0(x, r) do { } while (0)
#define CC4_FT(x, r, k) do { } while (0)
#endif
#else
#define CC4_FT0(x, r) do { } while (0)
#define CC4_FT(x, r, k) do { } while (0)
#define CC4_AT0(x) do { } while (0)
#define CC4_AT(x, k) do { } while (0)
#define CC4_TICK(x, i) do { } while (0)
#define CC4_TICK0(x) do { } while (0)
#endif

// ------------------------------------------------------------------ small helpers
CC4_HD void set_err(Ctx x, uint32_t f) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_fetch_or(&x.s->err, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  x.s->err |= f;
#endif
}
// OR event bits into EnvState.hev[h] (one byte per host, four hosts per word); atomic on the device because agents resolved on
// different lanes may raise events on the same host
CC4_HD void ev_or(Ctx x, int h, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t* w = reinterpret_cast<uint32_t*>(x.s->hev) + (h >> 2);
  __hip_atomic_fetch_or(w, bits << (8 * (h & 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  x.s->hev[h] |= (uint8_t)bits;
#endif
}
// optional event log (EnvCold.evlog, enabled through cc4_enable_event_log): the content of the HostEvents entry behind an
// event flag.  `order` orders entries of different agents as the serial walk would (greens by host id, then reds by index).
CC4_HD void ev_log(Ctx x, int order, int host, int kind, int laddr, int lport, int raddr, int rport, int pid, int rep = 1) {
  if (!x.lg) return;
  EvLog& L = *x.lg;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t i = atomicAdd(&L.n, 1u);
#else
  uint32_t i = L.n++;
#endif
  if (i >= (uint32_t)MAX_EV) return;   // the count keeps growing: the reader sees n > MAX_EV and reports truncation
  EvRec e; e.host = (uint8_t)host; e.kind = (uint8_t)kind; e.laddr = (uint8_t)laddr; e.raddr = (uint8_t)raddr;
  e.lport = (uint16_t)lport; e.rport = (uint16_t)rport; e.pid = (uint16_t)pid; e.rep = (uint8_t)rep; e.order = (uint8_t)order;
  L.rec[i] = e;
}
CC4_HD int h_subnet(int h) { return h / SLOTS; }
CC4_HD int h_slot(int h) { return h % SLOTS; }
CC4_HD bool h_is_router(int h) { return h_slot(h) == 0; }          // incl. root_internet_host_0
CC4_HD bool h_is_user(int h) { int sl = h_slot(h); return sl >= 1 && sl <= 10; }
CC4_HD bool h_is_server(int h) { return h_slot(h) >= 11; }
CC4_HD constexpr int h_make(int s, int slot) { return s * SLOTS + slot; }
CC4_HD bool bit_get(const uint32_t* b, int i) { return (b[i >> 5] >> (i & 31)) & 1u; }
CC4_HD void bit_set(uint32_t* b, int i) { b[i >> 5] |= 1u << (i & 31); }
CC4_HD void bit_clr(uint32_t* b, int i) { b[i >> 5] &= ~(1u << (i & 31)); }
// for bitmaps shared by agents that may be resolved on different waves (EnvState.red_hosts)
CC4_HD void bit_set_shared(uint32_t* b, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_fetch_or(&b[i >> 5], 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  b[i >> 5] |= 1u << (i & 31);
#endif
}
// the HostDyn row of host h is about to be written (StepWork.hdirty; several lanes may mark)
CC4_HD void hd_touch(Ctx x, int h);
CC4_HD void bit_clr_shared(uint32_t* b, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_fetch_and(&b[i >> 5], ~(1u << (i & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  b[i >> 5] &= ~(1u << (i & 31));
#endif
}
CC4_HD void hd_touch(Ctx x, int h) { bit_set_shared(x.w->hdirty, h); }
CC4_HD uint32_t or_shared(uint32_t* p, uint32_t v) {   // returns the previous word
#if defined(__HIP_DEVICE_COMPILE__)
  return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  uint32_t o = *p; *p = o | v; return o;
#endif
}
// a 137-bit host bitmap held in registers: loaded / stored as one batch of independent LDS accesses (a `for w` loop that
// alternates loads and stores pays one LDS round trip per word)
struct B5 { uint32_t w[5]; };
CC4_HD B5 b5_load(const uint32_t* p) { B5 r; r.w[0] = p[0]; r.w[1] = p[1]; r.w[2] = p[2]; r.w[3] = p[3]; r.w[4] = p[4]; return r; }
CC4_HD void b5_store(uint32_t* p, const B5& v) { p[0] = v.w[0]; p[1] = v.w[1]; p[2] = v.w[2]; p[3] = v.w[3]; p[4] = v.w[4]; }
CC4_HD B5 b5_zero() { B5 r; r.w[0] = r.w[1] = r.w[2] = r.w[3] = r.w[4] = 0; return r; }
CC4_HD B5 b5_or(const B5& a, const B5& b) { B5 r; CC4_UNROLL for (int i = 0; i < 5; ++i) r.w[i] = a.w[i] | b.w[i]; return r; }
CC4_HD B5 b5_andn(const B5& a, const B5& b) { B5 r; CC4_UNROLL for (int i = 0; i < 5; ++i) r.w[i] = a.w[i] & ~b.w[i]; return r; }
CC4_HD bool b5_any(const B5& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3] | a.w[4]) != 0; }
CC4_HD void b5_set(B5& a, int i) {   // constant word indices only: a dynamically indexed register array would live in scratch memory
  const uint32_t bit = 1u << (i & 31); const int wi = i >> 5;
  CC4_UNROLL for (int k = 0; k < 5; ++k) a.w[k] |= (wi == k) ? bit : 0u;
}
CC4_HD int popc32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(v);
#else
  return __builtin_popcount(v);
#endif
}
CC4_HD int ctz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ffs((int)v) - 1;
#else
  return __builtin_ctz(v);
#endif
}
// position of the n-th (0-based) set bit of a small mask
CC4_HD int nth_bit(uint32_t m, int n) {
  for (int i = 0; i < n; ++i) m &= m - 1;
  return ctz32(m);
}
// index of the n-th (0-based) set bit of a multi-word bitmap
CC4_HD int nth_set(const uint32_t* bm, int nwords, int n) {
  for (int w = 0; w < nwords; ++w) {
    int c = popc32(bm[w]);
    if (n < c) return w * 32 + nth_bit(bm[w], n);
    n -= c;
  }
  return -1;
}
CC4_HD int last_set(const uint32_t* bm, int nwords) {
  for (int w = nwords - 1; w >= 0; --w) if (bm[w]) { uint32_t v = bm[w]; int b = 31; while (!((v >> b) & 1u)) --b; return w * 32 + b; }
  return -1;
}

// One bit of Host.ephemeral_ports: set it, tell whether it was set.  On the device every test-and-set of these words is one
// L2 atomic (the row is never staged): the serial walk and the lane-parallel green actions of the numpy-stream kernel
// (wave_green_exec) then agree on one coherence point, and the walk saves the separate store.
enum : uint32_t { EPH_RANGE = 60000 - 49152 };
CC4_HD bool eph_test_and_set(EnvCold* c, int h, uint32_t p) {
  uint32_t* w = &c->eph[h][p >> 5];
  const uint32_t bit = 1u << (p & 31);
#if defined(__HIP_DEVICE_COMPILE__)
  return (__hip_atomic_fetch_or(w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) != 0;
#else
  const bool was = (*w & bit) != 0; *w |= bit; return was;
#endif
}
// Host.get_ephemeral_port (Simulator/Host.py:175-187): one re-draw on collision, then remember the port.
CC4_HD int eph_port(Ctx x, int h, int salt = 0) {
  // Counter mode: the port value is unobservable on the flat-observation path, so nothing is drawn.  With the event log on
  // (dict observations) it is taken from a sibling of the acting agent's stream -- block (current draw position, salt) of stream
  // id | 0x8000 -- which consumes nothing of the agent's own stream: logging never changes a trajectory.  `salt` tells apart
  // the ports one action draws between two draws of its stream.  (No per-host uniqueness re-draw: agents resolved on different
  // lanes would race for it; a collision among 10848 ports is as unobservable as it is rare.)
  if (x.r->mode != 0) {
    if (!x.lg) return 49152;
    uint32_t c[4];
    rng_block(x.r, x.r->ndraw | 0x8000u, (uint32_t)(x.r->s_hi * 64u + (uint32_t)salt), c);
    return 49152 + (int)(((uint64_t)c[0] * (60000u - 49152u)) >> 32);
  }
  uint32_t p = rng_below(x.r, EPH_RANGE);
  if (eph_test_and_set(x.c, h, p)) { p = rng_below(x.r, EPH_RANGE); (void)eph_test_and_set(x.c, h, p); }
  return 49152 + (int)p;
}
CC4_HD void eph_clear(Ctx x, int h) {
  if (x.r->mode != 0) return;   // the bitmap only exists for numpy-stream parity (see eph_port)
  struct alignas(16) Q { uint32_t a, b, c, d; };       // EPH_WORDS * 4 = 1360 B = 85 x 16 B, 16-byte aligned rows
  Q* p = reinterpret_cast<Q*>(x.c->eph[h]);
  Q z; z.a = 0; z.b = 0; z.c = 0; z.d = 0;
  for (int i = 0; i < EPH_WORDS / 4; ++i) p[i] = z;
}
// ---- Host.processes: entries 0..PIN-1 in the hot row, PIN.. in the cold row (cold_povf).  Every scan reads eight
// records per round as independent word loads (pid | kind << 16 | flags << 24); PIN and the cold capacity are multiples of
// eight, so a round never straddles the two parts and may read allocated slots past the end of the list (masked by the index test).
CC4_HD int hd_nsvc(const HostDyn& d) { return d.nsf & 0xF; }
CC4_HD void hd_set_nsvc(HostDyn& d, int n) { d.nsf = (uint8_t)((d.nsf & 0xF0) | n); }
CC4_HD int hd_files(const HostDyn& d) { return d.nsf >> 4; }
CC4_HD void hd_set_files(HostDyn& d, int f) { d.nsf = (uint8_t)((d.nsf & 0x0F) | (f << 4)); }
struct P8 { uint32_t v[8]; };
CC4_HD const uint32_t* proc_round_ptr(Ctx x, int h, int i0) {
  return i0 < PIN ? reinterpret_cast<const uint32_t*>(x.hd[h].procs) + i0 : cold_povf(x.c, x.s->steps, h) + (i0 - PIN);
}
CC4_HD P8 proc_load8(Ctx x, int h, int i0) {
  const uint32_t* p = proc_round_ptr(x, h, i0);
  P8 q;
  CC4_UNROLL for (int k = 0; k < 8; ++k) q.v[k] = p[k];
  return q;
}
CC4_HD int pw_pid(uint32_t v) { return (int)(v & 0xFFFF); }
CC4_HD int pw_kind(uint32_t v) { return (int)((v >> 16) & 0xFF); }
CC4_HD int pw_flags(uint32_t v) { return (int)(v >> 24); }
CC4_HD uint32_t proc_get(Ctx x, int h, int i) { return proc_round_ptr(x, h, i & ~7)[i & 7]; }
// The list's length together with its first round, both from the host's own row (one cache line) and requested at once: a scan
// that reads the length first and the records after it spends two dependent round trips on what is nearly always (19 hosts in 20
// never exceed PIN processes) a one-round list.  Slots past the end are allocated and masked by the index test.
struct P8N { uint32_t v[8]; int n; int nsf; };   // nsf: HostDyn.nsf (service count | file flags), same line
CC4_HD P8 p8_of(const P8N& a) { P8 q; CC4_UNROLL for (int k = 0; k < 8; ++k) q.v[k] = a.v[k]; return q; }
CC4_HD P8N proc_head(Ctx x, int h) {
  const HostDyn& d = x.hd[h];
  P8N r;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(d.procs);
  CC4_UNROLL for (int k = 0; k < 8; ++k) r.v[k] = p[k];
  r.n = d.nproc;
  r.nsf = d.nsf;
  return r;
}
CC4_HD void proc_put(Ctx x, int h, int i, uint32_t v) { hd_touch(x, h); const_cast<uint32_t*>(proc_round_ptr(x, h, i & ~7))[i & 7] = v; }
// Host.create_pid (Simulator/Host.py:198-200)
CC4_HD int create_pid(Ctx x, int h, const P8N& hd0) {   // hd0: proc_head(x, h), when the caller holds it already
  int mx = 0;
  const int n = hd0.n;
  CC4_UNROLL for (int k = 0; k < 8; ++k) if (k < n && pw_pid(hd0.v[k]) > mx) mx = pw_pid(hd0.v[k]);
  for (int i0 = 8; i0 < n; i0 += 8) {
    const P8 q = proc_load8(x, h, i0);
    CC4_UNROLL for (int k = 0; k < 8; ++k) if (i0 + k < n && pw_pid(q.v[k]) > mx) mx = pw_pid(q.v[k]);
  }
  return mx + 1 + (int)rng_below(x.r, 9);
}
CC4_HD int create_pid(Ctx x, int h) { return create_pid(x, h, proc_head(x, h)); }
// n: the list's current length (HostDyn.nproc), which the caller holds
CC4_HD bool add_proc_n(Ctx x, int h, int n, int pid, int kind, int flags) {
  HostDyn& d = x.hd[h];
  if (n >= PIN && n >= PIN + cold_povf_cap(x.s->steps)) { set_err(x, E_PROC_OVERFLOW); return false; }
  proc_put(x, h, n, (uint32_t)pid | ((uint32_t)kind << 16) | ((uint32_t)flags << 24));
  d.nproc = (uint16_t)(n + 1);
  return true;
}
CC4_HD bool add_proc(Ctx x, int h, int pid, int kind, int flags) {
  HostDyn& d = x.hd[h];
  const int n = d.nproc;
  if (n >= PIN && n >= PIN + cold_povf_cap(x.s->steps)) { set_err(x, E_PROC_OVERFLOW); return false; }
  proc_put(x, h, n, (uint32_t)pid | ((uint32_t)kind << 16) | ((uint32_t)flags << 24));
  d.nproc = (uint16_t)(n + 1);
  return true;
}
CC4_HD int find_proc(Ctx x, int h, int pid, const P8N& hd0) {   // hd0: proc_head(x, h), when the caller holds it already
  const int n = hd0.n;
  {
    int hit = -1;
    CC4_UNROLL for (int k = 7; k >= 0; --k) if (k < n && pw_pid(hd0.v[k]) == pid) hit = k;
    if (hit >= 0) return hit;
  }
  for (int i0 = 8; i0 < n; i0 += 8) {
    const P8 q = proc_load8(x, h, i0);
    int hit = -1;
    CC4_UNROLL for (int k = 7; k >= 0; --k) if (i0 + k < n && pw_pid(q.v[k]) == pid) hit = i0 + k;
    if (hit >= 0) return hit;
  }
  return -1;
}
CC4_HD int find_proc(Ctx x, int h, int pid) { return find_proc(x, h, pid, proc_head(x, h)); }
// the union of the listening-port bits of the host's processes (Host.is_using_port over all ports at once)
CC4_HD int proc_ports(Ctx x, int h, const P8N& hd0) {   // hd0: proc_head(x, h), when the caller holds it already
  const int n = hd0.n;
  int used = 0;
  CC4_UNROLL for (int k = 0; k < 8; ++k) if (k < n) used |= kind_port(pw_kind(hd0.v[k]));
  for (int i0 = 8; i0 < n; i0 += 8) {
    const P8 q = proc_load8(x, h, i0);
    CC4_UNROLL for (int k = 0; k < 8; ++k) if (i0 + k < n) used |= kind_port(pw_kind(q.v[k]));
  }
  return used;
}
CC4_HD int proc_ports(Ctx x, int h) { return proc_ports(x, h, proc_head(x, h)); }
// records idx+1 .. n-1 move down by one: per round, the round and the first record of the next one are read before the
// round is rewritten
CC4_HD void remove_proc_at(Ctx x, int h, int idx) {
  hd_touch(x, h);
  HostDyn& d = x.hd[h];
  const int n = d.nproc;
  for (int i0 = idx & ~7; i0 < n; i0 += 8) {
    const P8 q = proc_load8(x, h, i0);
    const uint32_t nxt = (i0 + 8 < n) ? proc_get(x, h, i0 + 8) : 0u;
    uint32_t* p = const_cast<uint32_t*>(proc_round_ptr(x, h, i0));
    CC4_UNROLL for (int k = 0; k < 8; ++k) {
      const int i = i0 + k;
      if (i < idx || i + 1 >= n) continue;
      p[k] = k < 7 ? q.v[(k + 1) & 7] : nxt;
    }
  }
  d.nproc = (uint16_t)(n - 1);
}
// host.events.network_connections.append(...)
CC4_HD void ev_conn(Ctx x, int h) { ev_or(x, h, EV_CUR_CONN); }
// host.events.process_creation.append(...); pid > 0 when the event dict carries 'pid' (ExploitAction.py:264-275)
CC4_HD void ev_proc(Ctx x, int h, int pid) {
  ev_or(x, h, EV_CUR_PROC);
  if (pid > 0 && blue_of_subnet(h_subnet(h)) >= 0) {
    if (x.s->npend >= MAX_PEND) { set_err(x, E_PEND_OVERFLOW); return; }
    x.s->pend[x.s->npend++] = ((uint32_t)h << 16) | (uint32_t)pid;
  }
}
// red exploit variant: the pid-carrying event goes to the agent's own slot so that red agents can be resolved on different
// waves; step_red_merge() appends the slots in agent order (== the serial append order)
CC4_HD void ev_proc_red(Ctx x, int r, int h, int pid) {
  ev_or(x, h, EV_CUR_PROC);
  if (blue_of_subnet(h_subnet(h)) >= 0) x.w->pend_r[r] = ((uint32_t)h << 16) | (uint32_t)pid;
}
CC4_HD void step_red_merge(Ctx x) {
  EnvState* s = x.s;
  uint32_t pr[NRED];
  CC4_UNROLL for (int r = 0; r < NRED; ++r) pr[r] = x.w->pend_r[r];
  CC4_UNROLL for (int r = 0; r < NRED; ++r) {
    if (!pr[r]) continue;
    if (s->npend >= MAX_PEND) set_err(x, E_PEND_OVERFLOW); else s->pend[s->npend++] = pr[r];
    x.w->pend_r[r] = 0;
  }
}
CC4_HD void pend_drop_host(Ctx x, int h) {
  EnvState* s = x.s;
  int n = 0;
  for (int i = 0; i < s->npend; ++i) if ((int)(s->pend[i] >> 16) != h) s->pend[n++] = s->pend[i];
  s->npend = (uint8_t)n;
}
// RemoteAction.blocking_host (Simulator/Actions/Action.py:126-135): only subnet-level blocks exist in CC4
CC4_HD bool subnet_blocked(Ctx x, int src_sub, int other_sub) { return (x.s->blocks[other_sub] >> src_sub) & 1u; }

// ------------------------------------------------------------------ red sessions
// state.sessions[red_agent_r] is RedAgent.sord: pool slots in dict order.  A scan reads eight list bytes as one word, then the
// eight records with independent 8-byte loads (two LDS round trips per eight sessions).  A record as a little-endian word:
// id | pid << 16 | host << 32 | flags << 40.  MAX_RS is a multiple of 8 and list bytes are always valid slots, so a round may
// read past nsess; those lanes are masked by the index test.
struct S8 { uint64_t v[8]; uint64_t slots; };
CC4_HD uint64_t rs_word(const EnvState* s, int slot) { uint64_t v; __builtin_memcpy(&v, &s->spool[slot], 8); return v; }
CC4_HD void rs_word_put(EnvState* s, int slot, uint64_t v) { __builtin_memcpy(&s->spool[slot], &v, 8); }
CC4_HD S8 rs_load8(const EnvState* s, const RedAgent& a, int i0) {
  S8 q;
  __builtin_memcpy(&q.slots, &a.sord[i0], 8);
  CC4_UNROLL for (int k = 0; k < 8; ++k) q.v[k] = rs_word(s, (int)((q.slots >> (8 * k)) & 0xFF));
  return q;
}
CC4_HD int s8_slot(const S8& q, int k) { return (int)((q.slots >> (8 * k)) & 0xFF); }
CC4_HD int rsw_id(uint64_t v) { return (int)(v & 0xFFFF); }
CC4_HD int rsw_pid(uint64_t v) { return (int)((v >> 16) & 0xFFFF); }
CC4_HD int rsw_host(uint64_t v) { return (int)((v >> 32) & 0xFF); }
CC4_HD int rsw_flags(uint64_t v) { return (int)((v >> 40) & 0xFF); }
CC4_HD uint64_t rsw_make(int id, int pid, int host, int flags) {
  return (uint64_t)(id & 0xFFFF) | ((uint64_t)(pid & 0xFFFF) << 16) | ((uint64_t)(host & 0xFF) << 32) | ((uint64_t)(flags & 0xFF) << 40);
}
// the i-th session of the agent (list position -> record)
CC4_HD uint64_t rs_at(const EnvState* s, const RedAgent& a, int i) { return rs_word(s, a.sord[i]); }
CC4_HD int rs_find_id(const EnvState* s, const RedAgent& a, int id, int n) {
  for (int i0 = 0; i0 < n; i0 += 8) {
    const S8 q = rs_load8(s, a, i0);
    int hit = -1;
    CC4_UNROLL for (int k = 7; k >= 0; --k) if (i0 + k < n && rsw_id(q.v[k]) == id) hit = i0 + k;
    if (hit >= 0) return hit;
  }
  return -1;
}
CC4_HD int rs_find_id(const EnvState* s, const RedAgent& a, int id) { return rs_find_id(s, a, id, a.h.nsess); }
// sessions of the agent on host h: how many, the first one, the first root one
struct HostSess { int n, first, first_root; };
CC4_HD HostSess rs_on_host(const EnvState* s, const RedAgent& a, int h) {
  HostSess r; r.n = 0; r.first = -1; r.first_root = -1;
  const int n = a.h.nsess;
  for (int i0 = 0; i0 < n; i0 += 8) {
    const S8 q = rs_load8(s, a, i0);
    CC4_UNROLL for (int k = 0; k < 8; ++k)
      if (i0 + k < n && rsw_host(q.v[k]) == h) {
        r.n++;
        if (r.first < 0) r.first = i0 + k;
        if (r.first_root < 0 && (rsw_flags(q.v[k]) & RS_ROOT)) r.first_root = i0 + k;
      }
  }
  return r;
}
// index of the k-th (0-based) session on host h, or -1
CC4_HD int rs_kth_on_host(const EnvState* s, const RedAgent& a, int h, int kth) {
  const int n = a.h.nsess;
  for (int i0 = 0; i0 < n; i0 += 8) {
    const S8 q = rs_load8(s, a, i0);
    int hit = -1;
    CC4_UNROLL for (int k = 0; k < 8; ++k)
      if (hit < 0 && i0 + k < n && rsw_host(q.v[k]) == h) { if (kth == 0) hit = i0 + k; kth--; }
    if (hit >= 0) return hit;
  }
  return -1;
}
// index of the session with this (host, pid), or -1 (State.get_session_from_pid)
CC4_HD int rs_find_host_pid(const EnvState* s, const RedAgent& a, int h, int pid) {
  const int n = a.h.nsess;
  for (int i0 = 0; i0 < n; i0 += 8) {
    const S8 q = rs_load8(s, a, i0);
    int hit = -1;
    CC4_UNROLL for (int k = 7; k >= 0; --k) if (i0 + k < n && rsw_host(q.v[k]) == h && rsw_pid(q.v[k]) == pid) hit = i0 + k;
    if (hit >= 0) return hit;
  }
  return -1;
}
// list entries idx+1 .. nsess-1 move down by one: the 64-byte list is eight words, read as one batch and funnel-shifted
CC4_HD void rs_list_remove(RedAgent& a, int idx) {
  uint64_t w[MAX_RS / 8];
  __builtin_memcpy(w, a.sord, MAX_RS);
  const int wi = idx >> 3, sh = 8 * (idx & 7);
  CC4_UNROLL for (int k = 0; k < MAX_RS / 8; ++k) {
    if (k < wi) continue;
    const uint64_t nxt = k + 1 < MAX_RS / 8 ? w[k + 1] : 0ull;
    uint64_t v = (w[k] >> 8) | (nxt << 56);                                        // the whole word moves down one byte
    if (k == wi && sh) v = (w[k] & ((1ull << sh) - 1ull)) | (v & ~((1ull << sh) - 1ull));   // bytes below idx stay
    w[k] = v;
  }
  __builtin_memcpy(a.sord, w, MAX_RS);
}
// A free pool record.  Allocation order must not depend on how the red agents are spread over waves, so the one session an
// agent's exploit may create in a step gets its slot before the red actions run (rs_reserve, StepWork.rs_slot); every other
// creation site (reset, PhishingEmail) runs on one thread and takes the lowest free slot.
CC4_HD int rs_alloc_lowest(Ctx x) {
  for (int w = 0; w < RS_POOL / 32; ++w) {
    const uint32_t fr = ~x.s->spool_used[w];
    if (fr) return w * 32 + ctz32(fr);
  }
  return -1;
}
CC4_HD void rs_reserve(Ctx x) {
  uint32_t used[RS_POOL / 32];
  CC4_UNROLL for (int w = 0; w < RS_POOL / 32; ++w) used[w] = x.s->spool_used[w];
  int ty[NRED];
  CC4_UNROLL for (int r = 0; r < NRED; ++r) ty[r] = x.s->rexec[r].type;
  CC4_UNROLL for (int r = 0; r < NRED; ++r) {
    int slot = 0xFF;
    if (ty[r] == RA_EXPLOIT) {
      CC4_UNROLL for (int w = 0; w < RS_POOL / 32; ++w) {
        const uint32_t fr = ~used[w];
        if (slot == 0xFF && fr) { slot = w * 32 + ctz32(fr); used[w] |= 1u << (slot & 31); }
      }
    }
    x.w->rs_slot[r] = (uint8_t)slot;
  }
}
// State.add_session (Simulator/State.py:305-324): ident = max(existing)+1 (0 if none); appended (dict order).
// slot: the pool record to use (< 0: lowest free).  Returns the list index, or -1.
CC4_HD int rs_add(Ctx x, int r, int host, int pid, int flags, int slot = -1) {
  EnvState* s = x.s;
  RedAgent& a = s->red[r];
  if (slot < 0) slot = rs_alloc_lowest(x);
  if (a.h.nsess >= MAX_RS || slot < 0 || slot >= RS_POOL) { set_err(x, E_RSESS_OVERFLOW); return -1; }
  int id = 0;
  for (int i0 = 0; i0 < a.h.nsess; i0 += 8) {
    const S8 q = rs_load8(s, a, i0);
    CC4_UNROLL for (int k = 0; k < 8; ++k) if (i0 + k < a.h.nsess && rsw_id(q.v[k]) + 1 > id) id = rsw_id(q.v[k]) + 1;
  }
  (void)or_shared(&s->spool_used[slot >> 5], 1u << (slot & 31));
  rs_word_put(s, slot, rsw_make(id, pid, host, flags));
  if (flags & RS_ABSTRACT) {   // a fresh RedAbstractSession knows no ports
    uint64_t* row = reinterpret_cast<uint64_t*>(x.c->kports[slot]);   // 144-byte rows, 8-byte aligned
    for (int k = 0; k < (MAXH + 7) / 8; ++k) row[k] = 0;
  }
  a.sord[a.h.nsess++] = (uint8_t)slot;
  a.h.rsc_dirty = 1; a.h.fsm_dirty = 1;
  if (!bit_get(a.live_hosts, host)) { bit_set(a.live_hosts, host); a.h.nlive++; }
  bit_set_shared(s->red_hosts, host);
  return a.h.nsess - 1;
}
// the agent no longer holds a session on `gone`: its own bitmap, and the episode's if no other agent does either
CC4_HD void rs_host_left(Ctx x, int r, int gone) {
  RedAgent& a = x.s->red[r];
  bit_clr(a.live_hosts, gone); a.h.nlive--;
  uint32_t lw[NRED];   // the six agents' words for that host in one batch of loads (this agent's bit is already clear)
  CC4_UNROLL for (int q = 0; q < NRED; ++q) lw[q] = x.s->red[q].live_hosts[gone >> 5];
  uint32_t any_other = 0;
  CC4_UNROLL for (int q = 0; q < NRED; ++q) any_other |= lw[q];
  if (!((any_other >> (gone & 31)) & 1u)) bit_clr_shared(x.s->red_hosts, gone);
}
// the list entry goes; the pool record is released unless it moves on to another agent (keep_record)
CC4_HD void rs_remove_at(Ctx x, int r, int idx, bool keep_record = false) {
  EnvState* s = x.s;
  RedAgent& a = s->red[r];
  const int slot = a.sord[idx];
  const int gone = rsw_host(rs_word(s, slot));
  if (!keep_record) bit_clr_shared(s->spool_used, slot);
  rs_list_remove(a, idx);
  a.h.nsess--;
  a.h.rsc_dirty = 1; a.h.fsm_dirty = 1;
  if (rs_on_host(s, a, gone).n == 0) rs_host_left(x, r, gone);
}
// Every non-original session of agent r on host h, in one compaction pass (== rs_remove_at on each of them in list order:
// the survivors keep their order).  RestoreFromBackup.
CC4_HD void rs_remove_on_host(Ctx x, int r, int h) {
  EnvState* s = x.s;
  RedAgent& a = s->red[r];
  const int n = a.h.nsess;
  int out = 0, removed = 0;
  bool left = false;
  for (int i0 = 0; i0 < n; i0 += 8) {
    const S8 q = rs_load8(s, a, i0);      // the round is in registers before any of its list bytes is overwritten (out <= i0 + k)
    CC4_UNROLL for (int k = 0; k < 8; ++k) {
      if (i0 + k >= n) continue;
      const bool here = rsw_host(q.v[k]) == h;
      if (here && !(rsw_flags(q.v[k]) & RS_ORIG)) { bit_clr_shared(s->spool_used, s8_slot(q, k)); removed++; continue; }
      if (here) left = true;
      if (out != i0 + k) a.sord[out] = (uint8_t)s8_slot(q, k);
      out++;
    }
  }
  if (!removed) return;
  a.h.nsess = (uint8_t)out;
  a.h.rsc_dirty = 1; a.h.fsm_dirty = 1;
  if (!left) rs_host_left(x, r, h);
}
// dict pop + re-insert of the same session object (RedSessionCheck promotion, RestoreFromBackup of an original session):
// the entry moves to the end of the agent's order; which hosts hold sessions does not change
CC4_HD void rs_move_to_end(EnvState* s, RedAgent& a, int idx, int new_id) {
  const int slot = a.sord[idx];
  rs_list_remove(a, idx);
  a.sord[a.h.nsess - 1] = (uint8_t)slot;
  if (new_id >= 0) rs_word_put(s, slot, (rs_word(s, slot) & ~0xFFFFull) | (uint64_t)(new_id & 0xFFFF));
  a.h.rsc_dirty = 1;
}
// ks = the agent's ordered id list (EnvCold.known_sid[r]): read only for ids the bitmap does not cover
CC4_HD bool sid_known(const RedAgent& a, const uint16_t* ks, int id, int nknown) {
  if (id < 256) return bit_get(a.known_bm, id);
  for (int i = 0; i < nknown; ++i) if (ks[i] == id) return true;
  return false;
}
// ActionSpace.update: server_session[session_id] = True (Shared/ActionSpace.py:205-211)
CC4_HD void as_know_sid(Ctx x, int r, int id) {
  RedAgent& a = x.s->red[r];
  if (sid_known(a, x.c->known_sid[r], id, a.h.nknown)) return;
  if (a.h.nknown >= MAX_KS) { set_err(x, E_KS_OVERFLOW); return; }
  x.c->known_sid[r][a.h.nknown++] = (uint16_t)id;
  if (id < 256) bit_set(a.known_bm, id);
}
// one key of the agent's step observation (Shared/Observation.py add_* / combine_obs); also applies the
// ActionSpace.update side effects (ip / hostname / subnet known) eagerly -- nothing reads them before step end.
CC4_HD void obs_put(Ctx x, int r, bool key_ip, int host, int flags, bool subnet_known) {
  RedAgent& a = x.s->red[r];
  if (flags & OE_IFACE) bit_set(a.as_ip, host);
  if (flags & OE_SYSHN) bit_set(a.as_hn, host);
  if (subnet_known) a.h.as_subnet |= (uint16_t)(1u << h_subnet(host));
  uint8_t want = (uint8_t)(key_ip ? OE_KEY_IP : 0);
  uint32_t* has = a.obs_has[key_ip ? 1 : 0];
  if (bit_get(has, host)) {
    for (int i = 0; i < a.h.nobs; ++i)
      if (a.obs[i].host == host && (a.obs[i].flags & OE_KEY_IP) == want) { a.obs[i].flags |= (uint8_t)flags; return; }
  }
  bit_set(has, host);
  if (a.h.nobs >= MAX_OBS) { set_err(x, E_OBS_OVERFLOW); return; }
  a.obs[a.h.nobs].host = (uint8_t)host;
  a.obs[a.h.nobs].flags = (uint8_t)(flags | want);
  a.h.nobs++;
}
// observations[0] of the agent this step decides what the FSM sees as (action, success)
CC4_HD void obs_first(Ctx x, int r, int success, int atype, int ahost, int aarg) {
  RedAgent& a = x.s->red[r];
  if (a.h.obs_success != 0) return;
  a.h.obs_success = (uint8_t)success; a.h.obs_act_type = (uint8_t)atype; a.h.obs_act_host = (uint8_t)ahost; a.h.obs_act_arg = (uint8_t)aarg;
}

// ------------------------------------------------------------------ routes (tree of routers)
// RemoteAction.get_route (Action.py:100-116) on the fixed link diagram (EnterpriseScenarioGenerator.py:373-416):
// the graph is a tree, so the route is the unique path src .. dst (inclusive).
CC4_HD int path_to_root(int h, uint8_t* out) {
  int n = 0;
  out[n++] = (uint8_t)h;
  if (h == H_INTERNET) return n;
  int s = h_subnet(h);
  if (!h_is_router(h)) out[n++] = (uint8_t)h_make(s, 0);
  while (true) {
    int p = router_parent(s);
    if (p == S_INT) { out[n++] = (uint8_t)H_INTERNET; break; }
    out[n++] = (uint8_t)h_make(p, 0);
    s = p;
  }
  return n;
}
// work: >= 24 bytes (not a private array: see EnvState.scratch); the hops end up in work[12 .. 12+n)
CC4_HD int route(int src, int dst, uint8_t* work) {
  uint8_t* a = work; uint8_t* b = work + 6; uint8_t* out = work + 12;
  int na = path_to_root(src, a), nb = path_to_root(dst, b);
  // both end at the root; drop the shared tail but keep the lowest common ancestor
  while (na >= 2 && nb >= 2 && a[na - 2] == b[nb - 2]) { na--; nb--; }
  int n = 0;
  for (int i = 0; i < na; ++i) out[n++] = a[i];
  for (int i = nb - 2; i >= 0; --i) out[n++] = b[i];
  return n;
}

// ------------------------------------------------------------------ reset: EnterpriseScenarioGenerator + State.__init__
// Draw order follows create_scenario (EnterpriseScenarioGenerator.py:123-169) then State.__init__ (State.py:66-148).
CC4_HD int gen_pid(Ctx x, uint32_t* used) {  // _generate_pid (ESG.py:564-578)
  while (true) {
    int pid = rng_range(x.r, 1000, 10000);
    if (!bit_get(used, pid - 1000)) { bit_set(used, pid - 1000); return pid; }
  }
}
// The scenario is generated into the dynamic host rows (EnvState.hd: LDS on the device), and the backup images
// (Host.create_backup -> EnvCold.hs, HBM) are written from them once, with stores only: generating into the cold row
// made every read-modify-write of the build a global-memory round trip.
CC4_HD void gen_host(Ctx x, int h, uint32_t* used) {  // _generate_linux_host (ESG.py:470-528)
  EnvState* s = x.s;
  HostDyn& st = x.hd[h];
  st.nproc = 0; st.nsf = 0;
  bit_set(s->exists, h);
  st.gtmp = (uint8_t)rng_below(x.r, 2);  // OSDistribution choice (ESG.py:488-494): 0 UBUNTU, 1 KALI; parked in gtmp until the backup is written
  if (h_is_router(h)) return;
  // _generate_linux_host_services (ESG.py:530-562); services dict order = SSHD, [OTSERVICE], chosen add-ons
  int n = 0;
  auto put = [&](int kind, int pid) {
    st.svcs[n].kind = (uint8_t)kind; st.svcs[n].pid = (uint16_t)pid; st.svcs[n].st = (uint8_t)(SV_ACTIVE | 5);
    st.procs[n].kind = (uint8_t)kind; st.procs[n].pid = (uint16_t)pid; st.procs[n].flags = 0;
    n++;
  };
  put(K_SSHD, gen_pid(x, used));
  int sub = h_subnet(h);
  if (sub == S_OZA || sub == S_OZB) put(K_OT, gen_pid(x, used));
  int p_apache = gen_pid(x, used), p_mysql = gen_pid(x, used), p_smtp = gen_pid(x, used);  // all three are always drawn
  int n_add = (int)rng_below(x.r, 4);  // integers(0, 3, endpoint=True)
  uint32_t left = 7;                    // remaining add-on options {APACHE2, MYSQLD, SMTP} as a bit list
  for (int k = 0; k < n_add; ++k) {
    int c = (int)rng_below(x.r, (uint32_t)popc32(left));
    int o = nth_bit(left, c);
    left &= ~(1u << o);
    put(o == 0 ? K_APACHE : (o == 1 ? K_MYSQL : K_SMTP), o == 0 ? p_apache : (o == 1 ? p_mysql : p_smtp));
  }
  // _generate_linux_host_processes (ESG.py:580-629): one random() per service, never below 1.0
  for (int i = 0; i < n; ++i) (void)rng_random(x.r);
  hd_set_nsvc(st, n); st.nproc = (uint16_t)n;
}
// Host.add_session for a starting session (Host.py:189-196): Process(pid=create_pid(), name=session_type)
CC4_HD int start_session_proc(Ctx x, int h, int kind) {   // at generation time a host holds fewer than PIN processes
  HostDyn& st = x.hd[h];
  int mx = 0;
  for (int i = 0; i < st.nproc; ++i) if (st.procs[i].pid > mx) mx = st.procs[i].pid;
  int pid = mx + 1 + (int)rng_below(x.r, 9);
  st.procs[st.nproc].pid = (uint16_t)pid; st.procs[st.nproc].kind = (uint8_t)kind; st.procs[st.nproc].flags = 0;
  st.nproc++;
  return pid;
}
// Host.create_backup (Host.py:316-371): the freshly generated dynamic row becomes the backup image (stores only)
CC4_HD void host_backup(Ctx x, int h, int ip_octet) {
  const HostDyn& d = x.hd[h];
  HostStatic st;
  for (int i = 0; i < 8; ++i) st.procs[i] = d.procs[i];
  for (int i = 0; i < 5; ++i) st.svcs[i] = d.svcs[i];
  st.nproc = (uint8_t)d.nproc; st.nsvc = (uint8_t)hd_nsvc(d); st.exists = (uint8_t)(1 | ((d.gtmp & 1) << 1)); st.ip_octet = (uint8_t)ip_octet;
  if (d.nproc > PIN - 1 || hd_nsvc(d) > 5) set_err(x, E_PROC_OVERFLOW);   // slot PIN-1 carried the address during generation
  __builtin_memcpy(&x.c->hs[h], &st, sizeof(HostStatic));
}
CC4_HD void host_restore(Ctx x, int h) {  // Host.restore (Host.py:373-429)
  hd_touch(x, h);
  HostDyn& d = x.hd[h];
  // the backup image lives in the cold row (HBM): fetch its 56 bytes with independent wide loads, then unpack
  HostStatic st;
  __builtin_memcpy(&st, &x.c->hs[h], sizeof(HostStatic));   // 8-byte aligned POD -> 7 wide loads in flight
  for (int i = 0; i < 8; ++i) d.procs[i] = st.procs[i];
  for (int i = 0; i < 5; ++i) d.svcs[i] = st.svcs[i];
  d.nproc = st.nproc; d.gtmp = 0; d.nsf = st.nsvc;   // the cold part of the list is simply abandoned; Host.files is cleared
  x.s->hev[h] = 0;                                  // self.events = HostEvents()
  eph_clear(x, h);
  pend_drop_host(x, h);
}

// ------------------------------------------------------------------ counter-mode scenario generation, in phases
// The numpy-stream mode above is serial by definition (one shared generator).  In the counter-based mode every host draws
// from its own streams (ST_GEN_HOST / ST_GEN_REDRAW / ST_GEN_SESS + host id), so the generation is cut into phases whose
// per-host parts are independent: the device runs them one host per thread, env_reset runs the same phases as loops (the
// oracle), and both leave identical bytes.  Same scenario distribution as _generate_* (ESG.py:171-817); pids are unique
// network-wide as in _generate_pid: a pid drawn by several services stays with the first of them in host order, the
// others draw again.
struct ResetCarry { uint64_t env_key; };
enum : int { RESET_WS_SEEN = 0, RESET_WS_DUP = 288, RESET_WS_HOSTS = 576, RESET_WS_WORDS = 584 };   // work area (LDS on the device)

// phase 0, all threads: clear the row (except the generator) and the backup images
CC4_HD void reset_zero(EnvState* s, HostDyn* hd, EnvCold* c, int t, int nt) {
  static_assert(offsetof(EnvState, rng) == 0, "the generator leads the row");
  uint32_t* w = (uint32_t*)s;
  for (size_t i = sizeof(Rng) / 4 + (size_t)t; i < offsetof(EnvState, hd) / 4; i += (size_t)nt) w[i] = 0;
  uint32_t* hw = (uint32_t*)hd;
  for (size_t i = (size_t)t; i < sizeof(HostDyn) * MAXH / 4; i += (size_t)nt) hw[i] = 0;
  uint32_t* b = (uint32_t*)c->hs;
  for (size_t i = (size_t)t; i < sizeof(c->hs) / 4; i += (size_t)nt) b[i] = 0;
}
// the 9000-bit set of the generation's pids: lent from the episode's session pool (zero after reset_zero; no session exists before
// reset_finish), cleared again by all threads (reset_used_clear) once the pids are settled
static_assert(sizeof(RSess) * RS_POOL >= 4 * 282, "the idle session pool holds the 9000-bit pid set of the generation");
CC4_HD uint32_t* reset_used_set(EnvState* s) { return reinterpret_cast<uint32_t*>(s->spool); }
CC4_HD void reset_used_clear(EnvState* s, int t, int nt) { uint32_t* u = reset_used_set(s); for (int i = t; i < 282; i += nt) u[i] = 0; }
// phase 1, one thread: generator, mission phases, subnets, host counts and addresses (main reset stream)
CC4_HD ResetCarry reset_topology(Ctx x, uint64_t seed, int steps, bool continue_stream, int policy, uint32_t topo_seed, uint32_t* ws,
                                 bool rng_is_copy) {
  EnvState* s = x.s;
  if (!continue_stream) rng_seed(&s->rng, seed, 1u);   // a fresh seed needs x.r == &s->rng (see env_reset)
  s->rng_mode = 1;
  s->policy = (uint8_t)policy;
  rng_begin_episode(x.r);
  ResetCarry k; k.env_key = x.r->s_lo;
  if (topo_seed) x.r->s_lo = (uint64_t)topo_seed;
  s->steps = steps;
  { int q = steps / 3, rem = steps % 3; s->phase_len[0] = q + (rem >= 1 ? 1 : 0); s->phase_len[1] = q + (rem == 2 ? 1 : 0); s->phase_len[2] = q; }
  for (int i = 0; i < RESET_WS_WORDS; ++i) ws[i] = 0;
  {
    uint32_t* avail = x.w->scratch;
    for (int i = 0; i < 8; ++i) avail[i] = 0xFFFFFFFFu;
    int n = 256;
    for (int sn = 0; sn < NSUB; ++sn) {
      int v = nth_set(avail, 8, (int)rng_below(x.r, (uint32_t)n));
      s->cidr_octet[sn] = (uint8_t)v;
      bit_clr(avail, v);
      n--;
    }
  }
  for (int sn = 0; sn < NSUB; ++sn) {
    uint32_t* ips = x.w->scratch;
    for (int i = 0; i < 8; ++i) ips[i] = 0xFFFFFFFFu;
    ips[0] &= ~1u; ips[7] &= 0x7FFFFFFFu;
    int n = 254;
    auto place = [&](int h, int v) { bit_clr(ips, v); n--; bit_set(s->exists, h); x.hd[h].procs[PIN - 1].pid = (uint16_t)v; };   // ip parked in the last inline process slot until the backup
    if (sn == S_INT) { place(H_INTERNET, nth_set(ips, 8, (int)rng_below(x.r, (uint32_t)n))); continue; }
    place(h_make(sn, 0), nth_set(ips, 8, (int)rng_below(x.r, (uint32_t)n)));
    int nu = 3 + (int)rng_below(x.r, 8);
    for (int i = 0; i < nu; ++i) place(h_make(sn, 1 + i), nth_set(ips, 8, (int)rng_below(x.r, (uint32_t)n)));
    int ns = 1 + (int)rng_below(x.r, 6);
    for (int i = 0; i < ns; ++i) place(h_make(sn, 11 + i), last_set(ips, 8));
    s->n_users[sn] = (uint8_t)nu; s->n_servers[sn] = (uint8_t)ns;
  }
  if (rng_is_copy) s->rng = *x.r;   // the per-host generators fork from the row (key, episode, reset step word)
  return k;
}
// phase 2, per host: services and candidate pids (_generate_linux_host, ESG.py:470-629) from the host's own stream
CC4_HD void reset_gen_host(Ctx x, int h) {
  EnvState* s = x.s;
  if (!bit_get(s->exists, h)) return;
  rng_set_stream(x.r, ST_GEN_HOST + (uint32_t)h);
  HostDyn& st = x.hd[h];
  st.gtmp = (uint8_t)rng_below(x.r, 2);   // OSDistribution, parked in gtmp until the backup
  if (h_is_router(h)) return;
  int n = 0;
  auto put = [&](int kind, int pid) {
    st.svcs[n].kind = (uint8_t)kind; st.svcs[n].pid = (uint16_t)pid; st.svcs[n].st = (uint8_t)(SV_ACTIVE | 5);
    st.procs[n].kind = (uint8_t)kind; st.procs[n].pid = (uint16_t)pid; st.procs[n].flags = 0;
    n++;
  };
  put(K_SSHD, rng_range(x.r, 1000, 10000));
  int sub = h_subnet(h);
  if (sub == S_OZA || sub == S_OZB) put(K_OT, rng_range(x.r, 1000, 10000));
  int p_apache = rng_range(x.r, 1000, 10000), p_mysql = rng_range(x.r, 1000, 10000), p_smtp = rng_range(x.r, 1000, 10000);
  int n_add = (int)rng_below(x.r, 4);
  uint32_t left = 7;
  for (int k = 0; k < n_add; ++k) {
    int o = nth_bit(left, (int)rng_below(x.r, (uint32_t)popc32(left)));
    left &= ~(1u << o);
    put(o == 0 ? K_APACHE : (o == 1 ? K_MYSQL : K_SMTP), o == 0 ? p_apache : (o == 1 ? p_mysql : p_smtp));
  }
  for (int i = 0; i < n; ++i) (void)rng_random(x.r);
  hd_set_nsvc(st, n); st.nproc = (uint16_t)n;
  // the three add-on candidates in draw order, installed or not, for the uniqueness pass (reset_pid_serial clears them again): a
  // freshly generated host has at most five services and five processes, slots 5 and 6 are free
  st.procs[5].pid = (uint16_t)p_apache; st.procs[6].pid = (uint16_t)p_mysql; st.svcs[6].pid = (uint16_t)p_smtp;
}
// phase 3, one thread: the network-wide pid uniqueness of _generate_pid (ESG.py:564-578), in the REFERENCE'S ORDER.  The reference
// draws the pids host after host -- SSHD, [OT service], then the three add-on candidates apache / mysql / smtp, installed or not
// (ESG.py:540-556) -- each against the set of pids drawn so far: a value that is already in the set is drawn again, at once, by the
// same caller.  Here the hosts drew their candidates side by side (reset_gen_host, each from its own stream); this pass walks
// them in the same host / draw order with the same running set: the first holder of a value keeps it, a later one draws again --
// from the re-draw stream of ITS host (ST_GEN_REDRAW + h), so that nothing else the host drew moves -- until it finds a value no
// EARLIER position holds.  (r03 resolved against the installed pids of all hosts at once: the same marginals, but another
// trajectory than the reference's whenever a re-draw lands on a later host's value, or an uninstalled candidate collides.)
// `used`: a 9000-bit set, >= 282 words, zeroed by the caller (the idle session pool of the row lends it: no session exists yet).
CC4_HD void reset_pid_serial(Ctx x, uint32_t* used) {
  EnvState* s = x.s;
  for (int h = 0; h < MAXH; ++h) {
    if (!bit_get(s->exists, h) || h_is_router(h)) continue;
    HostDyn& st = x.hd[h];
    const int n = hd_nsvc(st);
    const int sub = h_subnet(h);
    const int nfix = (sub == S_OZA || sub == S_OZB) ? 2 : 1;          // SSHD, [OT]: services 0 .. nfix-1
    Rng t; bool forked = false;
    auto settle = [&](int v0, bool* moved) {
      int v = v0 - 1000;
      *moved = false;
      while (bit_get(used, v)) {
        if (!forked) { rng_fork(&t, x.r, ST_GEN_REDRAW + (uint32_t)h); forked = true; }
        v = (int)rng_below(&t, 9000);
        *moved = true;
      }
      bit_set(used, v);
      return v + 1000;
    };
    for (int i = 0; i < nfix; ++i) {
      bool moved;
      const int v = settle(st.svcs[i].pid, &moved);
      if (moved) { st.svcs[i].pid = (uint16_t)v; st.procs[i].pid = (uint16_t)v; }
    }
    // the add-on candidates, parked by reset_gen_host in draw order: apache in procs[5].pid, mysql in procs[6].pid, smtp in svcs[6].pid
    for (int k = 0; k < 3; ++k) {
      const int kind = k == 0 ? K_APACHE : (k == 1 ? K_MYSQL : K_SMTP);
      const int v0 = k == 0 ? st.procs[5].pid : (k == 1 ? st.procs[6].pid : st.svcs[6].pid);
      bool moved;
      const int v = settle(v0, &moved);
      if (moved) for (int i = nfix; i < n; ++i) if (st.svcs[i].kind == kind) { st.svcs[i].pid = (uint16_t)v; st.procs[i].pid = (uint16_t)v; }
    }
    st.procs[5].pid = 0; st.procs[6].pid = 0; st.svcs[6].pid = 0;
  }
}
// phase 4, one thread: blue parents, green agents, red start hosts (_generate_blue/green/red_agents; main reset stream)
CC4_HD void reset_agents(Ctx x) {
  EnvState* s = x.s;
  for (int b = 0; b < NBLUE; ++b) {
    int nsub = blue_nsub(b);
    (void)rng_below(x.r, (uint32_t)nsub);
    int cnt = 0;
    for (int i = 0; i < nsub; ++i) { int sn = blue_subnet_alloc(b, i); cnt += 1 + s->n_users[sn] + s->n_servers[sn]; }
    int c = (int)rng_below(x.r, (uint32_t)cnt);
    int ph = -1;
    for (int i = 0; i < nsub && ph < 0; ++i) {   // existing slots of a subnet: 0, 1..nu, 11..11+ns-1
      int sn = blue_subnet_alloc(b, i), nu = s->n_users[sn], ns = s->n_servers[sn];
      if (c < 1 + nu) ph = h_make(sn, c);
      else if (c < 1 + nu + ns) ph = h_make(sn, 11 + (c - 1 - nu));
      else c -= 1 + nu + ns;
    }
    s->blue[b].parent_host = (uint8_t)ph;
  }
  {
    int g = 0;
    for (int sn = 0; sn < NSUB - 1; ++sn) for (int i = 0; i < s->n_users[sn]; ++i) s->green_host[g++] = (uint8_t)h_make(sn, 1 + i);
    s->n_green = (uint8_t)g;
  }
  for (int r = 0; r < NRED; ++r) {
    int sn = red_subnet_alloc(r, (int)rng_below(x.r, (uint32_t)red_nsub(r)));
    int c = (int)rng_below(x.r, (uint32_t)(s->n_users[sn] + s->n_servers[sn]));
    s->red[r].h.start_host = (uint8_t)(c < s->n_users[sn] ? h_make(sn, 1 + c) : h_make(sn, 11 + (c - s->n_users[sn])));
    s->red[r].h.new_sess_host = 0xFF;
  }
}
// phase 5, per host: starting sessions (State.__init__, State.py:103-136) from the host's stream, then the backup image
CC4_HD void reset_host_sessions(Ctx x, int h) {
  EnvState* s = x.s;
  if (!bit_get(s->exists, h)) return;
  rng_set_stream(x.r, ST_GEN_SESS + (uint32_t)h);
  if (blue_of_subnet(h_subnet(h)) >= 0) (void)start_session_proc(x, h, K_SESS_BLUE);
  if (h != H_INTERNET && h_is_user(h)) (void)start_session_proc(x, h, K_SESS_GREEN);
  if (h == s->red[0].h.start_host) (void)start_session_proc(x, h, K_SESS_RED);
  host_backup(x, h, x.hd[h].procs[PIN - 1].pid);
  x.hd[h].procs[PIN - 1].pid = 0; x.hd[h].gtmp = 0;
}
// phase 6, one thread: red_agent_0's session, initial observations, counters
CC4_HD void reset_finish(Ctx x, ResetCarry k, int steps, uint32_t topo_seed, bool rng_is_copy) {
  EnvState* s = x.s;
  {
    const HostDyn& d = x.hd[s->red[0].h.start_host];
    int red0_pid = 0;
    for (int i = 0; i < d.nproc; ++i) if (d.procs[i].kind == K_SESS_RED) red0_pid = d.procs[i].pid;
    (void)rs_add(x, 0, s->red[0].h.start_host, red0_pid, RS_ABSTRACT | RS_ORIG);
    s->red[0].h.active = 1;
  }
  for (int r = 0; r < NRED; ++r) {
    RedAgent& a = s->red[r];
    int h = a.h.start_host;
    bit_set(a.as_ip, h); bit_set(a.as_hn, h); a.h.as_subnet |= (uint16_t)(1u << h_subnet(h));
    if (r == 0) {
      as_know_sid(x, 0, 0);
      obs_put(x, 0, false, h, OE_SESS | OE_IFACE | OE_SYSHN, true);
      obs_first(x, 0, T_UNKNOWN, RA_NONE, 0, 0);
    }
    a.h.exec_type = RA_SLEEP;
  }
  s->done = (uint8_t)(0 >= steps - 1);
  s->n_actions = NBLUE + s->n_green + NRED;
  if (topo_seed) x.r->s_lo = k.env_key;
  if (rng_is_copy) s->rng = *x.r;
  rng_park(&s->rng);
}
// the same phases as loops (the oracle; any single-threaded caller)
CC4_HD void env_reset_counter_mode(Ctx x, uint64_t seed, int steps, bool continue_stream, int policy, uint32_t topo_seed, uint32_t* ws,
                                   bool rng_is_copy) {
  reset_zero(x.s, x.hd, x.c, 0, 1);
  ResetCarry k = reset_topology(x, seed, steps, continue_stream, policy, topo_seed, ws, rng_is_copy);
  {
    Rng t; rng_fork(&t, x.r, ST_GEN_HOST);
    Ctx xh = x; xh.r = &t;
    for (int h = 0; h < MAXH; ++h) reset_gen_host(xh, h);
  }
  reset_pid_serial(x, reset_used_set(x.s));
  reset_used_clear(x.s, 0, 1);
  reset_agents(x);
  {
    Rng t; rng_fork(&t, x.r, ST_GEN_SESS);
    Ctx xh = x; xh.r = &t;
    for (int h = 0; h < MAXH; ++h) reset_host_sessions(xh, h);
  }
  reset_finish(x, k, steps, topo_seed, rng_is_copy);
}

// continue_stream = true restates CybORG.reset(seed=None) (env.py:218-243): the same Generator keeps going.
// topo_seed != 0 (counter-based RNG mode only): every episode draws its scenario from the reset stream of the key
// `topo_seed` instead of its own key, i.e. all episodes of a batch share topology, services and pids and differ only in
// their dynamics (SURVEY 8(d)-5 "uniform topology" contrast; the reference always randomises per reset).
// pid_ws: counter mode: RESET_WS_WORDS words of work area; numpy-stream mode: optional 288-word area for the used-pid bitmap
// of the generation (default: lent from the idle session pool of the row).
CC4_HD void env_reset(Ctx x, uint64_t seed, int rng_mode, int steps, bool continue_stream, int policy = 0, uint32_t topo_seed = 0,
                      uint32_t* pid_ws = nullptr, bool rng_is_copy = false) {
  EnvState* s = x.s;
  if (rng_mode == 1) {   // counter-based mode: generation in per-host phases; pid_ws must then hold RESET_WS_WORDS words
    env_reset_counter_mode(x, seed, steps, continue_stream, policy, topo_seed, pid_ws, rng_is_copy);
    return;
  }
  Rng keep = s->rng;
  {  // zero everything (POD)
    uint32_t* w = (uint32_t*)s;
    for (size_t i = 0; i < offsetof(EnvState, hd) / 4; ++i) w[i] = 0;
    uint32_t* hw = (uint32_t*)x.hd;
    for (size_t i = 0; i < sizeof(HostDyn) * MAXH / 4; ++i) hw[i] = 0;
  }
  if (continue_stream) s->rng = keep; else rng_seed(&s->rng, seed, (uint32_t)rng_mode);  // NOTE: a fresh seed needs x.r == &s->rng;
                                                                                          // a continued stream may be walked on a copy of it
  s->rng_mode = (uint8_t)rng_mode;
  s->policy = (uint8_t)policy;
  rng_begin_episode(x.r);  // philox: the reset stream uses its own (step, episode) counter words
  const uint64_t env_key = x.r->s_lo;
  if (topo_seed && rng_mode == 1) x.r->s_lo = (uint64_t)topo_seed;
  s->steps = steps;
  {  // _generate_mission_phases (ESG.py:854-860)
    int q = steps / 3, rem = steps % 3;
    s->phase_len[0] = q + (rem >= 1 ? 1 : 0); s->phase_len[1] = q + (rem == 2 ? 1 : 0); s->phase_len[2] = q;
  }
  {  // backup images of the previous episode
    uint32_t* w = (uint32_t*)x.c->hs;
    for (size_t i = 0; i < sizeof(x.c->hs) / 4; ++i) w[i] = 0;
  }
  // bitmap of used_pids over 1000..9999: by default lent from the session pool, which stays empty until red_agent_0's
  // session is added at the very end (records 8 .. 151; zeroed again below)
  static_assert(8 + 288 * 4 / sizeof(RSess) <= RS_POOL, "the used-pid bitmap fits the idle session pool");
  uint32_t* used = pid_ws ? pid_ws : reinterpret_cast<uint32_t*>(&s->spool[8]);
  for (int i = 0; i < 288; ++i) used[i] = 0;

  // _generate_subnets (ESG.py:171-266): choice(len(remaining /24 blocks)) per subnet, pop.  The remaining list stays
  // ascending, so "pop(c)" is the c-th set bit of a 256-bit availability map
  {
    uint32_t* avail = x.w->scratch;  // 8 words
    for (int i = 0; i < 8; ++i) avail[i] = 0xFFFFFFFFu;
    int n = 256;
    for (int sn = 0; sn < NSUB; ++sn) {
      int c = (int)rng_below(x.r, (uint32_t)n);
      int v = nth_set(avail, 8, c);
      s->cidr_octet[sn] = (uint8_t)v;
      bit_clr(avail, v);
      n--;
    }
  }
  // _generate_hosts (ESG.py:312-371): ip_addresses = hosts .1 .. .254 of the /24, ascending
  for (int sn = 0; sn < NSUB; ++sn) {
    uint32_t* ips = x.w->scratch;  // bit v set <=> 10.0.X.v still unassigned
    for (int i = 0; i < 8; ++i) ips[i] = 0xFFFFFFFFu;
    ips[0] &= ~1u; ips[7] &= 0x7FFFFFFFu;  // .0 and .255 are not host addresses
    int n = 254;
    auto pop_at = [&](int c) { int v = nth_set(ips, 8, c); bit_clr(ips, v); n--; return (uint8_t)v; };
    if (sn == S_INT) {
      int c = (int)rng_below(x.r, (uint32_t)n);
      uint8_t ip = pop_at(c);
      gen_host(x, H_INTERNET, used);
      x.hd[H_INTERNET].procs[PIN - 1].pid = ip;   // parked in the last inline process slot until the backup image is written
      continue;
    }
    int hr = h_make(sn, 0);
    { int c = (int)rng_below(x.r, (uint32_t)n); uint8_t ip = pop_at(c); gen_host(x, hr, used); x.hd[hr].procs[PIN - 1].pid = ip; }
    int nu = 3 + (int)rng_below(x.r, 8);  // integers(3, 10, endpoint=True)
    for (int i = 0; i < nu; ++i) {
      int h = h_make(sn, 1 + i);
      int c = (int)rng_below(x.r, (uint32_t)n); uint8_t ip = pop_at(c);
      gen_host(x, h, used); x.hd[h].procs[PIN - 1].pid = ip;
    }
    int ns = 1 + (int)rng_below(x.r, 6);  // integers(1, 6, endpoint=True)
    for (int i = 0; i < ns; ++i) {
      int h = h_make(sn, 11 + i);
      int v = last_set(ips, 8); bit_clr(ips, v); n--;  // ip_addresses.pop()
      gen_host(x, h, used); x.hd[h].procs[PIN - 1].pid = (uint16_t)v;
    }
    s->n_users[sn] = (uint8_t)nu; s->n_servers[sn] = (uint8_t)ns;
  }
  // _generate_blue_agents (ESG.py:631-698)
  for (int b = 0; b < NBLUE; ++b) {
    int nsub = blue_nsub(b);
    (void)rng_below(x.r, (uint32_t)nsub);  // starting_subnet = choice(allowed_subnets): unused
    int cnt = 0;
    for (int i = 0; i < nsub; ++i) { int sn = blue_subnet_alloc(b, i); cnt += 1 + s->n_users[sn] + s->n_servers[sn]; }
    int c = (int)rng_below(x.r, (uint32_t)cnt);  // parent_host = choice(allowed_hosts)
    int k = 0, ph = -1;
    for (int i = 0; i < nsub && ph < 0; ++i) {
      int sn = blue_subnet_alloc(b, i);
      for (int sl = 0; sl < SLOTS; ++sl) { int h = h_make(sn, sl); if (bit_get(s->exists, h)) { if (k == c) { ph = h; break; } k++; } }
    }
    s->blue[b].parent_host = (uint8_t)ph;
  }
  // _generate_green_agents (ESG.py:700-749): one per user host, host order
  {
    int g = 0;
    for (int h = 0; h < MAXH; ++h) if (bit_get(s->exists, h) && h_is_user(h)) s->green_host[g++] = (uint8_t)h;
    s->n_green = (uint8_t)g;
  }
  // _generate_red_agents (ESG.py:751-817)
  for (int r = 0; r < NRED; ++r) {
    int nsub = red_nsub(r);
    int sn = red_subnet_alloc(r, (int)rng_below(x.r, (uint32_t)nsub));
    int cnt = s->n_users[sn] + s->n_servers[sn];
    int c = (int)rng_below(x.r, (uint32_t)cnt);  // choice(non-router hosts): users then servers
    int h = c < s->n_users[sn] ? h_make(sn, 1 + c) : h_make(sn, 11 + (c - s->n_users[sn]));
    s->red[r].h.start_host = (uint8_t)h;
    s->red[r].h.new_sess_host = 0xFF;
    s->red[r].h.queue.busy = 0;
  }
  // State.__init__ (State.py:103-136): starting sessions in agent order; parent-less first
  for (int b = 0; b < NBLUE; ++b) {
    int ph = s->blue[b].parent_host;
    (void)start_session_proc(x, ph, K_SESS_BLUE);
    for (int i = 0; i < blue_nsub(b); ++i) {
      int sn = blue_subnet_alloc(b, i);
      for (int sl = 0; sl < SLOTS; ++sl) {
        int h = h_make(sn, sl);
        if (!bit_get(s->exists, h) || h == ph) continue;
        (void)start_session_proc(x, h, K_SESS_BLUE);
      }
    }
  }
  for (int g = 0; g < s->n_green; ++g) (void)start_session_proc(x, s->green_host[g], K_SESS_GREEN);
  int red0_pid = start_session_proc(x, s->red[0].h.start_host, K_SESS_RED);
  // host.create_backup() for every host (State.py:137-138) -> dynamic state := static
  for (int h = 0; h < MAXH; ++h) {
    if (bit_get(s->exists, h)) { host_backup(x, h, x.hd[h].procs[PIN - 1].pid); x.hd[h].procs[PIN - 1].pid = 0; x.hd[h].gtmp = 0; }
    eph_clear(x, h);
  }
  s->npend = 0;
  if (!pid_ws) for (int i = 0; i < 288; ++i) used[i] = 0;
  // red_agent_0 starts active with session 0 (ESG.py:791-801)
  {
    int idx = rs_add(x, 0, s->red[0].h.start_host, red0_pid, RS_ABSTRACT | RS_ORIG);
    (void)idx;
    s->red[0].h.active = 1;
  }
  // AgentInterface.set_init_obs (Shared/AgentInterface.py:110-117) with the OSINT observation of the start host
  for (int r = 0; r < NRED; ++r) {
    RedAgent& a = s->red[r];
    int h = a.h.start_host;
    bit_set(a.as_ip, h); bit_set(a.as_hn, h); a.h.as_subnet |= (uint16_t)(1u << h_subnet(h));
    a.h.nobs = 0; a.h.obs_success = 0;
    if (r == 0) {
      as_know_sid(x, 0, 0);
      // SimulationController.reset_observation (SC:767-773): first FSM call sees hostname-keyed OSINT obs
      obs_put(x, 0, false, h, OE_SESS | OE_IFACE | OE_SYSHN, true);
      obs_first(x, 0, T_UNKNOWN, RA_NONE, 0, 0);
    }
    a.h.exec_type = RA_SLEEP;
  }
  s->step_count = 0; s->phase = 0; s->done = (uint8_t)(0 >= steps - 1); s->reward = 0.f;
  s->n_actions = NBLUE + s->n_green + NRED;
  if (topo_seed && rng_mode == 1) x.r->s_lo = env_key;
  if (rng_is_copy) s->rng = *x.r;      // the generator was walked on a caller-side (register) copy
  rng_park(&s->rng);
}

// ------------------------------------------------------------------ blue actions
// wrapper index -> action (Agents/Wrappers/BlueFixedActionWrapper.py:233-309)
enum : int { BLUE_RAW_ACTION = 0x10000 };   // idx = BLUE_RAW_ACTION | BA_* type << 8 | host id: see blue_decode
// What slot idx of blue agent b's action list names before the episode's topology is consulted (BlueFixedActionWrapper.py:241-300: Analyse x hosts,
// Monitor, Remove x hosts, Restore x hosts, Sleep, Allow x (8 other subnets per own subnet), Block x same, DeployDecoy x hosts; hosts = the 16 server /
// user positions of each of the agent's subnets in sorted order): type | host id << 8 for the host actions, type | to-subnet << 8 | from-subnet << 16 for
// Allow / Block.  A pure function of (b, idx): the device reads it from a table the compiler fills (blue_slot), the host computes it.
CC4_HD constexpr uint32_t blue_slot_shape(int b, int idx) {
  const int nsub = blue_nsub(b), nh = ZONE_HOSTS * nsub, nc = 8 * nsub;
  int t = BA_SLEEP, j = 0;
  if (idx < 0 || idx >= 4 * nh + 2 + 2 * nc) return (uint32_t)BA_SLEEP;
  if (idx < nh) { t = BA_ANALYSE; j = idx; }
  else if (idx == nh) return (uint32_t)BA_MONITOR;
  else if (idx < 2 * nh + 1) { t = BA_REMOVE; j = idx - nh - 1; }
  else if (idx < 3 * nh + 1) { t = BA_RESTORE; j = idx - 2 * nh - 1; }
  else if (idx == 3 * nh + 1) return (uint32_t)BA_SLEEP;
  else if (idx < 3 * nh + 2 + nc) { t = BA_ALLOW; j = idx - 3 * nh - 2; }
  else if (idx < 3 * nh + 2 + 2 * nc) { t = BA_BLOCK; j = idx - 3 * nh - 2 - nc; }
  else { t = BA_DECOY; j = idx - 3 * nh - 2 - 2 * nc; }
  if (t == BA_ALLOW || t == BA_BLOCK) {
    const int dst = blue_subnet_sorted(b, j / 8), k = j % 8;
    // the 8 other subnets in alphabetical order
    int src = 0, c = 0;
    for (int i = 0; i < NSUB; ++i) { const int sn = sorted_subnet(i); if (sn == dst) continue; if (c == k) { src = sn; break; } c++; }
    return (uint32_t)t | ((uint32_t)dst << 8) | ((uint32_t)src << 16);
  }
  const int sn = blue_subnet_sorted(b, j / ZONE_HOSTS), hs = j % ZONE_HOSTS;
  const int h = hs < MAX_SERVERS ? h_make(sn, 11 + hs) : h_make(sn, 1 + (hs - MAX_SERVERS));
  return (uint32_t)t | ((uint32_t)h << 8);
}
#if defined(__HIP_DEVICE_COMPILE__)
// (the decode's divisions and subnet search were ~60 vector instructions of every step on the blue lanes; the step kernels are bound by vector issue slots)
struct BlueSlotTab { uint32_t v[NBLUE][ACT_LONG]; };
constexpr BlueSlotTab make_blue_slot_tab() {
  BlueSlotTab t{};
  for (int b = 0; b < NBLUE; ++b) for (int i = 0; i < ACT_LONG; ++i) t.v[b][i] = blue_slot_shape(b, i);
  return t;
}
static __device__ const BlueSlotTab blue_slot_tab = make_blue_slot_tab();
__device__ __forceinline__ uint32_t blue_slot(int b, int idx) { return blue_slot_tab.v[b][idx]; }      // callers pass 0 <= idx < the agent's list length
#else
CC4_HD uint32_t blue_slot(int b, int idx) { return blue_slot_shape(b, idx); }
#endif
CC4_HD Act blue_decode(const EnvState* s, int b, int idx) {
  Act a; a.type = BA_SLEEP; a.host = 0; a.arg = 0; a.ticks = 1; a.sid = 0; a.busy = 0;
  int nsub = blue_nsub(b), nh = ZONE_HOSTS * nsub, nc = 8 * nsub;
  int total = 4 * nh + 2 + 2 * nc;
  if (idx >= BLUE_RAW_ACTION) {
    // a host action given as (type, host id) instead of a slot of the wrapper's list: what CybORG.step / parallel_step forward
    // when they are handed an Action OBJECT (env.py:95-161) -- the list has no slot for a zone's router, the simulator takes one
    // (as it does from cc4BlueRandomAgent).  A host outside the agent's subnets or the episode's topology is not in its action
    // space: InvalidAction (SimulationController.py:1068-1112), which resolves like Sleep.
    const int t = (idx >> 8) & 0xF, h = idx & 0xFF;
    if (t == BA_BLOCK || t == BA_ALLOW) {
      // Block/AllowTrafficZone(from_subnet, to_subnet) as an object may name ANY pair: neither parameter is an ActionSpace key, so
      // the validity check lets it through (SC:1094-1096; Tests/test_cc4/test_BlueRewardMachine.py:133-137 has blue_agent_0 block
      // every subnet); host id field = to-subnet | from-subnet << 4
      const int to = h & 0xF, from = h >> 4;
      if (to >= NSUB || from >= NSUB) return a;
      a.type = (uint8_t)t; a.host = (uint8_t)to; a.arg = (uint8_t)from;
      return a;
    }
    if (t == BA_MONITOR) { a.type = BA_MONITOR; return a; }
    if (t < BA_ANALYSE || t > BA_DECOY || h >= H_INTERNET || !bit_get(s->exists, h) || blue_of_subnet(h_subnet(h)) != b) return a;
    a.type = (uint8_t)t; a.host = (uint8_t)h;
    return a;
  }
  if (idx < 0 || idx >= total) return a;  // padding / "no action submitted" -> Sleep
  const uint32_t sh = blue_slot(b, idx);
  const int t = (int)(sh & 0xFF), h = (int)((sh >> 8) & 0xFF);
  if (t == BA_SLEEP) return a;
  if (t == BA_MONITOR) { a.type = BA_MONITOR; return a; }
  if (t == BA_ALLOW || t == BA_BLOCK) { a.type = (uint8_t)t; a.host = (uint8_t)h; a.arg = (uint8_t)(sh >> 16); return a; }
  if (!bit_get(s->exists, h)) return a;  // "[Invalid] ..." slot -> Sleep() (BlueFixedActionWrapper.py:295-298)
  a.type = (uint8_t)t; a.host = (uint8_t)h;
  return a;
}
CC4_HD void blue_action_mask(const EnvState* s, uint8_t* mask /* MASK_TOTAL */) {
  int o = 0;
  for (int b = 0; b < NBLUE; ++b) {
    int n = b == 4 ? ACT_LONG : ACT_SHORT;
    for (int i = 0; i < n; ++i) {
      Act a = blue_decode(s, b, i);
      int nh = ZONE_HOSTS * blue_nsub(b);
      bool is_sleep_slot = (i == 3 * nh + 1);
      mask[o++] = (uint8_t)((a.type != BA_SLEEP) || is_sleep_slot);
    }
  }
}

// Monitor.execute (Simulator/Actions/AbstractActions/Monitor.py:35-74)
CC4_HD void blue_monitor(Ctx x, int b) {
  EnvState* s = x.s;
  BlueAgent& A = s->blue[b];
  for (int i = 0; i < blue_nsub(b); ++i) {
    int sn = blue_subnet_alloc(b, i);
    for (int sl = 0; sl < SLOTS; ++sl) {
      int h = h_make(sn, sl);
      if (!bit_get(s->exists, h)) continue;
      uint8_t ev = s->hev[h];
      uint8_t nev = 0;
      if (ev & EV_CUR_CONN) nev |= EV_OLD_CONN;
      if (ev & EV_CUR_PROC) nev |= EV_OLD_PROC;
      s->hev[h] = nev;
    }
  }
  // session.add_sus_pids for process_creation events that carry a pid
  int n = 0;
  for (int i = 0; i < s->npend; ++i) {
    int h = (int)(s->pend[i] >> 16);
    if (blue_of_subnet(h_subnet(h)) == b) {
      if (A.nsus >= cold_sus_cap(s->steps)) set_err(x, E_SUS_OVERFLOW); else { cold_sus(x.c, s->steps, b)[A.nsus++] = s->pend[i]; bit_set(A.sus_hosts, h); }
    } else s->pend[n++] = s->pend[i];
  }
  s->npend = (uint8_t)n;
}
// StopProcess.kill_process (ConcreteActions/StopProcess.py:36-58) after get_process / root check (:23-34)
// the table index of the host's service whose process is `pid`, or -1
CC4_HD int svc_of_pid(const HostDyn& d, int pid) {
  uint32_t sv[MAXSV];
  __builtin_memcpy(sv, d.svcs, sizeof(sv));
  const int n = hd_nsvc(d);
  int si = -1;
  CC4_UNROLL for (int i = MAXSV - 1; i >= 0; --i) if (i < n && (int)(sv[i] & 0xFFFF) == pid) si = i;
  return si;
}
// the host's service table (pid | kind << 16 | st << 24 per entry) and its length, requested together: one round trip to the host's row
struct SvTab { uint32_t w[MAXSV]; int n; };
CC4_HD SvTab svc_load(const HostDyn& d) {
  SvTab t;
  __builtin_memcpy(t.w, d.svcs, sizeof(t.w));
  t.n = hd_nsvc(d);
  return t;
}
CC4_HD int svw_kind(uint32_t w) { return (int)((w >> 16) & 0xFF); }
CC4_HD int svw_st(uint32_t w) { return (int)(w >> 24); }
// The process `pi` of host h (record word pw) dies, whoever asked: the list entry goes, a service process respawns under a
// new pid, and the session running in it ends (state.get_session_from_pid, State.py:420-443: a blue or green session is
// recognised by the process kind Host.add_session gave it, a red one by its (host, pid)).
CC4_HD void kill_process(Ctx x, int h, int pi, uint32_t pw) {
  EnvState* s = x.s;
  const int pid = pw_pid(pw), kind = pw_kind(pw);
  int owner = -1, owner_idx = -1;  // owner: 0 blue, 1 green, 2+r red
  if (kind == K_SESS_BLUE) owner = 0;
  else if (kind == K_SESS_GREEN) owner = 1;
  else if (bit_get(s->red_hosts, h)) {
    uint32_t lw[NRED];
    CC4_UNROLL for (int r = 0; r < NRED; ++r) lw[r] = s->red[r].live_hosts[h >> 5];
    for (int r = 0; r < NRED && owner < 0; ++r) {
      if (!((lw[r] >> (h & 31)) & 1u)) continue;
      int i = rs_find_host_pid(s, s->red[r], h, pid);
      if (i >= 0) { owner = 2 + r; owner_idx = i; }
    }
  }
  remove_proc_at(x, h, pi);
  HostDyn& d = x.hd[h];
  const int si = svc_of_pid(d, pid);
  if (si >= 0) {  // service process respawns under a new pid
    const P8N hd0 = proc_head(x, h);
    int np = create_pid(x, h, hd0);
    add_proc_n(x, h, hd0.n, np, kind, pw_flags(pw));
    d.svcs[si].pid = (uint16_t)np;     // (the row is marked: remove_proc_at / add_proc above)
  }
  if (owner < 0) return;
  if (owner < 2) { set_err(x, E_BLUE_GREEN_SESSION_KILLED); return; }
  rs_remove_at(x, owner - 2, owner_idx);
  if (si >= 0) set_err(x, E_UNREACHABLE);  // session re-created on a service process: never happens in CC4
}
// StopProcess.kill_process (ConcreteActions/StopProcess.py:36-58) after get_process / root check (:23-34)
CC4_HD void stop_process(Ctx x, int h, int pid) {
  int pi = find_proc(x, h, pid);
  if (pi < 0) return;
  const uint32_t pw = proc_get(x, h, pi);
  if (pw_flags(pw) & PF_ROOT) return;
  kill_process(x, h, pi, pw);
}
// Remove.execute (AbstractActions/Remove.py:42-71)
CC4_HD void blue_remove(Ctx x, int b, int h) {
  BlueAgent& A = x.s->blue[b];
  if (!bit_get(A.sus_hosts, h)) return;   // parent_session.sus_pids has no entry for this hostname
  // The list lives in the cold row (HBM).  It is filtered with 16 independent loads in flight per round into a small LDS
  // work area (12 words per blue agent), then the matching pids are stopped in list order.
  const uint32_t* list = cold_sus(x.c, x.s->steps, b);
  uint16_t* hit = reinterpret_cast<uint16_t*>(x.w->scratch + 12 * b);
  const int cap = 24, n = A.nsus;
  int i0 = 0;
  while (i0 < n) {
    int nh = 0;
    for (; i0 < n && nh + 16 <= cap; i0 += 16) {   // the list's capacity is a multiple of 16: the tail of a round reads allocated slots
      uint32_t v[16];                              // 16 independent loads in flight: one HBM round trip per 16 entries
      CC4_UNROLL for (int k = 0; k < 16; ++k) v[k] = list[i0 + k];
      CC4_UNROLL for (int k = 0; k < 16; ++k) if (i0 + k < n && (int)(v[k] >> 16) == h) hit[nh++] = (uint16_t)v[k];
    }
    for (int k = 0; k < nh; ++k) stop_process(x, h, (int)hit[k]);
  }
}
// Restore.execute -> RestoreFromBackup (AbstractActions/Restore.py:38-71, ConcreteActions/RestoreFromBackup.py:9-19)
CC4_HD void blue_restore(Ctx x, int h) {
  EnvState* s = x.s;
  uint32_t lw[NRED];   // which agents hold sessions on the host: one batch of loads
  CC4_UNROLL for (int r = 0; r < NRED; ++r) lw[r] = s->red[r].live_hosts[h >> 5];
  CC4_UNROLL for (int r = 0; r < NRED; ++r) {
    if (!((lw[r] >> (h & 31)) & 1u)) continue;
    RedAgent& a = s->red[r];
    // every non-original session on the host goes; an original one is popped and re-added => moves to the end
    rs_remove_on_host(x, r, h);
    {
      const int n = a.h.nsess;
      int orig = -1;
      for (int i0 = 0; i0 < n; i0 += 8) {
        const S8 q = rs_load8(s, a, i0);
        CC4_UNROLL for (int k = 0; k < 8; ++k) if (i0 + k < n && rsw_host(q.v[k]) == h && (rsw_flags(q.v[k]) & RS_ORIG)) orig = i0 + k;   // the last one, as the walk left it
      }
      if (orig >= 0) rs_move_to_end(s, a, orig, -1);
    }
  }
  host_restore(x, h);
}
// DecoyAction.execute (ConcreteActions/DecoyActions/DecoyAction.py:47-114) with DeployDecoy candidates (DeployDecoy.py:8-31)
CC4_HD void blue_decoy(Ctx x, int h) {
  uint32_t cand = 8;  // bit i <=> K_DEC_APACHE + i is compatible; vsftpd checks port 21, which nothing uses (DecoyVsftpd.py:17-20)
  // everything the action reads of the host -- process list head and length, service table -- is one cache line: one round trip
  HostDyn& d = x.hd[h];
  const P8N hd0 = proc_head(x, h);
  const SvTab sv = svc_load(d);
  const int used = proc_ports(x, h, hd0);   // Host.is_using_port per factory
  if (!(used & PB_80)) cand |= 1;
  if (!(used & PB_443)) cand |= 2;
  if (!(used & PB_25)) cand |= 4;
  int kind = K_DEC_APACHE + nth_bit(cand, (int)rng_below(x.r, (uint32_t)popc32(cand)));
  int pid = create_pid(x, h, hd0);
  if (!add_proc_n(x, h, hd0.n, pid, kind, 0)) return;
  ev_log(x, 250, h, 2, kind, 0, 0xFF, 0, pid);   // the action's own observation: obs.add_process(pid, parent_pid=1, ...) (DecoyAction.py:105-113)
  int si = -1;
  const int nsv = sv.n;
  CC4_UNROLL for (int i = MAXSV - 1; i >= 0; --i) if (i < nsv && svw_kind(sv.w[i]) == kind) si = i;   // the first one
  if (si < 0) {
    if (nsv >= MAXSV) { set_err(x, E_UNREACHABLE); return; }   // excluded by the port checks (see MAXSV)
    si = nsv; hd_set_nsvc(d, nsv + 1);
  }
  d.svcs[si].kind = (uint8_t)kind; d.svcs[si].pid = (uint16_t)pid; d.svcs[si].st = (uint8_t)(SV_ACTIVE | 5);
}
CC4_HD void blue_execute(Ctx x, int b, const Act& a) {
  EnvState* s = x.s;
  switch (a.type) {
    case BA_MONITOR: blue_monitor(x, b); break;
    case BA_ANALYSE: break;  // DensityScout/SigCheck only fill the dict observation (Analyse.py:40-71)
    case BA_REMOVE: blue_remove(x, b, a.host); break;
    case BA_RESTORE: blue_restore(x, a.host); break;
    case BA_DECOY: blue_decoy(x, a.host); break;
    // Observation(False) when the pair is already blocked / not blocked (ControlTraffic.py:111-113, :176-178)
    // (obs_dirty only when the pair's bit really changes: under a random policy half of these actions name a pair that is already as asked)
    case BA_BLOCK: { const bool was = (s->blocks[a.host] >> a.arg) & 1u;
                   s->blue[b].last_ok = (uint8_t)(was ? T_FALSE : T_TRUE);
                   if (!was) { s->blocks[a.host] |= (uint16_t)(1u << a.arg); s->obs_dirty |= OD_BLOCKS; } break; }   // ControlTraffic.py:88-116
    case BA_ALLOW: { const bool was = (s->blocks[a.host] >> a.arg) & 1u;
                   s->blue[b].last_ok = (uint8_t)(was ? T_TRUE : T_FALSE);
                   if (was) { s->blocks[a.host] &= (uint16_t)~(1u << a.arg); s->obs_dirty |= OD_BLOCKS; } break; }   // ControlTraffic.py:160-185
    default: break;
  }
}

// ------------------------------------------------------------------ green actions
// PhishingEmail._create_new_session (ConcreteActions/PhishingEmail.py:42-113)
CC4_HD void phishing(Ctx x, int gh) {
  EnvState* s = x.s;
  // EnvState.red_hosts = hosts where some red agent holds a session; RedAgent.live_hosts the same per agent (both exact)
  if (bit_get(s->red_hosts, gh)) return;  // a red agent already has a session on the green host (PhishingEmail.py:57-59)
  const int gsub = h_subnet(gh);
  const int lo = gsub * SLOTS, hi = lo + SLOTS - 1;   // host ids of the green host's subnet
  // hosts of the same subnet overwrite red_agent_src in host order -> the LAST such host with red sessions decides, and on
  // it the FIRST red agent (`break` leaves only the inner loop, PhishingEmail.py:63-69)
  int src = -1;
  {
    // the 17 ids lo..hi straddle at most two bitmap words
    int wl = lo >> 5, wh = hi >> 5;
    uint32_t mh = s->red_hosts[wh] & (0xFFFFFFFFu >> (31 - (hi & 31))) & (wl == wh ? (0xFFFFFFFFu << (lo & 31)) : 0xFFFFFFFFu);
    int best = -1;
    if (mh) best = wh * 32 + (31 - __builtin_clz(mh));
    else if (wl != wh) { uint32_t ml = s->red_hosts[wl] & (0xFFFFFFFFu << (lo & 31)); if (ml) best = wl * 32 + (31 - __builtin_clz(ml)); }
    if (best >= 0) {
      uint32_t lw[NRED];   // the six agents' bitmap words for that host, one batch of loads
      CC4_UNROLL for (int r = 0; r < NRED; ++r) lw[r] = s->red[r].live_hosts[best >> 5];
      CC4_UNROLL for (int r = NRED - 1; r >= 0; --r) if ((lw[r] >> (best & 31)) & 1u) src = r;   // first agent in order
    }
  }
  if (src < 0) {
    // red_agents = [(agent, host)] over hosts outside the subnet (there is no red session inside it here), host-major then
    // agent order; choice(red_agents, replace=False) is one bounded draw.  All counts and bitmaps are read up front.
    int nl[NRED]; uint32_t lh[NRED][5], rh[5];
    CC4_UNROLL for (int r = 0; r < NRED; ++r) nl[r] = s->red[r].h.nlive;
    CC4_UNROLL for (int w = 0; w < 5; ++w) rh[w] = s->red_hosts[w];
    CC4_UNROLL for (int r = 0; r < NRED; ++r) { CC4_UNROLL for (int w = 0; w < 5; ++w) lh[r][w] = s->red[r].live_hosts[w]; }
    int nc = 0;
    CC4_UNROLL for (int r = 0; r < NRED; ++r) nc += nl[r];
    if (nc == 0) return;
    int c = (int)rng_below(x.r, (uint32_t)nc);
    CC4_UNROLL for (int w = 0; w < 5; ++w) {
      uint32_t m = rh[w];
      if (src >= 0 || !m) continue;
      int inw = 0;
      CC4_UNROLL for (int r = 0; r < NRED; ++r) inw += popc32(lh[r][w]);
      if (c >= inw) { c -= inw; continue; }
      while (m && src < 0) {
        int b = ctz32(m); m &= m - 1;
        CC4_UNROLL for (int r = 0; r < NRED; ++r)
          if (src < 0 && ((lh[r][w] >> b) & 1u)) { if (c-- == 0) src = r; }
      }
    }
  }
  const P8N hd0 = proc_head(x, gh);
  int pid = create_pid(x, gh, hd0);
  if (!add_proc_n(x, gh, hd0.n, pid, K_SESS_RED, 0)) return;
  rs_add(x, src, gh, pid, RS_ABSTRACT);
}
// What a green agent's action reads from the state, as one 8-byte word that can be computed ahead of the (ordered) resolution
// by any lane -- the numpy-stream kernel does that for all agents at once while its walking lane would otherwise do it agent by
// agent between draws.  Nothing a green action writes (event bits, rewards, phishing sessions) feeds into these words.
//   GreenLocalWork:     byte i = status byte of service i of the agent's host (active bit | reliability/20), 0 beyond the table
//   GreenAccessService: byte sn = number of servers in the allowed subnets 0..sn (running total; byte 7 = all candidates)
CC4_HD uint64_t green_prepare(Ctx x, int g, int act) {
  EnvState* s = x.s;
  const int gh = s->green_host[g];
  uint64_t w = 0;
  if (act == 1) {
    const HostDyn& d = x.hd[gh];
    uint32_t sv[MAXSV];
    __builtin_memcpy(sv, d.svcs, sizeof(sv));
    const int nsvc = hd_nsvc(d);
    CC4_UNROLL for (int i = 0; i < MAXSV; ++i) if (i < nsvc) w |= (uint64_t)(sv[i] >> 24) << (8 * i);
  } else if (act == 0) {
    // agent_interface.allowed_subnets of the mission phase (EnterpriseGreenAgent hands them to the action), or the list a submitted
    // GreenAccessService came with (ExtAct.sid as a subnet mask: a green action's session_id is always 0)
    // (a source subnet outside the list reaches only itself: GreenAccessService.py:96-103)
    uint32_t allowed = green_allowed_mask(s->phase, h_subnet(gh));
    if (x.ext && x.ext[NRED + g].type == XG_ACCESS && x.ext[NRED + g].sid) {
      allowed = (uint32_t)x.ext[NRED + g].sid;
      if (!((allowed >> h_subnet(gh)) & 1u)) allowed = 1u << h_subnet(gh);
    }
    uint64_t ns;   // server counts of subnets 0..7 (the internet subnet has none), one batch of loads
    __builtin_memcpy(&ns, s->n_servers, 8);
    int n = 0;
    CC4_UNROLL for (int sn = 0; sn < NSUB - 1; ++sn) { if ((allowed >> sn) & 1u) n += (int)((ns >> (8 * sn)) & 0xFF); w |= (uint64_t)n << (8 * sn); }
  }
  return w;
}
// the c-th server over a GreenAccessService agent's allowed subnets in subnet order (pre: green_prepare's running totals): the
// first subnet whose total exceeds c
CC4_HD int green_as_dest(uint64_t pre, int c, int* sn_out) {
  int sn = 0, before = 0;
  CC4_UNROLL for (int k = 0; k < NSUB - 2; ++k) { const int tot = (int)((pre >> (8 * k)) & 0xFF); if (tot <= c) { sn = k + 1; before = tot; } }
  *sn_out = sn;
  return h_make(sn, 11 + (c - before));
}
// the active services of a GreenLocalWork agent's host as a bit mask over the service table (pre: green_prepare's status bytes)
CC4_HD uint32_t green_lw_active(uint64_t pre) {
  uint32_t act = 0;
  CC4_UNROLL for (int i = 0; i < MAXSV; ++i) if ((pre >> (8 * i)) & SV_ACTIVE) act |= 1u << i;
  return act;
}
// GreenLocalWork.execute (GreenActions/GreenLocalWork.py:60-125). returns success.  pre: green_prepare's word
CC4_HD bool green_local_work(Ctx x, int gh, uint64_t pre, bool* want_phish, double fp_rate = 0.01, double phish_rate = 0.01) {
  const uint32_t act = green_lw_active(pre);
  if (!act) return false;
  const int c = nth_bit(act, (int)rng_below(x.r, (uint32_t)popc32(act)));   // choice over the active services, table order
  const uint32_t st = (uint32_t)(pre >> (8 * c)) & 0xFF;
  int rel = (int)(st & 0x7F) * 20;
  if ((int)rng_below(x.r, 100) >= rel) return false;
  if (rng_random_lt(x.r, fp_rate)) { int port = eph_port(x, gh); ev_proc(x, gh, 0); ev_log(x, gh, gh, 1, gh, port, 0xFF, 0, 0); }   // pc = {local_address, local_port} (GreenLocalWork.py:112-115)
  if (rng_random_lt(x.r, phish_rate)) *want_phish = true;  // PhishingEmail sub-action: executed by the caller (ordering, see P5)
  return true;
}
// GreenAccessService.execute (GreenActions/GreenAccessService.py:137-217). returns success.  pre: green_prepare's word
CC4_HD bool green_access_service(Ctx x, int gh, uint64_t pre, double fp_rate = 0.01) {
  int own = h_subnet(gh);
  const int n = (int)(pre >> 56);
  const int c = (int)rng_below(x.r, (uint32_t)n);
  int sn;
  const int dest = green_as_dest(pre, c, &sn);
  const int dest_port = eph_port(x, dest);
  int ds = sn;
  // events land on the destination server (`from_host` in the reference, GreenAccessService.py:176-214)
  if (subnet_blocked(x, ds, own) || subnet_blocked(x, own, ds)) { ev_conn(x, dest); ev_log(x, gh, dest, 0, dest, 0, gh, 8800, 0); return false; }
  if (rng_random_lt(x.r, fp_rate)) { ev_conn(x, dest); ev_log(x, gh, dest, 0, gh, 0, dest, dest_port, 0); }
  return true;
}

// ------------------------------------------------------------------ red actions
CC4_HD void red_result(Ctx x, int r, const Act& a, int success) {
  int arg = a.type == RA_DRS ? a.arg : 0;
  obs_first(x, r, success, a.type, a.host, arg);
}
// DiscoverRemoteSystems -> Pingsweep.execute (ConcreteActions/Pingsweep.py:31-64)
CC4_HD void red_drs(Ctx x, int r, const Act& a) {
  EnvState* s = x.s;
  int sn = a.arg; bool any = false;
  // the session can die between filter_actions and execution (a blue Remove/Restore runs earlier in the same step)
  if (rs_find_id(s, s->red[r], a.sid) < 0) { red_result(x, r, a, T_FALSE); return; }
  bool allowed = (red_allowed_mask(r) >> sn) & 1u;  // SimulationController._filter_obs drops foreign-subnet interfaces
  // the subnet's non-router hosts are the 16 ids lo .. lo+15; they straddle at most two bitmap words.  All of them get the same
  // observation entry (obs_put(ip key, OE_IFACE, subnet known)), so the bitmaps are updated per word and the new entries
  // are appended with independent stores, in host order
  RedAgent& A = s->red[r];
  const int lo = sn * SLOTS + 1, w0 = lo >> 5, off = lo & 31, w1 = (w0 + 1 < 5) ? w0 + 1 : w0;
  const uint32_t e0 = s->exists[w0], e1 = s->exists[w1], h0 = A.obs_has[1][w0], h1 = A.obs_has[1][w1];
  auto field = [&](uint32_t a0, uint32_t a1) { uint32_t v = a0 >> off; if (off > 16 && w1 != w0) v |= a1 << (32 - off); return v & 0xFFFFu; };
  const uint32_t ex = field(e0, e1);
  any = ex != 0;
  if (allowed && ex) {
    const uint32_t had = field(h0, h1) & ex;
    uint32_t fresh = ex & ~had;
    const uint32_t m0 = ex << off, m1 = (off > 16 && w1 != w0) ? ex >> (32 - off) : 0u;
    A.as_ip[w0] |= m0; A.obs_has[1][w0] = h0 | m0;
    if (m1) { A.as_ip[w1] |= m1; A.obs_has[1][w1] = h1 | m1; }
    A.h.as_subnet |= (uint16_t)(1u << sn);
    int n = A.h.nobs;
    while (fresh) {
      const int b = ctz32(fresh); fresh &= fresh - 1;
      if (n >= MAX_OBS) { set_err(x, E_OBS_OVERFLOW); break; }
      A.obs[n].host = (uint8_t)(lo + b); A.obs[n].flags = (uint8_t)(OE_IFACE | OE_KEY_IP);
      n++;
    }
    A.h.nobs = (uint8_t)n;
    uint32_t again = had;   // already keyed this step (does not happen on the FSM path: the list is empty when an action starts)
    while (again) { const int b = ctz32(again); again &= again - 1; obs_put(x, r, true, lo + b, OE_IFACE, true); }
  }
  red_result(x, r, a, any ? T_TRUE : T_UNKNOWN);
}
// DiscoverNetworkServices.execute + Portscan.execute (AbstractActions/DiscoverNetworkServices.py:44-86, Portscan.py:23-65)
CC4_HD int port_of_bit(int pb) { return pb == PB_22 ? 22 : (pb == PB_80 ? 80 : (pb == PB_3390 ? 3390 : (pb == PB_25 ? 25 : (pb == PB_1 ? 1 : 443)))); }
CC4_HD void red_scan(Ctx x, int r, const Act& a, double rate) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  int si = rs_find_id(s, A, a.sid);
  if (si < 0) { red_result(x, r, a, T_FALSE); return; }
  const int slot = A.sord[si];
  const uint64_t sw = rs_word(s, slot);
  if (!(rsw_flags(sw) & RS_ABSTRACT)) { red_result(x, r, a, T_FALSE); return; }
  int src = rsw_host(sw), tgt = a.host;
  if (subnet_blocked(x, h_subnet(src), h_subnet(tgt))) { red_result(x, r, a, T_FALSE); return; }
  const bool fixed_le_rate = rng_random_le(x.r, rate);   // fixed = random(); `fixed <= rate` per port below
  int ports = 0;
  const P8N hd0 = proc_head(x, tgt);   // length and first round in one round trip
  const int np = hd0.n;
  for (int i0 = 0; i0 < np; i0 += 8) {
    P8 q = p8_of(hd0);
    if (i0) q = proc_load8(x, tgt, i0);
    CC4_UNROLL for (int k8 = 0; k8 < 8; ++k8) {
      if (i0 + k8 >= np) continue;
      int k = pw_kind(q.v[k8]);
      int pb = kind_port(k);
      if (!pb) continue;
      ports |= pb;
      if (fixed_le_rate || kind_is_decoy(k)) { ev_conn(x, tgt); int ep = eph_port(x, tgt, (i0 + k8) & 63); ev_log(x, 200 + r, tgt, 0, tgt, port_of_bit(pb), src, ep, 0); }   // Portscan.py:59-64
    }
  }
  if (ports) {
    obs_put(x, r, true, tgt, OE_IFACE, false);
    x.c->kports[slot][tgt] = (uint8_t)(PB_HAS | ports);
  }
  red_result(x, r, a, T_TRUE);
}
// ExploitAction._create_new_session (ExploitActions/ExploitAction.py:212-262): returns new session index or -1
// hd0: proc_head(x, tgt) as the action read it (nothing has written the target's row since)
CC4_HD int exploit_new_session(Ctx x, int r, int parent_sid, int tgt, const P8N& hd0) {
  (void)parent_sid;
  int pid = create_pid(x, tgt, hd0);
  if (!add_proc_n(x, tgt, hd0.n, pid, K_SHELL, 0)) return -1;
  x.hd[tgt].nsf = (uint8_t)((hd0.nsf & 0x0F) | ((((hd0.nsf >> 4) | HF_CMD) & ~HF_ESC_LAST) << 4));   // target_host.files.append(File('cmd.sh', density 0.9)) (ExploitAction.py:230-238)
  return rs_add(x, r, tgt, pid, RS_CHILD, x.w->rs_slot[r]);   // Session(parent=self.session) (ExploitAction.py:250-259); slot reserved by rs_reserve
}
// ExploitRemoteService.execute (AbstractActions/ExploitRemoteService.py:149-202) + selector (:37-69)
CC4_HD void red_exploit(Ctx x, int r, const Act& a) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  int si = rs_find_id(s, A, a.sid);
  if (si < 0) { red_result(x, r, a, T_FALSE); return; }
  const int slot = A.sord[si];
  const uint64_t sw = rs_word(s, slot);
  if (!(rsw_flags(sw) & RS_ABSTRACT)) { red_result(x, r, a, T_FALSE); return; }
  int src = rsw_host(sw), tgt = a.host;
  // two independent round trips side by side: the session's port knowledge of the target (cold row, HBM) and the target's
  // process list head (host table: length and first round)
  const P8N thd0 = proc_head(x, tgt);
  int known = x.c->kports[slot][tgt];
  if (!(known & PB_HAS)) { red_result(x, r, a, T_FALSE); return; }
  if (subnet_blocked(x, h_subnet(src), h_subnet(tgt))) { red_result(x, r, a, T_FALSE); return; }
  // DefaultExploitActionSelector: options in list order with non-zero weight
  // option list order: HTTPRFI(3.0) HTTPSRFI(4.0) SSH(0.1) SQLi(5.0) Haraka(6.0) -> bit i of `opts`
  enum { X_HTTPRFI = 0, X_HTTPSRFI = 1, X_SSH = 2, X_SQLI = 3, X_HARAKA = 4 };
  uint32_t opts = 0;
  if (known & PB_80) opts |= 1u << X_HTTPRFI;
  if (known & PB_443) opts |= 1u << X_HTTPSRFI;
  if (known & PB_22) opts |= 1u << X_SSH;
  if ((known & PB_3390) && (known & (PB_80 | PB_443))) opts |= 1u << X_SQLI;
  if (known & PB_25) opts |= 1u << X_HARAKA;
  if (opts == 0) { red_result(x, r, a, T_FALSE); return; }
  int sel = ctz32(opts);
  if (popc32(opts) > 1) {
    // top weight: Haraka > SQLi > HTTPSRFI > HTTPRFI > SSH; it is discarded, then a uniform choice among the rest
    int top = (opts >> X_HARAKA) & 1u ? X_HARAKA : ((opts >> X_SQLI) & 1u ? X_SQLI : ((opts >> X_HTTPSRFI) & 1u ? X_HTTPSRFI : X_HTTPRFI));
    opts &= ~(1u << top);
    sel = nth_bit(opts, (int)rng_below(x.r, (uint32_t)popc32(opts)));
    (void)rng_random(x.r);  // `elif random() < odds_of_top_choice` with odds 0
  }
  const int tnp = thd0.n;
  if (sel == X_SSH) {
    // SSHBruteForce.execute (ExploitActions/SSHBruteForce.py:24-84)
    uint8_t* work = reinterpret_cast<uint8_t*>(x.w->scratch + 6 * r);   // 24 bytes per red agent
    int nh = route(src, tgt, work);
    for (int i = 0; i < nh; ++i) if (!rng_random_le(x.r, 0.050000000000000044)) { ev_conn(x, work[12 + i]); ev_log(x, 200 + r, work[12 + i], 0, tgt, 0, src, 22, 0); }  // 1 - 0.95 in float64
    int vp = -1;
    for (int i0 = 0; i0 < tnp && vp < 0; i0 += 8) {
      P8 q = p8_of(thd0);
      if (i0) q = proc_load8(x, tgt, i0);
      CC4_UNROLL for (int k = 7; k >= 0; --k) if (i0 + k < tnp && pw_kind(q.v[k]) == K_SSHD) vp = i0 + k;
    }
    if (vp < 0) { red_result(x, r, a, T_FALSE); return; }
    obs_put(x, r, true, tgt, OE_IFACE, false);   // obs.add_process(target_process) -> interface of the target
    const int bf_port = eph_port(x, tgt);         // local_port
    ev_conn(x, tgt);                              // _create_brute_force_event: 10 connection events
    ev_log(x, 200 + r, tgt, 0, tgt, 22, src, bf_port, 0, 10);
    int ni = exploit_new_session(x, r, a.sid, tgt, thd0);
    if (ni < 0) { red_result(x, r, a, T_FALSE); return; }
    const uint64_t nw = rs_at(s, A, ni);
    ev_proc_red(x, r, tgt, rsw_pid(nw));       // _create_new_session_event (always for SSH)
    ev_log(x, 200 + r, tgt, 1, 0xFF, 0, 0xFF, 0, rsw_pid(nw));
    obs_put(x, r, true, tgt, OE_SESS | OE_IFACE | OE_SYSHN, false);
    obs_put(x, r, true, src, OE_IFACE, false);
    A.h.new_sess_host = (uint8_t)tgt; A.h.new_sess_id = (uint16_t)rsw_id(nw);
    red_result(x, r, a, T_TRUE);
    return;
  }
  // ExploitAction.sim_exploit (ExploitActions/ExploitAction.py:48-116)
  int vp = -1, vk = 0;
  {
    // the process kinds that answer the chosen exploit, as a bit set over K_*
    const uint32_t want = sel == X_HTTPRFI ? ((1u << K_APACHE) | (1u << K_DEC_APACHE) | (1u << K_DEC_VSFTPD))   // WEBSERVER @80
                        : sel == X_HTTPSRFI ? (1u << K_DEC_TOMCAT)                                              // WEBSERVER @443
                        : sel == X_SQLI ? (1u << K_MYSQL)                                                       // MYSQL @3390
                        : ((1u << K_SMTP) | (1u << K_DEC_HARAKA));                                              // SMTP @25
    for (int i0 = 0; i0 < tnp && vp < 0; i0 += 8) {
      P8 q = p8_of(thd0);
      if (i0) q = proc_load8(x, tgt, i0);
      CC4_UNROLL for (int k = 7; k >= 0; --k) if (i0 + k < tnp && ((want >> pw_kind(q.v[k])) & 1u)) { vp = i0 + k; vk = pw_kind(q.v[k]); }
    }
  }
  if (vp < 0) { red_result(x, r, a, T_FALSE); return; }
  bool decoy = kind_is_decoy(vk);
  const int xport = sel == X_HTTPRFI ? 80 : (sel == X_HTTPSRFI ? 443 : (sel == X_SQLI ? 3390 : 25));
  if (decoy) { ev_conn(x, tgt); int ep = eph_port(x, tgt, 1); ev_log(x, 200 + r, tgt, 0, tgt, xport, src, ep, 0); }  // _get_vulnerable_process alert (ExploitAction.py:189-196)
  obs_put(x, r, true, tgt, OE_IFACE, false);
  bool works = (sel == X_SQLI) || decoy;  // rfi only on decoys; real Haraka is 2.8.9 (HarakaRCE.py:19, HTTPRFI.py:18)
  if (!works) { red_result(x, r, a, T_FALSE); return; }
  if (rng_random_le(x.r, 0.0)) { red_result(x, r, a, T_FALSE); return; }  // (1 - success_rate) >= random()
  const int lport = eph_port(x, tgt, 2);  // local_port
  if (decoy) { ev_conn(x, tgt); int ep = eph_port(x, tgt, 3); ev_log(x, 200 + r, tgt, 0, tgt, lport, src, ep, 0); red_result(x, r, a, T_FALSE); return; }
  int ni = exploit_new_session(x, r, a.sid, tgt, thd0);
  if (ni < 0) { red_result(x, r, a, T_FALSE); return; }
  const uint64_t nw = rs_at(s, A, ni);
  if (!rng_random_le(x.r, 0.050000000000000044)) { ev_proc_red(x, r, tgt, rsw_pid(nw)); ev_log(x, 200 + r, tgt, 1, 0xFF, 0, 0xFF, 0, rsw_pid(nw)); }
  obs_put(x, r, true, tgt, OE_SESS | OE_IFACE | OE_SYSHN, false);
  obs_put(x, r, true, src, OE_IFACE, false);
  A.h.new_sess_host = (uint8_t)tgt; A.h.new_sess_id = (uint16_t)rsw_id(nw);
  red_result(x, r, a, T_TRUE);
}
// PrivilegeEscalate.execute (AbstractActions/PrivilegeEscalate.py:127-179)
CC4_HD void red_privesc(Ctx x, int r, const Act& a) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  int h = a.host;
  const HostSess hs = rs_on_host(s, A, h);
  const int n = hs.n;
  int target = hs.first_root;
  if (n == 0) { red_result(x, r, a, T_FALSE); return; }
  if (target < 0) {
    target = rs_kth_on_host(s, A, h, (int)rng_below(x.r, (uint32_t)n));   // choice(sessions on the host)
    // DefaultEscalateActionSelector (PrivilegeEscalate.py:52-66): self.session must exist and be a RedAbstractSession,
    // else no sub-action -> Observation(False); then V4L2KernelExploit via TargetedLocalAction.execute
    { int ss = rs_find_id(s, A, a.sid);
      if (ss < 0 || !(rsw_flags(rs_at(s, A, ss)) & RS_ABSTRACT)) { red_result(x, r, a, T_FALSE); return; } }
    const int tslot = A.sord[target];
    s->spool[tslot].flags |= RS_ROOT;  // EscalateAction.__upgrade_session (EscalateAction.py:57-87)
    hd_touch(x, h);
    hd_set_files(x.hd[h], hd_files(x.hd[h]) | HF_ESC | HF_ESC_LAST);   // ... which also drops File('escalate.sh', density 0.9) on the host (:70-77)
    int pi = find_proc(x, h, s->spool[tslot].pid);
    if (pi >= 0) proc_put(x, h, pi, proc_get(x, h, pi) | ((uint32_t)PF_ROOT << 24));
  }
  obs_put(x, r, false, h, OE_SESS, false);
  { const uint64_t tw = rs_at(s, A, target); if (rsw_flags(tw) & RS_ABSTRACT) as_know_sid(x, r, rsw_id(tw)); }
  // ExploreHost (EscalateAction.py:90-106): host.info links exist only on server_host_0 (ESG.py:418-468)
  if (rs_find_id(s, A, a.sid) >= 0 && h_slot(h) == 11) {
    uint32_t m = info_links(h_subnet(h));
    for (int sn = 0; sn < NSUB; ++sn) if ((m >> sn) & 1u) obs_put(x, r, false, h_make(sn, 11), OE_IFACE, false);
  }
  red_result(x, r, a, T_TRUE);
}
// Impact.execute (AbstractActions/Impact.py:38-88)
CC4_HD void red_impact(Ctx x, int r, const Act& a) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  int h = a.host;
  if (rs_on_host(s, A, h).first_root < 0) { red_result(x, r, a, T_FALSE); return; }   // no session, or none of them root
  HostDyn& d = x.hd[h];
  const SvTab sv = svc_load(d);
  int si = -1;
  uint32_t sw = 0;   // the entry found (kept beside its index: a run-time index into the table would put the table into scratch memory)
  CC4_UNROLL for (int i = MAXSV - 1; i >= 0; --i) if (i < sv.n && svw_kind(sv.w[i]) == K_OT && (svw_st(sv.w[i]) & SV_ACTIVE)) { si = i; sw = sv.w[i]; }   // the first one
  if (si < 0) { red_result(x, r, a, T_FALSE); return; }
  if (rs_find_id(s, A, a.sid) < 0) { red_result(x, r, a, T_FALSE); return; }  // StopService needs self.session too
  hd_touch(x, h);
  d.svcs[si].st = (uint8_t)(svw_st(sw) & ~SV_ACTIVE);          // Host.stop_service (Host.py:295-300)
  int pi = find_proc(x, h, (int)(sw & 0xFFFF));                 // State.remove_process (State.py:390-418)
  if (pi >= 0) remove_proc_at(x, h, pi);
  obs_put(x, r, false, h, 0, false);
  red_result(x, r, a, T_TRUE);
}
// DegradeServices.execute (AbstractActions/DegradeServices.py:38-82)
CC4_HD void red_degrade(Ctx x, int r, const Act& a) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  int h = a.host;
  if (rs_on_host(s, A, h).first_root < 0) { red_result(x, r, a, T_FALSE); return; }   // no session, or none of them root
  HostDyn& d = x.hd[h];
  int n = 0;
  hd_touch(x, h);
  const SvTab sv = svc_load(d);
  CC4_UNROLL for (int i = 0; i < MAXSV; ++i) if (i < sv.n && (svw_st(sv.w[i]) & SV_ACTIVE)) {
    n++;
    int rel = svw_st(sv.w[i]) & 0x7F;
    if (rel > 0) rel--;                                       // Service.degrade_service_reliability (Service.py:33-40)
    d.svcs[i].st = (uint8_t)(SV_ACTIVE | rel);
  }
  if (n == 0) { red_result(x, r, a, T_FALSE); return; }
  obs_put(x, r, false, h, 0, false);
  red_result(x, r, a, T_TRUE);
}
// DiscoverDeception.execute (AbstractActions/DiscoverDeception.py:44-101)
CC4_HD void red_deception(Ctx x, int r, const Act& a) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  if (rs_find_id(s, A, a.sid) < 0) { red_result(x, r, a, T_FALSE); return; }
  int tgt = a.host;
  // detection_rate / fp_rate (DiscoverDeception.py:40-41), or the ones the action object came with (ExtAct)
  const double det_rate = (x.ext && (a.busy & AQ_RATE0)) ? x.c->xrate[r][0] : 0.5;
  const double fp_rate = (x.ext && (a.busy & AQ_RATE1)) ? x.c->xrate[r][1] : 0.1;
  const P8N hd0 = proc_head(x, tgt);   // length and first round in one round trip
  const int np = hd0.n;
  for (int i0 = 0; i0 < np; i0 += 8) {
    P8 q = p8_of(hd0);
    if (i0) q = proc_load8(x, tgt, i0);
    CC4_UNROLL for (int k = 0; k < 8; ++k) {
      if (i0 + k >= np) continue;
      bool decoy = kind_is_decoy(pw_kind(q.v[k]));
      bool rep = false;
      if (rng_random_le(x.r, det_rate) && decoy) rep = true;
      else if (rng_random_le(x.r, fp_rate) && !decoy) rep = true;
      if (rep) obs_put(x, r, false, tgt, OE_IFACE, false);
    }
  }
  red_result(x, r, a, T_TRUE);
}
// RedSessionCheck.execute (ConcreteActions/RedSessionCheck.py:9-65)
CC4_HD void red_session_check(Ctx x, int r) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  obs_first(x, r, T_TRUE, RA_NONE, 0, 0);
  if (A.h.nsess == 0) return;
  // the primary (id 0) sits first, or last after a promotion / restore re-insert: look there before scanning
  if (rsw_id(rs_at(s, A, 0)) != 0 && rsw_id(rs_at(s, A, A.h.nsess - 1)) != 0 && rs_find_id(s, A, 0) < 0) {
    int c = (int)rng_below(x.r, (uint32_t)A.h.nsess);
    rs_move_to_end(s, A, c, 0);   // active_sessions.pop(old_id); ident = 0; re-inserted last (RedSessionCheck.py:36-45)
    // every other session's parent becomes new_primary.name, which is None unless the promoted session is the scenario's
    // own 'red_session_0' (never the case: that one holds id 0 from the start) (RedSessionCheck.py:52-55)
    for (int i = 0; i < A.h.nsess; ++i) s->spool[A.sord[i]].flags &= (uint8_t)~RS_CHILD;
  }
  // The observation lists every session as a hostname-keyed entry with Sessions / Interface{ip, Subnet} / System info.
  // Instead of materialising one entry per session, its three effects are applied in bulk from RedAgent.live_hosts (the
  // exact set of hosts in the listing): ActionSpace knowledge here, FSM knowledge in fsm_observe (rsc_listed).
  A.h.rsc_listed = 1;
  if (!A.h.rsc_dirty) return;      // same session table as at the last listing: nothing new to learn
  const B5 live = b5_load(A.live_hosts);
  { const B5 ip = b5_load(A.as_ip), hn = b5_load(A.as_hn); b5_store(A.as_ip, b5_or(ip, live)); b5_store(A.as_hn, b5_or(hn, live)); }
  {  // the subnets of the session hosts = the subnets with a live host (ids 17 sn .. 17 sn + 16)
    uint32_t sub = 0;
    CC4_UNROLL for (int sn = 0; sn < NSUB; ++sn) {
      const int lo = sn * SLOTS, w = lo >> 5, off = lo & 31;
      uint32_t bits = live.w[w] >> off;
      if (off > 32 - SLOTS && w + 1 < 5) bits |= live.w[w + 1] << (32 - off);
      if (bits & ((1u << SLOTS) - 1u)) sub |= 1u << sn;
    }
    A.h.as_subnet |= (uint16_t)sub;
  }
  // session ids new to the ActionSpace, in session order: eight records and their eight known-id words per round, so the
  // common case (everything known) costs two LDS round trips per eight sessions
  const int n = A.h.nsess;
  for (int i0 = 0; i0 < n; i0 += 8) {
    const S8 q = rs_load8(s, A, i0);
    uint32_t kw[8];
    CC4_UNROLL for (int k = 0; k < 8; ++k) kw[k] = A.known_bm[(rsw_id(q.v[k]) >> 5) & 7];
    CC4_UNROLL for (int k = 0; k < 8; ++k) {
      if (i0 + k >= n || !(rsw_flags(q.v[k]) & RS_ABSTRACT)) continue;
      const int id = rsw_id(q.v[k]);
      if (id < 256 && ((kw[k] >> (id & 31)) & 1u)) continue;
      as_know_sid(x, r, id);   // rare; two new records of one round never share an id, so the words read above stay valid
    }
  }
  A.h.rsc_dirty = 0;
}
// Withdraw.execute (ConcreteActions/Withdraw.py:38-91) + StopProcess(stop_all=True).  a.host = ip_address (unused beyond the
// route, which always exists), a.arg = hostname
CC4_HD void red_withdraw(Ctx x, int r, const Act& a) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  if (rs_find_id(s, A, a.sid) < 0) { red_result(x, r, a, T_FALSE); return; }
  const int h = a.arg;
  // all_agents_sessions = child sessions on the host + parent sessions with ident != 0 + (the acting session if it sits there)
  // held as ids, because killing one shifts the others
  uint16_t* ids = reinterpret_cast<uint16_t*>(x.w->scratch + 36);   // after the per-agent route work areas
  int n = 0;
  const int cap = (int)((sizeof(x.w->scratch) - 36 * 4) / 2);
  for (int pass = 0; pass < 2; ++pass)
    for (int i = 0; i < A.h.nsess; ++i) {
      const uint64_t w = rs_at(s, A, i);
      if (rsw_host(w) != h) continue;
      bool child = (rsw_flags(w) & RS_CHILD) != 0;
      if ((pass == 0 && child) || (pass == 1 && !child && rsw_id(w) != 0)) { if (n < cap) ids[n++] = (uint16_t)rsw_id(w); else set_err(x, E_RSESS_OVERFLOW); }
    }
  { int self = rs_find_id(s, A, a.sid); if (rsw_host(rs_at(s, A, self)) == h) { if (n < cap) ids[n++] = (uint16_t)a.sid; else set_err(x, E_RSESS_OVERFLOW); } }
  if (n == 0) { red_result(x, r, a, T_FALSE); return; }
  int ok = T_FALSE;
  for (int k = 0; k < n; ++k) {
    // TargetedLocalAction.execute: both sessions must still exist (the acting one may have been killed earlier in this loop)
    int self = rs_find_id(s, A, a.sid), ti = rs_find_id(s, A, ids[k]);
    if (self < 0 || ti < 0) { ok = T_FALSE; break; }
    int pid = rsw_pid(rs_at(s, A, ti));
    int pi = find_proc(x, h, pid);
    if (pi < 0) { ok = T_FALSE; break; }
    kill_process(x, h, pi, proc_get(x, h, pi));   // StopProcess.kill_process with stop_all: root processes die too
    ok = T_TRUE;
  }
  red_result(x, r, a, ok);
}
CC4_HD void red_execute(Ctx x, int r, const Act& a) {
  switch (a.type) {
    case RA_DRS: red_drs(x, r, a); break;
    // detection_rate: the class's own, or the one the action object came with (ExtAct: the reference's tests set it to 0 and 1)
    case RA_AGGR: red_scan(x, r, a, (x.ext && (a.busy & AQ_RATE0)) ? x.c->xrate[r][0] : 0.75); break;
    case RA_STEALTH: red_scan(x, r, a, (x.ext && (a.busy & AQ_RATE0)) ? x.c->xrate[r][0] : 0.25); break;
    case RA_DECEPTION: red_deception(x, r, a); break;
    case RA_EXPLOIT: red_exploit(x, r, a); break;
    case RA_PRIVESC: red_privesc(x, r, a); break;
    case RA_IMPACT: red_impact(x, r, a); break;
    case RA_DEGRADE: red_degrade(x, r, a); break;
    case RA_INVALID: obs_first(x, r, T_FALSE, RA_NONE, 0, 0); break;
    case RA_WITHDRAW: red_withdraw(x, r, a); break;
    default: obs_first(x, r, T_UNKNOWN, RA_NONE, 0, 0); break;  // Sleep -> Observation()
  }
}

// ------------------------------------------------------------------ FiniteStateRedAgent (Agents/SimpleAgents/FiniteStateRedAgent.py)
CC4_HD int fsm_next(int cur, int act, bool success) {
  // state_transitions_success / _failure (:441-452, :481-492), one row per state packed as 9 nibbles (column = action index
  // in action_list order, 0xF = None = keep), selected without a memory table:
  //   success  K:[KD,S,S,-,-,-,-,-,-] KD:[KD,SD,SD,..] S:[SD,-,-,S,U,..] SD:[SD,-,-,SD,UD,..] U:[UD,-,-,-,-,R,-,-,S]
  //            UD:[UD,..,RD,-,-,SD] R:[RD,-,-,-,-,-,R,R,S] RD:[RD,..,RD,RD,SD] F:[F,-..]
  //   failure  every defined entry keeps the current state
  uint64_t row;
  if (success) switch (cur) {
    case FS_K: row = 0xffffff221ull; break;  case FS_KD: row = 0xffffff331ull; break;
    case FS_S: row = 0xffff42ff3ull; break;  case FS_SD: row = 0xffff53ff3ull; break;
    case FS_U: row = 0x2ff6ffff5ull; break;  case FS_UD: row = 0x3ff7ffff5ull; break;
    case FS_R: row = 0x266fffff7ull; break;  case FS_RD: row = 0x377fffff7ull; break;
    default: row = 0xffffffff8ull; break;
  } else switch (cur) {
    case FS_K: row = 0xffffff000ull; break;  case FS_KD: row = 0xffffff111ull; break;
    case FS_S: row = 0xffff22ff2ull; break;  case FS_SD: row = 0xffff33ff3ull; break;
    case FS_U: row = 0x4ff4ffff4ull; break;  case FS_UD: row = 0x5ff5ffff5ull; break;
    case FS_R: row = 0x666fffff6ull; break;  case FS_RD: row = 0x777fffff7ull; break;
    default: row = 0xffffffff8ull; break;
  }
  int v = (int)((row >> (4 * act)) & 0xF);
  return v == 0xF ? 0xFF : v;
}
// (bitmap first, byte store last: with the opposite order hipcc 7.2 -O3 turns the uniform `fsm_step == 0 ? U : K` select
// feeding the byte store into an s_cselect on a stale SCC in one unrolled copy of the caller's loop -- tools/isa_scan.py
// checks the built ISA for that pattern)
// host_states[h] as a nibble: FS_* + 1, 0 = absent (so a zeroed row knows no host)
CC4_HD int fsm_get(const RedAgent& A, int h) {
  const int v = (A.fsm_st4[h >> 1] >> (4 * (h & 1))) & 0xF;
  return v ? v - 1 : FS_NONE;
}
CC4_HD void fsm_put(RedAgent& A, int h, int st) {
  const int sh = 4 * (h & 1);
  A.fsm_st4[h >> 1] = (uint8_t)((A.fsm_st4[h >> 1] & ~(0xF << sh)) | ((st + 1) << sh));
}
CC4_HD void fsm_set_state(RedAgent& A, int h, int st) {
  if (st >= FS_U && st <= FS_RD) bit_set(A.fsm_ur, h); else bit_clr(A.fsm_ur, h);
  if ((st & 1) || st == FS_F) bit_set(A.fsm_nodrs, h); else bit_clr(A.fsm_nodrs, h);
  fsm_put(A, h, st);
}
CC4_HD void fsm_apply(Ctx x, int r, RedHdr& H, int h, int act, bool success) {  // _host_state_transition inner loop (:147-167)
  RedAgent& A = x.s->red[r];
  int cur = fsm_get(A, h);
  if (cur == FS_NONE) return;
  int nx = fsm_next(cur, act, success);
  if (nx == FS_U) nx = ((red_allowed_mask(r) >> h_subnet(h)) & 1u) ? FS_U : FS_F;
  if (nx == 0xFF) nx = cur;
  if (nx != cur) fsm_set_state(A, h, nx);
  if (nx == FS_F && cur != FS_F) {  // leaves known_hosts for good
    int n = 0;
    for (int i = 0; i < H.fsm_n; ++i) if (A.fsm_order[i] != h) A.fsm_order[n++] = A.fsm_order[i];
    H.fsm_n = (uint8_t)n;
  }
}
// H: the agent's scalar fields in registers (RedHdr); the caller loaded them and stores them back
CC4_HD void fsm_observe(Ctx x, int r, RedHdr& H) {
  RedAgent& A = x.s->red[r];
  CC4_AT0(x);
  // Mid-action agent whose observation is just the RedSessionCheck listing of an unchanged session table: nothing below can
  // change anything.  No transition (success is IN_PROGRESS), no entry to process, the listing's hosts were merged when the
  // table last changed (live <= known, hostnames known), and the removal check found fsm_ur & ~live empty then -- neither
  // set has moved since (fsm_ur only moves in this function, live only with fsm_dirty).
  if (H.obs_success == T_IN_PROGRESS && H.nobs == 0 && H.rsc_listed && !H.fsm_dirty) return;
  // 1. _host_state_transition (:124-167)
  if (H.obs_act_type <= RA_WITHDRAW && H.obs_success != T_IN_PROGRESS && H.obs_success != 0) {
    bool ok = H.obs_success == T_TRUE;
    int t = H.obs_act_type;
    if (t == RA_DRS) {
      // every known host of the pinged subnet (ids subnet*17 .. +16); the order of the per-host transitions is immaterial
      // on success only K, S, U, R move (to KD, SD, UD, RD): visit just those hosts
      const int lo = H.obs_act_arg * SLOTS, hi = lo + SLOTS - 1;
      for (int w = lo >> 5; w <= (hi >> 5) && w < 5; ++w) {
        uint32_t m = A.fsm_known[w];
        if (ok) m &= ~A.fsm_nodrs[w];
        const int l = lo - 32 * w, u = hi - 32 * w;
        if (l > 0) m &= 0xFFFFFFFFu << l;
        if (u < 31) m &= (2u << u) - 1u;
        while (m) { int b = ctz32(m); m &= m - 1; fsm_apply(x, r, H, w * 32 + b, t, ok); }
      }
    } else if (t == RA_PRIVESC || t == RA_IMPACT || t == RA_DEGRADE) {
      int h = H.obs_act_host;  // matched through host_states[ip]['hostname']
      if (bit_get(A.fsm_known, h) && bit_get(A.fsm_hn, h)) fsm_apply(x, r, H, h, t, ok);
    } else {
      fsm_apply(x, r, H, H.obs_act_host, t, ok);
    }
  }
  CC4_AT(x, 3);
  // 2. _process_new_observations (:190-250)
  B5 seen = b5_zero();   // hosts whose observation entry carries Sessions
  for (int i = 0; i < H.nobs; ++i) {
    int h = A.obs[i].host; int f = A.obs[i].flags;
    if (f & OE_SESS) b5_set(seen, h);
    bool hn = !(f & OE_KEY_IP) || (f & OE_SYSHN);
    bool ip = (f & OE_KEY_IP) || (f & OE_IFACE);
    if (!ip) {
      // ip looked up through a known hostname; unknown -> reference would key host_states[None]
      if (!(bit_get(A.fsm_known, h) && bit_get(A.fsm_hn, h))) { set_err(x, E_UNREACHABLE); }
      continue;
    }
    if (!bit_get(A.fsm_known, h)) {
      fsm_set_state(A, h, H.fsm_step == 0 ? FS_U : FS_K);
      A.fsm_order[H.fsm_n++] = (uint8_t)h;
      bit_set(A.fsm_known, h);
    }
    if (hn) bit_set(A.fsm_hn, h);
  }
  CC4_AT(x, 4);
  // the RedSessionCheck listing (comes last in the observation's key order): every session host is hostname-known, carries
  // Sessions, and is new to host_states iff it is not in fsm_known -- added in session (dict) order
  if (H.rsc_listed) {
    const B5 live = b5_load(A.live_hosts), known = b5_load(A.fsm_known), hn = b5_load(A.fsm_hn);
    seen = b5_or(seen, live);
    if (b5_any(b5_andn(live, known)))
      for (int i = 0; i < H.nsess; ++i) {
        int h = rsw_host(rs_at(x.s, A, i));
        if (bit_get(A.fsm_known, h)) continue;
        fsm_set_state(A, h, H.fsm_step == 0 ? FS_U : FS_K);
        A.fsm_order[H.fsm_n++] = (uint8_t)h;
        bit_set(A.fsm_known, h);
      }
    b5_store(A.fsm_hn, b5_or(hn, live));
    H.fsm_dirty = 0;
  }
  CC4_AT(x, 5);
  // 3. _session_removal_state_change (:169-188): hosts in U/UD/R/RD without a Sessions entry in the observation -> KD
  {
    const B5 lost = b5_andn(b5_load(A.fsm_ur), seen);
    if (b5_any(lost))
      CC4_UNROLL for (int w = 0; w < 5; ++w) {
        uint32_t m = lost.w[w];
        while (m) { int b = ctz32(m); m &= m - 1; fsm_set_state(A, w * 32 + b, FS_KD); }
      }
  }
  CC4_AT(x, 6);
}
// DiscoveryFSRed._choose_host (FiniteStateRedAgent.py:252-293 with FSMRedVariants.py:95-110): host-state priorities
// {K,KD,S,SD: 20, U,UD: 10, R,RD: 0} and prioritise_servers.  The float arithmetic restates the Python expressions
// operation by operation (probs = (p/100) * (1/(sum/100)); numpy choice(p) = cumsum, /= last, searchsorted right).
CC4_HD int fsm_choose_host_discovery(Ctx x, int r, RedHdr& H) {
  RedAgent& A = x.s->red[r];
  // available_states in order of first appearance among the known hosts, packed as nibbles
  uint32_t order = 0; int nst = 0; uint32_t seen = 0;
  for (int i = 0; i < H.fsm_n && nst < 8; ++i) {
    int st = fsm_get(A, A.fsm_order[i]);
    if (!((seen >> st) & 1u)) { seen |= 1u << st; order |= (uint32_t)st << (4 * nst); nst++; }
  }
  int sum = 0;
  for (int k = 0; k < nst; ++k) { int st = (order >> (4 * k)) & 0xF; sum += st <= FS_SD ? 20 : (st <= FS_UD ? 10 : 0); }
  int chosen_state;
  if (sum > 0) {
    double mult = 1.0 / ((double)sum / 100.0);
    double last = 0.0;
    for (int k = 0; k < nst; ++k) { int st = (order >> (4 * k)) & 0xF; int p = st <= FS_SD ? 20 : (st <= FS_UD ? 10 : 0); last = last + ((double)p / 100.0) * mult; }
    double u = rng_random(x.r);
    double run = 0.0; int idx = 0;
    for (int k = 0; k < nst; ++k) {
      int st = (order >> (4 * k)) & 0xF; int p = st <= FS_SD ? 20 : (st <= FS_UD ? 10 : 0);
      run = run + ((double)p / 100.0) * mult;
      if (run / last <= u) idx++;
    }
    if (idx >= nst) idx = nst - 1;
    chosen_state = (order >> (4 * idx)) & 0xF;
  } else {
    chosen_state = (order >> (4 * (int)rng_below(x.r, (uint32_t)nst))) & 0xF;
  }
  int n_all = 0, n_srv = 0;
  for (int i = 0; i < H.fsm_n; ++i) {
    int h = A.fsm_order[i];
    if (fsm_get(A, h) != chosen_state) continue;
    n_all++;
    if (h_is_server(h) && bit_get(A.fsm_hn, h)) n_srv++;
  }
  int want_srv = -1;  // -1: any host of the state, 1: servers only, 0: non-servers only
  if (n_all > 1 && n_srv > 0) {
    double i01 = rng_random(x.r);
    want_srv = (i01 <= 0.75 || n_srv == n_all) ? 1 : 0;
  }
  int cnt = want_srv < 0 ? n_all : (want_srv ? n_srv : n_all - n_srv);
  int c = (int)rng_below(x.r, (uint32_t)cnt);
  for (int i = 0; i < H.fsm_n; ++i) {
    int h = A.fsm_order[i];
    if (fsm_get(A, h) != chosen_state) continue;
    bool srv = h_is_server(h) && bit_get(A.fsm_hn, h);
    if (want_srv >= 0 && (int)srv != want_srv) continue;
    if (c-- == 0) return h;
  }
  return A.fsm_order[0];
}
// get_action (:58-122) incl. _choose_host (:252-293) and _choose_host_and_action (:296-336)
// observed: fsm_observe (which draws nothing) has already run for this step (step_red_observe)
CC4_HD Act fsm_get_action(Ctx x, int r, RedHdr& H, bool observed = false) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  Act out; out.type = RA_SLEEP; out.host = 0; out.arg = 0; out.ticks = 1; out.sid = 0; out.busy = 0;
  CC4_FT0(x, r);
  if (!observed) fsm_observe(x, r, H);
  CC4_FT(x, r, 5);
  if (H.obs_success == T_IN_PROGRESS) { H.fsm_step++; return out; }
  int n = H.fsm_n;  // fsm_order holds exactly the non-'F' hosts, in host_states insertion order
  if (n == 0) { set_err(x, E_FSM_NO_HOST); H.fsm_step++; return out; }
  const bool discovery = (s->policy & 3) == RP_DISCOVERY;
  int host;
  if (!discovery) host = A.fsm_order[rng_below(x.r, (uint32_t)n)];
  else host = fsm_choose_host_discovery(x, r, H);
  // options in red_actions list order (ESG.py:764-768) with state_transitions_probability (:540-549).  All probabilities
  // are multiples of 1/4, so cdf.searchsorted(u, 'right') == #{i : 4*cdf[i] <= floor(4u)}.  Packed per state:
  // low 16 bits = option nibbles, high 16 bits = 4*cdf nibbles (unused slots = 15)
  CC4_FT(x, r, 6);
  uint32_t pk;
  const int host_state = fsm_get(A, host);
  if (!discovery) switch (host_state) {
    case FS_K:  pk = 0xF432u << 16 | (RA_DRS | RA_AGGR << 4 | RA_STEALTH << 8); break;                          // .5 .25 .25
    case FS_KD: pk = 0xFF42u << 16 | (RA_AGGR | RA_STEALTH << 4); break;                                         // .5 .5
    case FS_S:  pk = 0xF431u << 16 | (RA_DRS | RA_EXPLOIT << 4 | RA_DECEPTION << 8); break;                     // .25 .5 .25
    case FS_SD: pk = 0xFF43u << 16 | (RA_EXPLOIT | RA_DECEPTION << 4); break;                                    // .75 .25
    case FS_U:  pk = 0xF442u << 16 | (RA_DRS | RA_PRIVESC << 4 | RA_WITHDRAW << 8); break;                      // .5 .5 0
    case FS_UD: pk = 0xFF44u << 16 | (RA_PRIVESC | RA_WITHDRAW << 4); break;                                     // 1 0
    case FS_R:  pk = 0x4432u << 16 | (RA_DRS | RA_DEGRADE << 4 | RA_IMPACT << 8 | RA_WITHDRAW << 12); break;    // .5 .25 .25 0
    default:    pk = 0xF442u << 16 | (RA_DEGRADE | RA_IMPACT << 4 | RA_WITHDRAW << 8); break;                   // RD: .5 .5 0
  } else switch (host_state) {  // DiscoveryFSRed.state_transitions_probability (FSMRedVariants.py:111-122)
    case FS_K:  pk = 0xF441u << 16 | (RA_DRS | RA_AGGR << 4 | RA_STEALTH << 8); break;                          // .25 .75 0
    case FS_KD: pk = 0xFF44u << 16 | (RA_AGGR | RA_STEALTH << 4); break;                                         // 1 0
    case FS_S:  pk = 0xF441u << 16 | (RA_DRS | RA_EXPLOIT << 4 | RA_DECEPTION << 8); break;                     // .25 .75 0
    case FS_SD: pk = 0xFF44u << 16 | (RA_EXPLOIT | RA_DECEPTION << 4); break;                                    // 1 0
    case FS_U:  pk = 0xF440u << 16 | (RA_DRS | RA_PRIVESC << 4 | RA_WITHDRAW << 8); break;                      // 0 1 0
    case FS_UD: pk = 0xFF44u << 16 | (RA_PRIVESC | RA_WITHDRAW << 4); break;                                     // 1 0
    case FS_R:  pk = 0x4444u << 16 | (RA_DRS | RA_DEGRADE << 4 | RA_IMPACT << 8 | RA_WITHDRAW << 12); break;    // 1 0 0 0
    default:    pk = 0xF442u << 16 | (RA_DEGRADE | RA_IMPACT << 4 | RA_WITHDRAW << 8); break;                   // RD: .5 .5 0
  }
  const int fl = rng_random_quarter(x.r);   // floor(4 u)
  int k = 0;
  for (int i = 0; i < 4; ++i) if ((int)((pk >> (16 + 4 * i)) & 0xF) <= fl) k++;
  int t = (int)((pk >> (4 * k)) & 0xF);
  out.type = (uint8_t)t; out.host = (uint8_t)host;
  // parameters in constructor-signature order; only `subnet` and `session` can draw
  bool bad = false;
  if (t == RA_DRS) {
    uint32_t known = H.as_subnet;  // known subnets in dict (= SUBNET enum) order
    if (known == 0) bad = true; else out.arg = (uint8_t)nth_bit(known, (int)rng_below(x.r, (uint32_t)popc32(known)));
  }
  if ((t == RA_PRIVESC || t == RA_IMPACT || t == RA_DEGRADE) && !bit_get(A.fsm_hn, host)) bad = true;
  if (!bad) {
    if (H.nknown == 0) bad = true; else out.sid = x.c->known_sid[r][rng_below(x.r, (uint32_t)H.nknown)];
  }
  if (bad) { set_err(x, E_UNREACHABLE); out.type = RA_SLEEP; }  // reference would re-draw with p not summing to 1 and raise
  out.ticks = (uint8_t)red_duration(out.type);
  H.fsm_step++;
  CC4_FT(x, r, 7);
  return out;
}
// RandomSelectRedAgent.get_action (Agents/SimpleAgents/RandomSelectRedAgent.py:33-103): uniform command, then uniform
// parameters from the ActionSpace entries that are True.  Commands in red_actions order (ESG.py:764-768), parameters in
// constructor order; a draw is made only when there is more than one option.
CC4_HD Act random_red_get_action(Ctx x, int r, RedHdr& H) {
  RedAgent& A = x.s->red[r];
  Act out; out.type = RA_SLEEP; out.host = 0; out.arg = 0; out.ticks = 1; out.sid = 0; out.busy = 0;
  const int nip = popc32(A.as_ip[0]) + popc32(A.as_ip[1]) + popc32(A.as_ip[2]) + popc32(A.as_ip[3]) + popc32(A.as_ip[4]);
  const int nhn = popc32(A.as_hn[0]) + popc32(A.as_hn[1]) + popc32(A.as_hn[2]) + popc32(A.as_hn[3]) + popc32(A.as_hn[4]);
  const int nsub = popc32(H.as_subnet), nsid = H.nknown;
  // valid_commands: a command is listed only if each of its parameters has at least one True option
  // order: DRS, Aggressive, Stealth, Exploit, PrivEsc, Degrade, DiscoverDeception, Impact, Withdraw, Sleep
  const uint8_t types[10] = {RA_DRS, RA_AGGR, RA_STEALTH, RA_EXPLOIT, RA_PRIVESC, RA_DEGRADE, RA_DECEPTION, RA_IMPACT, RA_WITHDRAW, RA_SLEEP};
  uint32_t valid = 0;
  for (int c = 0; c < 10; ++c) {
    int t = types[c];
    bool ok = true;
    if (t != RA_SLEEP && nsid == 0) ok = false;
    if (t == RA_DRS && nsub == 0) ok = false;
    if ((t == RA_AGGR || t == RA_STEALTH || t == RA_EXPLOIT || t == RA_DECEPTION || t == RA_WITHDRAW) && nip == 0) ok = false;
    if ((t == RA_PRIVESC || t == RA_DEGRADE || t == RA_IMPACT || t == RA_WITHDRAW) && nhn == 0) ok = false;
    if (ok) valid |= 1u << c;
  }
  const int t = types[nth_bit(valid, (int)rng_below(x.r, (uint32_t)popc32(valid)))];
  auto pick_sid = [&]() { return (int)x.c->known_sid[r][rng_below(x.r, (uint32_t)nsid)]; };
  auto pick_ip = [&]() { return nth_set(A.as_ip, 5, (int)rng_below(x.r, (uint32_t)nip)); };
  auto pick_hn = [&]() { return nth_set(A.as_hn, 5, (int)rng_below(x.r, (uint32_t)nhn)); };
  out.type = (uint8_t)t;
  switch (t) {
    case RA_DRS: out.arg = (uint8_t)nth_bit(H.as_subnet, (int)rng_below(x.r, (uint32_t)nsub)); out.sid = (uint16_t)pick_sid(); break;
    case RA_AGGR: case RA_STEALTH: case RA_DECEPTION: out.sid = (uint16_t)pick_sid(); out.host = (uint8_t)pick_ip(); break;
    case RA_EXPLOIT: out.host = (uint8_t)pick_ip(); out.sid = (uint16_t)pick_sid(); break;
    case RA_PRIVESC: case RA_DEGRADE: case RA_IMPACT: out.host = (uint8_t)pick_hn(); out.sid = (uint16_t)pick_sid(); break;
    case RA_WITHDRAW: out.sid = (uint16_t)pick_sid(); out.host = (uint8_t)pick_ip(); out.arg = (uint8_t)pick_hn(); break;
    default: break;
  }
  out.ticks = (uint8_t)red_duration(out.type);
  H.fsm_step++;   // self.step
  return out;
}
// SimulationController.replace_action_if_invalid (SC:1068-1112) for a red action
CC4_HD void red_validate(Ctx x, int r, const RedHdr& H, Act& a) {
  RedAgent& A = x.s->red[r];
  if (a.type >= RA_SLEEP) return;
  bool ok = true;
  if (!sid_known(A, x.c->known_sid[r], a.sid, H.nknown)) ok = false;
  if (a.type == RA_DRS) { if (!((H.as_subnet >> a.arg) & 1u)) ok = false; }
  else if (a.type == RA_PRIVESC || a.type == RA_IMPACT || a.type == RA_DEGRADE) { if (!bit_get(A.as_hn, a.host)) ok = false; }
  else if (a.type == RA_WITHDRAW) { if (!bit_get(A.as_ip, a.host) || !bit_get(A.as_hn, a.arg)) ok = false; }
  else { if (!bit_get(A.as_ip, a.host)) ok = false; }
  if (!ok) { a.type = RA_INVALID; a.ticks = 1; }
}

// ------------------------------------------------------------------ different_subnet_agent_reassignment (SC:820-903)
// word w of the host-id bitmap of the hosts in red agent r's allowed subnets
CC4_HD uint32_t red_zone_hosts(int r, int w) {
  uint32_t m = 0;
  for (int i = 0; i < red_nsub(r); ++i) {
    int lo = red_subnet_alloc(r, i) * SLOTS - 32 * w, hi = lo + SLOTS - 1;   // the subnet's id range relative to this word
    if (hi < 0 || lo > 31) continue;
    uint32_t up = hi >= 31 ? 0xFFFFFFFFu : ((2u << hi) - 1u);
    uint32_t dn = lo <= 0 ? 0xFFFFFFFFu : (0xFFFFFFFFu << lo);
    m |= up & dn;
  }
  return m;
}
CC4_HD void red_reassign(Ctx x, uint32_t foreign) {   // foreign: red_foreign_agents(s), != 0
  EnvState* s = x.s;
  // moves are collected first (the reference builds the list, then applies it), one packed word each in the shared work
  // area: from | to << 3 | list index << 6 | pool slot << 12
  uint32_t* mv = x.w->scratch; int nm = 0;
  const int cap = (int)(sizeof(x.w->scratch) / 4);
  for (int r = 0; r < NRED; ++r) {
    if (!((foreign >> r) & 1u)) continue;
    const RedAgent& A = s->red[r];
    const int n = A.h.nsess;
    for (int i0 = 0; i0 < n; i0 += 8) {   // 8 session records per round (see rs_load8)
      const S8 q = rs_load8(s, A, i0);
      CC4_UNROLL for (int k = 0; k < 8; ++k) {
        if (i0 + k >= n) continue;
        const int host = rsw_host(q.v[k]);
        const int sn = h_subnet(host);
        if ((red_allowed_mask(r) >> sn) & 1u) continue;
        const int to = red_of_subnet(sn);
        if (to < 0) { set_err(x, E_UNREACHABLE); continue; }
        if (nm < cap) mv[nm++] = (uint32_t)r | ((uint32_t)to << 3) | ((uint32_t)(i0 + k) << 6) | ((uint32_t)s8_slot(q, k) << 12);
        else set_err(x, E_RSESS_OVERFLOW);
      }
    }
  }
  // A move takes the entry out of the source agent's list and appends the same pool record to the target's (new ident, a fresh
  // RedAbstractSession: no port knowledge).  A source agent's moves come in list order and sessions gained on the way are
  // appended behind, so the entry sits at its collection index minus the number of this agent's earlier removals (checked
  // against the slot; the scan is the fallback).
  int gone_from[NRED];
  CC4_UNROLL for (int r = 0; r < NRED; ++r) gone_from[r] = 0;
  for (int m = 0; m < nm; ++m) {
    const uint32_t w = mv[m];
    const int from = (int)(w & 7u), to = (int)((w >> 3) & 7u), slot = (int)((w >> 12) & 0xFFu);
    RedAgent& F = s->red[from];
    int shift = 0;
    CC4_UNROLL for (int r = 0; r < NRED; ++r) if (r == from) shift = gone_from[r];
    int i = (int)((w >> 6) & 0x3Fu) - shift;
    if (i < 0 || i >= F.h.nsess || F.sord[i] != slot) {
      i = -1;
      for (int k = 0; k < F.h.nsess; ++k) if (F.sord[k] == slot) { i = k; break; }
      if (i < 0) continue;
    }
    CC4_UNROLL for (int r = 0; r < NRED; ++r) if (r == from) gone_from[r]++;
    const uint64_t rec = rs_word(s, slot);
    const int old_host = rsw_host(rec), old_pid = rsw_pid(rec), old_flags = rsw_flags(rec), old_id = rsw_id(rec);
    rs_remove_at(x, from, i, true);
    int ni = rs_add(x, to, old_host, old_pid, RS_ABSTRACT | (old_flags & RS_ROOT), slot);
    if (ni < 0) { bit_clr_shared(s->spool_used, slot); continue; }
    // observation hand-over: only if the creating action's observation carries the host ip key with this session
    if (F.h.new_sess_host == old_host && F.h.new_sess_id == old_id) {
      obs_first(x, to, T_UNKNOWN, RA_NONE, 0, 0);
      obs_put(x, to, true, old_host, OE_SESS | OE_IFACE | OE_SYSHN, false);
      as_know_sid(x, to, rsw_id(rs_at(s, s->red[to], ni)));
    }
  }
  int ns[NRED];
  CC4_UNROLL for (int r = 0; r < NRED; ++r) ns[r] = s->red[r].h.nsess;
  CC4_UNROLL for (int r = 0; r < NRED; ++r) s->red[r].h.active = (uint8_t)(ns[r] > 0);
}

// ------------------------------------------------------------------ the step (SimulationController.step, SC:211-315)
// The step is cut into phases that exchange data only through EnvState, so that the serial walk (env_step: the CPU
// oracle and the PCG64 device mode) and the lane-parallel Philox kernel run the very same phase bodies.
//   P0 step_phase + submit   lane 0      phase check, blue decode (or built-in policy) + queue  (SC:224-248)
//   P1 step_green_policy(g)  per green   EnterpriseGreenAgent.get_action             (EnterpriseGreenAgent.py:60)
//   P2 step_red_policy_tick(r) per red   FSM get_action + validity + queue, then the agent's duration-queue tick (SC:236-265)
//   P3 step_tick_blue(b), step_blue_exec   duration queue of the blue agents, (pcg: shuffle), blue execution
//   P4 step_green_exec(g)    per green   green actions except PhishingEmail          (GreenLocalWork/GreenAccessService)
//   P5 step_phishing         lane 0      deferred PhishingEmail in agent order
//   P6 step_red_exec         lane 0      red actions in agent order, reassignment    (SC:271-278)
//   P7 step_monitor_host(h)  per host    end-turn Monitor event roll-over             (Monitor.py:35-74)
//      step_monitor_pend     lane 0      sus pid hand-over
//   P8 step_rsc(r)           per red     end-turn RedSessionCheck
//   P9 step_end              lane 0      counters, done, reward, messages            (SC:297-311)
// Equivalence of the per-green split with the serial order: a green action reads only its own host's services, the
// blocks and the server counts, and writes event bits (OR) and the reward (sum); the one order-dependent effect,
// PhishingEmail (new red session), is deferred and replayed in agent order (P5).

// returns false when the episode is stepped past its end (State.py:539-540 raises ValueError)
// State.check_next_phase_on_update_step (State.py:514-544): the mission phase of step `st`, or -1 past the last phase
CC4_HD int step_phase_of(int st, int len0, int len1, int len2) {
  if (st < len0) return 0;
  if (st < len0 + len1) return 1;
  if (st < len0 + len1 + len2) return 2;
  return -1;
}
// The accumulators of a step (brm, n_restore, n_actions) are also left initialised by step_end / the reset, so the
// lane-parallel kernel may add to them before its thread 0 has finished this function's writes.
CC4_HD bool step_phase(Ctx x, bool init_accumulators = true) {
  EnvState* s = x.s;
  {
    const int st = s->step_count;
    const int ph = step_phase_of(st, s->phase_len[0], s->phase_len[1], s->phase_len[2]);
    if (ph < 0) { set_err(x, E_STEP_PAST_END); return false; }
    s->obs_dirty = (uint8_t)(ph > s->phase ? OD_PHASE : 0);   // a new mission phase changes the phase words and the comms policy of the observation
    if (ph > s->phase) s->phase = ph;
  }
  rng_begin_step(&s->rng, (uint32_t)s->step_count);
  if (x.lg) { x.lg->n = 0; x.lg->step = (uint32_t)s->step_count; }
  if (x.ext) { x.c->gfail[0] = 0; x.c->gfail[1] = 0; x.c->gfail[2] = 0; }
  s->action_cost = 0.f;
  if (init_accumulators) {
    s->brm = 0; s->n_restore = 0;
    s->n_actions = NBLUE + s->n_green + NRED;   // minus the actions filter_actions drops (step_tick_agent)
  }
  return true;
}
// one blue agent's submitted action: decode, cost, queue (SC:236-248); independent across agents
// cc4BlueRandomAgent.get_action (Agents/SimpleAgents/RandomAgent.py:28-52,66-69) for a blue agent nobody submitted an action
// for (SimulationController.py:236-248 asks the scenario's agent object): epsilon = 1 still costs the random() of the test;
// a uniform action class out of the agent's actions minus Block/AllowTrafficZone, in blue_actions order
// (EnterpriseScenarioGenerator.py:642: Monitor, Analyse, Restore, Remove, DeployDecoy, Sleep); then one uniform value per
// constructor parameter with more than one option: `session` and `agent` have one, `hostname` ranges over the agent's hosts
// in ActionSpace.hostname's insertion order = the initial observation's: per allowed subnet the router, the user hosts, the
// server hosts (routers are valid targets here, unlike in the wrapper's action list).
CC4_HD Act blue_random_policy(Ctx x, int b) {
  EnvState* s = x.s;
  Act a; a.type = BA_SLEEP; a.host = 0; a.arg = 0; a.ticks = 1; a.sid = 0; a.busy = 0;
  rng_set_stream(x.r, ST_BLUE_POL + (uint32_t)b);
  (void)rng_random(x.r);
  const int c = (int)rng_below(x.r, 6);
  const int t = c == 0 ? BA_MONITOR : (c == 1 ? BA_ANALYSE : (c == 2 ? BA_RESTORE : (c == 3 ? BA_REMOVE : (c == 4 ? BA_DECOY : BA_SLEEP))));
  a.type = (uint8_t)t;
  if (t == BA_MONITOR || t == BA_SLEEP) return a;
  int nh = 0;
  for (int i = 0; i < blue_nsub(b); ++i) { const int sn = blue_subnet_alloc(b, i); nh += 1 + s->n_users[sn] + s->n_servers[sn]; }
  int j = (int)rng_below(x.r, (uint32_t)nh);
  for (int i = 0; i < blue_nsub(b); ++i) {
    const int sn = blue_subnet_alloc(b, i), nu = s->n_users[sn], cnt = 1 + nu + s->n_servers[sn];
    if (j >= cnt) { j -= cnt; continue; }
    a.host = (uint8_t)(j == 0 ? h_make(sn, 0) : (j - 1 < nu ? h_make(sn, 1 + (j - 1)) : h_make(sn, 11 + (j - 1 - nu))));
    break;
  }
  return a;
}
CC4_HD void step_blue_submit(Ctx x, int b, int action_index) {
  EnvState* s = x.s;
  const bool builtin = action_index < 0 && (s->policy & BP_RANDOM_BIT);
  // `action.duration` set by the caller (the reference's scripted tests shorten Analyse / Restore / DeployDecoy to one tick) rides
  // in bits 20.. of a non-negative index
  const int dur = action_index >= 0 ? (action_index >> BLUE_DUR_SHIFT) & 0xFF : 0;
  if (action_index >= 0) action_index &= BLUE_IDX_MASK;
  Act a = builtin ? blue_random_policy(x, b) : blue_decode(s, b, action_index);
  // Restore.cost = -1, charged on submission even while busy -- for SUBMITTED actions only: the sum runs over the step's
  // `actions` argument, which a default agent's choice never enters (SC:236-240,310)
  if (a.type == BA_RESTORE && !builtin) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(&s->n_restore, 1);
#else
    s->n_restore++;
#endif
  }
  a.ticks = (uint8_t)(dur ? dur : blue_duration(a.type));
  if (!s->blue[b].queue.busy) { s->blue[b].queue = a; s->blue[b].queue.busy = 1; }
}
// The policies' generator while a set_seed split is in force (EnvState.rng2; numpy-stream mode only -- in the counter mode
// cc4_set_seed re-keys every stream): the walking generator and rng2 trade places around the policy loops.
CC4_HD void rng_policy_swap(Ctx x, bool back) {
  EnvState* s = x.s;
  (void)back;
  if (!s->rng_split) return;
  // field by field: a whole-struct copy through x.r pins the caller's register copy of the generator to memory (measured on
  // the numpy-stream kernel: +15 % per step)
  Rng* a = x.r; Rng* b = &x.c->rng2;
  { uint64_t t = a->s_hi; a->s_hi = b->s_hi; b->s_hi = t; }
  { uint64_t t = a->s_lo; a->s_lo = b->s_lo; b->s_lo = t; }
  { uint64_t t = a->inc_hi; a->inc_hi = b->inc_hi; b->inc_hi = t; }
  { uint64_t t = a->inc_lo; a->inc_lo = b->inc_lo; b->inc_lo = t; }
  { uint32_t t = a->has32; a->has32 = b->has32; b->has32 = t; }
  { uint32_t t = a->u32; a->u32 = b->u32; b->u32 = t; }
  { uint32_t t = a->ndraw; a->ndraw = b->ndraw; b->ndraw = t; }
}
// pre: block 0 of the agent's policy stream when the caller has it already (rng_preload), else null
CC4_HD void step_green_policy(Ctx x, int g, const uint32_t* pre = nullptr) {
  if (x.ext && x.ext[NRED + g].type != XA_NONE) {   // the step's `actions` dict holds this agent's action: its policy is not asked (SC:236-240)
    const ExtAct& ea = x.ext[NRED + g];
    int t = ea.type;
    if ((t == XG_ACCESS || t == XG_LOCAL) && !(ea.flags & XF_SKIP_VALID)) {
      // replace_action_if_invalid (SC:1068-1112) over the action's attributes that are ActionSpace keys: the class (a SleepAgent
      // green agent's action space holds Sleep only, EnterpriseScenarioGenerator.py:742-743), ip_address (the agent's own host is
      // what it may act from), and -- GreenAccessService -- every entry of allowed_subnets must be in the agent's CURRENT list
      // (ActionSpace.get_action_space carries 'allowed_subnets', ActionSpace.py:115): a list from before a mission-phase change
      // turns the action into an InvalidAction.  The phase of this step = max(stored, from the step count): the same value
      // whether or not step_phase has stored it yet.
      const EnvState* s = x.s;
      const int own = h_subnet(s->green_host[g]);
      int ph = step_phase_of(s->step_count, s->phase_len[0], s->phase_len[1], s->phase_len[2]);
      if (ph < s->phase) ph = s->phase;
      if (((s->policy & GP_SLEEP_BIT) && !(s->policy & GP_OPEN_BIT)) || ea.host != s->green_host[g] || (t == XG_ACCESS && (ea.sid & ~green_allowed_mask(ph, own)))) t = XG_INVALID;
    }
    // InvalidAction: Observation(False), nothing the reward reads -- resolved like Sleep (3: step_end reports its success as False)
    x.w->green_act[g] = (uint8_t)(t == XG_ACCESS ? 0 : (t == XG_LOCAL ? 1 : (t == XG_INVALID ? 3 : 2)));
    return;
  }
  if (x.s->policy & GP_SLEEP_BIT) { x.w->green_act[g] = 2; return; }   // green_agent_class=SleepAgent: no draw, Sleep
  rng_set_stream(x.r, ST_GREEN_POL + (uint32_t)g);
  if (pre) rng_preload(x.r, pre);
  x.w->green_act[g] = (uint8_t)rng_below(x.r, 3);  // choice([GreenAccessService, GreenLocalWork, Sleep])
}
// One red agent's policy (AgentInterface.get_action, SC:236-248) followed by its own duration-queue tick (SC:251-265; a tick
// touches only its agent: queue, observation reset, filter_actions against its own session table).  The agent's scalar fields
// (RedHdr) are read once into registers, worked on there and written back once.  Returns 1 if the agent's action was dropped
// by filter_actions (SC:466-485: it names a dead session).
// The observation half of an FSM agent's policy on its own (no draws, only the agent's own tables): the numpy-stream kernel
// runs it for the six agents side by side before its walking lane draws their choices in order.
CC4_HD void step_red_observe(Ctx x, int r) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  const int pol = s->policy & 3;
  if (!A.h.active || pol == RP_RANDOM || pol == RP_SLEEP) return;
  if (x.ext && x.ext[r].type != XA_NONE) return;   // an action was submitted for this agent: get_action is not called, nothing is observed
  RedHdr H = A.h;
  fsm_observe(x, r, H);
  A.h = H;
}
// True if agent r's end-of-turn RedSessionCheck will draw (it promotes a session to primary: no session holds ident 0)
CC4_HD bool rsc_draws(const EnvState* s, int r) {
  const RedAgent& A = s->red[r];
  if (!A.h.active || A.h.nsess == 0) return false;
  return rsw_id(rs_at(s, A, 0)) != 0 && rsw_id(rs_at(s, A, A.h.nsess - 1)) != 0 && rs_find_id(s, A, 0) < 0;
}
// pre: block 0 of the agent's policy stream when the caller has it already (rng_preload), else null
CC4_HD int step_red_policy_tick(Ctx x, int r, bool observed = false, const uint32_t* pre = nullptr) {
  EnvState* s = x.s;
  RedAgent& A = s->red[r];
  CC4_FT0(x, r);
  RedHdr H = A.h;
  Act a; a.type = RA_SLEEP; a.host = 0; a.arg = 0; a.ticks = 1; a.sid = 0; a.busy = 0;
  rng_set_stream(x.r, ST_RED_POL + (uint32_t)r);
  if (pre) rng_preload(x.r, pre);
  CC4_FT(x, r, 0);
  const ExtAct* ea = (x.ext && x.ext[r].type != XA_NONE) ? &x.ext[r] : nullptr;
  if (ea) {
    // actions.get(agent_name) (SC:236-240): the submitted action is taken whether or not the agent is active, the agent object is
    // not asked -- no draw, no FSM observation, no step count -- and it goes through the same validity check (SC:241-242)
    a.type = (uint8_t)ea->type; a.host = ea->host; a.arg = ea->arg; a.sid = ea->sid;
    a.ticks = (uint8_t)(ea->ticks ? ea->ticks : red_duration(a.type));
    if (!(ea->flags & XF_SKIP_VALID)) red_validate(x, r, H, a);
    if (a.type < RA_SLEEP && (ea->flags & (XF_RATE0 | XF_RATE1))) a.busy = (uint16_t)((ea->flags & XF_RATE0 ? AQ_RATE0 : 0) | (ea->flags & XF_RATE1 ? AQ_RATE1 : 0));
  }
  else if (H.active && (s->policy & 3) == RP_RANDOM) { a = random_red_get_action(x, r, H); red_validate(x, r, H, a); }
  else if (H.active && (s->policy & 3) != RP_SLEEP) { a = fsm_get_action(x, r, H, observed); CC4_AT0(x); red_validate(x, r, H, a); CC4_AT(x, 7); }   // AgentInterface.get_action (:120-143); SleepAgent -> Sleep
  CC4_FT(x, r, 1);
  if (!H.queue.busy) {
    H.queue = a; H.queue.busy = (uint16_t)(AQ_BUSY | a.busy);
    if (x.ext && (a.busy & (AQ_RATE0 | AQ_RATE1))) { x.c->xrate[r][0] = ea->rate0; x.c->xrate[r][1] = ea->rate1; }
  }
  // ---- tick: a new step's observation starts empty
  H.nobs = 0; H.obs_success = 0; H.obs_act_type = RA_NONE; H.new_sess_host = 0xFF; H.rsc_listed = 0;
  for (int w = 0; w < 5; ++w) { A.obs_has[0][w] = 0; A.obs_has[1][w] = 0; }
  Act q = H.queue, ex;
  q.ticks--;
  if (q.ticks < 1) { ex = q; q.busy = 0; }
  else {
    ex.type = RA_SLEEP; ex.host = 0; ex.arg = 0; ex.ticks = 0; ex.sid = 0; ex.busy = 0;
    H.obs_success = T_IN_PROGRESS; H.obs_act_type = RA_NONE; H.obs_act_host = 0; H.obs_act_arg = 0;   // obs_first(IN_PROGRESS)
  }
  H.queue = q;
  H.exec_type = ex.type; H.exec_host = ex.host;
  int dropped = 0;
  CC4_FT(x, r, 2);
  if (ex.type <= RA_WITHDRAW && rs_find_id(s, A, ex.sid, H.nsess) < 0) { ex.type = RA_NONE; dropped = 1; }
  CC4_FT(x, r, 3);
  s->rexec[r] = ex;
  A.h = H;
  CC4_FT(x, r, 4);
  return dropped;
}
// shuffle_done: the caller has already consumed the shuffle's draws (the numpy-stream kernel does that across the wave)
CC4_HD void step_blue_exec(Ctx x, bool shuffle_done = false) {
  EnvState* s = x.s;
  if (!shuffle_done) {
    CC4_TICK(x, 3);
    // ---- sort_action_order (SC:398-464): the shuffle only consumes the shared numpy stream; the Philox streams are
    // per agent, so there is nothing to consume there
    if (x.r->mode == 0) rng_shuffle_consume(x.r, s->n_actions);
    CC4_TICK(x, 4);
  }
  // ---- execute: priority 1 (ControlTraffic) first, then agent order
  for (int b = 0; b < NBLUE; ++b) if (s->bexec[b].type == BA_BLOCK || s->bexec[b].type == BA_ALLOW) blue_execute(x, b, s->bexec[b]);
  for (int b = 0; b < NBLUE; ++b)
    if (!(s->bexec[b].type == BA_BLOCK || s->bexec[b].type == BA_ALLOW)) {
      rng_set_stream(x.r, ST_BLUE_EXE + (uint32_t)b);
      blue_execute(x, b, s->bexec[b]);
    }
  CC4_TICK(x, 5);
}
// One blue agent's action on its own generator stream.  Blue agents own disjoint zones: their actions touch only hosts of the
// zone, the session table of the zone's red agent and commutative shared bits, so in the counter-based RNG mode they can be
// resolved concurrently -- except Monitor, which edits the shared pending-event list (step_blue_exec keeps the serial order).
CC4_HD void step_blue_exec_agent(Ctx x, int b, const uint32_t* pre = nullptr) {
  rng_set_stream(x.r, ST_BLUE_EXE + (uint32_t)b);
  if (pre) rng_preload(x.r, pre);
  blue_execute(x, b, x.s->bexec[b]);
}
CC4_HD bool blue_exec_independent(const EnvState* s) {
  if (s->npend) return false;
  for (int b = 0; b < NBLUE; ++b) if (s->bexec[b].type == BA_MONITOR) return false;
  return true;
}
// duration queue of one blue agent (SC:251-265)
CC4_HD void step_tick_blue(Ctx x, int b) {
  EnvState* s = x.s;
  Act& q = s->blue[b].queue;
  q.ticks--;
  if (q.ticks < 1) { s->bexec[b] = q; q.busy = 0; }
  else { Act z; z.type = BA_SLEEP; z.host = 0; z.arg = 0; z.ticks = 0; z.sid = 0; z.busy = 0; s->bexec[b] = z; }
}
// returns the BlueRewardMachine penalty of this green agent's action (<= 0)
// pre: block 0 of the agent's stream when the caller has it already (rng_preload), else null
CC4_HD int step_green_exec(Ctx x, int g, const uint32_t* pre = nullptr) {
  EnvState* s = x.s;
  int gh = s->green_host[g];
  int own = h_subnet(gh);
  rng_set_stream(x.r, ST_GREEN_EXE + (uint32_t)g);
  if (pre) rng_preload(x.r, pre);
  const int act = x.w->green_act[g];
  if (act >= 2) return 0;
  const uint64_t gp = x.gpre ? x.gpre[g] : green_prepare(x, g, act);
  const ExtAct* ea = (x.ext && x.ext[NRED + g].type != XA_NONE) ? &x.ext[NRED + g] : nullptr;   // its own rates, if it came with any
  if (act == 0) {
    const bool ok = green_access_service(x, gh, gp, (ea && (ea->flags & XF_RATE0)) ? ea->rate0 : 0.01);
    if (x.ext && !ok) bit_set_shared(x.c->gfail, g);
    return ok ? 0 : reward_table(s->phase, own, RW_ASF);
  }
  if (act == 1) {
    bool want_phish = false;
    bool ok = green_local_work(x, gh, gp, &want_phish, (ea && (ea->flags & XF_RATE0)) ? ea->rate0 : 0.01, (ea && (ea->flags & XF_RATE1)) ? ea->rate1 : 0.01);
    if (want_phish) bit_set_shared(x.w->phish_mask, g);   // green agents may be resolved on different lanes
    if (x.ext && !ok) bit_set_shared(x.c->gfail, g);
    return ok ? 0 : reward_table(s->phase, own, RW_LWF);
  }
  return 0;
}
CC4_HD void step_phishing(Ctx x) {
  EnvState* s = x.s;
  for (int w = 0; w < 3; ++w) {   // green agent order
    uint32_t m = x.w->phish_mask[w];
    if (!m) continue;
    x.w->phish_mask[w] = 0;
    while (m) {
      int g = w * 32 + ctz32(m); m &= m - 1;
      rng_set_stream(x.r, ST_GREEN_PHISH + (uint32_t)g);
      phishing(x, s->green_host[g]);
    }
  }
}
// true if some red agent holds a session outside its allowed subnets (work for different_subnet_agent_reassignment): live_hosts
// against the host-id ranges of each agent's subnets (subnet sn = ids sn*17 .. sn*17+16); all six agents at once:
// 30 independent loads, masks folded to constants by the unrolling
CC4_HD uint32_t red_foreign_agents(const EnvState* s) {   // bit r: agent r holds a session outside its zone
  uint32_t foreign = 0;
  CC4_UNROLL for (int r = 0; r < NRED; ++r) {
    uint32_t acc = 0;
    CC4_UNROLL for (int w = 0; w < 5; ++w) acc |= s->red[r].live_hosts[w] & ~red_zone_hosts(r, w);
    if (acc) foreign |= 1u << r;
  }
  return foreign;
}
CC4_HD bool red_any_foreign_session(const EnvState* s) { return red_foreign_agents(s) != 0; }
// pre: block 0 of the agent's action stream when the caller has it already (rng_preload), else null
CC4_HD void step_red_exec_agent(Ctx x, int r, const uint32_t* pre = nullptr) {
  EnvState* s = x.s;
  if (s->rexec[r].type == RA_NONE) return;
  rng_set_stream(x.r, ST_RED_EXE + (uint32_t)r);
  if (pre) rng_preload(x.r, pre);
  red_execute(x, r, s->rexec[r]);
}
// Red actions of different agents commute when they name different hosts (each action reads/writes its own agent's
// tables, its target host and commutative event bits); DiscoverRemoteSystems only reads the topology.  Returns the set of
// agents whose actions do NOT commute with some other agent's and therefore keep the serial agent order among themselves.
CC4_HD uint32_t red_conflict_mask(const EnvState* s) {
  int ty[NRED], ho[NRED];   // the six actions in one batch of loads (Act = type | host << 8 | ... as one 8-byte record)
  CC4_UNROLL for (int a = 0; a < NRED; ++a) { uint64_t v; __builtin_memcpy(&v, &s->rexec[a], 8); ty[a] = (int)(v & 0xFF); ho[a] = (int)((v >> 8) & 0xFF); }
  uint32_t m = 0;
  bool withdraw = false;
  CC4_UNROLL for (int a = 0; a < NRED; ++a) {
    if (ty[a] == RA_WITHDRAW) withdraw = true;   // kills sessions: everything serial
    CC4_UNROLL for (int b = a + 1; b < NRED; ++b)
      if (ty[a] >= RA_AGGR && ty[a] <= RA_DEGRADE && ty[b] >= RA_AGGR && ty[b] <= RA_DEGRADE && ho[a] == ho[b]) m |= (1u << a) | (1u << b);
  }
  return withdraw ? (1u << NRED) - 1u : m;
}
CC4_HD void step_red_exec(Ctx x) {
  rs_reserve(x);
  for (int r = 0; r < NRED; ++r) step_red_exec_agent(x, r);
  step_red_merge(x);
  CC4_TICK(x, 7);
}
// different_subnet_agent_reassignment (SC:820-903): `any_foreign` = some red agent holds a session outside its zone
CC4_HD void step_reassign(Ctx x, uint32_t foreign) {   // foreign = red_foreign_agents(s)
  EnvState* s = x.s;
  if (foreign) red_reassign(x, foreign);
  else {
    int ns[NRED];
    CC4_UNROLL for (int r = 0; r < NRED; ++r) ns[r] = s->red[r].h.nsess;
    CC4_UNROLL for (int r = 0; r < NRED; ++r) s->red[r].h.active = (uint8_t)(ns[r] > 0);
  }
  CC4_TICK(x, 8);
}
// the per-host part of Monitor.execute: this step's event bits become last step's on the hosts a blue agent watches
// ... and four hosts' bytes at a time (word w of EnvState.hev = hosts 4w .. 4w+3): the same function on each byte, the watched hosts as a byte mask
CC4_HD constexpr uint32_t monitor_watch_mask(int w) {
  uint32_t m = 0;
  for (int i = 0; i < 4; ++i) { const int h = 4 * w + i; if (h < MAXH && ((0xf444f3210ull >> (4 * (h / SLOTS))) & 0xF) != 0xF) m |= 0xFFu << (8 * i); }   // blue_of_subnet(h_subnet(h)) >= 0
  return m;
}
CC4_HD uint32_t monitor_roll4(uint32_t ev4, uint32_t watch) {
  static_assert(EV_OLD_CONN == EV_CUR_CONN << 2 && EV_OLD_PROC == EV_CUR_PROC << 2 && (EV_CUR_CONN | EV_CUR_PROC) == 3, "current bits move up two places");
  return (ev4 & ~watch) | (((ev4 & 0x03030303u) << 2) & watch);
}
CC4_HD uint8_t monitor_roll(int h, uint8_t ev) {
  if (blue_of_subnet(h_subnet(h)) < 0) return ev;
  uint8_t nev = 0;
  if (ev & EV_CUR_CONN) nev |= EV_OLD_CONN;
  if (ev & EV_CUR_PROC) nev |= EV_OLD_PROC;
  return nev;
}
CC4_HD int step_monitor_host(Ctx x, int h) {  // returns the host's event bits afterwards
  EnvState* s = x.s;
  if (!bit_get(s->exists, h)) return 0;                     // rows of hosts that do not exist stay zero
  const uint8_t ev = s->hev[h];
  const uint8_t nev = monitor_roll(h, ev);
  if (nev != ev) s->hev[h] = nev;
  return nev;
}
CC4_HD void step_monitor_pend(Ctx x) {  // session.add_sus_pids for the pid-carrying process_creation events
  EnvState* s = x.s;
  for (int i = 0; i < s->npend; ++i) {
    int b = blue_of_subnet(h_subnet((int)(s->pend[i] >> 16)));
    if (b < 0) continue;
    BlueAgent& A = s->blue[b];
    if (A.nsus >= cold_sus_cap(s->steps)) set_err(x, E_SUS_OVERFLOW); else { cold_sus(x.c, s->steps, b)[A.nsus++] = s->pend[i]; bit_set(A.sus_hosts, (int)(s->pend[i] >> 16)); }
  }
  s->npend = 0;
}
CC4_HD void step_rsc(Ctx x, int r) {
  if (!x.s->red[r].h.active) return;
  rng_set_stream(x.r, ST_RED_RSC + (uint32_t)r);
  red_session_check(x, r);
}
// the messages submitted with this step, agent b's row (read back by the observation encode of the same step only, so the
// lane-parallel kernel stores them when the actions are submitted and step_end skips the copy)
CC4_HD void step_messages(EnvState* s, const uint8_t* messages, int b) {
  static_assert(MSG_LEN == 8, "a message is one 8-byte word");
  uint64_t w = 0;
  if (messages) for (int i = 0; i < MSG_LEN; ++i) w |= (uint64_t)(messages[b * MSG_LEN + i] ? 1 : 0) << (8 * i);
  __builtin_memcpy(&s->msg[b][0], &w, 8);       // (no messages, the usual case of a vectorised batch: one store instead of eight byte stores)
}
CC4_HD void step_end(Ctx x, const uint8_t* messages, bool copy_msgs = true) {
  EnvState* s = x.s;
  s->step_count++;
  s->done = (uint8_t)(s->step_count >= s->steps - 1);
  int brm = s->brm;
  int et[NRED], ns[NRED];
  CC4_UNROLL for (int r = 0; r < NRED; ++r) { et[r] = s->red[r].h.exec_type; ns[r] = s->red[r].h.nsess; }
  CC4_UNROLL for (int r = 0; r < NRED; ++r)
    if (et[r] == RA_IMPACT && ns[r] > 0)
      brm += reward_table(s->phase, h_subnet(s->red[r].h.exec_host), RW_RIA);  // charged for any executed Impact (App. B.2)
  s->action_cost = -(float)s->n_restore;
  s->reward = (float)brm + s->action_cost;
  if (x.ext) for (int g = 0; g < s->n_green; ++g) if (x.w->green_act[g] == 3) bit_set_shared(x.c->gfail, g);   // submitted green actions that were invalid
  if (copy_msgs) for (int b = 0; b < NBLUE; ++b) step_messages(s, messages, b);
  s->brm = 0; s->n_restore = 0; s->n_actions = NBLUE + s->n_green + NRED;   // the next step's accumulators (see step_phase)
  rng_park(&s->rng);
}

// the serial walk: actions[5] = wrapper action index per blue agent (negative = no action submitted -> SleepAgent)
CC4_HD void env_step(Ctx x, const int32_t* actions, const uint8_t* messages /* [5][8] or null */) {
  EnvState* s = x.s;
  CC4_TICK0(x);
  if (!step_phase(x)) return;
  rng_policy_swap(x, false);     // every agent's policy -- a built-in blue one included -- draws from the generator it was created with
  for (int b = 0; b < NBLUE; ++b) step_blue_submit(x, b, actions ? actions[b] : -1);
  CC4_TICK(x, 0);
  for (int g = 0; g < s->n_green; ++g) step_green_policy(x, g);
  CC4_TICK(x, 1);
  for (int r = 0; r < NRED; ++r) s->n_actions -= step_red_policy_tick(x, r);
  rng_policy_swap(x, true);
  CC4_TICK(x, 2);
  for (int b = 0; b < NBLUE; ++b) step_tick_blue(x, b);
  step_blue_exec(x);
  for (int g = 0; g < s->n_green; ++g) {
    s->brm += step_green_exec(x, g);
    if (bit_get(x.w->phish_mask, g)) { bit_clr(x.w->phish_mask, g); rng_set_stream(x.r, ST_GREEN_PHISH + (uint32_t)g); phishing(x, s->green_host[g]); }
  }
  CC4_TICK(x, 6);
  step_red_exec(x);
  {
    step_reassign(x, red_foreign_agents(s));
  }
  for (int h = 0; h < MAXH; ++h) step_monitor_host(x, h);
  step_monitor_pend(x);
  CC4_TICK(x, 9);
  for (int r = 0; r < NRED; ++r) step_rsc(x, r);
  CC4_TICK(x, 10);
  step_end(x, messages);
}

// ------------------------------------------------------------------ direct edits of an episode between steps
// What the reference's own scripted tests do by hand to env.environment_controller.state before they step (no simulator API):
//   SE_SET_PHASE        state.mission_phase = a0                          (Tests/test_cc4/test_BlueRewardMachine.py:48,70,113,142)
//   SE_ADD_SERVICE      pid = host.create_pid(); host.processes.append(Process(pid, ...)); host.add_service(name, Service(pid))
//                       on host a0, service kind a1 (K_*), process owned by root if a2   (test_Red/test_Impact.py:89-101, test_BlueRewardMachine.py:38-42)
//   SE_SET_RELIABILITY  every service of host a0: _percent_reliable = a1 (a multiple of 20)   (test_BlueRewardMachine.py:75-77)
//   SE_CLEAR_HOST       host a0: processes = [], services = {}            (test_blue_actions.py:223-224)
//   SE_DEPLOY_DECOY     Decoy<kind a1>(hostname a0).execute(state): one specific decoy factory instead of DeployDecoy's random
//                       choice (test_Red/test_DiscoverDeception.py:43,94,142,190); returns 1 on success, 0 when the factory's port is taken
// Returns >= 0, or -1 for an unknown op / bad argument.  (Host code: libcc4 applies them to a row fetched with cc4_get_state.)
//   SE_ADD_RED_SESSION  state.add_session(Session(hostname a1, username, agent red_agent_<a0>, parent, session_type='shell', pid=None)):
//                       a plain shell session -- a2 bit 0: username root / SYSTEM, bit 1: parent is not None -- with the ident
//                       State.add_session would pick (max + 1; the tests pass exactly those) and a fresh 'shell' process
//                       (test_Red/test_Withdraw.py:10-17, test_RedSessionCheck.py:10-23); returns the ident
//                       a2 bit 2: a RedAbstractSession (session_type='RedAbstractSession': Tests/test_cc4/test_Acceptance/test_priority.py:172-176)
//   SE_BLOCK            a2 ? state.blocks.setdefault(to a0, []).append(from a1) : the pair removed   (test_issue22_blocks.py:73-88)
//   SE_SET_STEP         environment_controller.step_count = a0: the next step's mission phase follows it, never backwards
//                       (State.check_next_phase_on_update_step; test_issue22_blocks.py:91-97)
//   SE_ADD_EVENT        host a0: events.network_connections (a1 = 0) / events.process_creation (a1 = 1) gets an entry whose port is
//                       host.get_ephemeral_port() -- one draw, a second on a collision (test_issue26_monitor.py:134-142,178-185)
//   SE_SET_RED_ACTIVE   agent_interfaces[red_agent_<a0>].active = bool(a1)   (test_Acceptance/test_challenge_details.py:253-256, test_priority.py:177)
enum : int { SE_SET_PHASE = 0, SE_ADD_SERVICE = 1, SE_SET_RELIABILITY = 2, SE_CLEAR_HOST = 3, SE_DEPLOY_DECOY = 4, SE_ADD_RED_SESSION = 5,
             SE_BLOCK = 6, SE_SET_STEP = 7, SE_ADD_EVENT = 8, SE_SET_RED_ACTIVE = 9 };
inline int state_edit(Ctx x, int op, int a0, int a1, int a2) {
  EnvState* s = x.s;
  auto host_ok = [&](int h) { return h >= 0 && h < MAXH && bit_get(s->exists, h); };
  switch (op) {
    case SE_SET_PHASE:
      if (a0 < 0 || a0 > 2) return -1;
      s->phase = a0; s->obs_dirty = OD_PHASE;
      return 0;
    case SE_ADD_SERVICE: {
      if (!host_ok(a0) || a1 < K_SSHD || a1 > K_DEC_VSFTPD) return -1;
      HostDyn& d = x.hd[a0];
      const int nsv = hd_nsvc(d);
      int si = -1;
      for (int i = nsv - 1; i >= 0; --i) if (d.svcs[i].kind == a1) si = i;   // add_service on an existing name replaces the entry
      if (si < 0 && nsv >= MAXSV) return -1;
      const int pid = create_pid(x, a0);
      if (!add_proc(x, a0, pid, K_PLAIN, a2 ? PF_ROOT : 0)) return -1;   // Process(pid=pid, process_name=...): no open_ports
      if (si < 0) { si = nsv; hd_set_nsvc(d, nsv + 1); }
      d.svcs[si].kind = (uint8_t)a1; d.svcs[si].pid = (uint16_t)pid; d.svcs[si].st = (uint8_t)(SV_ACTIVE | 5);
      return pid;
    }
    case SE_SET_RELIABILITY: {
      if (!host_ok(a0) || a1 < 0 || a1 > 100 || a1 % 20) return -1;
      HostDyn& d = x.hd[a0];
      for (int i = 0; i < hd_nsvc(d); ++i) d.svcs[i].st = (uint8_t)((d.svcs[i].st & SV_ACTIVE) | (a1 / 20));
      return 0;
    }
    case SE_CLEAR_HOST: {
      if (!host_ok(a0)) return -1;
      HostDyn& d = x.hd[a0];
      d.nproc = 0; hd_set_nsvc(d, 0);
      for (int i = 0; i < PIN; ++i) { d.procs[i].pid = 0; d.procs[i].kind = 0; d.procs[i].flags = 0; }
      for (int i = 0; i < MAXSV; ++i) { d.svcs[i].pid = 0; d.svcs[i].kind = 0; d.svcs[i].st = 0; }
      return 0;
    }
    case SE_DEPLOY_DECOY: {
      if (!host_ok(a0) || a1 < K_DEC_APACHE || a1 > K_DEC_VSFTPD) return -1;
      HostDyn& d = x.hd[a0];
      const int used = proc_ports(x, a0);   // DecoyFactory.is_host_compatible: the factory's port is free (vsftpd checks 21: always)
      const int need = a1 == K_DEC_APACHE ? PB_80 : (a1 == K_DEC_TOMCAT ? PB_443 : (a1 == K_DEC_HARAKA ? PB_25 : 0));
      if (used & need) return 0;
      const int pid = create_pid(x, a0);
      if (!add_proc(x, a0, pid, a1, 0)) return -1;
      const int nsv = hd_nsvc(d);
      int si = -1;
      for (int i = nsv - 1; i >= 0; --i) if (d.svcs[i].kind == a1) si = i;
      if (si < 0) { if (nsv >= MAXSV) return -1; si = nsv; hd_set_nsvc(d, nsv + 1); }
      d.svcs[si].kind = (uint8_t)a1; d.svcs[si].pid = (uint16_t)pid; d.svcs[si].st = (uint8_t)(SV_ACTIVE | 5);
      return 1;
    }
    case SE_ADD_RED_SESSION: {
      if (a0 < 0 || a0 >= NRED || !host_ok(a1)) return -1;
      const int pid = create_pid(x, a1);                      // Host.add_session: pid = create_pid(), Process(name=session_type, username)
      if (!add_proc(x, a1, pid, (a2 & 4) ? K_SESS_RED : K_SHELL, (a2 & 1) ? PF_ROOT : 0)) return -1;
      const int idx = rs_add(x, a0, a1, pid, ((a2 & 1) ? RS_ROOT : 0) | ((a2 & 2) ? RS_CHILD : 0) | ((a2 & 4) ? RS_ABSTRACT : 0));
      if (idx < 0) return -1;
      return rsw_id(rs_at(s, s->red[a0], idx));
    }
    case SE_BLOCK:
      if (a0 < 0 || a0 >= NSUB - 1 || a1 < 0 || a1 >= NSUB - 1) return -1;
      if (a2) s->blocks[a0] |= (uint16_t)(1u << a1); else s->blocks[a0] &= (uint16_t)~(1u << a1);
      s->obs_dirty = OD_BLOCKS;
      return 0;
    case SE_SET_STEP:
      if (a0 < 0 || a0 >= s->steps) return -1;
      s->step_count = a0;
      return 0;
    case SE_ADD_EVENT: {
      if (!host_ok(a0) || a1 < 0 || a1 > 1) return -1;
      const int port = eph_port(x, a0);
      ev_or(x, a0, a1 ? EV_CUR_PROC : EV_CUR_CONN);
      return port;
    }
    case SE_SET_RED_ACTIVE:
      if (a0 < 0 || a0 >= NRED) return -1;
      s->red[a0].h.active = (uint8_t)(a1 != 0);
      return 0;
    default: return -1;
  }
}

// ------------------------------------------------------------------ BlueFlatWrapper.observation_change (BlueFlatWrapper.py:172-256)
// out: OBS_TOTAL values, agents 0..3 (92 each) then agent 4 (210).  The vector splits into 12 independent parts:
// parts 0..6 = the seven 59-value subnet blocks (agents 0..3 own one, agent 4 owns three), parts 7..11 = agent b's
// phase word + 32 message bits.  The device encodes the parts on separate lanes; the host loops over them.
enum : int { OBS_PARTS = 12 };
template <typename T>
CC4_HD void env_flat_obs_part(const EnvState* s, T* out, int part) {
  if (part < 7) {
    int b = part < 4 ? part : 4, i = part < 4 ? 0 : part - 4;
    int o = (b < 4 ? b * OBS_SHORT : 4 * OBS_SHORT) + 1 + 59 * i;
    int sn = blue_subnet_sorted(b, i);
    uint32_t blk = s->blocks[sn];
    uint32_t adj = comms_adjacent(s->phase, sn);
    for (int k = 0; k < NSUB; ++k) {
      int ss = sorted_subnet(k);
      out[o + k] = (T)(ss == sn);
      out[o + 9 + k] = (T)((blk >> ss) & 1u);
      out[o + 18 + k] = (T)(!((adj >> ss) & 1u));
    }
    for (int hs = 0; hs < ZONE_HOSTS; ++hs) {
      int h = hs < MAX_SERVERS ? h_make(sn, 11 + hs) : h_make(sn, 1 + (hs - MAX_SERVERS));
      int ev = bit_get(s->exists, h) ? s->hev[h] : 0;
      out[o + 27 + hs] = (T)((ev & (EV_CUR_PROC | EV_OLD_PROC)) != 0);
      out[o + 43 + hs] = (T)((ev & (EV_CUR_CONN | EV_OLD_CONN)) != 0);
    }
  } else {
    int b = part - 7;
    int base = b < 4 ? b * OBS_SHORT : 4 * OBS_SHORT;
    int len = b < 4 ? OBS_SHORT : OBS_LONG;
    out[base] = (T)s->phase;
    int o = base + len - 32;
    for (int j = 0; j < NBLUE; ++j) { if (j == b) continue; for (int i = 0; i < MSG_LEN; ++i) out[o++] = (T)s->msg[j][i]; }
  }
}
// the same vector, one value at a time (value `idx` of the 578): what the device encodes with one value per thread
CC4_HD int env_flat_obs_at(const EnvState* s, int idx) {
  const int b = idx < 4 * OBS_SHORT ? idx / OBS_SHORT : 4;
  const int j = idx - (b < 4 ? b * OBS_SHORT : 4 * OBS_SHORT);
  const int len = b < 4 ? OBS_SHORT : OBS_LONG;
  if (j == 0) return s->phase;
  if (j >= len - 32) {
    const int m = j - (len - 32), jj = m / MSG_LEN;
    return s->msg[jj < b ? jj : jj + 1][m % MSG_LEN];
  }
  const int q = j - 1, i = q / 59, k = q % 59;
  const int sn = blue_subnet_sorted(b, i);
  if (k < 9) return sorted_subnet(k) == sn;
  if (k < 18) return (s->blocks[sn] >> sorted_subnet(k - 9)) & 1u;
  if (k < 27) return !((comms_adjacent(s->phase, sn) >> sorted_subnet(k - 18)) & 1u);
  const int hs = k < 43 ? k - 27 : k - 43;
  const int h = hs < MAX_SERVERS ? h_make(sn, 11 + hs) : h_make(sn, 1 + (hs - MAX_SERVERS));
  const int ev = s->hev[h];   // bytes of hosts that do not exist stay zero (env_reset), so no existence test is needed
  return k < 43 ? ((ev & (EV_CUR_PROC | EV_OLD_PROC)) != 0) : ((ev & (EV_CUR_CONN | EV_OLD_CONN)) != 0);
}
// The same 578 values enumerated kind by kind (v = 0..577), so that the lanes of a wave take the same branch, the values that
// can change every step first:
//   [0,224) host events (7 subnet blocks x {16 process, 16 connection}), [224,384) message bits (5 agents x 32)  -- OBS_FAST --
//   [384,447) blocked bits, [447,510) comms policy, [510,573) subnet one-hot, [573,578) the 5 phase words (EnvState.obs_dirty).
// *idx = position in the vector.
enum : int { OBS_FAST = 384 };
CC4_HD int env_flat_obs_sorted(const EnvState* s, int v, int* idx) {
  if (v >= 224 && v < OBS_FAST) {
    const int w = v - 224, b = w >> 5, m = w & 31, jj = m / MSG_LEN;
    *idx = (b < 4 ? b * OBS_SHORT + OBS_SHORT : 4 * OBS_SHORT + OBS_LONG) - 32 + m;
    return s->msg[jj < b ? jj : jj + 1][m % MSG_LEN];
  }
  if (v < 573) {
    int sb, k;       // subnet block 0..6 (agents 0..3 own one, agent 4 owns three), offset inside the 59-value block
    if (v < 224) { sb = v >> 5; const int r = v & 31; k = 27 + (r & 15) + ((r >> 4) ? 16 : 0); }
    else { const int w = v - OBS_FAST, kind = w / 63, q = w % 63; sb = q / 9; k = 9 * (kind == 0 ? 1 : (kind == 1 ? 2 : 0)) + q % 9; }
    const int b = sb < 4 ? sb : 4, i = sb < 4 ? 0 : sb - 4;
    *idx = (b < 4 ? b * OBS_SHORT : 4 * OBS_SHORT) + 1 + 59 * i + k;
    const int sn = blue_subnet_sorted(b, i);
    if (k >= 27) {
      const int hs = k < 43 ? k - 27 : k - 43;
      const int h = hs < MAX_SERVERS ? h_make(sn, 11 + hs) : h_make(sn, 1 + (hs - MAX_SERVERS));
      const int ev = s->hev[h];   // bytes of hosts that do not exist stay zero (env_reset)
      return k < 43 ? ((ev & (EV_CUR_PROC | EV_OLD_PROC)) != 0) : ((ev & (EV_CUR_CONN | EV_OLD_CONN)) != 0);
    }
    if (k < 9) return sorted_subnet(k) == sn;
    if (k < 18) return (s->blocks[sn] >> sorted_subnet(k - 9)) & 1u;
    return !((comms_adjacent(s->phase, sn) >> sorted_subnet(k - 18)) & 1u);
  }
  const int b = v - 573;
  *idx = b < 4 ? b * OBS_SHORT : 4 * OBS_SHORT;
  return s->phase;
}
// The OBS_FAST values as a table (the device kernels read it instead of redoing the index arithmetic for every value of every
// step): entry v = position in the vector | source byte << 10 | bit mask << 18; the value is (byte & mask) != 0.  Source byte
// 0..136: the event bits of host h (EnvState.hev); 137 + 8 j + i: message bit i of blue agent j (EnvState.msg[j][i]).  Same enumeration as
// env_flat_obs_sorted (tests/test_host_logic.py checks the two against each other).
CC4_HD constexpr uint32_t obs_fast_entry(int v) {
  if (v >= 224) {
    const int w = v - 224, b = w >> 5, m = w & 31, jj = m / MSG_LEN;
    const int idx = (b < 4 ? b * OBS_SHORT + OBS_SHORT : 4 * OBS_SHORT + OBS_LONG) - 32 + m;
    const int src = MAXH + MSG_LEN * (jj < b ? jj : jj + 1) + m % MSG_LEN;
    return (uint32_t)idx | ((uint32_t)src << 10) | (1u << 18);
  }
  const int sb = v >> 5, r = v & 31, k = 27 + (r & 15) + ((r >> 4) ? 16 : 0);
  const int b = sb < 4 ? sb : 4, i = sb < 4 ? 0 : sb - 4;
  const int idx = (b < 4 ? b * OBS_SHORT : 4 * OBS_SHORT) + 1 + 59 * i + k;
  const int sn = blue_subnet_sorted(b, i);
  const int hs = k < 43 ? k - 27 : k - 43;
  const int h = hs < MAX_SERVERS ? h_make(sn, 11 + hs) : h_make(sn, 1 + (hs - MAX_SERVERS));
  const uint32_t mask = k < 43 ? (uint32_t)(EV_CUR_PROC | EV_OLD_PROC) : (uint32_t)(EV_CUR_CONN | EV_OLD_CONN);
  return (uint32_t)idx | ((uint32_t)h << 10) | (mask << 18);
}
CC4_HD int obs_fast_value(uint32_t entry, const EnvState* s) {
  const int src = (int)((entry >> 10) & 0xFF);
  const uint32_t byte = src < MAXH ? s->hev[src] : (&s->msg[0][0])[src - MAXH];
  return (byte & (entry >> 18)) != 0 ? 1 : 0;
}
template <typename T>
CC4_HD void env_flat_obs(const EnvState* s, T* out) {
  for (int p = 0; p < OBS_PARTS; ++p) env_flat_obs_part<T>(s, out, p);
}

}  // namespace cc4
