// cc4_oracle.cpp -- CPU oracle for the CC4 step engine.  TEST INFRASTRUCTURE, NOT PRODUCT.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from
// this file (oracle/liboracle.so).  The product path (cage_challenge_4_amd/csrc/*.hip -> libcc4.so)
// never links or calls it and fails loudly without a GPU.
//
// What it is: a host (g++) build of the transition restated in cage_challenge_4_amd/csrc/cc4_engine.h
// -- every function there cites the reference file:line it follows -- driven serially, one episode at a
// time.  How it is pinned: in PCG mode it must reproduce, bit for bit, the golden trajectories under
// tests/golden/ that oracle/refgen/make_golden.py recorded from the real reference (CybORG v4, imported
// from /root/reference in the build container): per-step flat observations, rewards, dones and the PCG64
// stream position.  tests/test_oracle_golden.py checks that on CPU; tests/test_hip_parity.py then checks
// the HIP path both against this oracle and directly against the same golden files.
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../cage_challenge_4_amd/csrc/cc4_engine.h"
#include "../cage_challenge_4_amd/csrc/cc4_export.h"

using namespace cc4;

struct Oracle {
  int n;
  uint32_t topo = 0;   // cc4_config.topology_seed
  int evlog = 0;       // cc4_enable_event_log
  int steps = 500;     // the episode length the cold rows were sized for (cc4o_create2)
  size_t cold_row = 0;
  std::vector<EnvState> st;
  std::vector<unsigned char> cold_mem;   // n rows of cold_row_bytes(steps): fixed part + steps-sized containers (csrc/cc4_state.h)
  EnvCold* cold(int i) { return cold_at(reinterpret_cast<EnvCold*>(cold_mem.data()), (size_t)i, cold_row); }
};

extern "C" {

// steps: EnterpriseScenarioGenerator(steps=...) of every episode this oracle will hold (sizes the cold containers, as cc4_create does)
void* cc4o_create2(int n, int steps) {
  Oracle* o = new Oracle();
  o->n = n;
  o->steps = steps;
  o->cold_row = cold_row_bytes(steps);
  o->st.resize(n);
  o->cold_mem.assign(o->cold_row * (size_t)n + 16, 0);
  memset(o->st.data(), 0, sizeof(EnvState) * n);
  return o;
}
void* cc4o_create(int n) { return cc4o_create2(n, 500); }
void cc4o_destroy(void* h) { delete (Oracle*)h; }
size_t cc4o_state_bytes() { return sizeof(EnvState); }
size_t cc4o_cold_bytes(void* h) { return ((Oracle*)h)->cold_row; }
void* cc4o_state_ptr(void* h, int i) { return &((Oracle*)h)->st[i]; }
void* cc4o_cold_ptr(void* h, int i) { return ((Oracle*)h)->cold(i); }

void cc4o_reset(void* h, int i, uint64_t seed, int rng_mode, int steps, int continue_stream, int policy) {
  Oracle* o = (Oracle*)h;
  if (cold_row_bytes(steps) != o->cold_row) { fprintf(stderr, "cc4o_reset: steps=%d needs other cold containers than this oracle was created for (steps=%d): use cc4o_create2\n", steps, o->steps); abort(); }
  StepWork w; memset(&w, 0, sizeof(w));
  Ctx x{&o->st[i], o->cold(i), &o->st[i].rng, o->st[i].hd, &w};
  uint32_t ws[RESET_WS_WORDS];   // work area of the counter-mode generation (the device kernels use LDS)
  env_reset(x, seed, rng_mode, steps, continue_stream != 0, policy, o->topo, rng_mode == 1 ? ws : nullptr);
}
void cc4o_set_seed(void* h, int i, uint64_t seed, int rng_mode) {   // CybORG.set_seed: the restatement of k_set_seed (csrc/cc4_k_misc.hip)
  EnvState& st = ((Oracle*)h)->st[i];
  if (rng_mode == 0) { if (!st.rng_split) ((Oracle*)h)->cold(i)->rng2 = st.rng; st.rng_split = 1; }
  Rng* r = &st.rng;
  rng_seed(r, seed, (uint32_t)rng_mode);
  if (rng_mode == 1) { rng_begin_episode(r); rng_park(r); }
}
void cc4o_set_rng_state(void* h, int i, const uint64_t* w) {   // restatement of k_set_rng_state (csrc/cc4_k_misc.hip)
  EnvState& st = ((Oracle*)h)->st[i];
  Rng r; rng_seed(&r, 0, 0);
  r.s_hi = w[0]; r.s_lo = w[1]; r.inc_hi = w[2]; r.inc_lo = w[3]; r.has32 = (uint32_t)w[4]; r.u32 = (uint32_t)w[5];
  st.rng = r; st.rng_split = 0;
}
void cc4o_set_topology_seed(void* h, uint32_t seed) { ((Oracle*)h)->topo = seed; }
void cc4o_enable_event_log(void* h, int on) { Oracle* o = (Oracle*)h; o->evlog = on ? 1 : 0; for (int i = 0; i < o->n; ++i) { o->cold(i)->evlog.enabled = on ? 1u : 0u; o->cold(i)->evlog.n = 0; } }
void cc4o_step(void* h, int i, const int32_t* actions, const uint8_t* msgs) {
  Oracle* o = (Oracle*)h;
  StepWork w; memset(&w, 0, sizeof(w));
  Ctx x{&o->st[i], o->cold(i), &o->st[i].rng, o->st[i].hd, &w};
  x.lg = o->evlog ? &o->cold(i)->evlog : nullptr;
  env_step(x, actions, msgs);
}
// cc4_step_ex (include/cc4.h): the step with the red / green entries of the `actions` dict; ext = EXT_PER_ENV records
// (NRED red, then MAXG green; type XA_NONE = nothing submitted for that agent), or null
void cc4o_step_ex(void* h, int i, const int32_t* actions, const uint8_t* msgs, const ExtAct* ext) {
  Oracle* o = (Oracle*)h;
  StepWork w; memset(&w, 0, sizeof(w));
  Ctx x{&o->st[i], o->cold(i), &o->st[i].rng, o->st[i].hd, &w};
  x.lg = o->evlog ? &o->cold(i)->evlog : nullptr;
  // once an episode has taken a submitted action its later steps keep serving the rates a queued action came with
  // (libcc4: the handle stays on the full builds of its step kernels)
  static const ExtAct none[EXT_PER_ENV] = {};
  static ExtAct all_none[EXT_PER_ENV];
  static bool init = false;
  if (!init) { memset(all_none, 0xFF, sizeof(all_none)); init = true; }
  (void)none;
  x.ext = ext ? ext : all_none;
  env_step(x, actions, msgs);
}
size_t cc4o_ext_bytes() { return sizeof(ExtAct) * EXT_PER_ENV; }
// cc4_edit_state (include/cc4.h)
int cc4o_edit_state(void* h, int i, int op, int a0, int a1, int a2) {
  Oracle* o = (Oracle*)h;
  StepWork w; memset(&w, 0, sizeof(w));
  Ctx x{&o->st[i], o->cold(i), &o->st[i].rng, o->st[i].hd, &w};
  return state_edit(x, op, a0, a1, a2);
}
// cc4o_step that also checks the engine's dirty-row marks (StepWork.hdirty, hd_touch): returns the number of HostDyn rows the
// step changed WITHOUT marking them (must be 0: the four-wave kernel writes back only marked rows), *marked = rows marked
int cc4o_step_check_marks(void* h, int i, const int32_t* actions, const uint8_t* msgs, int* marked) {
  Oracle* o = (Oracle*)h;
  StepWork w; memset(&w, 0, sizeof(w));
  Ctx x{&o->st[i], o->cold(i), &o->st[i].rng, o->st[i].hd, &w};
  x.lg = o->evlog ? &o->cold(i)->evlog : nullptr;
  HostDyn before[MAXH];
  memcpy(before, o->st[i].hd, sizeof(before));
  env_step(x, actions, msgs);
  int bad = 0, m = 0;
  for (int k = 0; k < MAXH; ++k) {
    const bool mk = (w.hdirty[k >> 5] >> (k & 31)) & 1u;
    m += mk;
    if (!mk && memcmp(&before[k], &o->st[i].hd[k], sizeof(HostDyn)) != 0) ++bad;
  }
  if (marked) *marked = m;
  return bad;
}
// whole-batch step, OpenMP over envs when built with -fopenmp (bench.py cpu_baseline)
void cc4o_step_all(void* h, const int32_t* actions /* [n][5] */) {
  Oracle* o = (Oracle*)h;
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < o->n; ++i) {
    StepWork w; memset(&w, 0, sizeof(w));
    Ctx x{&o->st[i], o->cold(i), &o->st[i].rng, o->st[i].hd, &w};
    x.lg = o->evlog ? &o->cold(i)->evlog : nullptr;
    env_step(x, actions + 5 * i, nullptr);
  }
}
// whole-batch step with the results gathered in one call (OpenMP over episodes): what OracleVecEnv.step does episode by
// episode -- an episode that reported done is regenerated instead of stepped when `autoreset` is set (CybORG.reset(seed=None):
// the stream continues) -- so that the full-size GPU parity tests (8192 episodes, every step) finish in seconds
void cc4o_step_batch(void* h, const int32_t* actions /* [n][5] or null */, const uint8_t* msgs /* [n][5][8] or null */, int autoreset,
                     int rng_mode, int steps, int policy, int32_t* obs /* [n][578] */, float* rew, uint8_t* done, uint32_t* err) {
  Oracle* o = (Oracle*)h;
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < o->n; ++i) {
    StepWork w; memset(&w, 0, sizeof(w));
    Ctx x{&o->st[i], o->cold(i), &o->st[i].rng, o->st[i].hd, &w};
    x.lg = o->evlog ? &o->cold(i)->evlog : nullptr;
    bool was_reset = false;
    if (autoreset && o->st[i].done) {
      uint32_t ws[RESET_WS_WORDS];
      env_reset(x, 0, rng_mode, steps, true, policy, o->topo, rng_mode == 1 ? ws : nullptr);
      was_reset = true;
    } else env_step(x, actions ? actions + 5 * i : nullptr, msgs ? msgs + 5 * MSG_LEN * i : nullptr);
    const EnvState* s = &o->st[i];
    env_flat_obs<int32_t>(s, obs + (size_t)OBS_TOTAL * i);
    rew[i] = was_reset ? 0.f : s->reward; done[i] = s->done; err[i] = s->err;
  }
}
int cc4o_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void cc4o_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
// same document as cc4_get_true_state (include/cc4.h); returns bytes needed incl. NUL
long long cc4o_true_state(void* h, int i, char* json, size_t cap) {
  Oracle* o = (Oracle*)h;
  std::string doc = export_true_state(o->st[i], *o->cold(i));
  if (json && cap >= doc.size() + 1) memcpy(json, doc.c_str(), doc.size() + 1);
  return (long long)doc.size() + 1;
}
// same layout as cc4_get_topology (include/cc4.h)
void cc4o_topology(void* h, int i, uint8_t* out) {
  Oracle* o = (Oracle*)h;
  const EnvState& s = o->st[i];
  for (int k = 0; k < NSUB; ++k) { out[k] = s.cidr_octet[k]; out[9 + k] = s.n_users[k]; out[18 + k] = s.n_servers[k]; }
  for (int k = 0; k < MAXH; ++k) { out[27 + 2 * k] = bit_get(s.exists, k) ? 1 : 0; out[28 + 2 * k] = o->cold(i)->hs[k].ip_octet; }
}
void cc4o_obs(void* h, int i, int32_t* out) { const EnvState* s = &((Oracle*)h)->st[i]; env_flat_obs<int32_t>(s, out); }
// the two per-value enumerations of the same vector (by position / by kind), for the host-logic test
void cc4o_obs_variants(void* h, int i, int32_t* by_pos, int32_t* by_kind) {
  const EnvState* s = &((Oracle*)h)->st[i];
  for (int k = 0; k < OBS_TOTAL; ++k) by_pos[k] = env_flat_obs_at(s, k);
  for (int v = 0; v < OBS_TOTAL; ++v) { int idx = -1; int val = env_flat_obs_sorted(s, v, &idx); by_kind[idx] = val; }
}
// the table form of the fast part (obs_fast_entry / obs_fast_value: what the device kernels encode from), values beyond OBS_FAST = -1
void cc4o_obs_by_table(void* h, int i, int32_t* out) {
  const EnvState* s = &((Oracle*)h)->st[i];
  for (int k = 0; k < OBS_TOTAL; ++k) out[k] = -1;
  for (int v = 0; v < OBS_FAST; ++v) { const uint32_t e = obs_fast_entry(v); out[e & 0x3FF] = obs_fast_value(e, s); }
}
float cc4o_reward(void* h, int i) { return ((Oracle*)h)->st[i].reward; }
int cc4o_done(void* h, int i) { return ((Oracle*)h)->st[i].done; }
uint32_t cc4o_err(void* h, int i) { return ((Oracle*)h)->st[i].err; }
void cc4o_mask(void* h, int i, uint8_t* out) { blue_action_mask(&((Oracle*)h)->st[i], out); }
void cc4o_rng_state(void* h, int i, uint64_t* out /* s_hi s_lo inc_hi inc_lo has32 u32 ndraw */) {
  const Rng& r = ((Oracle*)h)->st[i].rng;
  out[0] = r.s_hi; out[1] = r.s_lo; out[2] = r.inc_hi; out[3] = r.inc_lo; out[4] = r.has32; out[5] = r.u32; out[6] = r.ndraw;
}

// RNG conformance hook: run a script of draws on a fresh stream. ops: 0 random() 1 below(arg) 2 shuffle_consume(arg)
// 3 next64 4 next32. out receives the value (doubles bit-cast) per op.
void cc4o_rng_script(uint64_t seed, int mode, int n, const int32_t* ops, const uint32_t* args, uint64_t* out) {
  Rng r; rng_seed(&r, seed, (uint32_t)mode);
  for (int i = 0; i < n; ++i) {
    switch (ops[i]) {
      case 0: { double d = rng_random(&r); memcpy(&out[i], &d, 8); break; }
      case 1: out[i] = rng_below(&r, args[i]); break;
      case 2: rng_shuffle_consume(&r, (int)args[i]); out[i] = 0; break;
      case 3: out[i] = rng_next64(&r); break;
      default: out[i] = rng_next32(&r); break;
    }
  }
}

// counter mode: the threshold forms of a uniform draw (cc4_rng.h rng_random_lt / rng_random_le / rng_random_quarter) and the raw words, one op per entry:
// 0: rng_random_lt(t[i])  1: rng_random_le(t[i])  2: rng_random_quarter  3: rng_next32  4: rng_set_stream((uint32)t[i])
void cc4o_rng_script2(uint64_t seed, int mode, int n, const int32_t* ops, const double* t, uint64_t* out) {
  Rng r; rng_seed(&r, seed, (uint32_t)mode);
  rng_begin_step(&r, 7);
  for (int i = 0; i < n; ++i) {
    switch (ops[i]) {
      case 0: out[i] = rng_random_lt(&r, t[i]) ? 1 : 0; break;
      case 1: out[i] = rng_random_le(&r, t[i]) ? 1 : 0; break;
      case 2: out[i] = (uint64_t)rng_random_quarter(&r); break;
      case 4: rng_set_stream(&r, (uint32_t)t[i]); out[i] = 0; break;
      default: out[i] = rng_next32(&r); break;
    }
  }
}
// the device's tables against the functions they are filled from: blue_slot_shape(b, idx) and obs_fast_entry(v)
uint32_t cc4o_blue_slot_shape(int b, int idx) { return blue_slot_shape(b, idx); }
uint32_t cc4o_obs_fast_entry(int v) { return obs_fast_entry(v); }
uint32_t cc4o_monitor_roll4(uint32_t ev4, int w) { return monitor_roll4(ev4, monitor_watch_mask(w)); }
uint32_t cc4o_monitor_roll(int h, uint32_t ev) { return monitor_roll(h, (uint8_t)ev); }

// "name offset" lines for EnvState members (maps a differing byte offset back to a field when bisecting)
int cc4o_layout(char* buf, int cap) {
  int n = 0;
#define F(m) n += snprintf(buf + n, cap - n, #m " %zu\n", offsetof(EnvState, m))
  F(rng); F(step_count); F(steps); F(phase); F(phase_len); F(err); F(reward); F(done); F(blocks); F(cidr_octet); F(n_users);
  F(n_servers); F(green_host); F(pend); F(npend); F(exists); F(red_hosts); F(spool_used); F(msg); F(bexec); F(rexec); F(brm);
  F(blue); F(spool); F(red); F(hev); F(hd);
#undef F
#define G(m) n += snprintf(buf + n, cap - n, "red." #m " %zu\n", offsetof(RedAgent, m))
  G(sord); G(known_bm); G(fsm_order); G(fsm_st4); G(fsm_hn); G(as_ip); G(as_hn); G(obs);
#undef G
#define G(m) n += snprintf(buf + n, cap - n, "red." #m " %zu\n", offsetof(RedAgent, h) + offsetof(RedHdr, m))
  G(queue); G(as_subnet); G(fsm_step); G(nsess); G(nknown); G(fsm_n); G(nobs); G(active); G(obs_success); G(exec_type); G(new_sess_host); G(new_sess_id); G(start_host);
#undef G
  n += snprintf(buf + n, cap - n, "sizeof.RedAgent %zu\nsizeof.BlueAgent %zu\nsizeof.HostDyn %zu\nsizeof.HostStatic %zu\nsizeof.EnvState %zu\n",
                sizeof(RedAgent), sizeof(BlueAgent), sizeof(HostDyn), sizeof(HostStatic), sizeof(EnvState));
  return n;
}

// canonical text dump of one episode's state (parity bisecting against oracle/refgen/ref_dump.py)
int cc4o_dump(void* h, int i, char* buf, int cap) {
  const EnvState& s = ((Oracle*)h)->st[i];
  const EnvCold& cold = *((Oracle*)h)->cold(i);
  int n = 0;
#define P(...) do { if (n < cap) n += snprintf(buf + n, cap - n, __VA_ARGS__); } while (0)
  P("step %d phase %d blocks", s.step_count, s.phase);
  for (int k = 0; k < NSUB; ++k) P(" %u", s.blocks[k]);
  P("\n");
  for (int hh = 0; hh < MAXH; ++hh) {
    if (!bit_get(s.exists, hh)) continue;
    const HostDyn& d = s.hd[hh];
    P("host %d procs", hh);
    for (int k = 0; k < d.nproc; ++k) { const Proc pr = export_proc(s, cold, hh, k); P(" (%d,%d,%d)", pr.pid, pr.kind, pr.flags & 1); }
    P(" svcs");
    for (int k = 0; k < hd_nsvc(d); ++k) P(" (%d,%d,%d,%d)", d.svcs[k].kind, (d.svcs[k].st & SV_ACTIVE) ? 1 : 0, (d.svcs[k].st & 0x7F) * 20, d.svcs[k].pid);
    { const int ev = s.hev[hh]; P(" ev %d%d%d%d\n", (ev & EV_CUR_CONN) ? 1 : 0, (ev & EV_CUR_PROC) ? 1 : 0, (ev & EV_OLD_CONN) ? 1 : 0, (ev & EV_OLD_PROC) ? 1 : 0); }
  }
  for (int r = 0; r < NRED; ++r) {
    const RedAgent& a = s.red[r];
    P("red %d active %d sess", r, a.h.active);
    for (int k = 0; k < a.h.nsess; ++k) { const RSess& q = s.spool[a.sord[k]]; P(" (%d,%d,%d,%d,%d)", q.id, q.host, q.pid, (q.flags & RS_ABSTRACT) ? 1 : 0, (q.flags & RS_ROOT) ? 1 : 0); }
    P(" known");
    for (int k = 0; k < a.h.nknown; ++k) P(" %d", cold.known_sid[r][k]);
    P(" fsmstep %d fsm", a.h.fsm_step);
    for (int k = 0; k < a.h.fsm_n; ++k) { int hh = a.fsm_order[k]; P(" (%d,%d,%d)", hh, fsm_get(a, hh), bit_get(a.fsm_hn, hh) ? 1 : 0); }
    for (int hh = 0; hh < MAXH; ++hh) if (fsm_get(a, hh) == FS_F) P(" (%d,%d,%d)", hh, FS_F, bit_get(a.fsm_hn, hh) ? 1 : 0);  // 'F' hosts after the live list
    P(" subnets %u busy %d qt %d\n", a.h.as_subnet, a.h.queue.busy ? 1 : 0, a.h.queue.busy ? a.h.queue.type : -1);
  }
  for (int b = 0; b < NBLUE; ++b) {
    const BlueAgent& a = s.blue[b];
    P("blue %d sus", b);
    for (int hh = 0; hh < MAXH; ++hh)  // grouped by host, chronological within a host
      for (int k = 0; k < a.nsus; ++k) if ((int)(cold_sus(&cold, s.steps, b)[k] >> 16) == hh) P(" (%d,%d)", hh, (int)(cold_sus(&cold, s.steps, b)[k] & 0xFFFF));
    P("\n");
  }
#undef P
  return n;
}

}  // extern "C"
