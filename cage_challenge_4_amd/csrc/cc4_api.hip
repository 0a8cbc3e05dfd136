// cc4_api.hip -- the host side of libcc4.so: the handle, the launch schedules, the exchange, the C ABI (include/cc4.h, include/cc4_debug.h).
#include "cc4_kernels.h"
#include "cc4_export.h"
#include "cc4_kernel_decls.h"

struct cc4_handle {
  cc4_config cfg;
  hipStream_t stream = nullptr;
  EnvState* d_state = nullptr; EnvCold* d_cold = nullptr;
  size_t cold_row = 0;             // bytes per cold row: fixed part + the containers sized from cfg.steps (cold_row_bytes)
  int32_t* d_actions = nullptr; uint8_t* d_msgs = nullptr; uint64_t* d_seeds = nullptr; uint8_t* d_envmask = nullptr;
  int32_t* d_obs = nullptr; float* d_reward = nullptr; uint8_t* d_done = nullptr; uint32_t* d_err = nullptr;
  uint8_t* d_mask = nullptr; uint64_t* d_rng = nullptr;
  // d_obs | d_reward | d_err | d_done are ONE allocation (base d_obs), d_actions | d_msgs another (base d_actions): cc4_step_fetch moves
  // a step's inputs and outputs with one copy each; small batches go through pinned staging buffers (a copy to or from pageable
  // memory is staged by the runtime anyway, synchronously and per call)
  size_t out_bytes = 0, in_bytes = 0;
  uint8_t* pin_out = nullptr; uint8_t* pin_in = nullptr;
  // Handles of up to SMALL_IO_ENVS episodes (the single-episode wrapper surface) keep both blocks in pinned HOST memory the device reads
  // and writes directly: the step kernel fetches its five action indices over PCIe and posts its results there, so a step is a launch and
  // one host wait -- no copy engine in either direction (each DMA costs ~10 us of latency for a few hundred bytes).  CC4_SMALL_IO=0: off.
  static constexpr int SMALL_IO_ENVS = 16;
  bool small_io = false;
  // cc4_keep_previous / cc4_replay_logged (small handles): the rows as they stood before the last step, so that the step can be repeated
  // with the event log on when -- and only when -- somebody asks what happened in it (the single-episode wrapper surface: flat
  // observations need no log, and the logging build of the numpy-stream kernel walks its green actions serially: +30 us per step)
  bool keep_prev = false, prev_valid = false;
  EnvState* d_prev_state = nullptr; EnvCold* d_prev_cold = nullptr; uint8_t* d_prev_out = nullptr;
  const int32_t* prev_actions = nullptr; const uint8_t* prev_msgs = nullptr; bool prev_full_obs = false, prev_ext = false;
  // byte observations and gathered observations ([world*N][578]) in a ring of OBS_RING buffers: the all-gather of step t
  // overlaps later steps, and the compute stream waits for the communication stream only once per OBS_WAIT_EVERY steps
  // (a cross-stream wait in front of every launch costs the stream ~10 us)
  static constexpr int OBS_RING = 8, OBS_WAIT_EVERY = 4;
  static constexpr int MAX_GROUPS = cc4_handle_max_groups;          // launches per step (episode groups, below); CC4_GROUPS may ask for up to this many
  uint8_t* d_obs8[OBS_RING] = {};
  uint8_t* d_all_obs8[OBS_RING] = {};
  long long gather_seq[OBS_RING] = {};           // sequence number of the last all-gather that read buffer b (0 = none)
  long long gathers_issued = 0, gathers_waited = 0;
  hipEvent_t tev_start[cc4_handle_max_groups] = {}, tev_stop[cc4_handle_max_groups] = {};   // timing events the NEXT launch of a group carries (cc4_run_random_steps)
  long long comm_delay_ticks = 0;                // debug: spin this long on the communication stream ahead of every all-gather
  long long gather_stalls = 0;                   // a step launch found the all-gather it had to wait for still running
  long long stat_steps = 0; double stat_launch_us = 0, stat_gather_us = 0;   // cc4_host_stats
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_step[OBS_RING][MAX_GROUPS] = {}, ev_comm[OBS_RING] = {};   // ev_step[b][g]: group g's launch that wrote buffer b; ev_comm[q % OBS_RING]: all-gather number q has completed
  int obs_buf = 0;                               // buffer written by the most recent step
  int gather_buf = -1;                           // buffer of the most recent all-gather (-1: none issued)
  bool step_event_attached = false;              // ev_step[obs_buf] was recorded by the launch of that step itself
  // A step of a large batch is issued as `ngroups` launches, one per contiguous group of episodes, each group on its own HIP
  // stream (group 0 on `stream`): episodes are independent, a group's next step depends only on its own previous one, so while
  // one group's launch drains -- its last blocks running on a half-empty chip -- the other group's launch fills the free
  // slots, and the chip stays full across step boundaries.  Measured on MI355X (r03, 8192 episodes, counter mode): one launch
  // per step 507 M agent-env steps/s, two groups of 4096 on two streams 639 M (a single 32768-episode launch per step: 605 M).
  int ngroups = 1;
  int cus = 256;                                 // compute units of the device
  int glo[MAX_GROUPS + 1] = {};                  // group g = episodes [glo[g], glo[g + 1])
  hipStream_t gstream[MAX_GROUPS] = {};          // gstream[0] == stream
  hipEvent_t gev[MAX_GROUPS] = {};               // group stream -> main stream ordering (join_groups)
  hipEvent_t mev = nullptr;                      // main stream -> group streams ordering (fork_groups)
  bool auto_groups = true;                       // the number of groups is the library's choice (no CC4_GROUPS)
  bool groups_busy = false;                      // a group stream other than the main one may hold unfinished step launches
  bool joined_between = false;                   // something ordered the main stream behind all groups (or waited for them) since the last step launches:
                                                 // the caller works on the WHOLE batch between steps (launch_step: one launch then, not one per group)
  bool main_ahead = false;                       // the main stream holds work the group streams have not been ordered behind
  unsigned long long* d_prof = nullptr;
  int dbg_stop = 0;                  // cc4_debug_stop_phase
  uint32_t* d_reset_ws = nullptr;    // k_step_philox1's generation work area, [num_envs][RESET_WS_WORDS]
  uint8_t* d_unpacked = nullptr;                 // [world*N][578] bytes: cc4_unpack_obs_device
  int evlog_on = 0;               // cc4_enable_event_log
  // externally submitted red / green actions (cc4_step_ex).  Once a handle has taken any, its steps run the full builds of the
  // kernels (an action queued for several ticks carries its own rates into later steps), with d_ext all XA_NONE for the steps
  // that submit nothing
  // the persistent run kernel (k_run_philox1: K steps of the batch in one launch; RunArgs): per-partition ticket
  // counters, per-episode progress, partition owners in ONE buffer (cleared by one memset per call), the CU table
  uint32_t* d_run = nullptr;      // [P ticket | P owner | n progress]
  int32_t* d_slot_part = nullptr; // [CC4_SLOTS] CU slot id -> 1 + partition (persist_setup)
  unsigned long long* d_timeline = nullptr;   // CC4_PERSIST_TIMELINE: per-wave time stamps of the current persistent launch
  size_t run_words = 0;           // words of d_run
  int run_P = 0, run_grid = 0;    // partitions (= CUs that take waves; XCD pools: = XCDs), waves per launch; 0: the persistent path is off
  int run_G = 0;                  // exchange groups of the persistent kernel (episode e counts in group e % run_G): the device's CUs
  int run_pool = 2;               // the persistent kernel's schedule (RunArgs.pool; CC4_PERSIST_SCHED): 2 = per-CU partitions balanced inside the XCD,
                                  // 0 = the per-CU partitions of r04 / r05 (only a call's tail is shared), 1 = XCD pools (experiment)
  uint8_t xcc_lo[8] = {0}, xcc_n[8] = {0};
  int run_thr = 16;               // schedule 2: a wave helps the partition that lags most once its own is more than this many tickets ahead (CC4_PERSIST_THR)
  uint8_t xcc_pool[8] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
  uint32_t* d_pool = nullptr;     // [2][8][TK_STRIDE] the pools' ticket counters, one set per call parity
  int run_SA = 0, run_SB = 1, run_nB = 0, run_single = 0;   // runs of steps (RunArgs.SA ..; CC4_PERSIST_RUNS="SA,SB,nB,single"; SA = 1: every step an item, as in r05;
                                                            // SA = 0: chosen per call -- 4 steps, 8 in calls of 64 steps and more: profiles/r06_runs_ab.txt, r06_sched_ab2.txt)
  uint32_t pool_base = 0;         // steps every episode's progress word stands at (XCD pools: the words are not cleared between calls)
  int pool_parity = 0;
  int persist_state = -1;         // -1 off / unavailable, 0 not set up yet (persist_setup on first use), 1 on
  bool whole_batch_steps = true;  // CC4_WHOLE_BATCH_STEPS=0: the step entry points always launch per group (A/B)
  int persist_order = 0;          // RunArgs.order (CC4_PERSIST_ORDER)
  int run_margin = 0;             // episode blocks per CU the one-launch forms leave free (choose_run_form)
  // the per-step hand-off out of the one-launch kernels (XchgArgs): with a communicator, cc4_run_random_steps stays ONE launch and the
  // communication stream follows the kernel's per-step counters (xchg_*)
  static constexpr int XRING = 32;
  bool xchg_on = false;           // cc4_comm_init; CC4_EXCHANGE_INKERNEL=0 keeps the per-step launches
  int xchg_chunk = 8;             // steps per gate / publish on the communication stream (CC4_EXCHANGE_CHUNK; their slabs go out in ONE all-gather: the host
                                  // pays ~25 us to enqueue a wait, an all-gather and a publish -- more than a step of a small batch lasts)
  uint8_t* d_xslab = nullptr;     // [XRING][n][OBS_PACKED]
  uint8_t* d_xall = nullptr;      // [XRING][world * n][OBS_PACKED]
  int khz = 0;                    // wall-clock rate (hipDeviceAttributeWallClockRate), asked once
  // ---- rollouts with the policy in the loop (cc4_rollout_begin .. cc4_rollout_end)
  int32_t* d_ract = nullptr;      // [2][n][5] action slots (step j reads slot j % 2)
  uint32_t* d_rready = nullptr;   // [P][32] words: word g of partition p's line = actions of steps < value are published for policy group g (every line holds the same)
  uint32_t* d_rcnt = nullptr;     // [P][RPG][XRING] episodes of (partition, policy group) whose packed row of step j is in memory (slot j % XRING)
  uint32_t* d_rfail = nullptr;    // [1] a gate gave up
  hipStream_t policy_stream = nullptr;   // = gpolicy[0]
  hipStream_t gpolicy[4] = {nullptr, nullptr, nullptr, nullptr};   // one policy stream per policy group: the groups' gate -> policy -> publish chains run side by side
  hipEvent_t rev = nullptr;       // the rollout's starting observations are packed (slab XRING - 1)
  int rollout_k = 0;              // > 0: a rollout of that many steps is in flight
  bool rollout_entering = false;  // cc4_rollout_end is draining it (its own calls may pass join_groups)
  int rollout_watchdog_ms = 2000;
  int rollout_margin = 1;
  int rpg = 4;                    // policy groups (CC4_ROLLOUT_GROUPS, 1 .. RPG_MAX)
  int obs8_from_slab = -1;        // >= 0: the per-step ring's current buffer is to be filled from this slab of the exchange ring (xchg_end), when somebody reads it
  uint32_t* d_xflags = nullptr;   // [0] gathered, [1] timeout (what the waits poll)
  uint32_t* d_xgcnt = nullptr;    // [groups][XRING] group counters (xchg_count)
  uint32_t* h_xtimeout = nullptr; // pinned host word the kernel raises when a wait gives up (read without a copy)
  uint32_t* d_xtimeout = nullptr; // its device address
  int xflags_clean = 0;           // the flags are cleared already (behind the previous call) and xev says when
  hipEvent_t xev = nullptr;
  long long xchg_calls = 0, xchg_timeouts = 0;
  int xchg_watchdog_ms = 2000;
  uint8_t* last_gathered = nullptr;   // gathered rows of the most recent all-gather, whichever path issued it
  uint8_t* d_xlog = nullptr;      // debug (cc4_debug_gather_log): every gathered slab in issue order, [xlog_cap][world * n][OBS_PACKED]
  int xlog_cap = 0, xlog_n = 0;
  bool persist_refused = false;   // persist_setup found an unexpected picture (said so on stderr; cc4_run_kernel reports the per-step kernel)
  // CC4_PERSIST_VERIFY=1: every one-launch call of cc4_run_random_steps is repeated with per-step launches on a shadow handle that starts
  // from a copy of this handle's rows, and the two results are compared episode by episode (verify_*)
  bool verify = false, is_shadow = false;
  int verify_every = 1024;        // without CC4_PERSIST_VERIFY: every verify_every-th persistent call is checked all the same (CC4_PERSIST_VERIFY_EVERY; 0: never)
  uint64_t persist_calls = 0;
  cc4_handle* shadow = nullptr;
  uint64_t* d_digest = nullptr;   // [num_envs] per-episode digest
  long long verify_calls = 0, verify_mismatches = 0;
  int persist_min_k = 10;         // shorter calls keep the per-step launches: a launch's ramp and tail cost a few steps' worth (with the tail's items shared
                                  // among the CUs of an XCD: K = 10: 733 vs 685 M, K = 20: 813 vs 742 M, K = 32: 857 vs 756 M; CC4_PERSIST_MIN_K)
  struct EnqPool* pool = nullptr; // one enqueue thread per group stream beyond the first (cc4_run_random_steps; enq_*)
  bool enq_threads = false;
  bool run1m = false;             // cc4_run_random_steps as ONE launch of k_run_philox1m (batches of the one-wave kernel that one launch holds)
  int multistep_minb = 5;         // which build of it: 5 (k_run_philox) or 8 blocks per CU (k_run_philox8)
  bool multistep = false;         // k_run_philox: cc4_run_random_steps as ONE launch, every block looping over the steps of its episode
  ExtAct* d_ext = nullptr;        // [num_envs][EXT_PER_ENV]
  bool ext_seen = false, ext_dirty = false;   // dirty: d_ext holds the records of an earlier step
  std::vector<ExtAct> h_ext;
  bool full_obs_next = true;      // the next step launch rewrites every observation value (fresh handle, restored state)
  uint32_t full_obs_gmask = 0;    // ... per group, for the group-wise launches of cc4_step_group_device
  bool philox_lean = false;       // k_step_philox1 (one wave per episode) instead of k_step_philox (cc4_create)
  int philox_minw = 1;            // which register budget of k_step_philox this batch size runs (1, 7 or 8 blocks per CU; cc4_create)
  ncclComm_t comm = nullptr; int rank = 0, world = 1;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> evs;                   // timing events of cc4_run_random_steps: [group][2 * timed group of launches + {start, stop}]
  std::string err;
};

static thread_local std::string g_create_err;

// The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a
// queue run their kernels one after the other.  Three launches per step fit the default; a fourth stream needs more queues
// (measured, r03 profiles/r03_hwq_sweep.txt: 8192 episodes, 3 / 4 launches per step: 715 / 445 M with 4 queues, 717 / 742 M with
// 8).  The variable is read when the runtime initialises, so it is set when this library is loaded (never overriding the
// user's choice) -- and whether the four streams of a handle really run side by side is measured on those very streams when the
// handle is created, not assumed (streams_run_concurrently): a process that initialised HIP earlier, or one whose other streams
// already occupy the queues (a second handle next to a busy first one), keeps three launches per step.
__attribute__((constructor)) static void cc4_runtime_env() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

// 1 if kernels launched on the n given streams at the same time run side by side, 0 if some of them share a hardware queue and
// run one after the other (or the probe failed).  ~1 ms.
static int streams_run_concurrently(hipStream_t* st, int n) {
  if (n < 2) return 1;
  int khz = 100000, dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  if (khz <= 0) khz = 100000;
  const long long ticks = 300LL * khz / 1000;            // 300 us per kernel
  bool ok = true;
  double one = 0, all = 0;
  for (int pass = 0; pass < 2 && ok; ++pass) {             // pass 0: one stream (also warms the kernel up), pass 1: all of them
    const int m = pass ? n : 1;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < m; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, st[i], ticks);
    for (int i = 0; i < m && ok; ++i) ok = hipStreamSynchronize(st[i]) == hipSuccess;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    (pass ? all : one) = us;
  }
  return ok && all < 1.5 * (one > 300.0 ? one : 300.0);
}

#define HIPCHK(h, call)                                                                        \
  do {                                                                                         \
    hipError_t _e = (call);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(_e);                            \
      return -1;                                                                               \
    }                                                                                          \
  } while (0)

// How a step of this handle is cut into launches, and which build of the counter-mode kernel they run (the kernels are
// chosen from what one LAUNCH puts on a CU and from what the whole batch does).
static void configure_groups(cc4_handle* h, int ng) {
  const int n = h->cfg.num_envs, cus = h->cus;
  if (ng < 1) ng = 1;
  if (ng > cc4_handle::MAX_GROUPS) ng = cc4_handle::MAX_GROUPS;
  if (ng > n) ng = n;
  h->ngroups = ng;
  for (int g = 0; g <= ng; ++g) h->glo[g] = (int)(((long long)n * g) / ng);
  const int gsize = (n + ng - 1) / ng;                                 // episodes per launch
  const int bpc = (gsize + cus - 1) / cus;                             // episode blocks of one launch per CU
  const int bpc_all = (n + cus - 1) / cus;                             // ... of all launches of a step
  // four-wave kernel: one round of <= 5 blocks per CU runs the unconstrained build; else the build whose residency fills whole
  // rounds best (exactly 8 per CU -- 2048 episodes on 256 CUs -- is one round of the 8-block build)
  h->philox_minw = bpc <= 5 ? 1 : (bpc == 8 ? 8 : 7);
  // ... and with several launches per step what counts is what they put on a CU together (r03 profiles: three launches, register
  // budget 1 / 7 / 8: 1024 episodes 176 / 174 / 164 M, 2048: 295 / 302 / 278, 4096: 355 / 442 / 417)
  // (r04 flags, four launches: 1536 episodes 269 / 266 / 257, 2048: 329 / 333 / 323, 3072: 364 / 433 / 423, 4096: 369 / 471 / 490 -- one wave: 507)
  if (ng > 1) h->philox_minw = bpc_all <= 6 ? 1 : 7;
  if (const char* v = getenv("CC4_PHILOX_MINW")) h->philox_minw = atoi(v);   // tuning override: 1, 7 or 8
  // The one-wave-per-episode build when a single launch puts more than eight episodes on a CU, or the launches of a step
  // together more than thirteen.  Measured on MI355X (M agent-env steps/s, four waves / one wave per episode; r02, one launch per
  // step): 1024 episodes 168 / 131, 2048: 262 / 238, 2304: 249 / 260, 3072: 294 / 322, 4096: 319 / 395, 8192: 391 / 510;
  // (r03, three launches per step): 1024: 175 / 137, 2048: 295 / 249, 3072: 347 / 345, 4096: 442 / 422, 6144: 455 / 556,
  // 8192: 466 / 659.  (The same kernel with the host table in LDS as well is no faster anywhere.)
  // (after the r03 changes to the one-wave kernel -- event bytes staged, rows on cache-line boundaries -- with four launches per
  // step: 2048 episodes 312 / 276, 3072: 401 / 388, 4096: 442 / 487, 5120: 453 / 560, 6144: 458 / 627)
  h->philox_lean = bpc > 8 || bpc_all > 13;
  if (const char* v = getenv("CC4_PHILOX_LEAN")) h->philox_lean = atoi(v) != 0;   // tuning / test override
}

// Every API call other than the step launches works on the main stream: order it behind whatever the group streams still
// hold (device-side waits, no host synchronisation), and remember that the next step launches must be ordered behind it.
static int join_groups(cc4_handle* h) {
  // (nearly every entry point comes through here: while a rollout's kernel is running the handle's rows and streams are its alone)
  if (h->rollout_k > 0 && !h->rollout_entering) { h->err = "a rollout is in flight on this handle: cc4_rollout_end first"; return -1; }
  if (h->ngroups > 1) {
    if (h->groups_busy) {
      for (int g = 1; g < h->ngroups; ++g) {
        HIPCHK(h, hipEventRecord(h->gev[g], h->gstream[g]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->gev[g], 0));
      }
      h->groups_busy = false;
    }
    h->main_ahead = true;
    h->joined_between = true;
  }
  return 0;
}
static int sync_all(cc4_handle* h) {
  // (!groups_busy: whatever the group streams were given, the main stream already waits for -- join_groups, or the joined end of
  // cc4_run_random_steps -- and a host wait on an idle stream is not free: ~8 us each inside a short timed region)
  for (int g = h->ngroups - 1; g >= 1; --g) if (h->groups_busy) HIPCHK(h, hipStreamSynchronize(h->gstream[g]));
  // (a query first: behind a call that ended with a synchronisation of its own -- cc4_run_random_steps -- the stream is idle, and asking is cheaper than waiting)
  if (hipStreamQuery(h->stream) != hipSuccess) { (void)hipGetLastError(); HIPCHK(h, hipStreamSynchronize(h->stream)); }
  h->groups_busy = false;
  h->joined_between = true;
  return 0;
}

// rand: draw the blue actions inside the step kernel from (seed0, t) and record them in the handle's action buffer
// one group's launch of a step: the kernel cc4_create picked for this handle, on the group's stream, carrying `start` / `stop` as the
// launch's own timing events (or null)
static void launch_range(cc4_handle* h, StepArgs a, int e0, int e1, hipStream_t st, bool full, hipEvent_t start, hipEvent_t stop);
static void launch_group(cc4_handle* h, StepArgs a, int g, bool full, hipEvent_t start, hipEvent_t stop) {
  launch_range(h, a, h->glo[g], h->glo[g + 1], h->gstream[g], full, start, stop);
}
static void launch_range(cc4_handle* h, StepArgs a, int e0, int e1, hipStream_t st, bool full, hipEvent_t start, hipEvent_t stop) {
  a.e0 = e0; a.n = e1; a.dbg_stop = h->dbg_stop;
  if (h->dbg_stop) a.full_obs = 0;     // (a restored batch would get every observation value rewritten: the measurement wants the steady-state encode)
  const size_t lds1 = offsetof(EnvState, hd);     // one-wave kernels: the agent part
  const dim3 grid(a.n - a.e0);
  if (h->cfg.rng_mode == 1) {
    if (h->philox_lean) {
      if (full || h->d_prof || (h->dbg_stop && !getenv("CC4_DEBUG_STOP_FAST"))) hipExtLaunchKernelGGL(k_step_philox1<true>, grid, dim3(WAVE), lds1, st, start, stop, 0, a);
      else hipExtLaunchKernelGGL(k_step_philox1<false>, grid, dim3(WAVE), lds1, st, start, stop, 0, a);
    }
    else if (full) hipExtLaunchKernelGGL((k_step_philox<true, 1>), grid, dim3(PT), sizeof(EnvState), st, start, stop, 0, a);
    else if (h->philox_minw == 8) hipExtLaunchKernelGGL((k_step_philox<false, 8>), grid, dim3(PT), sizeof(EnvState), st, start, stop, 0, a);
    else if (h->philox_minw == 7) hipExtLaunchKernelGGL((k_step_philox<false, 7>), grid, dim3(PT), sizeof(EnvState), st, start, stop, 0, a);
#ifndef CC4_SMALL_MINW
#define CC4_SMALL_MINW 1
#endif
    else hipExtLaunchKernelGGL((k_step_philox<false, CC4_SMALL_MINW>), grid, dim3(PT), sizeof(EnvState), st, start, stop, 0, a);
  } else {
    if (full || h->d_prof) hipExtLaunchKernelGGL(k_step<true>, grid, dim3(WAVE), lds1, st, start, stop, 0, a);
    else hipExtLaunchKernelGGL(k_step<false>, grid, dim3(WAVE), lds1, st, start, stop, 0, a);
  }
}

// ---- one enqueue thread per group stream (cc4_run_random_steps without a communicator).  The groups of a batch never wait for each
// other, so their launches need not come from one thread: the first launch on a stream that has been synchronised costs the calling
// thread ~10 us (3.5 us in the steady state), four in a row delay the last group's first kernel by 30-45 us in every timed region;
// issued side by side they cost one.  A worker spins for 200 us after a call (CC4_ENQ_SPIN_US; a loop of calls keeps it hot), then sleeps.
struct EnqPool {
  std::vector<std::thread> th;
  std::mutex mu; std::condition_variable cv;
  std::atomic<uint64_t> gen{0};
  std::atomic<int> pending{0}, failed{0};
  std::atomic<bool> quit{false};
  int spin_us = 200;            // how long a worker spins for the next call before it parks on the condition variable (CC4_ENQ_SPIN_US): a loop of
                                // calls with nothing in between keeps it hot, a caller that does host work between bursts gets its cores back
  StepArgs a{}; int k = 0; uint32_t t0 = 0; bool full = false, first_full_obs = false, join = false;
  hipEvent_t start[cc4_handle_max_groups] = {}, stop[cc4_handle_max_groups] = {};
};
static void enq_run_group(cc4_handle* h, EnqPool* P, int g) {
  StepArgs a = P->a;
  for (int i = 0; i < P->k; ++i) {
    a.rand_t = P->t0 + (uint32_t)i;
    a.full_obs = (i == 0 && P->first_full_obs) ? 1 : 0;
    launch_group(h, a, g, P->full, i == 0 ? P->start[g] : nullptr, i == P->k - 1 ? P->stop[g] : nullptr);
  }
  if (g > 0 && P->join && hipEventRecord(h->gev[g], h->gstream[g]) != hipSuccess) P->failed.fetch_add(1);   // the main stream waits for it: one host wait per call
  if (hipGetLastError() != hipSuccess) P->failed.fetch_add(1);
}
static void enq_worker(cc4_handle* h, EnqPool* P, int g) {
  (void)hipSetDevice(h->cfg.device_id);
  uint64_t seen = 0;
  for (;;) {
    auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (P->gen.load(std::memory_order_acquire) == seen && !P->quit.load(std::memory_order_relaxed)) {
      __builtin_ia32_pause();
      if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(P->spin_us)) {
        std::unique_lock<std::mutex> lk(P->mu);
        P->cv.wait(lk, [&] { return P->gen.load(std::memory_order_acquire) != seen || P->quit.load(); });
      }
    }
    if (P->quit.load()) return;
    seen = P->gen.load(std::memory_order_acquire);
    enq_run_group(h, P, g);
    P->pending.fetch_sub(1, std::memory_order_release);
  }
}
static void enq_pool_start(cc4_handle* h) {
  if (h->pool || h->ngroups < 2) return;
  h->pool = new EnqPool;
  if (const char* v = getenv("CC4_ENQ_SPIN_US")) h->pool->spin_us = atoi(v) > 0 ? atoi(v) : 0;
  for (int g = 1; g < h->ngroups; ++g) h->pool->th.emplace_back(enq_worker, h, h->pool, g);
}
static void enq_pool_stop(cc4_handle* h) {
  if (!h->pool) return;
  { std::lock_guard<std::mutex> lk(h->pool->mu); h->pool->quit.store(true); }
  h->pool->cv.notify_all();
  for (auto& t : h->pool->th) t.join();
  delete h->pool; h->pool = nullptr;
}

// api_step: one of the step entry points (cc4_step / _ex / _fetch / _device), as opposed to the loop of cc4_run_random_steps.  When the caller did
// something with the WHOLE batch since the last step (an upload, a fetch, a policy kernel over all observations: anything that went through
// join_groups or waited for the streams), the groups cannot run ahead of each other anyway -- a launch per group then pays a fork and a join
// across streams per step for nothing: ONE launch on the main stream (8192 episodes, k_random_actions + cc4_step_device per step: 334 -> 583 M;
// a loop of cc4_step_device with nothing in between keeps the groups and their overlap across steps).
static int launch_step(cc4_handle* h, const int32_t* d_actions, const uint8_t* d_msgs, bool rand = false, uint64_t seed0 = 0,
                       uint32_t t = 0, bool ext_uploaded = false, bool api_step = false) {
  if (h->ext_seen && h->ext_dirty && !ext_uploaded) {     // this step submits no red / green action: every record says so
    if (join_groups(h)) return -1;
    HIPCHK(h, hipMemsetAsync(h->d_ext, 0xFF, (size_t)h->cfg.num_envs * EXT_PER_ENV * sizeof(ExtAct), h->stream));
    h->ext_dirty = false;
  }
  const bool full = h->evlog_on || h->ext_seen;
  // the byte-observation buffer about to be overwritten may still be read by an overlapped all-gather
  int buf = h->comm ? (h->obs_buf + 1) % cc4_handle::OBS_RING : 0;
  if (h->comm && h->gather_seq[buf] > h->gathers_waited) {
    // the last all-gather that read this buffer must be complete; wait for a slightly newer one (the communication stream is
    // in order), so the next OBS_WAIT_EVERY-1 launches need no wait of their own -- but never for the newest one, which is
    // the one meant to overlap this step
    long long q = h->gather_seq[buf] + cc4_handle::OBS_WAIT_EVERY - 1;
    if (q > h->gathers_issued - 1) q = h->gathers_issued - 1;
    if (q < h->gather_seq[buf]) q = h->gather_seq[buf];
    // waited for by the host, not by the stream: the all-gather in question is several steps old and normally complete, and
    // a wait packet in the compute queue costs stream time whether or not it has to wait
    hipError_t qs = hipEventQuery(h->ev_comm[q % cc4_handle::OBS_RING]);
    if (qs == hipErrorNotReady) { h->gather_stalls++; HIPCHK(h, hipEventSynchronize(h->ev_comm[q % cc4_handle::OBS_RING])); }
    else HIPCHK(h, qs);
    h->gathers_waited = q;
  }
  const bool whole = api_step && h->whole_batch_steps && h->ngroups > 1 && h->joined_between && !h->groups_busy && !h->comm;
  h->joined_between = false;
  if (!whole && h->ngroups > 1 && h->main_ahead) {   // e.g. an action upload or a reset on the main stream: the group streams start behind it
    HIPCHK(h, hipEventRecord(h->mev, h->stream));
    for (int g = 1; g < h->ngroups; ++g) HIPCHK(h, hipStreamWaitEvent(h->gstream[g], h->mev, 0));
    h->main_ahead = false;
  }
  if (h->keep_prev) {      // the rows as they stand before this step (cc4_replay_logged); a small handle: one launch per step, main stream
    const size_t n = (size_t)h->cfg.num_envs;
    HIPCHK(h, hipMemcpyAsync(h->d_prev_state, h->d_state, n * sizeof(EnvState), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_prev_cold, h->d_cold, n * h->cold_row, hipMemcpyDeviceToDevice, h->stream));
    h->prev_actions = d_actions; h->prev_msgs = d_msgs; h->prev_full_obs = h->full_obs_next; h->prev_ext = h->ext_seen;
    h->prev_valid = !rand;
  }
  StepArgs a{h->d_state, h->d_cold, d_actions, d_msgs, h->d_obs, h->d_reward, h->d_done, h->d_err,
             h->comm ? h->d_obs8[buf] : nullptr, rand ? h->d_actions : nullptr, seed0, t,
             h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), h->full_obs_next ? 1 : 0,
             (uint32_t)h->cfg.topology_seed, h->d_prof, h->d_reset_ws, h->ext_seen ? h->d_ext : nullptr, 0};
  h->full_obs_next = false;
  if (whole) {
    hipEvent_t stop = h->tev_stop[0], start = h->tev_start[0];
    for (int g = 0; g < h->ngroups; ++g) h->tev_start[g] = h->tev_stop[g] = nullptr;
    // (fewer waves per CU so that the batch runs in whole rounds -- 8192 episodes: 16 per CU, two even rounds instead of 1.6 at 20 -- is slower at every
    // residency tried: 578 M at 20, 567 at 18, 538 at 16, 503 at 14; tools/ab/ab_whole_residency.sh)
    launch_range(h, a, 0, h->cfg.num_envs, h->stream, full, start, stop);
    HIPCHK(h, hipGetLastError());
    h->step_event_attached = false;
    h->main_ahead = true;            // (the group streams have not been ordered behind this launch)
    h->obs_buf = buf;
    return 0;
  }
  for (int g = 0; g < h->ngroups; ++g) {
    // with a communicator, the launch carries ev_step[buf][g] as its stop event: the event rides on the kernel's own completion
    // signal, where a separate hipEventRecord would put a marker packet between two step kernels (~5 us of idle stream time)
    hipEvent_t stop = h->comm ? h->ev_step[buf][g] : h->tev_stop[g];
    hipEvent_t start = h->comm ? nullptr : h->tev_start[g];        // timing rides on the kernels' own signals too: no marker packets
    h->tev_start[g] = h->tev_stop[g] = nullptr;
    launch_group(h, a, g, full, start, stop);
    HIPCHK(h, hipGetLastError());
  }
  h->step_event_attached = h->comm != nullptr;
  if (h->ngroups > 1) h->groups_busy = true;
  h->obs_buf = buf;
  h->obs8_from_slab = -1;           // (this step's packed rows are in the ring buffer it wrote)
  return 0;
}

// the persistent kernel of a handle's mode
static const void* persist_kernel(const cc4_handle* h) {
  if (h->cfg.rng_mode == 0) return reinterpret_cast<const void*>(k_run_pcg);
  return h->comm ? reinterpret_cast<const void*>(k_run_philox1x) : reinterpret_cast<const void*>(k_run_philox1);
}


// Which form cc4_run_random_steps takes on this handle (decided at cc4_create, again at cc4_comm_init): the multi-step form of the four-wave
// kernel (k_run_philox / k_run_philox8) for batches the chip holds at once, the plain multi-step form of the one-wave kernel (k_run_philox1m)
// up to 20 episodes per CU, the persistent kernel beyond.  `margin` = episode blocks per CU the multi-step kernels leave free,
// `persist_margin` = waves per CU the persistent kernel's grid leaves free (see cc4_comm_init).
static int choose_run_form(cc4_handle* h, int margin, int persist_margin = -1) {
  const cc4_config* cfg = &h->cfg;
  if (persist_margin < 0) persist_margin = margin;
  h->multistep = false; h->run1m = false;
  if (cfg->rng_mode == 1 && !h->philox_lean) {
    // the multi-step form of the four-wave kernel (k_run_philox): for batches the chip holds at once
    int per_cu = 0, per_cu8 = 0;
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_run_philox, PT, sizeof(EnvState)));
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu8, k_run_philox8, PT, sizeof(EnvState)));
    h->multistep = per_cu - margin > 0 && cfg->num_envs <= (per_cu - margin) * h->cus;
    h->multistep_minb = 5;
    if (!h->multistep && per_cu8 > per_cu && cfg->num_envs <= (per_cu8 - margin) * h->cus) { h->multistep = true; h->multistep_minb = 8; }
    if (const char* v = getenv("CC4_MULTISTEP")) {            // 0: off; 1: on (the build that holds the batch); 5 / 8: that build
      const int m = atoi(v);
      h->multistep = m != 0;
      if (m == 5 || m == 8) h->multistep_minb = m;
    }
    if (getenv("CC4_PERSIST_DEBUG")) fprintf(stderr, "[cc4] k_run_philox: %d / %d blocks per CU resident (margin %d), multistep %d (build %d)\n", per_cu, per_cu8, margin, (int)h->multistep, h->multistep_minb);
  }
  if (cfg->rng_mode == 1 && !h->multistep) {
    // (whichever per-step kernel the handle runs: a batch of 2049-5120 episodes that cc4_step serves with the four-wave kernel is served here by the one-wave loop)
    // the plain multi-step form of the one-wave kernel (k_run_philox1m) where one launch holds the whole batch: 20 waves per CU
    // (4096 episodes 507 -> 709 M, 5120: 586 -> 811 M; beyond the residency the second round runs on a half-empty chip and four
    // streams of per-step launches win: 8192: 740 vs 789 M, 16384: 812 vs 864 M -- profiles/r04_run1m_ab.txt)
    int per_cu = 0;
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_run_philox1m, WAVE, offsetof(EnvState, hd)));
    h->run1m = per_cu - margin > 0 && cfg->num_envs <= (per_cu - margin) * h->cus;
    if (const char* v = getenv("CC4_RUN1")) h->run1m = atoi(v) != 0;
  }
  // one enqueue thread per group stream in cc4_run_random_steps (EnqPool): on where the host has cores to spare; CC4_ENQ_THREADS=0/1 decides otherwise
  h->enq_threads = std::thread::hardware_concurrency() >= 8;
  if (const char* v = getenv("CC4_ENQ_THREADS")) h->enq_threads = atoi(v) != 0;
  // the persistent run kernel of large batches (k_run_philox1): set up on first use (persist_setup); CC4_PERSIST=0 keeps it off
  h->persist_state = -1;
  h->run_margin = persist_margin;
  bool persist_mode = cfg->rng_mode == 1 && !h->multistep && !h->run1m;
  persist_mode = persist_mode || cfg->rng_mode == 0;
  if (persist_mode) {
    int per_cu = 0;
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persist_kernel(h), WAVE, offsetof(EnvState, hd)));
    const int grid = (per_cu - persist_margin) * h->cus;
    // batches of more than the chip holds at once (with the tail shared, also just more).  The numpy-stream mode has no other one-launch form: there
    // the persistent kernel also serves batches from half the residency up (a partition of fewer episodes than the CU has waves just leaves waves idle)
    // (counter mode: what neither multi-step kernel holds -- 5121 .. 6144 episodes at 20 / 24 waves per CU -- is the persistent kernel's as well)
    if (per_cu - persist_margin > 0 && (cfg->num_envs > grid || 2 * cfg->num_envs > grid)) h->persist_state = 0;
  }
  if (const char* v = getenv("CC4_PERSIST")) { if (atoi(v) == 0) h->persist_state = -1; }
  return 0;
}

extern "C" {

const char* cc4_last_error(cc4_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }
size_t cc4_state_bytes(void) { return sizeof(EnvState); }
int cc4_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
size_t cc4_algorithmic_bytes_per_env_step(void) {
  // state row in + out, flat obs out (int32), actions in, reward + done + err out (DESIGN.md "algorithmic bytes")
  return 2 * sizeof(EnvState) + 4 * OBS_TOTAL + 4 * NBLUE + 4 + 1 + 4;
}
size_t cc4_hot_bytes(void) { return offsetof(EnvState, hd); }
const char* cc4_step_kernel(cc4_handle* h) {
  if (!h) return "";
  if (h->cfg.rng_mode != 1) return "k_step";
  return h->philox_lean ? "k_step_philox1" : "k_step_philox";
}

// the kernel cc4_run_random_steps launches on this handle as it stands (no communicator, no event log): the step kernel, once per
// step and group -- or one of the one-launch forms
const char* cc4_run_kernel(cc4_handle* h) {
  if (!h) return "";
  const bool plain = (!h->comm || h->xchg_on) && !h->evlog_on && !h->ext_seen && !h->d_prof && !h->dbg_stop;
  if (plain && h->multistep) return h->multistep_minb == 8 ? "k_run_philox8" : "k_run_philox";
  if (plain && h->run1m) return "k_run_philox1m";
  if (plain && h->persist_state >= 0) return h->cfg.rng_mode == 0 ? "k_run_pcg" : (h->comm ? "k_run_philox1x" : "k_run_philox1");      // (calls of fewer than persist_min_k steps: the per-step launches)
  return cc4_step_kernel(h);
}
static int persist_setup(cc4_handle* h);
static int ensure_shadow(cc4_handle* h);
const char* cc4_run_kernel_for(cc4_handle* h, int32_t k) {
  if (!h) return "";
  // (the persistent kernel's discovery pass runs on first use: asking which kernel a call of k steps will launch is such a use -- the answer depends on it,
  // and a caller that asks before its timed region keeps it out of that region)
  if (h->persist_state == 0 && !h->run1m && !h->multistep && (!h->comm || h->xchg_on) && !h->evlog_on && !h->ext_seen && !h->d_prof && k >= h->persist_min_k && hipSetDevice(h->cfg.device_id) == hipSuccess) (void)persist_setup(h);
  const char* r = cc4_run_kernel(h);
  if (k < 2) return cc4_step_kernel(h);
  if (!h->multistep && !h->run1m && h->persist_state >= 0 && k < h->persist_min_k) return cc4_step_kernel(h);
  return r;
}

int cc4_create(const cc4_config* cfg, cc4_handle** out) {
  if (!cfg || !out || cfg->num_envs <= 0 || cfg->steps <= 0 || cfg->red_policy < 0 || cfg->red_policy > 3 ||
      cfg->green_policy < 0 || cfg->green_policy > 2 || cfg->blue_policy < 0 || cfg->blue_policy > 1 || cfg->rng_mode < 0 || cfg->rng_mode > 1) { g_create_err = "cc4_create: bad config"; return -2; }
  if (cfg->topology_seed != 0 && cfg->rng_mode != 1) { g_create_err = "cc4_create: topology_seed needs rng_mode 1 (the numpy stream draws scenario and dynamics from one generator)"; return -2; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = "cc4_create: no HIP device available (libcc4 has no CPU fallback)";
    return -3;
  }
  if (cfg->device_id < 0 || cfg->device_id >= ndev) { g_create_err = "cc4_create: device_id out of range"; return -2; }
  cc4_handle* h = new cc4_handle();
  h->cfg = *cfg;
  *out = h;
  // how a host thread waits for the device is the runtime's default unless CC4_HOST_WAIT=spin|yield|block says otherwise (spinning by default was
  // tried in r06: no measurable gain on a 20-step call, and the exchange's soak test failed once under it); ignored (hipErrorSetOnActiveProcess) when
  // the process has initialised the device some other way already
  if (const char* v = getenv("CC4_HOST_WAIT")) {
    static std::once_flag once;
    std::call_once(once, [v] {
      unsigned flags = hipDeviceScheduleAuto;
      if (!strcmp(v, "spin")) flags = hipDeviceScheduleSpin; else if (!strcmp(v, "yield")) flags = hipDeviceScheduleYield; else if (!strcmp(v, "block")) flags = hipDeviceScheduleBlockingSync;
      (void)hipSetDeviceFlags(flags);
      (void)hipGetLastError();
    });
  }
  HIPCHK(h, hipSetDevice(cfg->device_id));
  if (cc4_upload_pcg_tables() != hipSuccess) { h->err = "cc4_create: the numpy-stream kernel's jump table could not be uploaded"; return -1; }
  {
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, cfg->device_id));
    h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  {
    // Episode groups (see cc4_handle::ngroups).  Measured on MI355X (r03, profiles/r03_groups_sweep_philox.txt; M agent-env steps/s,
    // counter mode, 1 / 2 / 3 launches per step): 1024 episodes 165 / 173 / 175, 2048: 257 / 289 / 295, 4096: 392 / 413 / 442,
    // 8192: 503 / 629 / 659, 16384: 570 / 710 / 701; numpy stream, 8192 episodes: 272 / 352 / 363.  A fourth stream halves the
    // rate (the runtime's hardware queues), so three it is.  CC4_GROUPS overrides (1 .. 4).
    int ng = cfg->num_envs >= 1024 ? 3 : (cfg->num_envs >= 512 ? 2 : 1);
    // (a fourth launch where four streams really overlap: decided below, once the streams exist)
    h->auto_groups = getenv("CC4_GROUPS") == nullptr;
    if (const char* v = getenv("CC4_GROUPS")) ng = atoi(v);
    configure_groups(h, ng);
  }
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  h->gstream[0] = h->stream;
  for (int g = 1; g < h->ngroups; ++g) {   // only the streams that are used: the runtime spreads streams over few hardware queues
    HIPCHK(h, hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->gev[g], hipEventDisableTiming));
  }
  HIPCHK(h, hipEventCreateWithFlags(&h->mev, hipEventDisableTiming));
  if (h->auto_groups && h->ngroups == 3) {
    // a fourth launch per step where this handle's four streams really run side by side (8192 episodes 717 -> 742 M, 2048: 310 ->
    // 314 M, 1024: 179 -> 183 M; with two of them on one hardware queue: 445 M)
    HIPCHK(h, hipStreamCreateWithFlags(&h->gstream[3], hipStreamNonBlocking));
    if (streams_run_concurrently(h->gstream, 4)) {
      HIPCHK(h, hipEventCreateWithFlags(&h->gev[3], hipEventDisableTiming));
      configure_groups(h, 4);
    } else {
      (void)hipStreamDestroy(h->gstream[3]);
      h->gstream[3] = nullptr;
    }
  }
  if (!h->auto_groups && getenv("CC4_EXP_PROBE")) {   // experiment: explicit groups, but the streams / probe of the automatic path exist as well
    hipStream_t tmp[4];
    for (int g = 0; g < 4; ++g) { if (g < h->ngroups) tmp[g] = h->gstream[g]; else HIPCHK(h, hipStreamCreateWithFlags(&tmp[g], hipStreamNonBlocking)); }
    if (atoi(getenv("CC4_EXP_PROBE")) > 1) (void)streams_run_concurrently(tmp, 4);
  }
  size_t n = (size_t)cfg->num_envs;
  // experiment (CC4_EXP_MEM=1 fine-grained, 2 uncached): the episodes' rows in memory whose lines the vector L1 does not keep
  static const int exp_mem = getenv("CC4_EXP_MEM") ? atoi(getenv("CC4_EXP_MEM")) : 0;
  auto row_alloc = [&](void** p, size_t bytes) -> hipError_t {
    if (exp_mem == 1) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    if (exp_mem == 2) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
    return hipMalloc(p, bytes);
  };
  HIPCHK(h, row_alloc((void**)&h->d_state, n * sizeof(EnvState)));
  h->cold_row = cold_row_bytes(cfg->steps);
  HIPCHK(h, row_alloc((void**)&h->d_cold, n * h->cold_row));
  if (cfg->rng_mode == 1) HIPCHK(h, hipMalloc(&h->d_reset_ws, n * RESET_WS_WORDS * sizeof(uint32_t)));   // the one-wave kernel's generation work area
  else HIPCHK(h, hipMalloc(&h->d_reset_ws, n * 128 * sizeof(uint64_t)));                                 // numpy stream: the LCG window of the green actions (wave_green_exec), 1 KB per episode
  h->in_bytes = n * NBLUE * sizeof(int32_t) + n * NBLUE * MSG_LEN;
  h->small_io = cfg->num_envs <= cc4_handle::SMALL_IO_ENVS;
  if (const char* v = getenv("CC4_SMALL_IO")) h->small_io = h->small_io && atoi(v) != 0;
  if (h->small_io) {
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), h->in_bytes, hipHostMallocDefault));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_actions), h->pin_in, 0));
    memset(h->pin_in, 0, h->in_bytes);
  } else
  HIPCHK(h, hipMalloc(&h->d_actions, h->in_bytes));
  h->d_msgs = reinterpret_cast<uint8_t*>(h->d_actions) + n * NBLUE * sizeof(int32_t);
  HIPCHK(h, hipMalloc(&h->d_seeds, n * sizeof(uint64_t)));
  HIPCHK(h, hipMalloc(&h->d_envmask, n));
  h->out_bytes = n * OBS_TOTAL * sizeof(int32_t) + n * sizeof(float) + n * sizeof(uint32_t) + n;
  if (h->small_io) {
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), h->out_bytes, hipHostMallocDefault));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_obs), h->pin_out, 0));
  } else
  HIPCHK(h, row_alloc((void**)&h->d_obs, h->out_bytes));
  h->d_reward = reinterpret_cast<float*>(h->d_obs + n * OBS_TOTAL);
  h->d_err = reinterpret_cast<uint32_t*>(h->d_reward + n);
  h->d_done = reinterpret_cast<uint8_t*>(h->d_err + n);
  HIPCHK(h, hipMalloc(&h->d_mask, n * MASK_TOTAL));
  HIPCHK(h, hipMalloc(&h->d_rng, n * 7 * sizeof(uint64_t)));
  HIPCHK(h, hipMemsetAsync(h->d_state, 0, n * sizeof(EnvState), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_cold, 0, n * h->cold_row, h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_obs, 0, n * OBS_TOTAL * sizeof(int32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_done, 0, n, h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_err, 0, n * sizeof(uint32_t), h->stream));
  HIPCHK(h, hipEventCreate(&h->ev0));
  HIPCHK(h, hipEventCreate(&h->ev1));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (choose_run_form(h, 0)) return -1;
  if (const char* v = getenv("CC4_PERSIST_MIN_K")) h->persist_min_k = atoi(v);
  if (const char* v = getenv("CC4_PERSIST_ORDER")) h->persist_order = atoi(v);
  if (const char* v = getenv("CC4_WHOLE_BATCH_STEPS")) h->whole_batch_steps = atoi(v) != 0;
  if (const char* v = getenv("CC4_PERSIST_VERIFY")) h->verify = atoi(v) != 0;
  if (const char* v = getenv("CC4_PERSIST_VERIFY_EVERY")) h->verify_every = atoi(v) > 0 ? atoi(v) : 0;
  return 0;
}

void cc4_destroy(cc4_handle* h) {
  if (!h) return;
  enq_pool_stop(h);
  (void)hipSetDevice(h->cfg.device_id);
  for (int g = cc4_handle::MAX_GROUPS - 1; g >= 0; --g) if (h->gstream[g]) (void)hipStreamSynchronize(h->gstream[g]);
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  if (h->comm) ncclCommDestroy(h->comm);
  for (int b = 0; b < cc4_handle::OBS_RING; ++b) { for (int g = 0; g < cc4_handle::MAX_GROUPS; ++g) if (h->ev_step[b][g]) (void)hipEventDestroy(h->ev_step[b][g]); if (h->ev_comm[b]) (void)hipEventDestroy(h->ev_comm[b]); }
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  for (int g = 0; g < 4; ++g) if (h->gpolicy[g]) (void)hipStreamDestroy(h->gpolicy[g]);
  if (h->rev) (void)hipEventDestroy(h->rev);
  for (void* p : {(void*)h->d_ract, (void*)h->d_rready, (void*)h->d_rcnt, (void*)h->d_rfail}) if (p) (void)hipFree(p);
  void* ptrs[] = {h->d_state, h->d_cold, h->small_io ? nullptr : (void*)h->d_actions, h->d_seeds, h->d_envmask, h->small_io ? nullptr : (void*)h->d_obs,
                  h->d_mask, h->d_rng, h->d_reset_ws, h->d_ext, h->d_run, h->d_slot_part, h->d_pool};     // (d_msgs, d_reward, d_err, d_done live inside d_actions / d_obs; small handles: pinned host memory, freed below)
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->shadow) { cc4_destroy(h->shadow); h->shadow = nullptr; (void)hipSetDevice(h->cfg.device_id); }
  for (void* p : {(void*)h->d_prev_state, (void*)h->d_prev_cold, (void*)h->d_prev_out}) if (p) (void)hipFree(p);
  if (h->d_digest) (void)hipFree(h->d_digest);
  if (h->pin_in) (void)hipHostFree(h->pin_in);
  if (h->pin_out) (void)hipHostFree(h->pin_out);
  for (int b = 0; b < cc4_handle::OBS_RING; ++b) { if (h->d_obs8[b]) (void)hipFree(h->d_obs8[b]); if (h->d_all_obs8[b]) (void)hipFree(h->d_all_obs8[b]); }
  if (h->d_unpacked) (void)hipFree(h->d_unpacked);
  for (void* p : {(void*)h->d_xslab, (void*)h->d_xall, (void*)h->d_xflags, (void*)h->d_xlog, (void*)h->d_xgcnt}) if (p) (void)hipFree(p);
  if (h->h_xtimeout) (void)hipHostFree(h->h_xtimeout);
  if (h->xev) (void)hipEventDestroy(h->xev);
  for (hipEvent_t e : h->evs) if (e) (void)hipEventDestroy(e);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (int g = 1; g < cc4_handle::MAX_GROUPS; ++g) { if (h->gev[g]) (void)hipEventDestroy(h->gev[g]); if (h->gstream[g]) (void)hipStreamDestroy(h->gstream[g]); }
  if (h->mev) (void)hipEventDestroy(h->mev);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int cc4_reset(cc4_handle* h, const uint64_t* seeds, const uint8_t* env_mask) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  size_t n = (size_t)h->cfg.num_envs;
  if (seeds) HIPCHK(h, hipMemcpyAsync(h->d_seeds, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  if (env_mask) HIPCHK(h, hipMemcpyAsync(h->d_envmask, env_mask, n, hipMemcpyHostToDevice, h->stream));
  ResetArgs a{h->d_state, h->d_cold, seeds ? h->d_seeds : nullptr, env_mask ? h->d_envmask : nullptr, h->d_obs, h->d_reward,
              h->d_done, h->d_err, h->d_mask, h->cfg.num_envs, h->cfg.steps, h->cfg.rng_mode,
              (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), (uint32_t)h->cfg.topology_seed,
              h->comm ? h->d_obs8[h->obs_buf] : nullptr};
  // with a communicator the reset also writes the packed exchange row of its observations into the current ring buffer; an
  // overlapped all-gather may still be reading that buffer
  if (h->comm && h->gathers_issued > h->gathers_waited) { HIPCHK(h, hipStreamSynchronize(h->comm_stream)); h->gathers_waited = h->gathers_issued; }
  if (h->comm && h->obs8_from_slab >= 0) {      // the reset writes the current ring buffer's packed rows itself -- all of them, unless it is masked
    const size_t row = (size_t)h->cfg.num_envs * OBS_PACKED;
    if (env_mask) HIPCHK(h, hipMemcpyAsync(h->d_obs8[h->obs_buf], h->d_xslab + (size_t)h->obs8_from_slab * row, row, hipMemcpyDeviceToDevice, h->stream));
    h->obs8_from_slab = -1;
  }
  hipLaunchKernelGGL(k_reset, dim3(h->cfg.num_envs), dim3(WAVE), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  h->prev_valid = false;            // (cc4_replay_logged: no step to repeat)
  h->step_event_attached = false;   // the buffer's event must be recorded again before the next all-gather
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int cc4_step(cc4_handle* h, const int32_t* actions, const uint8_t* messages) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  size_t n = (size_t)h->cfg.num_envs;
  if (actions) HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, n * NBLUE * sizeof(int32_t), hipMemcpyDefault, h->stream));
  if (messages) HIPCHK(h, hipMemcpyAsync(h->d_msgs, messages, n * NBLUE * MSG_LEN, hipMemcpyDefault, h->stream));
  if (launch_step(h, actions ? h->d_actions : nullptr, messages ? h->d_msgs : nullptr, false, 0, 0, false, true)) return -1;
  return sync_all(h);
}

// The outputs of the last step (or reset) in one copy and one host synchronisation: observations, reward, done and error flags live in
// one device allocation.  Batches of up to PIN_MAX_ENVS episodes come through a pinned staging buffer (the copy is a real asynchronous
// DMA; a copy into pageable memory is staged by the runtime, call by call).  Any of the four pointers may be null.
constexpr int PIN_MAX_ENVS = 4096;
static int fetch_outputs(cc4_handle* h, int32_t* obs, float* reward, uint8_t* done, uint32_t* err) {
  if (join_groups(h)) return -1;
  const size_t n = (size_t)h->cfg.num_envs;
  const size_t b_obs = n * OBS_TOTAL * sizeof(int32_t), b_rew = n * sizeof(float), b_err = n * sizeof(uint32_t);
  if (h->small_io) {                                          // the kernels wrote into pinned host memory: wait, then read it
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (obs) memcpy(obs, h->pin_out, b_obs);
    if (reward) memcpy(reward, h->pin_out + b_obs, b_rew);
    if (err) memcpy(err, h->pin_out + b_obs + b_rew, b_err);
    if (done) memcpy(done, h->pin_out + b_obs + b_rew + b_err, n);
    return 0;
  }
  if (h->cfg.num_envs <= PIN_MAX_ENVS) {
    if (!h->pin_out) HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), h->out_bytes, hipHostMallocDefault));
    const size_t lo = obs ? 0 : b_obs;                         // (a caller that wants no observations does not pay for them)
    HIPCHK(h, hipMemcpyAsync(h->pin_out + lo, reinterpret_cast<const uint8_t*>(h->d_obs) + lo, h->out_bytes - lo, hipMemcpyDefault, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (obs) memcpy(obs, h->pin_out, b_obs);
    if (reward) memcpy(reward, h->pin_out + b_obs, b_rew);
    if (err) memcpy(err, h->pin_out + b_obs + b_rew, b_err);
    if (done) memcpy(done, h->pin_out + b_obs + b_rew + b_err, n);
    return 0;
  }
  if (obs) HIPCHK(h, hipMemcpyAsync(obs, h->d_obs, b_obs, hipMemcpyDefault, h->stream));
  if (reward) HIPCHK(h, hipMemcpyAsync(reward, h->d_reward, b_rew, hipMemcpyDefault, h->stream));
  if (err) HIPCHK(h, hipMemcpyAsync(err, h->d_err, b_err, hipMemcpyDefault, h->stream));
  if (done) HIPCHK(h, hipMemcpyAsync(done, h->d_done, n, hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_fetch(cc4_handle* h, int32_t* obs, float* reward, uint8_t* done, uint32_t* err) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  return fetch_outputs(h, obs, reward, done, err);
}
// cc4_step + cc4_fetch with one host synchronisation in all: inputs up in one copy, the step's launches, outputs down in one copy
int cc4_step_fetch(cc4_handle* h, const int32_t* actions, const uint8_t* messages, int32_t* obs, float* reward, uint8_t* done, uint32_t* err) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  const size_t n = (size_t)h->cfg.num_envs;
  const size_t b_act = n * NBLUE * sizeof(int32_t), b_msg = n * NBLUE * MSG_LEN;
  if (h->small_io) {                                          // (every earlier launch has completed: each call of this surface ends with a host wait)
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (actions) memcpy(h->pin_in, actions, b_act);
    if (messages) memcpy(h->pin_in + b_act, messages, b_msg);
  } else if (h->cfg.num_envs <= PIN_MAX_ENVS && (actions || messages)) {
    if (!h->pin_in) HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), h->in_bytes, hipHostMallocDefault));
    if (actions) memcpy(h->pin_in, actions, b_act);
    if (messages) memcpy(h->pin_in + b_act, messages, b_msg);
    const size_t lo = actions ? 0 : b_act, hi = messages ? b_act + b_msg : b_act;
    HIPCHK(h, hipMemcpyAsync(reinterpret_cast<uint8_t*>(h->d_actions) + lo, h->pin_in + lo, hi - lo, hipMemcpyDefault, h->stream));
  } else {
    if (actions) HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, b_act, hipMemcpyDefault, h->stream));
    if (messages) HIPCHK(h, hipMemcpyAsync(h->d_msgs, messages, b_msg, hipMemcpyDefault, h->stream));
  }
  if (launch_step(h, actions ? h->d_actions : nullptr, messages ? h->d_msgs : nullptr, false, 0, 0, false, true)) return -1;
  return fetch_outputs(h, obs, reward, done, err);
}

// cc4_step plus the red / green entries of the step's `actions` dict (SimulationController.py:236-240)
int cc4_step_ex(cc4_handle* h, const int32_t* actions, const uint8_t* messages, const cc4_agent_action* red, const cc4_agent_action* green) {
  static_assert(sizeof(cc4_agent_action) == sizeof(ExtAct) && offsetof(cc4_agent_action, session) == offsetof(ExtAct, sid) &&
                offsetof(cc4_agent_action, rate0) == offsetof(ExtAct, rate0) && offsetof(cc4_agent_action, flags) == offsetof(ExtAct, flags),
                "cc4_agent_action (include/cc4.h) is ExtAct (csrc/cc4_state.h)");
  if (!red && !green) return cc4_step(h, actions, messages);
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  const size_t n = (size_t)h->cfg.num_envs;
  if (!h->d_ext) HIPCHK(h, hipMalloc(&h->d_ext, n * EXT_PER_ENV * sizeof(ExtAct)));
  h->h_ext.resize(n * EXT_PER_ENV);
  memset(h->h_ext.data(), 0xFF, h->h_ext.size() * sizeof(ExtAct));      // type -1 everywhere: nothing submitted
  for (size_t e = 0; e < n; ++e) {
    if (red) memcpy(&h->h_ext[e * EXT_PER_ENV], red + e * NRED, NRED * sizeof(ExtAct));
    if (green) memcpy(&h->h_ext[e * EXT_PER_ENV + NRED], green + e * MAXG, MAXG * sizeof(ExtAct));
  }
  for (size_t i = 0; i < h->h_ext.size(); ++i) {     // what the kernels index with must be in range; everything else is the engine's validity check
    ExtAct& a = h->h_ext[i];
    const bool is_red = (i % EXT_PER_ENV) < (size_t)NRED;
    if (a.type == XA_NONE) continue;
    if (a.type < 0 || (is_red ? a.type > RA_INVALID : a.type > XG_INVALID) || a.host >= MAXH || (is_red && a.type == RA_DRS && a.arg >= NSUB) ||
        (is_red && a.type == RA_WITHDRAW && a.arg >= MAXH) || (!is_red && a.ticks > 1)) {
      h->err = "cc4_step_ex: action record " + std::to_string(i % EXT_PER_ENV) + " of episode " + std::to_string(i / EXT_PER_ENV) + " is out of range (type / host / subnet; a green action takes one tick)";
      return -2;
    }
  }
  HIPCHK(h, hipMemcpyAsync(h->d_ext, h->h_ext.data(), h->h_ext.size() * sizeof(ExtAct), hipMemcpyHostToDevice, h->stream));
  h->ext_seen = true; h->ext_dirty = true;
  if (actions) HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, n * NBLUE * sizeof(int32_t), hipMemcpyDefault, h->stream));
  if (messages) HIPCHK(h, hipMemcpyAsync(h->d_msgs, messages, n * NBLUE * MSG_LEN, hipMemcpyDefault, h->stream));
  if (launch_step(h, actions ? h->d_actions : nullptr, messages ? h->d_msgs : nullptr, false, 0, 0, true, true)) return -1;
  return sync_all(h);
}

// direct edits of one episode between steps (state_edit, csrc/cc4_engine.h): the row and its cold part come to the host, the
// engine's own host build edits them, they go back
int cc4_edit_state(cc4_handle* h, int32_t env, int32_t op, int32_t a0, int32_t a1, int32_t a2) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_edit_state: env out of range"; return -2; }
  EnvState* st = (EnvState*)malloc(sizeof(EnvState));
  EnvCold* cold = (EnvCold*)malloc(h->cold_row);
  int rc = cc4_get_state(h, env, st);
  if (rc == 0) rc = cc4_get_cold(h, env, cold);
  if (rc == 0) {
    StepWork w; memset(&w, 0, sizeof(w));
    Ctx x{st, cold, &st->rng, st->hd, &w};
    rc = state_edit(x, op, a0, a1, a2);
    if (rc < 0) { h->err = "cc4_edit_state: unknown op or bad argument"; rc = -2; }
    else { int r2 = cc4_set_state(h, env, st); if (r2 == 0) r2 = cc4_set_cold(h, env, cold); if (r2) rc = r2; }
  }
  free(st); free(cold);
  return rc;
}

int cc4_step_device(cc4_handle* h, const int32_t* d_actions, const uint8_t* d_messages) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  return launch_step(h, d_actions, d_messages, false, 0, 0, false, true);
}

// ---- group-wise stepping for a policy that lives on the GPU.  A step of a large batch is one launch per episode group, each group on its
// own stream, and the groups never wait for each other -- unless the caller's policy makes them: a policy kernel over the WHOLE batch
// between two steps is a barrier across the groups (cc4_step_device behind it: bench.py `policy_in_loop`).  A policy is batch-independent,
// though: applied per group, on the group's own stream, it keeps the groups' pipelines apart.  cc4_group_info says which episodes a group
// holds and which stream its launches run on (the caller enqueues its policy kernel for those episodes there); cc4_step_group_device
// launches that group's step behind it.  Without a communicator, event log or submitted red / green actions.
int cc4_group_info(cc4_handle* h, int32_t g, int32_t* lo, int32_t* hi, void** hip_stream) {
  if (g < 0 || g >= h->ngroups) { h->err = "cc4_group_info: no such group"; return -2; }
  if (lo) *lo = h->glo[g];
  if (hi) *hi = h->glo[g + 1];
  if (hip_stream) *hip_stream = reinterpret_cast<void*>(h->gstream[g]);
  return 0;
}
static int group_prologue(cc4_handle* h, int32_t g, const char* who) {
  if (g < 0 || g >= h->ngroups) { h->err = std::string(who) + ": no such group"; return -2; }
  if (h->comm || h->evlog_on || h->ext_seen) { h->err = std::string(who) + ": group-wise stepping serves handles without a communicator, event log or submitted red / green actions"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (h->ngroups > 1 && h->main_ahead) {     // e.g. a reset or an upload on the main stream: the group streams start behind it
    HIPCHK(h, hipEventRecord(h->mev, h->stream));
    for (int q = 1; q < h->ngroups; ++q) HIPCHK(h, hipStreamWaitEvent(h->gstream[q], h->mev, 0));
    h->main_ahead = false;
  }
  return 0;
}
int cc4_step_group_device(cc4_handle* h, int32_t g, const int32_t* d_actions, const uint8_t* d_messages) {
  if (int rc = group_prologue(h, g, "cc4_step_group_device")) return rc;
  if (h->full_obs_next) { h->full_obs_gmask = (1u << h->ngroups) - 1u; h->full_obs_next = false; }
  const bool full_obs = (h->full_obs_gmask >> g) & 1u;
  h->full_obs_gmask &= ~(1u << g);
  StepArgs a{h->d_state, h->d_cold, d_actions, d_messages, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, nullptr, 0, 0,
             h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), full_obs ? 1 : 0,
             (uint32_t)h->cfg.topology_seed, h->d_prof, h->d_reset_ws, nullptr, 0};
  launch_group(h, a, g, false, nullptr, nullptr);
  HIPCHK(h, hipGetLastError());
  if (h->ngroups > 1) h->groups_busy = true;
  return 0;
}
// the stand-in policy of bench.py for one group: uniform random action indices for the group's episodes into the handle's device action
// buffer, on the group's stream (what a policy network's kernel would do there)
int cc4_random_actions_group_device(cc4_handle* h, int32_t g, uint64_t seed0, uint32_t t) {
  if (int rc = group_prologue(h, g, "cc4_random_actions_group_device")) return rc;
  const int tot = (h->glo[g + 1] - h->glo[g]) * NBLUE;
  hipLaunchKernelGGL(k_random_actions, dim3((tot + 255) / 256), dim3(256), 0, h->gstream[g], h->d_actions, h->glo[g + 1], seed0, t, h->glo[g]);
  HIPCHK(h, hipGetLastError());
  if (h->ngroups > 1) h->groups_busy = true;
  return 0;
}

int cc4_get_obs(cc4_handle* h, int32_t* obs) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(obs, h->d_obs, (size_t)h->cfg.num_envs * OBS_TOTAL * sizeof(int32_t), hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_get_reward_done(cc4_handle* h, float* reward, uint8_t* done) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  size_t n = (size_t)h->cfg.num_envs;
  if (reward) HIPCHK(h, hipMemcpyAsync(reward, h->d_reward, n * sizeof(float), hipMemcpyDefault, h->stream));
  if (done) HIPCHK(h, hipMemcpyAsync(done, h->d_done, n, hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_get_action_mask(cc4_handle* h, uint8_t* mask) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(mask, h->d_mask, (size_t)h->cfg.num_envs * MASK_TOTAL, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_get_err(cc4_handle* h, uint32_t* err) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(err, h->d_err, (size_t)h->cfg.num_envs * sizeof(uint32_t), hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_get_rng_state(cc4_handle* h, uint64_t* out) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  int n = h->cfg.num_envs;
  hipLaunchKernelGGL(k_rng_state, dim3((n + 127) / 128), dim3(128), 0, h->stream, h->d_state, h->d_rng, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out, h->d_rng, (size_t)n * 7 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_set_seed(cc4_handle* h, const uint64_t* seeds) {
  h->prev_valid = false;        // (cc4_replay_logged would repeat a step from rows that have moved on)
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  int n = h->cfg.num_envs;
  HIPCHK(h, hipMemcpyAsync(h->d_seeds, seeds, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_set_seed, dim3((n + 127) / 128), dim3(128), 0, h->stream, h->d_state, h->d_cold, h->cold_row, h->d_seeds, n, h->cfg.rng_mode);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_set_rng_state(cc4_handle* h, const uint64_t* words) {
  h->prev_valid = false;
  if (h->cfg.rng_mode != 0) { h->err = "cc4_set_rng_state: a numpy PCG64 state needs rng_mode 0"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  int n = h->cfg.num_envs;
  HIPCHK(h, hipMemcpyAsync(h->d_rng, words, (size_t)n * 6 * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));   // d_rng holds 7 words per episode
  hipLaunchKernelGGL(k_set_rng_state, dim3((n + 127) / 128), dim3(128), 0, h->stream, h->d_state, h->d_rng, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_obs_device(cc4_handle* h, int32_t** p) { *p = h->d_obs; return 0; }
int cc4_reward_device(cc4_handle* h, float** p) { *p = h->d_reward; return 0; }
int cc4_done_device(cc4_handle* h, uint8_t** p) { *p = h->d_done; return 0; }
int cc4_actions_device(cc4_handle* h, int32_t** p) { *p = h->d_actions; return 0; }
// host copy of the handle's device action buffer: the indices cc4_step uploaded, or the ones the last step of
// cc4_run_random_steps / cc4_random_actions_device drew on the device
int cc4_get_actions(cc4_handle* h, int32_t* out) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(out, h->d_actions, (size_t)h->cfg.num_envs * NBLUE * sizeof(int32_t), hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int cc4_random_actions_device(cc4_handle* h, uint64_t seed0, uint32_t t) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  int tot = h->cfg.num_envs * NBLUE;
  hipLaunchKernelGGL(k_random_actions, dim3((tot + 255) / 256), dim3(256), 0, h->stream, h->d_actions, h->cfg.num_envs, seed0, t, 0);
  HIPCHK(h, hipGetLastError());
  return 0;
}
int cc4_synchronize(cc4_handle* h) {
  if (h->rollout_k > 0) { h->err = "cc4_synchronize: a rollout is in flight on this handle (its kernel ends when every pass is published): cc4_rollout_end"; return -1; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  return sync_all(h);
}
// The persistent run kernel (cc4_run_random_steps without a communicator, batches beyond what one launch holds): one wave per
// residency slot, the batch cut into one partition per CU.  Which CUs the device has is found once per handle (k_discover: many small
// waves reporting HW_REG_XCC_ID / HW_REG_HW_ID; the path stays off unless exactly as many CUs show up as the device properties
// promise -- a mis-decoded id would merge CUs and show here); how many waves of the run kernel a CU takes is the dispatcher's business.
// History: r04 built it with the step body as a call and measured it 18-38 % slower than four streams of per-step launches; the call
// was the brake (a kernel that contains one loses a quarter of its rate).  Inlined (lane id opaque per item) and compiled without
// machine LICM (which hoisted ~200 registers' worth of loop-invariant values across the item loop and spilled them) it is the faster
// schedule from ~20 steps per call on: 8192 episodes 795 -> 917 M at K = 500 (profiles/r04_persistent_kernel_ab.txt).
static int persist_setup(cc4_handle* h) {
  h->persist_state = -1;
  const size_t n = (size_t)h->cfg.num_envs;
  int per_cu = 0;
  HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persist_kernel(h), WAVE, offsetof(EnvState, hd)));
  // (the occupancy query divides 160 KB by the kernel's LDS bytes; the hardware allocates 1280-byte granules -- profiles/r05_lds_residency.txt)
  hipFuncAttributes fa{};
  HIPCHK(h, hipFuncGetAttributes(&fa, persist_kernel(h)));
  const int granules = (int)((offsetof(EnvState, hd) + fa.sharedSizeBytes + 1279) / 1280);
  if (granules > 0 && 128 / granules < per_cu) per_cu = 128 / granules;
  per_cu -= h->run_margin;            // (with ranks to talk to: a slot per CU stays free for RCCL's kernels)
  if (per_cu <= 0) return 0;
  // The hand-over between two items of an episode relies on what gfx942 / gfx950 do in their default (non-tgsplit) mode: the waves of a
  // CU share one write-through vector L1 (validated on MI355X in SPX mode, the only partition mode of this pool).  Any other
  // architecture keeps the per-step launches -- and says so.
  hipDeviceProp_t prop;
  HIPCHK(h, hipGetDeviceProperties(&prop, h->cfg.device_id));
  if (!(strncmp(prop.gcnArchName, "gfx942", 6) == 0 || strncmp(prop.gcnArchName, "gfx950", 6) == 0)) {
    fprintf(stderr, "[cc4] the persistent run kernel stays OFF for this handle (per-step launches instead): architecture %s is neither gfx942 nor gfx950\n", prop.gcnArchName);
    h->persist_refused = true;
    return 0;
  }
  if (join_groups(h)) return -1;
  int32_t* d_count = nullptr;
  HIPCHK(h, hipMalloc(&d_count, CC4_SLOTS * sizeof(int32_t)));
  HIPCHK(h, hipMemsetAsync(d_count, 0, CC4_SLOTS * sizeof(int32_t), h->stream));
  int khz = 100000;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id);
  hipLaunchKernelGGL(k_discover, dim3(24 * h->cus), dim3(WAVE), 0, h->stream, d_count, 100LL * (khz > 0 ? khz : 100000) / 1000);   // ~100 us each
  std::vector<int32_t> count(CC4_SLOTS);
  HIPCHK(h, hipMemcpyAsync(count.data(), d_count, CC4_SLOTS * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  (void)hipFree(d_count);
  std::vector<int32_t> table(CC4_SLOTS, 0);
  int P = 0;
  for (int sl = 0; sl < CC4_SLOTS; ++sl) if (count[sl] > 0) table[sl] = ++P;        // 1 + partition, in slot order: an XCD's CUs own neighbouring partitions
  if (getenv("CC4_PERSIST_DEBUG")) fprintf(stderr, "[cc4] persistent kernel: %d compute units seen (device: %d), %d LDS granules per wave, %d waves per CU\n", P, h->cus, granules, per_cu);
  {
    // the partition mode the hand-over was validated in: SPX -- one device, all eight XCDs, every CU of each (MI355X: 8 x 32).  Another
    // picture (CPX / DPX / QPX partitions, a part with CUs fused off differently per XCD) may well work -- an XCD's L2 is still the
    // coherence point of its CUs -- but nobody has run the self-check there: CC4_PERSIST_ANY_PARTITION=1 takes the responsibility.
    int nx = 0, per_x[8] = {0};
    for (int sl = 0; sl < 8 << 8; ++sl) if (count[sl] > 0) ++per_x[sl >> 8];
    bool even = true;
    for (int xc = 0; xc < 8; ++xc) { if (per_x[xc]) ++nx; if (per_x[xc] && per_x[xc] != per_x[0]) even = false; }
    const bool spx = nx == 8 && even && per_x[0] > 0;
    if (!spx && !(getenv("CC4_PERSIST_ANY_PARTITION") && atoi(getenv("CC4_PERSIST_ANY_PARTITION")) != 0)) {
      fprintf(stderr, "[cc4] the persistent run kernel stays OFF for this handle (per-step launches instead): the device shows %d XCD(s) with %d..CUs each -- not the SPX "
                      "picture (8 XCDs, equal CU counts) the hand-over between waves was validated in; CC4_PERSIST_ANY_PARTITION=1 overrides\n", nx, per_x[0]);
      h->persist_refused = true;
      return 0;
    }
  }
  if (P != h->cus) {       // a CU id that does not tell CUs apart would put two CUs on one partition: never run on a guess
    fprintf(stderr, "[cc4] the persistent run kernel stays OFF for this handle (per-step launches instead): its discovery pass saw %d compute units, the device has %d\n", P, h->cus);
    h->persist_refused = true;
    return 0;
  }
  h->run_P = P; h->run_G = P; h->run_grid = per_cu * h->cus;
  if (const char* v = getenv("CC4_PERSIST_THR")) h->run_thr = atoi(v);
  h->run_pool = 2;
  if (const char* v = getenv("CC4_PERSIST_RUNS")) {
    int q[4] = {h->run_SA, h->run_SB, h->run_nB, h->run_single};
    (void)sscanf(v, "%d,%d,%d,%d", &q[0], &q[1], &q[2], &q[3]);
    h->run_SA = q[0] < 0 ? 0 : q[0]; h->run_SB = q[1] < 1 ? 1 : q[1]; h->run_nB = q[2] < 0 ? 0 : q[2]; h->run_single = q[3] < 0 ? 0 : q[3];
  }
  if (h->run_pool == 2) {
    // the partitions of an XCD: a contiguous range (they are numbered in slot order, slot id = XCC id << 8 | CU)
    bool ok = true;
    for (int xc = 0; xc < 8; ++xc) {
      int lo = -1, cnt = 0;
      for (int sl = xc << 8; sl < (xc + 1) << 8; ++sl) if (table[sl] > 0) { if (lo < 0) lo = table[sl] - 1; ++cnt; }
      if (cnt > WAVE || lo > 255) ok = false;          // (one lane per partition of the XCD; the range's start travels as a byte)
      h->xcc_lo[xc] = (uint8_t)(lo < 0 ? 0 : lo); h->xcc_n[xc] = (uint8_t)(cnt > WAVE ? 0 : cnt);
    }
    for (int sl = 8 << 8; sl < CC4_SLOTS; ++sl) if (count[sl] > 0) ok = false;          // an XCC id beyond 7: not a device this schedule knows
    if (P > 510) ok = false;                            // the runner's id in the progress words: 9 bits
    if (!ok) {
      fprintf(stderr, "[cc4] the persistent run kernel stays OFF for this handle (per-step launches instead): more than 64 CUs in an XCD, or more than 510 CUs\n");
      h->persist_refused = true;
      return 0;
    } else {
      if (!h->d_pool) HIPCHK(h, hipMalloc(&h->d_pool, 2 * (size_t)CC4_SLOTS * TK_STRIDE * sizeof(uint32_t)));
      HIPCHK(h, hipMemset(h->d_pool, 0, 2 * (size_t)CC4_SLOTS * TK_STRIDE * sizeof(uint32_t)));
      h->pool_base = 0; h->pool_parity = 0;
    }
  }
  if (!h->d_slot_part) HIPCHK(h, hipMalloc(&h->d_slot_part, CC4_SLOTS * sizeof(int32_t)));
  HIPCHK(h, hipMemcpy(h->d_slot_part, table.data(), CC4_SLOTS * sizeof(int32_t), hipMemcpyHostToDevice));
  if (h->d_run) { (void)hipFree(h->d_run); h->d_run = nullptr; }
  h->run_words = 2 * (size_t)h->run_G + n;                                            // [P ticket | P owner | n progress]: one memset per call
  HIPCHK(h, hipMalloc(&h->d_run, h->run_words * sizeof(uint32_t)));
  HIPCHK(h, hipMemset(h->d_run, 0, h->run_words * sizeof(uint32_t)));
  h->persist_state = 1;
  // the sampled self-check's shadow handle is created HERE, with the path itself (the first persistent call of a handle: normally a warm-up) -- a
  // cc4_create inside the 1024th call would cost that call ~45 ms; the checks themselves then cost ~6 calls' worth each (copies and digests of the
  // cold rows), i.e. ~0.6 % of a long run.  CC4_PERSIST_VERIFY_EVERY=0: no sampling, no second copy of the rows.
  if (!h->is_shadow && !h->comm && (h->verify || h->verify_every > 0)) { if (ensure_shadow(h)) return -1; }
  return 0;
}
// ---- the exchange around a one-launch kernel (XchgArgs).  Before the launch: the call's flags cleared on the main stream, the
// communication stream ordered behind that.  After the launch: per chunk of steps, on the communication stream, wait for the chunk's last
// step to be complete (done[k] == episodes: the kernel counts an episode once its packed row is in memory), all-gather the chunk's
// slabs, publish gathered = k + 1.  After the main stream's synchronisation: the communication stream drained, the watchdog flag read.
static int xchg_begin(cc4_handle* h, int k, XchgArgs* x) {
  (void)k;
  const size_t groups = (size_t)h->cfg.num_envs / 32 + 1 > (size_t)h->cus ? (size_t)h->cfg.num_envs / 32 + 1 : (size_t)h->cus;
  {
    const size_t nb = (size_t)h->cfg.num_envs * OBS_PACKED;
    if (!h->d_xslab) HIPCHK(h, hipMalloc(&h->d_xslab, nb * cc4_handle::XRING));
    if (!h->d_xall) HIPCHK(h, hipMalloc(&h->d_xall, nb * (size_t)h->world * cc4_handle::XRING));
  }
  if (!h->d_xflags) { HIPCHK(h, hipMalloc(&h->d_xflags, 2 * sizeof(uint32_t))); h->xflags_clean = 0; }
  if (!h->d_xgcnt) { HIPCHK(h, hipMalloc(&h->d_xgcnt, groups * cc4_handle::XRING * sizeof(uint32_t))); h->xflags_clean = 0; }
  if (!h->h_xtimeout) {
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_xtimeout), sizeof(uint32_t), hipHostMallocDefault));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_xtimeout), h->h_xtimeout, 0));
  }
  *h->h_xtimeout = 0;
  if (!h->xflags_clean) {       // normally cleared behind the previous call already (xchg_end): nothing of it in front of this call's launch
    HIPCHK(h, hipMemsetAsync(h->d_xflags, 0, 2 * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_xgcnt, 0, groups * cc4_handle::XRING * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipEventRecord(h->xev, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->xev, 0));
  }
  // (clean -- the usual case: the previous call's communication stream zeroed both words behind its last publish, xchg_enqueue, and the host
  // has waited for that stream since -- nothing of this call's is ordered behind anything: no memset, no event, no cross-stream wait)
  h->xflags_clean = 0;
  if (h->khz <= 0) { int khz = 100000; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id); h->khz = khz > 0 ? khz : 100000; }
  const int khz = h->khz;
  *x = XchgArgs{h->d_xslab, h->d_xflags, h->d_xflags + 1, cc4_handle::XRING, (long long)h->xchg_watchdog_ms * (khz > 0 ? khz : 100000), h->d_xgcnt, h->d_xtimeout};
  return 0;
}
// form: 3 = the persistent kernel (groups = its partitions), else groups of 32 neighbouring episodes
static int xchg_enqueue(cc4_handle* h, int k, const XchgArgs& x, int form) {
  const size_t row = (size_t)h->cfg.num_envs * OBS_PACKED;
  const int C = h->xchg_chunk, n = h->cfg.num_envs;
  const int P = form == 3 ? h->run_G : 0, groups = form == 3 ? h->run_G : (n + 31) / 32;
  const long long gate_ticks = 30000LL * (h->khz > 0 ? h->khz : 100000);         // 30 s: a step kernel that never gets there (the host would wait for it forever anyway)
  for (int c0 = 0, hi = 0; c0 < k; c0 = hi + 1) {
    hi = (c0 + C < k ? c0 + C : k) - 1;
    if (hi == k - 1 && hi > c0) --hi;       // the call's last step is a chunk of its own: behind the kernel's end only ONE all-gather is left
    if (c0 % cc4_handle::XRING + (hi - c0) >= cc4_handle::XRING) hi = c0 + cc4_handle::XRING - 1 - c0 % cc4_handle::XRING;     // a chunk's slabs are neighbours in the ring
    hipLaunchKernelGGL(k_xchg_gate, dim3(1), dim3(WAVE), 0, h->comm_stream, x.gcnt, x.ring, groups, n, P, c0, hi, gate_ticks, x.timeout_host, hi == k - 1 ? 1 : 8);
    HIPCHK(h, hipGetLastError());
    if (h->comm_delay_ticks > 0) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, h->comm_stream, h->comm_delay_ticks); HIPCHK(h, hipGetLastError()); }
    // ONE all-gather for the chunk's m neighbouring slabs (an ncclAllGather costs the host ~10 us to enqueue, grouped or not: eight of them
    // per chunk were as much as the eight steps of a 1024-episode batch last).  The gathered block of a chunk is rank-major: rank r's rows of
    // the chunk's step j at ((r * m + j - c0) * N) -- for a chunk of one step, the call's last among them, plain [world * N] rows.
    const int s0 = c0 % cc4_handle::XRING, m = hi - c0 + 1;
    uint8_t* const block = h->d_xall + (size_t)s0 * row * (size_t)h->world;
    ncclResult_t r = ncclAllGather(h->d_xslab + (size_t)s0 * row, block, (size_t)m * row, ncclUint8, h->comm, h->comm_stream);
    if (r != ncclSuccess) { h->err = std::string("ncclAllGather: ") + ncclGetErrorString(r); return -1; }
    if (h->d_xlog) for (int j = c0; j <= hi && h->xlog_n < h->xlog_cap; ++j, ++h->xlog_n)     // debug: keep every step's gathered rows as [world * N] (cc4_debug_gather_log)
      for (int rk = 0; rk < h->world; ++rk)
        HIPCHK(h, hipMemcpyAsync(h->d_xlog + ((size_t)h->xlog_n * h->world + rk) * row, block + ((size_t)rk * m + (size_t)(j - c0)) * row, row, hipMemcpyDeviceToDevice, h->comm_stream));
    HIPCHK(h, hipStreamWriteValue32(h->comm_stream, x.gathered, (uint32_t)(hi + 1), 0));
  }
  // behind the call's last publish (every episode has counted its last step: nobody reads the two words any more) the communication stream
  // itself hands them back zeroed for the next call -- the host waits for this stream in xchg_end, so the next launch finds them clean
  HIPCHK(h, hipStreamWriteValue32(h->comm_stream, x.gathered, 0u, 0));
  HIPCHK(h, hipStreamWriteValue32(h->comm_stream, x.timeout, 0u, 0));
  h->gathers_issued += k;
  return 0;
}
static int xchg_end(cc4_handle* h, int k) {
  const size_t row = (size_t)h->cfg.num_envs * OBS_PACKED;
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  h->gathers_waited = h->gathers_issued;
  const uint32_t flag = *reinterpret_cast<volatile uint32_t*>(h->h_xtimeout);      // (both streams are drained: the kernel's system-scope store has landed)
  h->xchg_calls++;
  const int last = (k - 1) % cc4_handle::XRING;
  h->last_gathered = h->d_xall + last * row * (size_t)h->world;      // (the call's last step is a chunk of its own: plain [world * N] rows)
  h->gather_buf = -2;                               // (not one of the per-step ring's buffers: last_gathered says where)
  // the per-step path's current buffer is to hold the observations of the last step as well -- filled when an explicit cc4_allgather_obs asks
  // for it (a per-step launch or a reset that follows writes a buffer of its own)
  h->obs8_from_slab = last;
  h->step_event_attached = false;
  if (flag) {   // a watchdog fired: some counts may never have been collected -- everything cleared the long way before the next call
    h->xflags_clean = 0;
  } else h->xflags_clean = 1;   // (both words zeroed by the communication stream behind its last publish, the group counters by the gates)
  if (flag) {
    // an item waited longer than the watchdog for its slab: the exchange did not keep up at all (e.g. its kernels found no room beside the
    // one-launch kernel).  The episodes are intact -- a wait that gives up only stops protecting slabs, so gathers of this call may have
    // carried a later step's rows -- and the handle goes back to per-step launches, loudly.
    h->xchg_timeouts++;
    h->xchg_on = false;
    // what this call gathered is not published as valid: the gather log forgets the call's steps, and the observations of the call's last
    // step -- whose slab nothing overwrote -- are gathered again through the per-step path when somebody asks (obs8_from_slab stays)
    h->last_gathered = nullptr; h->gather_buf = -1;
    if (h->d_xlog) h->xlog_n = h->xlog_n >= k ? h->xlog_n - k : 0;
    h->err = "the in-kernel exchange timed out in the last cc4_run_random_steps call (episodes intact; its all-gathers are void; per-step launches from now on)";
    (void)hipFree(h->d_xall); h->d_xall = nullptr;      // (the gathered twin of the ring: world times the ring; the ring itself still holds the last step's rows)
    fprintf(stderr, "[cc4] the in-kernel exchange timed out (a step waited > %d ms for the all-gather of %d steps earlier): this handle returns to per-step launches with the exchange\n",
            h->xchg_watchdog_ms, cc4_handle::XRING);
  }
  return 0;
}
static int run_random_steps_impl(cc4_handle* h, uint64_t seed0, uint32_t t0, int32_t k, float* ms_step_kernels);
// CC4_PERSIST_VERIFY=1 (a self-check mode, not a fast one): a call that takes a one-launch form -- the persistent kernels, whose hand-over
// between the steps of an episode leans on how a CU's L1 behaves, and the plain multi-step kernels -- is run a second time
// from the same starting rows with per-step launches on a shadow handle, and the two outcomes are compared episode by episode.
static int verify_digest(cc4_handle* h, std::vector<uint64_t>& out) {
  const int n = h->cfg.num_envs;
  if (!h->d_digest) HIPCHK(h, hipMalloc(&h->d_digest, 3 * (size_t)n * sizeof(uint64_t)));
  if (join_groups(h)) return -1;
  hipLaunchKernelGGL(k_digest, dim3(n), dim3(WAVE), 0, h->stream, h->d_state, h->d_cold, h->cold_row, h->d_obs, h->d_reward, h->d_done, h->d_err, h->d_actions, h->d_digest, n);
  HIPCHK(h, hipGetLastError());
  out.resize(3 * (size_t)n);
  HIPCHK(h, hipMemcpyAsync(out.data(), h->d_digest, out.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
// the shadow handle of the self-check: a second copy of the batch's rows, stepped with per-step launches only
static int ensure_shadow(cc4_handle* h) {
  if (h->shadow) return 0;
  cc4_handle* sh = nullptr;
  if (cc4_create(&h->cfg, &sh) != 0) { h->err = std::string("CC4_PERSIST_VERIFY: the shadow handle could not be created: ") + cc4_last_error(sh); if (sh) cc4_destroy(sh); return -1; }
  sh->is_shadow = true; sh->verify = false; sh->verify_every = 0; sh->persist_state = -1; sh->multistep = false; sh->run1m = false;
  h->shadow = sh;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  return 0;
}
int cc4_run_random_steps(cc4_handle* h, uint64_t seed0, uint32_t t0, int32_t k, float* ms_step_kernels) {
  if (h->is_shadow || k < 2) return run_random_steps_impl(h, seed0, t0, k, ms_step_kernels);
  // The self-check: with CC4_PERSIST_VERIFY=1 every one-launch call is repeated on a shadow handle and compared; WITHOUT it every
  // verify_every-th call that takes the PERSISTENT form is (CC4_PERSIST_VERIFY_EVERY, default 1024, 0 = never; the communicator-less handles
  // only: a shadow handle cannot join the exchange) -- the hand-over between the waves of a CU rests on behaviour the memory model does not
  // promise, so the path keeps checking itself in production at < 1 % of its time (a checked call costs ~10 x a plain one; the shadow
  // handle -- a second copy of the batch's rows -- is allocated by the first checked call).
  bool check = h->verify;
  if (!check && h->verify_every > 0 && !h->comm && h->persist_state >= 0 && k >= h->persist_min_k && !h->run1m && !h->multistep) {
    if (++h->persist_calls % (uint64_t)h->verify_every == 0) check = true;
  }
  if (!check) return run_random_steps_impl(h, seed0, t0, k, ms_step_kernels);
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (strncmp(cc4_run_kernel_for(h, k), "k_run_", 6) != 0) return run_random_steps_impl(h, seed0, t0, k, ms_step_kernels);
  if (ensure_shadow(h)) return -1;
  cc4_handle* sh = h->shadow;
  const size_t n = (size_t)h->cfg.num_envs;
  if (join_groups(h) || join_groups(sh)) return -1;
  HIPCHK(h, hipStreamSynchronize(sh->stream));
  HIPCHK(h, hipMemcpyAsync(sh->d_state, h->d_state, n * sizeof(EnvState), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(sh->d_cold, h->d_cold, n * h->cold_row, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(sh->d_obs, h->d_obs, h->out_bytes, hipMemcpyDefault, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  sh->full_obs_next = h->full_obs_next; sh->main_ahead = sh->ngroups > 1;
  int rc = run_random_steps_impl(h, seed0, t0, k, ms_step_kernels);
  if (rc) return rc;
  rc = run_random_steps_impl(sh, seed0, t0, k, nullptr);
  if (rc) { h->err = "CC4_PERSIST_VERIFY: the shadow run failed: " + sh->err; return rc; }
  std::vector<uint64_t> a, b;
  if (verify_digest(h, a)) return -1;
  if (verify_digest(sh, b)) { h->err = "CC4_PERSIST_VERIFY: " + sh->err; return -1; }
  h->verify_calls++;
  for (size_t e = 0; e < n; ++e) {
    const bool hot = a[3 * e] != b[3 * e], cold = a[3 * e + 1] != b[3 * e + 1], outp = a[3 * e + 2] != b[3 * e + 2];
    if (hot || cold || outp) {
      h->verify_mismatches++;
      h->err = "CC4_PERSIST_VERIFY: " + std::string(cc4_run_kernel_for(h, k)) + " and the per-step launches disagree after " + std::to_string(k) + " steps: first episode " +
               std::to_string(e) + " (" + (hot ? "hot row " : "") + (cold ? "cold row " : "") + (outp ? "outputs" : "") + ")";
      fprintf(stderr, "[cc4] %s\n", h->err.c_str());
      return -5;
    }
  }
  return 0;
}
// out[0] calls checked, out[1] calls that disagreed (CC4_PERSIST_VERIFY)
int cc4_verify_stats(cc4_handle* h, int64_t* out /* [2] */) { out[0] = h->verify_calls; out[1] = h->verify_mismatches; return 0; }
// One launch of the persistent kernel for k steps of the whole batch (cc4_run_random_steps form 3; cc4_rollout_begin with rollout = true: every step
// an item of its own, the actions from the rollout's slots behind the caller's publishes).
static int persist_launch(cc4_handle* h, StepArgs a, int k, uint32_t t0, const XchgArgs& x, hipEvent_t e0, hipEvent_t e1, bool rollout) {
  if (h->pool_base + (uint32_t)k > 0x700000u) {      // (the progress words count steps since they were last cleared)
    HIPCHK(h, hipMemsetAsync(h->d_run, 0, h->run_words * sizeof(uint32_t), h->stream));
    h->pool_base = 0;
  }
  unsigned long long* d_tl = nullptr;
  if (getenv("CC4_PERSIST_TIMELINE")) { HIPCHK(h, hipMalloc(&d_tl, 4 * sizeof(unsigned long long) * (size_t)h->run_grid)); HIPCHK(h, hipMemsetAsync(d_tl, 0, 4 * sizeof(unsigned long long) * (size_t)h->run_grid, h->stream)); }
  h->d_timeline = d_tl;
  RunArgs ra{h->d_run, h->d_run + 2 * h->run_G, reinterpret_cast<int32_t*>(h->d_run + h->run_G), h->d_slot_part, h->run_P, k, h->run_G, t0, d_tl, h->persist_order,
             1, k, 1, 0, k, 0, 0u, nullptr, {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF}, {0}, {0}, h->run_thr};
  const int SA = rollout ? 1 : (h->run_SA > 0 ? h->run_SA : (k >= 64 ? 8 : 4));
  if (SA > 1) {
    // runs of steps: nB runs of SB and tail_single single steps close the call, runs of SA fill the rest (what is left over goes to the single steps)
    int single = h->run_single < k ? h->run_single : k;
    int nB = h->run_SB > 1 ? h->run_nB : 0;
    while (nB > 0 && single + nB * h->run_SB > k) --nB;
    const int nA = (k - single - nB * h->run_SB) / SA;
    single = k - nA * SA - nB * h->run_SB;
    ra.SA = SA; ra.nA = nA; ra.SB = h->run_SB > 1 ? h->run_SB : 1; ra.nB = nB; ra.nph = nA + nB + single;
  }
  if (h->run_pool) {
    ra.pool = h->run_pool; ra.base = h->pool_base;
    ra.ticket = h->d_pool + (size_t)h->pool_parity * CC4_SLOTS * TK_STRIDE;
    ra.ticket_next = h->d_pool + (size_t)(h->pool_parity ^ 1) * CC4_SLOTS * TK_STRIDE;
    memcpy(ra.xcc_pool, h->xcc_pool, 8); memcpy(ra.xcc_lo, h->xcc_lo, 8); memcpy(ra.xcc_n, h->xcc_n, 8);
    h->pool_parity ^= 1; h->pool_base += (uint32_t)k;
  }
  if (rollout) {
    ra.act_ready = h->d_rready; ra.act = h->d_ract; ra.PG = h->rpg;
    ra.act_wait_ticks = (long long)h->rollout_watchdog_ms * (h->khz > 0 ? h->khz : 100000);
  }
  if (h->cfg.rng_mode == 0) hipExtLaunchKernelGGL(k_run_pcg, dim3(h->run_grid), dim3(WAVE), offsetof(EnvState, hd), h->stream, e0, e1, 0, a, ra, x);
  else
  if (h->comm) hipExtLaunchKernelGGL(k_run_philox1x, dim3(h->run_grid), dim3(WAVE), offsetof(EnvState, hd), h->stream, e0, e1, 0, a, ra, x);
  else {
    // a rollout leaves `rollout_margin` waves per CU to the caller's policy kernels and the gates (CC4_ROLLOUT_MARGIN)
    const int grid = rollout ? h->run_grid - h->rollout_margin * h->cus : h->run_grid;
    if (rollout) hipExtLaunchKernelGGL(k_run_philox1r, dim3(grid > h->cus ? grid : h->cus), dim3(WAVE), offsetof(EnvState, hd), h->stream, e0, e1, 0, a, ra, x);
    else hipExtLaunchKernelGGL(k_run_philox1, dim3(grid), dim3(WAVE), offsetof(EnvState, hd), h->stream, e0, e1, 0, a, ra, x);
  }
  return 0;
}
static int run_random_steps_impl(cc4_handle* h, uint64_t seed0, uint32_t t0, int32_t k, float* ms_step_kernels) {
  h->prev_valid = false;        // (every form of this call moves the rows without refreshing the kept copy of cc4_keep_previous)
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (k <= 0) { if (ms_step_kernels) *ms_step_kernels = 0.f; return 0; }   // nothing to launch, no timing event to read
  const bool plain = (!h->comm || h->xchg_on) && !h->evlog_on && !h->ext_seen && !h->d_prof && !h->dbg_stop;
  if (plain && h->persist_state == 0 && !h->run1m && !h->multistep && k >= h->persist_min_k) { if (persist_setup(h)) return -1; }
  const int form = !plain ? 0 : (h->multistep && k >= 2) ? 1 : (h->run1m && k >= 2) ? 2 : (h->persist_state == 1 && h->run_P > 0 && k >= h->persist_min_k) ? 3 : 0;
  if (form) {
    // ONE launch for the k steps: 1 = every block loops over the steps of its episode (k_run_philox / k_run_philox8), 2 = the same on one wave
    // per episode (k_run_philox1m), 3 = the persistent form (k_run_philox1 / k_run_pcg: one wave per residency slot pulling (episode, step) items)
    if (join_groups(h)) return -1;
    StepArgs a{h->d_state, h->d_cold, nullptr, nullptr, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, h->d_actions, seed0, t0,
               h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
               (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0),
               h->full_obs_next ? 1 : 0, (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, nullptr, 0};
    XchgArgs x{};
    const bool exchange = h->comm != nullptr;
    static const bool xprof = getenv("CC4_EXCHANGE_PROF") != nullptr;      // debug: where the host's time goes around a one-launch call with the exchange
    static double xp[6] = {0}; static long xpn = 0;
    const auto xp0 = std::chrono::steady_clock::now();
    if (exchange && xchg_begin(h, k, &x)) return -1;
    const auto xp1 = std::chrono::steady_clock::now();
    if (ms_step_kernels && h->evs.size() < 2) { h->evs.resize(2, nullptr); for (auto& e : h->evs) if (!e) HIPCHK(h, hipEventCreate(&e)); }
    hipEvent_t e0 = ms_step_kernels ? h->evs[0] : nullptr, e1 = ms_step_kernels ? h->evs[1] : nullptr;
    auto c0 = std::chrono::steady_clock::now();
    if (form == 1) {
      if (h->multistep_minb == 8) hipExtLaunchKernelGGL(k_run_philox8, dim3(h->cfg.num_envs), dim3(PT), sizeof(EnvState), h->stream, e0, e1, 0, a, (int)k, t0, x);
      else hipExtLaunchKernelGGL(k_run_philox, dim3(h->cfg.num_envs), dim3(PT), sizeof(EnvState), h->stream, e0, e1, 0, a, (int)k, t0, x);
    } else if (form == 2) {
      hipExtLaunchKernelGGL(k_run_philox1m, dim3(h->cfg.num_envs), dim3(WAVE), offsetof(EnvState, hd), h->stream, e0, e1, 0, a, (int)k, t0, x);
    } else {
      if (persist_launch(h, a, k, t0, x, e0, e1, false)) return -1;
    }
    HIPCHK(h, hipGetLastError());
    h->stat_launch_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
    h->stat_steps += k;
    h->full_obs_next = false;     // (asked for, the first step of every episode rewrote all its observation values)
    h->main_ahead = h->ngroups > 1;
    const auto xp2 = std::chrono::steady_clock::now();
    if (exchange) {
      auto g0 = std::chrono::steady_clock::now();
      if (xchg_enqueue(h, k, x, form)) return -1;
      h->stat_gather_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g0).count();
    }
    const auto xp3 = std::chrono::steady_clock::now();
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const auto xp4 = std::chrono::steady_clock::now();
    if (h->d_timeline) {      // debug: where a call's time goes between the kernel's entry and its last item (ticks of the 100 MHz wall clock)
      std::vector<unsigned long long> tl(4 * (size_t)h->run_grid);
      HIPCHK(h, hipMemcpy(tl.data(), h->d_timeline, tl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      (void)hipFree(h->d_timeline); h->d_timeline = nullptr;
      unsigned long long e0 = ~0ull, e1 = 0, f1 = 0, l0 = ~0ull, l1 = 0; double fs = 0, ls = 0, items = 0; int nw = 0, idle = 0;
      for (int w = 0; w < h->run_grid; ++w) { const unsigned long long* t = &tl[4 * (size_t)w]; if (!t[0]) continue; ++nw; e0 = t[0] < e0 ? t[0] : e0; e1 = t[0] > e1 ? t[0] : e1; if (!(uint32_t)t[3]) { ++idle; continue; } f1 = t[1] > f1 ? t[1] : f1; l0 = t[2] < l0 ? t[2] : l0; l1 = t[2] > l1 ? t[2] : l1; fs += (double)t[1]; ls += (double)t[2]; items += (double)(uint32_t)t[3]; }
      const int busy = nw - idle;
      float ms = 0.f; if (ms_step_kernels) (void)hipEventElapsedTime(&ms, h->evs[0], h->evs[1]);
      fprintf(stderr, "[cc4 timeline] k=%d: %d waves reported (%d without an item); entry spread %.1f us; first item starts: mean +%.1f us, last +%.1f us after the first entry; "
                      "last item ends: earliest +%.1f us, mean +%.1f us, latest +%.1f us; items per busy wave %.1f; kernel (events) %.1f us\n",
              k, nw, idle, (e1 - e0) / 100.0, busy ? (fs / busy - (double)e0) / 100.0 : 0.0, (f1 - e0) / 100.0, (l0 - e0) / 100.0, busy ? (ls / busy - (double)e0) / 100.0 : 0.0, (l1 - e0) / 100.0, busy ? items / busy : 0.0, ms * 1000.0);
      // per CU: when its LAST wave ran dry, and how many items its waves executed (more than its own partition's = it helped out)
      { std::map<int, std::pair<unsigned long long, double>> cu;
        for (int w = 0; w < h->run_grid; ++w) { const unsigned long long* t = &tl[4 * (size_t)w]; if (!t[0] || !(uint32_t)t[3]) continue; auto& c = cu[(int)((t[3] >> 32) - 1)]; if (t[2] > c.first) c.first = t[2]; c.second += (double)(uint32_t)t[3]; }
        std::vector<double> last, its; for (auto& kv : cu) { last.push_back((kv.second.first - e0) / 100.0); its.push_back(kv.second.second); }
        std::sort(last.begin(), last.end()); std::sort(its.begin(), its.end());
        if (!last.empty()) { const size_t m = last.size(); fprintf(stderr, "[cc4 timeline]   per CU (%zu): last wave dry at min %.1f / 10%% %.1f / median %.1f / 90%% %.1f / max %.1f us; items executed min %.0f / median %.0f / max %.0f\n", m,
                                   last[0], last[m / 10], last[m / 2], last[m * 9 / 10], last[m - 1], its[0], its[m / 2], its[m - 1]); } }
      // per XCD: when its waves ran dry (intra-XCD sharing evens a tail out inside an XCD; what is left between XCDs is not shareable)
      { double xs[8] = {0}, xi[8] = {0}; unsigned long long xl[8] = {0}, xf[8]; int xn[8] = {0}; for (int i = 0; i < 8; ++i) xf[i] = ~0ull;
        for (int w = 0; w < h->run_grid; ++w) { const unsigned long long* t = &tl[4 * (size_t)w]; if (!t[0] || !(uint32_t)t[3]) continue; const int xc = (int)(((t[3] >> 32) - 1) >> 8) & 7;
          xs[xc] += (double)t[2]; xi[xc] += (double)(uint32_t)t[3]; ++xn[xc]; if (t[2] > xl[xc]) xl[xc] = t[2]; if (t[2] < xf[xc]) xf[xc] = t[2]; }
        for (int i = 0; i < 8; ++i) if (xn[i]) fprintf(stderr, "[cc4 timeline]   XCD %d: %d busy waves, %.0f items; waves ran dry: earliest +%.1f, mean +%.1f, latest +%.1f us\n", i, xn[i], xi[i],
                                                      (xf[i] - e0) / 100.0, (xs[i] / xn[i] - (double)e0) / 100.0, (xl[i] - e0) / 100.0); }
    }
    if (exchange && xchg_end(h, k)) return -1;
    if (xprof && exchange) {
      const auto xp5 = std::chrono::steady_clock::now();
      auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      xp[0] += us(xp0, xp1); xp[1] += us(xp1, xp2); xp[2] += us(xp2, xp3); xp[3] += us(xp3, xp4); xp[4] += us(xp4, xp5); xp[5] += us(xp0, xp5);
      if (++xpn % 200 == 0) {
        fprintf(stderr, "[cc4 exchange prof] k=%d, mean of 200 calls (us): begin %.1f, launch %.1f, enqueue of the chunks %.1f, wait for the kernel %.1f, then for the communication stream %.1f; call %.1f\n",
                k, xp[0] / 200, xp[1] / 200, xp[2] / 200, xp[3] / 200, xp[4] / 200, xp[5] / 200);
        for (double& v : xp) v = 0;
      }
    }
    if (ms_step_kernels) HIPCHK(h, hipEventElapsedTime(ms_step_kernels, h->evs[0], h->evs[1]));
    return 0;
  }
  if (h->enq_threads && !h->pool && h->ngroups > 1 && !h->comm) enq_pool_start(h);     // on first use: most handles never come here
  if (h->pool && (int)h->pool->th.size() == h->ngroups - 1 && !h->comm && !h->evlog_on && !h->ext_seen && !h->d_prof && k >= 1) {
    // every group's k launches from its own thread (EnqPool); this thread takes group 0
    const int G = h->ngroups;
    EnqPool* P = h->pool;
    if (ms_step_kernels && (int)h->evs.size() < 2 * G) {
      size_t old = h->evs.size();
      h->evs.resize(2 * (size_t)G, nullptr);
      for (size_t i = old; i < h->evs.size(); ++i) HIPCHK(h, hipEventCreate(&h->evs[i]));
    }
    if (h->main_ahead) {
      HIPCHK(h, hipEventRecord(h->mev, h->stream));
      for (int g = 1; g < G; ++g) HIPCHK(h, hipStreamWaitEvent(h->gstream[g], h->mev, 0));
      h->main_ahead = false;
    }
    P->a = StepArgs{h->d_state, h->d_cold, nullptr, nullptr, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, h->d_actions, seed0, t0,
                    h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
                    (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0),
                    0, (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, nullptr, 0};
    P->k = k; P->t0 = t0; P->full = false; P->first_full_obs = h->full_obs_next; P->join = !getenv("CC4_ENQ_NOJOIN");
    for (int g = 0; g < G; ++g) { P->start[g] = ms_step_kernels ? h->evs[2 * g] : nullptr; P->stop[g] = ms_step_kernels ? h->evs[2 * g + 1] : nullptr; }
    P->failed.store(0);
    P->pending.store(G - 1, std::memory_order_relaxed);
    auto c0 = std::chrono::steady_clock::now();
    { std::lock_guard<std::mutex> lk(P->mu); P->gen.fetch_add(1, std::memory_order_release); }
    P->cv.notify_all();
    enq_run_group(h, P, 0);
    while (P->pending.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    h->stat_launch_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
    h->stat_steps += k;
    h->full_obs_next = false;
    h->groups_busy = true;
    if (P->failed.load()) { h->err = "cc4_run_random_steps: a step launch failed"; return -1; }
    if (P->join) {
      for (int g = 1; g < G; ++g) HIPCHK(h, hipStreamWaitEvent(h->stream, h->gev[g], 0));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      h->groups_busy = false;
    } else if (sync_all(h)) return -1;
    if (ms_step_kernels) {
      float worst = 0.f;
      for (int g = 0; g < G; ++g) { float ms = 0.f; HIPCHK(h, hipEventElapsedTime(&ms, h->evs[2 * g], h->evs[2 * g + 1])); if (ms > worst) worst = ms; }
      *ms_step_kernels = worst;
    }
    return 0;
  }
  // Timing: HIP events on the launch streams around chunks of TIMED_CHUNK consecutive steps (an event pair around every single
  // launch costs the stream ~5 us of idle time per step); per stream, the sum over the chunks is the on-stream time of its k
  // launches, read back after the loop -- no host synchronisation inside the timed region.  With several episode groups
  // (cc4_handle::ngroups) every group's stream is timed; the slowest stream is reported: the on-stream time of the k STEPS.
  constexpr int TIMED_CHUNK = 25;
  const int G = h->ngroups;
  const int nchunks = ms_step_kernels ? (k + TIMED_CHUNK - 1) / TIMED_CHUNK : 0;
  if ((int)h->evs.size() < 2 * nchunks * G) {
    size_t old = h->evs.size();
    h->evs.resize(2 * (size_t)nchunks * G, nullptr);
    for (size_t i = old; i < h->evs.size(); ++i) HIPCHK(h, hipEventCreate(&h->evs[i]));
  }
  auto ev = [&](int chunk, int g, int which) { return h->evs[(size_t)(2 * (chunk * G + g) + which)]; };
  const bool hp = getenv("CC4_HOST_PROF") != nullptr;
  double t_launch = 0, t_ag = 0, t_first = 0;
  const long long stalls0 = h->gather_stalls;
  // Without a communicator the two timing events of a stream ride on its first and its last launch of the call (start / stop
  // event of hipExtLaunchKernelGGL: the kernels' own start and completion timestamps) -- marker packets from hipEventRecord cost
  // the streams 0.5 us per step at k = 500 and 1.2 us per step at k = 20 (tools/short_region_probe.py).  CC4_TIMING_MARKERS=1
  // keeps the marker form; with a communicator the launches' stop events belong to the exchange and the markers stay.
  const bool attach = ms_step_kernels && !h->comm && !getenv("CC4_TIMING_MARKERS");
  for (int i = 0; i < k; ++i) {
    if (attach) {
      if (i == 0) for (int g = 0; g < G; ++g) h->tev_start[g] = ev(0, g, 0);
      if (i == k - 1) for (int g = 0; g < G; ++g) h->tev_stop[g] = ev(0, g, 1);
    }
    if (ms_step_kernels && !attach && i % TIMED_CHUNK == 0) {
      if (G > 1 && h->main_ahead) {     // the group streams' first event must not be recorded ahead of what their first launch waits for
        HIPCHK(h, hipEventRecord(h->mev, h->stream));
        for (int g = 1; g < G; ++g) HIPCHK(h, hipStreamWaitEvent(h->gstream[g], h->mev, 0));
        h->main_ahead = false;
      }
      for (int g = 0; g < G; ++g) HIPCHK(h, hipEventRecord(ev(i / TIMED_CHUNK, g, 0), h->gstream[g]));
    }
    auto c0 = std::chrono::steady_clock::now();
    if (launch_step(h, nullptr, nullptr, true, seed0, t0 + (uint32_t)i)) return -1;   // actions drawn in-kernel
    auto c1 = std::chrono::steady_clock::now();
    if (ms_step_kernels && !attach && (i % TIMED_CHUNK == TIMED_CHUNK - 1 || i == k - 1))
      for (int g = 0; g < G; ++g) HIPCHK(h, hipEventRecord(ev(i / TIMED_CHUNK, g, 1), h->gstream[g]));
    auto c2 = std::chrono::steady_clock::now();
    if (h->comm) { if (cc4_allgather_obs(h, nullptr)) return -1; }                       // overlaps the next step
    auto c3 = std::chrono::steady_clock::now();
    t_launch += std::chrono::duration<double, std::micro>(c1 - c0).count();
    if (i == 0) t_first = std::chrono::duration<double, std::micro>(c1 - c0).count();
    t_ag += std::chrono::duration<double, std::micro>(c3 - c2).count();
  }
  h->stat_steps += k; h->stat_launch_us += t_launch; h->stat_gather_us += t_ag;
  if (hp) fprintf(stderr, "[cc4 host prof] k=%d launch_step %.2f us/step (%d launches per step), allgather enqueue %.2f us/step, %lld buffer-reuse stalls; the call's first launch_step %.1f us\n", k, t_launch / k, G, t_ag / k, h->gather_stalls - stalls0, t_first);
  auto p0 = std::chrono::steady_clock::now();
  if (h->comm) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  if (sync_all(h)) return -1;
  auto p1 = std::chrono::steady_clock::now();
  if (hp) {
    auto q0 = std::chrono::steady_clock::now();
    if (sync_all(h)) return -1;
    auto q1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[cc4 host prof] enqueue loop done -> all streams synchronised: %.1f us; a second sync_all on idle streams: %.1f us\n",
            std::chrono::duration<double, std::micro>(p1 - p0).count(), std::chrono::duration<double, std::micro>(q1 - q0).count());
  }
  if (ms_step_kernels) {
    float worst = 0.f;
    for (int g = 0; g < G; ++g) {
      float total = 0.f;
      for (int c = 0; c < (attach ? 1 : nchunks); ++c) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, ev(c, g, 0), ev(c, g, 1)));
        total += ms;
      }
      if (total > worst) worst = total;
    }
    *ms_step_kernels = worst;
    if (hp) fprintf(stderr, "[cc4 host prof] reading the timing events: %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - p1).count());
  }
  return 0;
}
// ---- rollouts with the policy in the loop (include/cc4.h).  ONE launch of the persistent kernel per k-step rollout; the caller's policy
// runs between the steps on the caller's stream, one policy group of episodes at a time, ordered against the stepping through device words only.
static int rollout_ready(cc4_handle* h, const char* who) {
  if (h->rollout_k <= 0) { h->err = std::string(who) + ": no rollout is in flight (cc4_rollout_begin)"; return -2; }
  return 0;
}
int cc4_rollout_begin(cc4_handle* h, int32_t k) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (h->rollout_k > 0) { h->err = "cc4_rollout_begin: a rollout is in flight (cc4_rollout_end)"; return -2; }
  if (k <= 0 || k > 0x100000) { h->err = "cc4_rollout_begin: 1 .. 2^20 steps"; return -2; }
  if (h->cfg.rng_mode != 1 || h->comm || h->evlog_on || h->ext_seen || h->d_prof) { h->err = "cc4_rollout_begin: for counter-mode handles without a communicator, event log or submitted red / green actions"; return -2; }
  if (h->persist_state == 0) { if (persist_setup(h)) return -1; }
  if (h->persist_state != 1 || h->run_pool != 2) { h->err = "cc4_rollout_begin: this handle has no persistent kernel (a batch the chip holds at once, or a device picture the schedule refuses): step it with cc4_step_device"; return -2; }
  if (join_groups(h)) return -1;
  h->prev_valid = false;
  const size_t n = (size_t)h->cfg.num_envs, row = n * OBS_PACKED;
  if (!h->d_ract) {
    HIPCHK(h, hipMalloc(&h->d_ract, 2 * n * NBLUE * sizeof(int32_t)));
    HIPCHK(h, hipMalloc(&h->d_rready, (size_t)CC4_SLOTS * 32 * sizeof(uint32_t)));
    HIPCHK(h, hipMalloc(&h->d_rcnt, (size_t)h->run_P * RPG_MAX * cc4_handle::XRING * sizeof(uint32_t)));
    HIPCHK(h, hipMalloc(&h->d_rfail, sizeof(uint32_t)));
    for (int g = 0; g < RPG_MAX; ++g) HIPCHK(h, hipStreamCreateWithFlags(&h->gpolicy[g], hipStreamNonBlocking));
    h->policy_stream = h->gpolicy[0];
    HIPCHK(h, hipEventCreateWithFlags(&h->rev, hipEventDisableTiming));
    if (const char* v = getenv("CC4_ROLLOUT_WATCHDOG_MS")) h->rollout_watchdog_ms = atoi(v) > 0 ? atoi(v) : 2000;
    if (const char* v = getenv("CC4_ROLLOUT_MARGIN")) h->rollout_margin = atoi(v) >= 0 ? atoi(v) : 1;
    if (const char* v = getenv("CC4_ROLLOUT_GROUPS")) { h->rpg = atoi(v); if (h->rpg < 1) h->rpg = 1; if (h->rpg > RPG_MAX) h->rpg = RPG_MAX; }
  }
  if (!h->d_xslab) HIPCHK(h, hipMalloc(&h->d_xslab, row * cc4_handle::XRING));
  if (!h->d_xflags) { HIPCHK(h, hipMalloc(&h->d_xflags, 2 * sizeof(uint32_t))); }
  if (!h->h_xtimeout) {
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_xtimeout), sizeof(uint32_t), hipHostMallocDefault));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_xtimeout), h->h_xtimeout, 0));
  }
  if (h->khz <= 0) { int khz = 100000; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id); h->khz = khz > 0 ? khz : 100000; }
  *h->h_xtimeout = 0;
  HIPCHK(h, hipMemsetAsync(h->d_xflags, 0, 2 * sizeof(uint32_t), h->stream));
  // (debug, CC4_ROLLOUT_PREPUBLISH=1: every pass counts as published from the start -- what the stepping itself costs in a rollout, without the waits)
  HIPCHK(h, hipMemsetAsync(h->d_rready, getenv("CC4_ROLLOUT_PREPUBLISH") ? 0x7F : 0, (size_t)h->run_P * 32 * sizeof(uint32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_rcnt, 0, (size_t)h->run_P * RPG_MAX * cc4_handle::XRING * sizeof(uint32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_rfail, 0, sizeof(uint32_t), h->stream));
  // what the first policy pass reads: the observations as they stand, packed into the slab in front of step 0's
  hipLaunchKernelGGL(k_pack_obs_rows, dim3((unsigned)n), dim3(WAVE), 0, h->stream, h->d_xslab + (size_t)(cc4_handle::XRING - 1) * row, h->d_obs, (int)n);
  HIPCHK(h, hipEventRecord(h->rev, h->stream));
  StepArgs a{h->d_state, h->d_cold, nullptr, nullptr, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, nullptr, 0, 0,
             h->cfg.num_envs, h->cfg.autoreset, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0),
             h->full_obs_next ? 1 : 0, (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, nullptr, 0};
  XchgArgs x{h->d_xslab, nullptr, h->d_xflags + 1, cc4_handle::XRING, 0, h->d_rcnt, h->d_xtimeout};
  if (persist_launch(h, a, k, 0u, x, nullptr, nullptr, true)) return -1;
  HIPCHK(h, hipGetLastError());
  h->stat_steps += k;
  h->full_obs_next = false;
  h->main_ahead = h->ngroups > 1;
  h->rollout_k = k;
  return 0;
}
int cc4_rollout_groups(cc4_handle* h, int32_t* groups, int32_t* block) {
  if (h->persist_state == 0) { HIPCHK(h, hipSetDevice(h->cfg.device_id)); if (persist_setup(h)) return -1; }
  *groups = h->rpg; *block = h->run_P > 0 ? h->run_P : h->cus;
  return 0;
}
int cc4_rollout_obs_packed(cc4_handle* h, int32_t j, const uint8_t** d_rows) {
  if (!h->d_xslab || !h->d_ract) { h->err = "cc4_rollout_obs_packed: no rollout was begun on this handle"; return -2; }      // (also behind cc4_rollout_end: the ring keeps the last 32 steps)
  *d_rows = h->d_xslab + (size_t)((j + cc4_handle::XRING - 1) % cc4_handle::XRING) * (size_t)h->cfg.num_envs * OBS_PACKED;
  return 0;
}
int cc4_rollout_actions(cc4_handle* h, int32_t j, int32_t** d_actions) {
  if (!h->d_ract) { h->err = "cc4_rollout_actions: no rollout was begun on this handle"; return -2; }
  *d_actions = h->d_ract + (size_t)(j & 1) * (size_t)h->cfg.num_envs * NBLUE;
  return 0;
}
int cc4_rollout_policy_stream(cc4_handle* h, void** hip_stream) {
  if (!h->policy_stream) { h->err = "cc4_rollout_policy_stream: no rollout was begun on this handle"; return -2; }
  *hip_stream = h->policy_stream;
  return 0;
}
int cc4_rollout_wait_obs(cc4_handle* h, int32_t g, int32_t j, void* hip_stream) {
  if (rollout_ready(h, "cc4_rollout_wait_obs")) return -2;
  if (g < 0 || g >= h->rpg || j < 0 || j >= h->rollout_k) { h->err = "cc4_rollout_wait_obs: group or step out of range"; return -2; }
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->gpolicy[g];
  if (j == 0) { HIPCHK(h, hipStreamWaitEvent(st, h->rev, 0)); return 0; }
  hipLaunchKernelGGL(k_rollout_gate, dim3(1), dim3(WAVE), 0, st, h->d_rcnt, h->run_P, h->rpg, (int)cc4_handle::XRING, (int)g, (int)((j - 1) % cc4_handle::XRING), h->cfg.num_envs,
                     (long long)h->rollout_watchdog_ms * h->khz, h->d_rfail);
  HIPCHK(h, hipGetLastError());
  return 0;
}
int cc4_rollout_sync(cc4_handle* h, int32_t pub_g, int32_t pub_j, int32_t gate_g, int32_t gate_j, void* hip_stream);
int cc4_rollout_publish(cc4_handle* h, int32_t g, int32_t j, void* hip_stream) {
  if (rollout_ready(h, "cc4_rollout_publish")) return -2;
  if (g < 0 || g >= h->rpg || j < 0 || j >= h->rollout_k) { h->err = "cc4_rollout_publish: group or step out of range"; return -2; }
  return cc4_rollout_sync(h, g, j, -1, 0, hip_stream);       // (a one-wave kernel: the word is published once per CU partition)
}
int cc4_rollout_sync(cc4_handle* h, int32_t pub_g, int32_t pub_j, int32_t gate_g, int32_t gate_j, void* hip_stream) {
  if (rollout_ready(h, "cc4_rollout_sync")) return -2;
  if (pub_g >= h->rpg || gate_g >= h->rpg || (pub_g >= 0 && (pub_j < 0 || pub_j >= h->rollout_k)) || (gate_g >= 0 && (gate_j < 0 || gate_j >= h->rollout_k))) { h->err = "cc4_rollout_sync: group or step out of range"; return -2; }
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->gpolicy[gate_g >= 0 ? gate_g : (pub_g >= 0 ? pub_g : 0)];
  if (gate_g >= 0 && gate_j == 0) { HIPCHK(h, hipStreamWaitEvent(st, h->rev, 0)); gate_g = -1; }      // (the observations as they stood: behind the event)
  if (pub_g < 0 && gate_g < 0) return 0;
  hipLaunchKernelGGL(k_rollout_sync, dim3(1), dim3(WAVE), 0, st, h->d_rready, (int)pub_g, (uint32_t)(pub_j + 1), h->d_rcnt, h->run_P, h->rpg, (int)cc4_handle::XRING, (int)gate_g,
                     (int)(gate_g >= 0 ? (gate_j - 1) % cc4_handle::XRING : 0), h->cfg.num_envs, (long long)h->rollout_watchdog_ms * h->khz, h->d_rfail);
  HIPCHK(h, hipGetLastError());
  return 0;
}
int cc4_rollout_random_policy(cc4_handle* h, int32_t g, int32_t j, uint64_t seed0, uint32_t t, void* hip_stream) {
  if (rollout_ready(h, "cc4_rollout_random_policy")) return -2;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->gpolicy[g];
  const int tot = h->cfg.num_envs * NBLUE;
  const int grp = ((h->cfg.num_envs + h->run_P - 1) / h->run_P + h->rpg - 1) / h->rpg * h->run_P * NBLUE;      // threads over the group's episodes (whole blocks of P)
  hipLaunchKernelGGL(k_rollout_random_policy, dim3((grp + WAVE - 1) / WAVE), dim3(WAVE), 0, st, h->d_ract + (size_t)(j & 1) * (size_t)tot, h->cfg.num_envs, h->run_P, h->rpg, (int)g, seed0, t);
  HIPCHK(h, hipGetLastError());
  return 0;
}
int cc4_rollout_hash_policy(cc4_handle* h, int32_t g, int32_t j, void* hip_stream) {
  if (rollout_ready(h, "cc4_rollout_hash_policy")) return -2;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->gpolicy[g];
  const int n = h->cfg.num_envs;
  const uint8_t* rows = h->d_xslab + (size_t)((j + cc4_handle::XRING - 1) % cc4_handle::XRING) * (size_t)n * OBS_PACKED;
  const int grp = ((n + h->run_P - 1) / h->run_P + h->rpg - 1) / h->rpg * h->run_P;
  hipLaunchKernelGGL(k_rollout_hash_policy, dim3((grp + WAVE - 1) / WAVE), dim3(WAVE), 0, st, h->d_ract + (size_t)(j & 1) * (size_t)n * NBLUE, rows, n, h->run_P, h->rpg, (int)g, (uint32_t)j);
  HIPCHK(h, hipGetLastError());
  return 0;
}
// debug: where a rollout stands / stood -- out[0..1] gate-failed flag and the kernel's timeout flag, out[2 + g] the groups' published step counts,
// out[6 + 4 * slot + g] = sum over the partitions of the count of (policy group g, ring slot), slots 0..3
int cc4_debug_rollout_state(cc4_handle* h, int64_t* out /* [22] */) {
  if (!h->d_rcnt) { h->err = "cc4_debug_rollout_state: no rollout was begun on this handle"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  std::vector<uint32_t> rd(32), cnt((size_t)h->run_P * RPG_MAX * cc4_handle::XRING);
  uint32_t fail = 0;
  HIPCHK(h, hipMemcpy(rd.data(), h->d_rready, rd.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(cnt.data(), h->d_rcnt, cnt.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(&fail, h->d_rfail, 4, hipMemcpyDeviceToHost));
  out[0] = fail; out[1] = *reinterpret_cast<volatile uint32_t*>(h->h_xtimeout);
  for (int g = 0; g < h->rpg; ++g) out[2 + g] = rd[g];
  for (int slot = 0; slot < 4; ++slot) for (int g = 0; g < h->rpg; ++g) {
    int64_t sum = 0;
    for (int p = 0; p < h->run_P; ++p) sum += cnt[((size_t)p * h->rpg + g) * cc4_handle::XRING + slot];
    out[6 + 4 * slot + g] = sum;
  }
  return 0;
}
int cc4_rollout_end(cc4_handle* h) {
  if (rollout_ready(h, "cc4_rollout_end")) return -2;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  h->rollout_entering = true;
  const hipError_t e1 = hipStreamSynchronize(h->stream);
  hipError_t e2 = hipSuccess;
  for (int g = 0; g < RPG_MAX; ++g) { const hipError_t e = hipStreamSynchronize(h->gpolicy[g]); if (e != hipSuccess) e2 = e; }
  h->rollout_entering = false;
  const int k = h->rollout_k;
  h->rollout_k = 0;
  HIPCHK(h, e1); HIPCHK(h, e2);
  uint32_t gate_failed = 0;
  HIPCHK(h, hipMemcpy(&gate_failed, h->d_rfail, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (*reinterpret_cast<volatile uint32_t*>(h->h_xtimeout) || gate_failed) {
    h->err = "cc4_rollout_end: a step of the " + std::to_string(k) + "-step rollout waited longer than " + std::to_string(h->rollout_watchdog_ms) +
             " ms for its actions (or a policy gate for its observations): not every group's policy pass of every step was published -- the episodes were "
             "stepped with whatever the action slots held (CC4_ROLLOUT_WATCHDOG_MS)";
    return -6;
  }
  return 0;
}
// A whole rollout with a stand-in policy (0: random indices, 1: hash of the observations), driven from here: begin, the passes of all steps -- two
// stream operations each (cc4_rollout_sync, the policy kernel) --, end.  What bench.py times as `policy_in_loop`, and what a trainer written against the C ABI would do.
int cc4_rollout_standin(cc4_handle* h, int32_t k, int32_t policy, uint64_t seed0, uint32_t t0) {
  int rc = cc4_rollout_begin(h, k);
  if (rc) return rc;
  // every policy group has a stream and a chain of its own: [publish of its pass of step j - 1 + gate of step j] -> policy of step j -> ...
  for (int j = 0; j < k && !rc; ++j)
    for (int g = 0; g < h->rpg && !rc; ++g) {
      rc = cc4_rollout_sync(h, j > 0 ? g : -1, j - 1, g, j, nullptr);
      if (!rc) rc = policy == 0 ? cc4_rollout_random_policy(h, g, j, seed0, t0 + (uint32_t)j, nullptr) : cc4_rollout_hash_policy(h, g, j, nullptr);
    }
  for (int g = 0; g < h->rpg && !rc; ++g) rc = cc4_rollout_sync(h, g, k - 1, -1, 0, nullptr);
  const int end = cc4_rollout_end(h);
  return rc ? rc : end;
}
int cc4_launches_per_step(cc4_handle* h) { return h ? h->ngroups : 0; }
// host-side counters since cc4_create: steps issued by cc4_run_random_steps, microseconds the host spent enqueueing their step
// launches and their all-gathers, all-gathers issued, and how many times a step had to WAIT for an old all-gather before it
// could reuse that observation buffer (0 = the exchange never held the compute stream up)
int cc4_host_stats(cc4_handle* h, double* out /* [5] */) {
  out[0] = (double)h->stat_steps; out[1] = h->stat_launch_us; out[2] = h->stat_gather_us; out[3] = (double)h->gathers_issued; out[4] = (double)h->gather_stalls;
  return 0;
}

int cc4_get_state(cc4_handle* h, int32_t env, void* buf) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_get_state: env out of range"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(buf, h->d_state + env, sizeof(EnvState), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_get_states(cc4_handle* h, int32_t first, int32_t count, void* buf) {
  if (first < 0 || count < 0 || first + count > h->cfg.num_envs) { h->err = "cc4_get_states: range out of the batch"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  if (count) HIPCHK(h, hipMemcpyAsync(buf, h->d_state + first, (size_t)count * sizeof(EnvState), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_set_state(cc4_handle* h, int32_t env, const void* buf) {
  h->prev_valid = false;        // (also every cc4_edit_state, which writes the rows back through here)
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_set_state: env out of range"; return -2; }
  {   // the cold containers of this handle were sized from cfg.steps; the row says how long ITS episode is (EnvState.steps)
    const int st_steps = static_cast<const EnvState*>(buf)->steps;
    // (0 = a never-reset, all-zero row; a negative length would turn into negative container capacities on the device)
    if (st_steps < 0 || (st_steps > 0 && cold_row_bytes(st_steps) != h->cold_row)) {
      h->err = "cc4_set_state: the row belongs to an episode of " + std::to_string(st_steps) + " steps, whose cold containers differ from this handle's (steps=" + std::to_string(h->cfg.steps) + ")";
      return -2;
    }
  }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(h->d_state + env, buf, sizeof(EnvState), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->full_obs_next = true;      // the observation buffer still holds the previous occupant's slowly varying values
  return 0;
}

size_t cc4_cold_bytes(cc4_handle* h) { return h ? h->cold_row : 0; }
int cc4_get_cold(cc4_handle* h, int32_t env, void* buf) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_get_cold: env out of range"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(buf, cold_at(h->d_cold, (size_t)env, h->cold_row), h->cold_row, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int cc4_set_cold(cc4_handle* h, int32_t env, const void* buf) {
  h->prev_valid = false;
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_set_cold: env out of range"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipMemcpyAsync(cold_at(h->d_cold, (size_t)env, h->cold_row), buf, h->cold_row, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int cc4_get_topology(cc4_handle* h, int32_t env, uint8_t* out) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_get_topology: env out of range"; return -2; }
  EnvState* tmp = (EnvState*)malloc(sizeof(EnvState));
  HostStatic* hs = (HostStatic*)malloc(sizeof(HostStatic) * MAXH);
  int rc = cc4_get_state(h, env, tmp);
  if (rc == 0) {
    hipError_t e = hipMemcpy(hs, cold_at(h->d_cold, (size_t)env, h->cold_row)->hs, sizeof(HostStatic) * MAXH, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { h->err = std::string("cc4_get_topology: ") + hipGetErrorString(e); rc = -1; }
  }
  if (rc == 0) {
    for (int i = 0; i < NSUB; ++i) { out[i] = tmp->cidr_octet[i]; out[9 + i] = tmp->n_users[i]; out[18 + i] = tmp->n_servers[i]; }
    for (int i = 0; i < MAXH; ++i) { out[27 + 2 * i] = bit_get(tmp->exists, i) ? 1 : 0; out[28 + 2 * i] = hs[i].ip_octet; }
  }
  free(tmp); free(hs);
  return rc;
}

int cc4_enable_event_log(cc4_handle* h, int32_t enable) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  hipLaunchKernelGGL(k_set_evlog, dim3((h->cfg.num_envs + 255) / 256), dim3(256), 0, h->stream, h->d_cold, h->cold_row, h->cfg.num_envs, enable ? 1u : 0u);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->evlog_on = enable ? 1 : 0;
  return 0;
}
// The event log on demand (handles of up to 16 episodes without a communicator).  cc4_keep_previous(1): every step launch is preceded by a
// device-side copy of the episodes' rows.  cc4_replay_logged: the LAST step again, on that copy, with the logging build of the step kernel
// and the same inputs (the action / message / submitted-action buffers still hold them) -- the copy ends where the live rows are, and its
// event log is copied into the live cold rows: cc4_get_true_state then reports the HostEvents entries of the last step although the step
// itself ran the fast build.  With no step since the reset the log is simply empty.  Same generator positions: logging draws nothing in
// the numpy-stream mode and only side streams in the counter mode.
int cc4_keep_previous(cc4_handle* h, int32_t on) {
  if (on && (h->comm || h->cfg.num_envs > 16 || h->cfg.autoreset)) { h->err = "cc4_keep_previous: for handles of up to 16 episodes without a communicator or autoreset"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t n = (size_t)h->cfg.num_envs;
  if (on && !h->d_prev_state) {
    HIPCHK(h, hipMalloc(&h->d_prev_state, n * sizeof(EnvState)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->d_prev_cold), n * h->cold_row));
    HIPCHK(h, hipMalloc(&h->d_prev_out, h->out_bytes));
  }
  h->keep_prev = on != 0;
  h->prev_valid = false;
  return 0;
}
int cc4_replay_logged(cc4_handle* h) {
  if (!h->keep_prev) { h->err = "cc4_replay_logged: cc4_keep_previous is off"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  const int n = h->cfg.num_envs;
  if (!h->prev_valid) {      // no step since the reset (or steps whose inputs are gone): an enabled, empty log
    hipLaunchKernelGGL(k_set_evlog, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_cold, h->cold_row, n, 1u);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
  }
  hipLaunchKernelGGL(k_set_evlog, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_prev_cold, h->cold_row, n, 1u);
  HIPCHK(h, hipGetLastError());
  int32_t* o = reinterpret_cast<int32_t*>(h->d_prev_out);
  float* rw = reinterpret_cast<float*>(o + (size_t)n * OBS_TOTAL);
  uint32_t* er = reinterpret_cast<uint32_t*>(rw + n);
  uint8_t* dn = reinterpret_cast<uint8_t*>(er + n);
  StepArgs a{h->d_prev_state, h->d_prev_cold, h->prev_actions, h->prev_msgs, o, rw, dn, er, nullptr, nullptr, 0, 0,
             n, 0, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), 1,
             (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, h->prev_ext ? h->d_ext : nullptr, 0};
  for (int g = 0; g < h->ngroups; ++g) launch_group(h, a, g, true, nullptr, nullptr);
  HIPCHK(h, hipGetLastError());
  if (h->ngroups > 1) { h->groups_busy = true; if (join_groups(h)) return -1; }
  hipLaunchKernelGGL(k_copy_evlog, dim3(n), dim3(64), 0, h->stream, h->d_cold, h->d_prev_cold, h->cold_row, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->prev_valid = false;          // (the copy has moved on: a second replay would repeat the step from the wrong rows)
  return 0;
}
int64_t cc4_get_true_state(cc4_handle* h, int32_t env, char* json, size_t cap) {
  if (env < 0 || env >= h->cfg.num_envs) { h->err = "cc4_get_true_state: env out of range"; return -2; }
  EnvState* st = (EnvState*)malloc(sizeof(EnvState));
  EnvCold* cold = (EnvCold*)malloc(h->cold_row);
  int64_t rc = cc4_get_state(h, env, st);
  if (rc == 0) rc = cc4_get_cold(h, env, cold);
  if (rc == 0) {
    std::string doc = export_true_state(*st, *cold);
    rc = (int64_t)doc.size() + 1;
    if (json && cap >= doc.size() + 1) memcpy(json, doc.c_str(), doc.size() + 1);
  }
  free(st); free(cold);
  return rc;
}

// debug: enable (buf != NULL first call allocates) / read per-episode cycle counters [N][64] (16 phase slots, then 8 per red agent)
int cc4_debug_profile(cc4_handle* h, int enable, unsigned long long* out) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  size_t bytes = (size_t)h->cfg.num_envs * PROF_SLOTS * sizeof(unsigned long long);
  if (enable && !h->d_prof) { HIPCHK(h, hipMalloc(&h->d_prof, bytes)); HIPCHK(h, hipMemsetAsync(h->d_prof, 0, bytes, h->stream)); }
  if (out && h->d_prof) { HIPCHK(h, hipMemcpyAsync(out, h->d_prof, bytes, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream)); }
  if (!enable && h->d_prof) { (void)hipFree(h->d_prof); h->d_prof = nullptr; }
  return 0;
}

// measurement (tools/valu_phases.py): from now on the per-step launches of k_step_philox1 (its full build) end behind phase `phase` of the step (1..13,
// csrc/cc4_philox1_body.h CC4_STOP) and write no row back; 14 = whole steps of the full build; 0 = whole steps of the usual build again.  The caller restores the batch (cc4_set_state / cc4_set_cold) after such a step.
int cc4_debug_stop_phase(cc4_handle* h, int phase) {
  if (join_groups(h)) return -1;
  if (phase < 0 || phase > 14 || !h->philox_lean) { h->err = "cc4_debug_stop_phase: phase 0..14, on a handle whose step kernel is k_step_philox1"; return -2; }
  h->dbg_stop = phase;
  return 0;
}

// debug: the red policy phase of every episode with G episodes' agents per wave; out[0] = mean launch duration in us, out[1] = mean cycles
// of a wave in the phase, out[2] = waves per launch.  Reads the batch as it stands, writes nothing back.
int cc4_debug_policy_probe(cc4_handle* h, int32_t G, int32_t reps, double* out) {
#ifndef CC4_POLICY_PROBE
  (void)G; (void)reps; (void)out;
  h->err = "cc4_debug_policy_probe: this library was built without -DCC4_POLICY_PROBE (the experiment of DESIGN 3.4 is concluded; tools/policy_group_probe.py says how to build it)";
  return -2;
#else
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (h->cfg.rng_mode != 1) { h->err = "cc4_debug_policy_probe: counter mode only"; return -2; }
  if (join_groups(h)) return -1;
  const int n = h->cfg.num_envs, waves = (n + G - 1) / G;
  unsigned long long* d_cyc = nullptr;
  HIPCHK(h, hipMalloc(&d_cyc, (size_t)waves * sizeof(unsigned long long)));
  StepArgs a{h->d_state, h->d_cold, nullptr, nullptr, h->d_obs, h->d_reward, h->d_done, h->d_err, nullptr, nullptr, 0, 0,
             n, 0, h->cfg.steps, h->cfg.rng_mode,
             (h->cfg.red_policy & 3) | (h->cfg.green_policy ? GP_SLEEP_BIT : 0) | (h->cfg.green_policy == 2 ? GP_OPEN_BIT : 0) | (h->cfg.blue_policy ? BP_RANDOM_BIT : 0), 0,
             (uint32_t)h->cfg.topology_seed, nullptr, h->d_reset_ws, nullptr, 0};
  hipEvent_t e0, e1;
  HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
  const size_t dyn = (size_t)G * offsetof(EnvState, hd);
  auto launch = [&]() {
    switch (G) {
      case 1: hipLaunchKernelGGL(k_policy_probe<1>, dim3(waves), dim3(WAVE), dyn, h->stream, a, d_cyc); break;
      case 2: hipLaunchKernelGGL(k_policy_probe<2>, dim3(waves), dim3(WAVE), dyn, h->stream, a, d_cyc); break;
      case 4: hipLaunchKernelGGL(k_policy_probe<4>, dim3(waves), dim3(WAVE), dyn, h->stream, a, d_cyc); break;
      default: hipLaunchKernelGGL(k_policy_probe<8>, dim3(waves), dim3(WAVE), dyn, h->stream, a, d_cyc); break;
    }
  };
  if (G != 1 && G != 2 && G != 4 && G != 8) { h->err = "cc4_debug_policy_probe: G is 1, 2, 4 or 8"; (void)hipFree(d_cyc); return -2; }
  if (G == 8) HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_policy_probe<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  launch();                                                     // warm-up
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(e0, h->stream));
  for (int i = 0; i < reps; ++i) launch();
  HIPCHK(h, hipEventRecord(e1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f; HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc((size_t)waves);
  HIPCHK(h, hipMemcpy(cyc.data(), d_cyc, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double sum = 0; for (auto c : cyc) sum += (double)c;
  out[0] = (double)ms * 1000.0 / (reps > 0 ? reps : 1); out[1] = sum / waves; out[2] = waves;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d_cyc);
  return 0;
#endif
}
int cc4_debug_copy_from_device(cc4_handle* h, void* host_dst, const void* device_src, size_t bytes) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int g = 0; g < 4; ++g) if (h->gpolicy[g]) HIPCHK(h, hipStreamSynchronize(h->gpolicy[g]));
  HIPCHK(h, hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int cc4_comm_unique_id(void* id128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, 128);
  return 0;
}
int cc4_comm_init(cc4_handle* h, int32_t rank, int32_t world, const void* id128) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (join_groups(h)) return -1;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclResult_t r = ncclCommInitRank(&h->comm, world, id, rank);
  if (r != ncclSuccess) { h->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); return -1; }
  h->rank = rank; h->world = world;
  if (!getenv("CC4_GROUPS")) {
    // With the exchange every launch carries a completion event and the host guards the observation ring, so a launch costs the
    // host several times what it costs without; small shards then run into the host.  Measured on MI355X with the exchange on a
    // one-rank communicator (r03, profiles/r03_exchange_groups_world1.txt; M agent-env steps/s, 1 / 2 / 3 launches per step):
    // 1024 episodes 159 / 104 / 111, 2048: 242 / 179 / 220, 4096: 380 / 282 / 420, 8192: 478 / 498 / 607.
    int ng = h->cfg.num_envs >= 4096 ? 3 : 1;      // (8192 episodes with the exchange, 3 / 4 launches per step: 563 / 509 M)
    if (const char* v = getenv("CC4_EXCHANGE_GROUPS")) { ng = atoi(v); if (ng <= 0 || ng > h->ngroups) ng = h->ngroups; }   // tuning override: 0 = keep the handle's groups
    if (ng != h->ngroups) {
      if (sync_all(h)) return -1;
      const int old = h->ngroups;
      configure_groups(h, ng);
      for (int g = old; g < h->ngroups; ++g) {
        if (!h->gstream[g]) HIPCHK(h, hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking));
        if (!h->gev[g]) HIPCHK(h, hipEventCreateWithFlags(&h->gev[g], hipEventDisableTiming));
      }
      h->main_ahead = true;
    }
  }
  size_t nb = (size_t)h->cfg.num_envs * OBS_PACKED;
  HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  for (int b = 0; b < cc4_handle::OBS_RING; ++b) {
    HIPCHK(h, hipMalloc(&h->d_obs8[b], nb));
    HIPCHK(h, hipMalloc(&h->d_all_obs8[b], nb * (size_t)world));
    HIPCHK(h, hipMemsetAsync(h->d_obs8[b], 0, nb, h->stream));
    for (int g = 0; g < h->ngroups; ++g) HIPCHK(h, hipEventCreateWithFlags(&h->ev_step[b][g], hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_comm[b], hipEventDisableTiming));
  }
  // (the ring of step slabs the one-launch kernels write, XchgArgs, and its gathered twin -- 32 x (1 + world) x N x 148 B -- are allocated by
  // the first call that takes a one-launch form: xchg_begin)
  HIPCHK(h, hipEventCreateWithFlags(&h->xev, hipEventDisableTiming));
  int can_wait = 0;
  (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, h->cfg.device_id);
  h->xchg_on = can_wait != 0;
  if (const char* v = getenv("CC4_EXCHANGE_INKERNEL")) h->xchg_on = h->xchg_on && atoi(v) != 0;
  if (const char* v = getenv("CC4_EXCHANGE_CHUNK")) { h->xchg_chunk = atoi(v); if (h->xchg_chunk < 1) h->xchg_chunk = 1; if (h->xchg_chunk > cc4_handle::XRING / 2) h->xchg_chunk = cc4_handle::XRING / 2; }
  if (const char* v = getenv("CC4_EXCHANGE_WATCHDOG_MS")) { h->xchg_watchdog_ms = atoi(v) > 0 ? atoi(v) : 2000; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  // the one-launch forms again, now that a step of one episode waits for the slowest episode of sixteen steps earlier: the multi-step kernels
  // must hold the whole batch with a block per CU to spare (at exactly full residency one block that is placed late stalls everybody until
  // the watchdog: tools/micro/ring_protocol.hip), and with peers RCCL's kernels need that room on every form (the persistent kernel's waves
  // pull items, so on one rank it keeps every slot)
  // (the numpy-stream persistent kernel has ONE build, six waves of 80 registers per SIMD: with a communicator its grid leaves eight waves per CU free, so
  // that every SIMD keeps room for the all-gather's kernels -- the counter mode runs its five-waves-per-SIMD build, k_run_philox1x, instead)
  // (counter mode: one wave per CU less also on a one-rank communicator -- r05 ran that case on a full grid, and r06 saw the soak test time out once)
  if (h->xchg_on) { if (choose_run_form(h, 1, h->cfg.rng_mode == 0 ? 8 : 1)) return -1; }
  return 0;
}
// the in-kernel exchange of this handle: out[0] on (1) / off (0), out[1] ring depth in steps, out[2] steps per publish (CC4_EXCHANGE_CHUNK),
// out[3] calls of cc4_run_random_steps it served, out[4] calls whose watchdog fired (the handle then returns to per-step launches)
int cc4_exchange_info(cc4_handle* h, int32_t* out /* [5] */) {
  out[0] = h->xchg_on ? 1 : 0; out[1] = cc4_handle::XRING; out[2] = h->xchg_chunk; out[3] = (int32_t)h->xchg_calls; out[4] = (int32_t)h->xchg_timeouts;
  return 0;
}
// debug / test hook: keep the gathered rows of the next `steps` steps cc4_run_random_steps exchanges from inside a one-launch kernel
// ([steps][world * N] packed rows, in step order), so that a test can check EVERY step's all-gather, not only the last of a burst.
// steps = 0 frees the log.
int cc4_debug_gather_log(cc4_handle* h, int32_t steps) {
  if (!h->comm) { h->err = "cc4_debug_gather_log: cc4_comm_init was not called"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  if (h->d_xlog) { (void)hipFree(h->d_xlog); h->d_xlog = nullptr; }
  h->xlog_cap = 0; h->xlog_n = 0;
  if (steps > 0) {
    HIPCHK(h, hipMalloc(&h->d_xlog, (size_t)steps * h->world * h->cfg.num_envs * OBS_PACKED));
    h->xlog_cap = steps;
  }
  return 0;
}
// host copy of the log: out [count][world * N][CC4_OBS_PACKED_BYTES]; returns the number of steps logged so far (< 0: error)
int cc4_get_gather_log(cc4_handle* h, uint8_t* out, int32_t first, int32_t count) {
  if (!h->d_xlog || first < 0 || count < 0 || first + count > h->xlog_n) { h->err = "cc4_get_gather_log: no log, or the range was not logged"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  const size_t row = (size_t)h->world * h->cfg.num_envs * OBS_PACKED;
  if (count) HIPCHK(h, hipMemcpy(out, h->d_xlog + (size_t)first * row, (size_t)count * row, hipMemcpyDeviceToHost));
  return h->xlog_n;
}
// What a multi-GPU run needs to PROVE its scaling line: RCCL's own view of the communicator (how many ranks it spans, which one this
// is, which device it is bound to) and the identity of the device this handle runs on.  out[0] ncclCommCount (1 without a
// communicator), out[1] ncclCommUserRank (0), out[2] ncclCommCuDevice (-1), out[3] the handle's HIP device ordinal, out[4] PCI
// domain, out[5] PCI bus, out[6] PCI device, out[7] compute units; uuid_hex: 32 hex digits + NUL of hipDeviceProp_t::uuid.
int cc4_comm_info(cc4_handle* h, int32_t* out, char* uuid_hex) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int count = 1, urank = 0, cudev = -1;
  if (h->comm) {
    if (ncclCommCount(h->comm, &count) != ncclSuccess || ncclCommUserRank(h->comm, &urank) != ncclSuccess || ncclCommCuDevice(h->comm, &cudev) != ncclSuccess) {
      h->err = "cc4_comm_info: RCCL did not answer"; return -1;
    }
  }
  hipDeviceProp_t prop;
  HIPCHK(h, hipGetDeviceProperties(&prop, h->cfg.device_id));
  out[0] = count; out[1] = urank; out[2] = cudev; out[3] = h->cfg.device_id;
  out[4] = prop.pciDomainID; out[5] = prop.pciBusID; out[6] = prop.pciDeviceID; out[7] = prop.multiProcessorCount;
  if (uuid_hex) { for (int i = 0; i < 16; ++i) snprintf(uuid_hex + 2 * i, 3, "%02x", (unsigned)(unsigned char)prop.uuid.bytes[i]); }
  return 0;
}
// All-gather of the observations written by the most recent step (as bytes, [world*N][578]) over RCCL/xGMI on the
// handle's communication stream: it waits for that step's kernel, runs concurrently with whatever is enqueued next on
// the compute stream (later steps write other buffers of the ring), and is awaited by cc4_allgather_wait / the step that
// reuses its buffer.  *d_all_obs8 is valid after cc4_allgather_wait().
int cc4_allgather_obs(cc4_handle* h, uint8_t** d_all_obs8) {
  if (!h->comm) { h->err = "cc4_allgather_obs: cc4_comm_init was not called"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int buf = h->obs_buf;
  if (h->obs8_from_slab >= 0) {    // the last step ran inside a one-launch kernel with the exchange: its packed rows are in the exchange ring
    const size_t row = (size_t)h->cfg.num_envs * OBS_PACKED;
    HIPCHK(h, hipMemcpyAsync(h->d_obs8[buf], h->d_xslab + (size_t)h->obs8_from_slab * row, row, hipMemcpyDeviceToDevice, h->stream));
    h->obs8_from_slab = -1;
  }
  if (!h->step_event_attached) {   // e.g. the observations of a reset: main-stream work, behind which the group streams' work was joined
    if (join_groups(h)) return -1;
    HIPCHK(h, hipEventRecord(h->ev_step[buf][0], h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_step[buf][0], 0));
  } else {
    for (int g = 0; g < h->ngroups; ++g) HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_step[buf][g], 0));
  }
  if (h->comm_delay_ticks > 0) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, h->comm_stream, h->comm_delay_ticks); HIPCHK(h, hipGetLastError()); }
  size_t cnt = (size_t)h->cfg.num_envs * OBS_PACKED;
  ncclResult_t r = ncclAllGather(h->d_obs8[buf], h->d_all_obs8[buf], cnt, ncclUint8, h->comm, h->comm_stream);
  if (r != ncclSuccess) { h->err = std::string("ncclAllGather: ") + ncclGetErrorString(r); return -1; }
  const long long q = ++h->gathers_issued;
  h->gather_seq[buf] = q;
  h->gather_buf = buf;
  h->last_gathered = h->d_all_obs8[buf];
  HIPCHK(h, hipEventRecord(h->ev_comm[q % cc4_handle::OBS_RING], h->comm_stream));
  if (d_all_obs8) *d_all_obs8 = h->d_all_obs8[buf];
  return 0;
}
// debug / test hook: every all-gather is preceded by a kernel that keeps the communication stream busy for about `us`
// microseconds -- an exchange slower than the step, which is what makes the observation ring's reuse guard work for its living
int cc4_debug_comm_delay_us(cc4_handle* h, int us) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int khz = 100000;   // wall_clock64 ticks at the constant 100 MHz reference clock
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id);
  if (khz <= 0) khz = 100000;
  h->comm_delay_ticks = (long long)us * khz / 1000;
  return 0;
}
int cc4_allgather_wait(cc4_handle* h) {
  if (!h->comm) { h->err = "cc4_allgather_wait: cc4_comm_init was not called"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  return 0;
}
// host copy of the gathered observations of the most recent cc4_allgather_obs (tests / debugging)
int cc4_get_allgathered_obs(cc4_handle* h, uint8_t* out /* [world*N][578] */) {
  if (!h->comm) { h->err = "cc4_get_allgathered_obs: cc4_comm_init was not called"; return -2; }
  if (!h->last_gathered) { h->err = "cc4_get_allgathered_obs: no all-gather has been issued"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  const size_t rows = (size_t)h->world * h->cfg.num_envs;
  std::vector<uint8_t> packed(rows * OBS_PACKED);
  HIPCHK(h, hipMemcpy(packed.data(), h->last_gathered, packed.size(), hipMemcpyDeviceToHost));
  for (size_t r = 0; r < rows; ++r)      // unpack to one byte per value for the host caller
    for (int i = 0; i < OBS_TOTAL; ++i) out[r * OBS_TOTAL + i] = (uint8_t)((packed[r * OBS_PACKED + (i >> 2)] >> (2 * (i & 3))) & 3u);
  return 0;
}
// Device-side consumer of the exchange format: the gathered rows of the most recent cc4_allgather_obs ([world*N] rows of
// CC4_OBS_PACKED_BYTES, 2 bits per value) unpacked to [world*N][578] bytes in a buffer owned by the handle -- what a shared
// on-GPU policy reads.  Enqueued on the communication stream behind the all-gather; *d_obs_u8 is valid after
// cc4_allgather_wait() (or after any later operation ordered behind ev_comm of that gather).
int cc4_unpack_obs_device(cc4_handle* h, uint8_t** d_obs_u8) {
  if (!h->comm) { h->err = "cc4_unpack_obs_device: cc4_comm_init was not called"; return -2; }
  if (!h->last_gathered) { h->err = "cc4_unpack_obs_device: no all-gather has been issued"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t rows = (size_t)h->world * h->cfg.num_envs;
  if (!h->d_unpacked) HIPCHK(h, hipMalloc(&h->d_unpacked, rows * OBS_TOTAL));
  hipLaunchKernelGGL(k_unpack_obs, dim3((unsigned)rows), dim3(192), 0, h->comm_stream, h->last_gathered, h->d_unpacked, (int)rows);
  HIPCHK(h, hipGetLastError());
  if (d_obs_u8) *d_obs_u8 = h->d_unpacked;
  return 0;
}
// host copy of that buffer (tests)
int cc4_get_unpacked_obs(cc4_handle* h, uint8_t* out /* [world*N][578] */) {
  if (!h->d_unpacked) { h->err = "cc4_get_unpacked_obs: cc4_unpack_obs_device was not called"; return -2; }
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  HIPCHK(h, hipMemcpy(out, h->d_unpacked, (size_t)h->world * h->cfg.num_envs * OBS_TOTAL, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"

