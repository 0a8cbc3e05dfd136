// cc4_state.h -- packed per-episode state of the CC4 (CybORG v4) step engine.
//
// One episode ("env") = one EnvState (hot, < 16 KB, staged in LDS) + one EnvCold (host backup images, process-list overflow,
// ephemeral-port bitmaps and per-red-session port knowledge; stays in HBM, touched a handful of times per step).  Everything
// is fixed-size POD so that a whole episode can be staged with coalesced loads and snapshot with memcpy.
//
// Process lists (Host.processes) are unbounded in the reference (a blue agent may stack decoys on one host without limit,
// DecoyAction.py:47-114 / DecoyVsftpd.py:10-20): the first PIN entries of a host's list live in the hot row, the rest in the
// cold row (cold_povf).  What bounds a list is the episode length -- DeployDecoy takes two ticks, so an episode of `steps`
// steps stacks at most steps/2 decoys on one host -- so the cold containers are sized from EnterpriseScenarioGenerator(steps=...)
// when the handle is created (cold_povf_cap / cold_sus_cap below): 500 steps -> 376 + 8 = 384 process slots per host (250 decoys,
// 7 generated processes, 127 red shells; the differential fuzz reached 257 with a decoy-only blue policy), 1000 steps -> 632 + 8.
// E_PROC_OVERFLOW beyond.
//
// Host id layout: h = subnet*17 + slot; slot 0 router, 1..10 user_host_0..9, 11..16 server_host_0..5;
// internet root host = 136.  Subnet index = SUBNET enum order of
// CybORG/Simulator/Scenarios/EnterpriseScenarioGenerator.py:40-51.  Iterating existing hosts by
// increasing id == the reference's dict iteration order over state.hosts / state.ip_addresses.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "cc4_rng.h"

namespace cc4 {

enum : int {
  NSUB = 9, SLOTS = 17, MAXH = 137, H_INTERNET = 136,
  MAX_USERS = 10, MAX_SERVERS = 6, ZONE_HOSTS = 16,
  MAXG = 80,            // green agents (one per user host)
  NBLUE = 5, NRED = 6,
  PIN = 8,              // process slots per host in the hot row (every freshly generated host fits: <= 7 processes)
  MAXSV = 7,            // services per host: sshd, OT, {apache | decoy apache} + decoy vsftpd (both port 80), mysql,
                        // decoy tomcat, {smtp | decoy haraka} -- the port checks of DecoyAction exclude any eighth
  MAX_RS = 64,          // sessions per red agent (its ordered list of pool slots)
  RS_POOL = 192,        // red session records per episode, shared by the six agents (r03's occupancy experiment with 96 / MAX_KS 32:
                        // profiles/r03_occupancy_experiment.txt; both generation bitmaps are lent from this pool and need >= 152 records)
  MAX_KS = 96,          // known server-session ids per red agent (ActionSpace.server_session)
  MAX_OBS = 32,         // red observation entries per agent per step (a subnet sweep adds 16, every other source <= 4)
  MAX_PEND = 8,         // process_creation events carrying a pid, per step (one per red agent)
  EPH_WORDS = 340,      // 10880-bit bitmap >= 60000-49152 ephemeral ports (Host.py:183); 1360 B = 85 x 16 B
  OBS_SHORT = 92, OBS_LONG = 210, OBS_TOTAL = 4 * 92 + 210,   // 578
  ACT_SHORT = 82, ACT_LONG = 242, MASK_TOTAL = 4 * 82 + 242,  // 570
  MSG_LEN = 8,
};

// subnet indices (SUBNET enum order)
enum : int { S_RZA = 0, S_OZA = 1, S_RZB = 2, S_OZB = 3, S_CON = 4, S_PUB = 5, S_ADM = 6, S_OFF = 7, S_INT = 8 };

// process / service kinds. Kinds 0..8 double as service-table keys (host.services dict keys).
enum : int {
  K_SSHD = 0, K_OT = 1, K_APACHE = 2, K_MYSQL = 3, K_SMTP = 4,
  K_DEC_APACHE = 5, K_DEC_TOMCAT = 6, K_DEC_HARAKA = 7, K_DEC_VSFTPD = 8,
  K_SESS_BLUE = 9, K_SESS_GREEN = 10, K_SESS_RED = 11,  // Process(name=session_type) of Host.add_session
  K_SHELL = 12,                                          // cmd.sh of ExploitAction._create_new_session
  K_PLAIN = 13,                                          // a process without listening ports (state_edit SE_ADD_SERVICE: the hand-made
                                                         // Process(pid, name) of the reference's tests has no open_ports, so no scan sees it)
};
// listening-port bit codes
enum : int { PB_22 = 1, PB_80 = 2, PB_3390 = 4, PB_25 = 8, PB_1 = 16, PB_443 = 32, PB_HAS = 128 };

// error / bound-overflow flags (cc4 never silently truncates: any bit set => results for that env are flagged)
enum : uint32_t {
  E_PROC_OVERFLOW = 1u << 0, E_RSESS_OVERFLOW = 1u << 1, E_KS_OVERFLOW = 1u << 2, E_KB_OVERFLOW = 1u << 3 /* unused since r02 */,
  E_SUS_OVERFLOW = 1u << 4, E_OBS_OVERFLOW = 1u << 5, E_PEND_OVERFLOW = 1u << 6,
  E_STEP_PAST_END = 1u << 7,      // reference raises ValueError (State.py:539-540)
  E_UNREACHABLE = 1u << 8,        // a path the reference would crash on (documented in DESIGN.md)
  E_BLUE_GREEN_SESSION_KILLED = 1u << 10,   // a stale sus pid / Withdraw hit a blue or green session process (not modelled)
  E_FSM_NO_HOST = 1u << 11,                 // FSM agent with no selectable host (reference: choice([]) raises)
};

struct alignas(4) Proc { uint16_t pid; uint8_t kind; uint8_t flags; };       // flags bit0: user == root
struct alignas(4) Svc  { uint16_t pid; uint8_t kind; uint8_t st; };          // st bit7 active, low bits reliability/20
enum : int { PF_ROOT = 1, SV_ACTIVE = 0x80 };

enum : int { EV_CUR_CONN = 1, EV_CUR_PROC = 2, EV_OLD_CONN = 4, EV_OLD_PROC = 8 };
enum : int { OD_BLOCKS = 1, OD_PHASE = 2, OD_ALL = 7 };   // EnvState.obs_dirty (OD_ALL: what a full rewrite covers on top: the subnet one-hots)

// HostDyn.nsf high nibble: the malware files Analyse reports (Host.files; cleared by Restore): cmd.sh present, escalate.sh present, and
// which of the two was appended last (Observation.add_file_info re-appends a repeated name, so only that order survives)
enum : int { HF_CMD = 1, HF_ESC = 2, HF_ESC_LAST = 4 };
struct alignas(16) HostDyn {        // 64 bytes = four 16-byte vectors
  Proc procs[PIN];                  // entries 0 .. PIN-1 of Host.processes; the rest in the cold row (cold_povf)
  Svc svcs[MAXSV];
  uint16_t nproc;                   // length of the whole list (hot + cold part)
  uint8_t gtmp;                     // scratch of the scenario generation (OS distribution, contested-pid marks); 0 outside a reset.
                                    // (The host's event bits live in EnvState.hev: they change every step, this row does not.)
  uint8_t nsf;                      // low nibble: number of services; high nibble: HF_* bits
};
static_assert(sizeof(HostDyn) == 64, "HostDyn is four 16-byte vectors");
struct alignas(8) HostStatic {                 // Host.create_backup (Host.py:316-371)
  Proc procs[8];
  Svc svcs[5];
  uint8_t nproc, nsvc, exists, ip_octet;   // exists: bit 0 host exists, bit 1 OSDistribution (0 UBUNTU, 1 KALI: ESG.py:488-494)
};

// red sessions (state.sessions[red_agent_k], dict order == array order)
enum : int { RS_ABSTRACT = 1, RS_ROOT = 2, RS_ORIG = 4, RS_CHILD = 8 };   // RS_CHILD: session.parent is not None
// One record of the episode's session pool (EnvState.spool).  An agent's sessions are the pool slots listed in
// RedAgent.sord, in dict order; moving a session between agents (different_subnet_agent_reassignment) moves a list entry,
// not the record.  The port knowledge of an abstract session (RedAbstractSession.ports) is the cold row EnvCold.kports[slot].
struct alignas(8) RSess { uint16_t id; uint16_t pid; uint8_t host; uint8_t flags; uint8_t pad[2]; };

// FSM host states (FiniteStateRedAgent.py:441-452)
enum : int { FS_K = 0, FS_KD = 1, FS_S = 2, FS_SD = 3, FS_U = 4, FS_UD = 5, FS_R = 6, FS_RD = 7, FS_F = 8, FS_NONE = 0xFF };
// red action types in FSM action_list column order (FiniteStateRedAgent.py:393-411) + non-FSM
enum : int { RA_DRS = 0, RA_AGGR = 1, RA_STEALTH = 2, RA_DECEPTION = 3, RA_EXPLOIT = 4, RA_PRIVESC = 5, RA_IMPACT = 6,
             RA_DEGRADE = 7, RA_WITHDRAW = 8, RA_SLEEP = 9, RA_INVALID = 10, RA_NONE = 11 };
// blue action types
enum : int { BA_SLEEP = 0, BA_MONITOR = 1, BA_ANALYSE = 2, BA_REMOVE = 3, BA_RESTORE = 4, BA_DECOY = 5, BA_BLOCK = 6, BA_ALLOW = 7 };
// built-in policies selectable through EnterpriseScenarioGenerator(red_agent_class=, green_agent_class=)
enum : int { RP_FSM = 0, RP_SLEEP = 1, RP_DISCOVERY = 2, RP_RANDOM = 3, GP_SLEEP_BIT = 0x10,
             GP_OPEN_BIT = 0x40 /* with GP_SLEEP_BIT: the green agents have no policy of their own here (a host-side agent class submits
                                   their actions) but, unlike SleepAgent's, their action space holds the green actions (ESG.py:714,742-746) */,
             BP_RANDOM_BIT = 0x20 /* blue_agent_class=cc4BlueRandomAgent: acts for every blue agent no action is submitted for */ };
// TernaryEnum (Shared/Enums.py:5-25)
enum : int { T_TRUE = 1, T_UNKNOWN = 2, T_FALSE = 3, T_IN_PROGRESS = 4 };

struct alignas(8) Act { uint8_t type; uint8_t host; uint8_t arg; uint8_t ticks; uint16_t sid; uint16_t busy; };
// Act.host: target host (blue/red) ; Act.arg: subnet (DRS) or from-subnet (Block/Allow) ; Act.busy: bit 0 queued, bits 1-2: the
// action came with its own rates (AQ_RATE0 / AQ_RATE1: EnvCold.xrate[agent], see ExtAct)
enum : int { AQ_BUSY = 1, AQ_RATE0 = 2, AQ_RATE1 = 4 };

// One externally submitted red or green action of a step: SimulationController.step takes `actions[agent_name]` for ANY agent
// and asks the scenario's agent object only for the agents the dict has no entry for (SimulationController.py:236-240;
// CybORG.step(agent, action), env.py:125-161).  The blue agents' actions arrive as wrapper indices (cc4_step); these records are
// the red and green entries of the dict: [0, NRED) red_agent_r, [NRED, NRED + MAXG) green_agent_g.
//   type   XA_NONE: no entry -- the agent's own policy acts (and draws).  Red: RA_* (RA_INVALID = an action the agent's action
//          space does not hold: InvalidAction, SimulationController.py:1085-1091).  Green: XG_*.
//   host   ip_address / hostname parameter as a host id; arg: DiscoverRemoteSystems subnet index, Withdraw hostname host id
//   sid    session parameter;  ticks: `action.duration` when the caller set one (the reference's tests do), 0 = the class's own
//   flags  XF_RATE0 / XF_RATE1: rate0 / rate1 replace the action's probability attributes -- Aggressive / StealthServiceDiscovery
//          detection_rate (rate0); DiscoverDeception detection_rate (rate0), fp_rate (rate1); GreenLocalWork fp_detection_rate
//          (rate0), phishing_error_rate (rate1); GreenAccessService fp_detection_rate (rate0).
//          XF_SKIP_VALID: skip_valid_action_check=True (SimulationController.py:241)
enum : int { XA_NONE = -1, XG_ACCESS = 0, XG_LOCAL = 1, XG_SLEEP = 2, XG_INVALID = 3 };
enum : int { XF_RATE0 = 1, XF_RATE1 = 2, XF_SKIP_VALID = 4 };
struct alignas(8) ExtAct { int8_t type; uint8_t host; uint8_t arg; uint8_t ticks; uint16_t sid; uint8_t flags; uint8_t pad; double rate0, rate1; };
static_assert(sizeof(ExtAct) == 24, "ExtAct mirrors cc4_agent_action (include/cc4.h)");
enum : int { EXT_PER_ENV = NRED + MAXG };
// a blue wrapper index may carry `action.duration` as well: bits 20..27 (0 = the class's own duration)
enum : int { BLUE_DUR_SHIFT = 20, BLUE_IDX_MASK = (1 << BLUE_DUR_SHIFT) - 1 };

// red observation entry (one dict key of the agent's combined Observation)
enum : int { OE_KEY_IP = 1, OE_SESS = 2, OE_IFACE = 4, OE_SYSHN = 8 };
struct alignas(2) ObsEnt { uint8_t host; uint8_t flags; };

// A red agent's scalar fields, 32 bytes = two 16-byte vectors: the policy / tick path of a step reads them as one batch into
// registers, works there, and writes them back once (every field read separately is an LDS round trip on the agent's
// dependency chain).
struct alignas(16) RedHdr {
  Act queue;                         // actions_in_progress[agent]
  uint16_t as_subnet;                // ActionSpace.subnet known bits
  uint16_t fsm_step;
  uint16_t new_sess_id;
  uint8_t nsess, nknown, fsm_n, nobs;
  uint8_t nlive;                     // popcount(live_hosts)
  uint8_t active;                    // AgentInterface.active
  uint8_t obs_success;               // success of observations[0]
  uint8_t obs_act_type, obs_act_host, obs_act_arg;   // 'action' of observations[0] (RA_NONE if absent / not FSM-relevant)
  uint8_t exec_type, exec_host;      // self.action[agent][0] of this step (for reward)
  uint8_t new_sess_host;             // host of a session created by this agent's exploit this step (0xFF none)
  uint8_t start_host;                // static per episode
  uint8_t rsc_dirty;                 // session table changed since the last full RedSessionCheck observation
  uint8_t rsc_listed;                // this step's observation includes the RedSessionCheck listing of every session
  uint8_t fsm_dirty;                 // the set of session hosts may have changed since fsm_observe last merged a listing
  uint8_t pad[1];
};
static_assert(sizeof(RedHdr) == 32, "RedHdr is two 16-byte vectors");
struct alignas(16) RedAgent {
  alignas(8) uint8_t sord[MAX_RS];   // state.sessions[agent] in dict order: pool slots (read eight at a time)
  uint32_t known_bm[8];              // ActionSpace.server_session keys with value True as a bitmap over ids 0..255.  The same set in
                                     // insertion order -- read once when a policy draws a session, scanned only for ids >= 256 -- is
                                     // EnvCold.known_sid[agent] (r06: 1152 bytes of the staged part moved behind the row; with rng2 gone
                                     // too the agent part is 5952 B, which with the kernels' statics is FIVE 1280-byte LDS granules)
  uint8_t fsm_order[MAXH];           // host_states dict insertion order, restricted to hosts whose state is not 'F'
                                     // ('F' is absorbing and excluded from known_hosts, FiniteStateRedAgent.py:114)
  uint8_t fsm_st4[(MAXH + 1) / 2];   // host_states[h]: one nibble per host, FS_* + 1 (0 = not in host_states): fsm_get / fsm_put
  uint32_t fsm_hn[5];                // host_states[ip]['hostname'] is not None
  uint32_t as_ip[5];                 // ActionSpace.ip_address[ip] == True
  uint32_t as_hn[5];                 // ActionSpace.hostname[name] == True
  ObsEnt obs[MAX_OBS];
  uint32_t obs_has[2][5];            // which (key type, host) pairs are already in obs[] this step
  uint32_t fsm_known[5];             // hosts present in host_states (fsm_state != FS_NONE)
  uint32_t fsm_ur[5];                // hosts whose state is U, UD, R or RD (the ones _session_removal_state_change looks at)
  uint32_t fsm_nodrs[5];             // hosts whose state a successful DiscoverRemoteSystems leaves alone (KD, SD, UD, RD, F)
  uint32_t live_hosts[5];            // hosts currently holding >= 1 session of this agent (kept exact by rs_add / rs_remove_at)
  RedHdr h;                          // the agent's scalars (32 bytes)
};

struct alignas(8) BlueAgent {
  uint32_t sus_hosts[5];             // hosts that have at least one entry in sus[] (entries are never removed)
  uint32_t pad2;
  Act queue;
  uint16_t nsus;
  uint8_t parent_host;
  uint8_t last_ok;                   // outcome of the last Block/AllowTrafficZone this agent resolved: 1 TRUE, 3 FALSE (T_*), 0 none
};

struct alignas(64) EnvState {
  Rng rng;
  int32_t step_count, steps, phase;
  int32_t phase_len[3];
  uint32_t err;
  float reward;                      // team reward of the last step (BlueRewardMachine + action_cost)
  uint8_t done, rng_mode, n_green;
  uint8_t policy;                    // bits 0-1 red policy (RP_*), bit 4 green policy (1 = SleepAgent)
  uint8_t rng_split;                 // 1 after CybORG.set_seed: the agents' policies keep drawing from rng2 (see there)
  uint8_t obs_dirty;                 // this step changed a slowly varying part of the flat observation (OD_BLOCKS: a pair of the block matrix, OD_PHASE:
                                     // the mission phase): the lane-parallel kernels rewrite those values only then (the output buffer persists)
  uint8_t pad0[2];
  uint16_t blocks[NSUB];             // blocks[to] bit from
  uint8_t cidr_octet[NSUB];
  uint8_t n_users[NSUB];
  alignas(8) uint8_t n_servers[NSUB];   // read as one 8-byte word by GreenAccessService
  uint8_t green_host[MAXG];
  uint32_t pend[MAX_PEND];           // (host<<16)|pid process_creation events not yet seen by Monitor
  uint8_t npend, pad1[3];
  uint32_t exists[5];                // host h is part of this episode's topology
  uint32_t red_hosts[5];             // hosts holding a session of ANY red agent (OR of RedAgent.live_hosts, kept incrementally)
  uint32_t spool_used[RS_POOL / 32]; // which records of spool[] are live
  uint8_t msg[NBLUE][MSG_LEN];       // messages submitted with the last step
  Act bexec[NBLUE];                  // self.action[blue_b][0] of the last step (also CybORG.get_last_action)
  Act rexec[NRED];
  int32_t brm;                       // BlueRewardMachine accumulator
  float action_cost;
  int32_t n_actions;                 // actions surviving filter_actions (length of the shuffled index list)
  int32_t n_restore;                 // Restore actions submitted this step (each costs -1)
  // (CybORG.set_seed's second generator -- the stream the policies keep drawing from until the next reset while rng_split is set --
  // is EnvCold.rng2: read by the numpy-stream kernel only, and only after a set_seed.)
  BlueAgent blue[NBLUE];
  RSess spool[RS_POOL];              // red session records of all six agents
  RedAgent red[NRED];
  // HostEvents of every host as EV_* bits (network_connections / process_creation of this step, and of the last one after the
  // end-turn Monitor rolled them over).  Every step raises a few dozen of them (ev_or), the Monitor rolls all 137 and the
  // observation encode reads all 137 -- in the part of the row that is staged in LDS, these are LDS operations; while they were a
  // byte of each HostDyn row, the kernels that leave the host table in HBM touched all 137 rows every step for them.
  alignas(4) uint8_t hev[MAXH + 39];  // (137 bytes + padding that puts the host table on a 64-byte boundary)
  // 64-byte alignment of the table and of the row (rows are allocated back to back from a 256-byte aligned base): every HostDyn is
  // exactly one cache line of HBM / L2 -- the kernels that leave the table there read and write single rows -- and the staged part
  // in front of it is whole lines, so stage-in and stage-out never share a line with a neighbour
  alignas(64) HostDyn hd[MAXH];      // last member: the numpy-stream kernel stages only the part in front of it
};
static_assert(offsetof(EnvState, hd) % 64 == 0 && sizeof(EnvState) % 64 == 0 && sizeof(EnvState) == offsetof(EnvState, hd) + sizeof(HostDyn) * MAXH, "the host table closes the row; row and table are whole cache lines");

// Work area of one step / one reset: temporaries the phases hand to each other.  Not part of the episode's state (LDS on the
// device, the caller's stack on the host); everything in it is dead between steps.
struct alignas(4) StepWork {        // (word accesses only; 404 bytes, not padded to 416: LDS is allocated in 1280-byte granules and the kernels sit at the edges)
  uint32_t scratch[64];              // ordered (lane 0) sections: small temporaries that would otherwise be dynamically
                                     // indexed private arrays (= scratch memory on the device)
  uint32_t phish_mask[4];            // bit g: green g's LocalWork asked for a PhishingEmail this step (word 3 unused)
  uint32_t pend_r[NRED];             // this step's pid-carrying event of red agent r (0 = none); merged into pend[] in agent order
  uint8_t rs_slot[NRED + 2];         // pool slot reserved for the session red agent r's exploit may create this step (rs_reserve)
  uint8_t green_act[MAXG];           // this step's green choice
  uint32_t hdirty[5];                // bit h: the HostDyn row of host h was written this step (hd_touch): the kernel that stages the host
                                     // table writes back only these rows -- one cache line each -- and a step touches a handful of the 137
};

// optional per-step event log (decoded Monitor observations with ports / peers / pids).  One record per
// HostEvents entry the step produced; the engine itself only needs the flags in HostDyn.ev.
enum : int { MAX_EV = 160 };
struct alignas(4) EvRec {
  uint8_t host;        // host whose events list receives the entry
  uint8_t kind;        // 0 network_connections, 1 process_creation
  uint8_t laddr;       // host whose address is the event's local_address
  uint8_t raddr;       // host whose address is the remote_address (0xFF: none)
  uint16_t lport;      // local_port (0: none)
  uint16_t rport;      // remote_port (0: none)
  uint16_t pid;        // pid (0: none)
  uint8_t rep;         // the entry is appended this many times (SSHBruteForce: 10)
  uint8_t order;       // acting agent in execution order: green host id (0..136), red 200 + r
};
struct alignas(16) EvLog { uint32_t n, step, enabled, pad; EvRec rec[MAX_EV]; };

// The cold row of an episode: a fixed part (this struct) followed by two containers whose capacity depends on the episode length
// the handle was created for (EnterpriseScenarioGenerator(steps=...)):
//   uint32_t sus[NBLUE][cold_sus_cap(steps)]    VelociraptorServer.sus_pids of blue agent b: (host << 16) | pid, chronological
//                                               (appended by Monitor, read by Remove; counts and per-host presence stay hot)
//   Proc     povf[MAXH][cold_povf_cap(steps)]   entries PIN.. of each host's process list (HostDyn.nproc > PIN)
// Rows are cold_row_bytes(steps) apart (cold_at).  The capacities are pure functions of `steps`, and every use reads `steps`
// from the episode's own row (EnvState.steps: LDS on the device), so no kernel carries them in registers; cc4_set_state refuses
// a row whose `steps` would give other capacities than the handle's.
struct alignas(16) EnvCold {
  HostStatic hs[MAXH];               // backup images (Host.create_backup): read by Restore and reset only
  uint8_t hs_pad[8];                 // keeps what follows 16-byte aligned (137 * 56 + 8 = 7680)
  uint32_t eph[MAXH][EPH_WORDS];     // Host.ephemeral_ports as a bitmap (port-49152)
  EvLog evlog;                       // events of the last step when EvLog.enabled (cc4_enable_event_log)
  uint8_t kports[RS_POOL][MAXH + 7]; // RedAbstractSession.ports[ip]: PB_* bits | PB_HAS; row = pool slot of the session
  double xrate[NRED][2];             // the probability attributes an externally submitted red action was queued with (ExtAct.rate0 / rate1:
                                     // valid while the agent's queued / executing Act carries AQ_RATE0 / AQ_RATE1)
  uint32_t gfail[4];                 // bit g: green_agent_g's action of the last step returned Observation(False) (what CybORG.step /
                                     // parallel_step report as its 'success'; written by the full builds of the step only; word 3 unused)
  uint16_t known_sid[NRED][MAX_KS];  // ActionSpace.server_session keys with value True of red agent r, insertion order; entries
                                     // [0, RedHdr.nknown) are valid (RedAgent.known_bm holds the ids < 256 as a bitmap)
  // CybORG.set_seed (env.py:316-325) hands the new Generator to the controller, the state and the hosts
  // (SimulationController.set_np_random, SC:317-320; State.set_np_random, State.py:241-251) but not to the agent objects,
  // whose np_random was bound when they were created (SC:1041): until the next reset the green / red POLICIES keep drawing
  // from the old stream while everything else draws from the new one.  rng2 is that old stream while EnvState.rng_split is set
  // (numpy-stream mode; the counter mode, which is not bit-comparable with the reference anyway, re-keys all streams).
  Rng rng2;
};
static_assert(sizeof(EnvCold) % 16 == 0, "the containers behind the fixed part start 16-byte aligned");
// Suspicious pids per blue agent: one per pid-carrying process_creation event in its zone, i.e. per successful red exploit
// (ExploitAction.py:264-275); an exploit takes four ticks, so six red agents produce at most 1.5 * steps (500 steps -> 768).
CC4_HD int cold_sus_cap(int steps) { const int c = (3 * steps) / 2 + 18; return (c + 15) & ~15; }
// Process slots per host behind the PIN inline ones: steps / 2 decoys + 7 generated + 127 red shells (500 steps -> 376).
CC4_HD int cold_povf_cap(int steps) { const int c = steps / 2 + 126; return (c + 7) & ~7; }
CC4_HD size_t cold_row_bytes(int steps) {
  return sizeof(EnvCold) + 4u * ((size_t)NBLUE * (size_t)cold_sus_cap(steps) + (size_t)MAXH * (size_t)cold_povf_cap(steps));
}
CC4_HD EnvCold* cold_at(EnvCold* base, size_t e, size_t row_bytes) { return reinterpret_cast<EnvCold*>(reinterpret_cast<char*>(base) + e * row_bytes); }
CC4_HD const EnvCold* cold_at(const EnvCold* base, size_t e, size_t row_bytes) { return reinterpret_cast<const EnvCold*>(reinterpret_cast<const char*>(base) + e * row_bytes); }
CC4_HD uint32_t* cold_sus(EnvCold* c, int steps, int b) { return reinterpret_cast<uint32_t*>(c + 1) + (size_t)b * (size_t)cold_sus_cap(steps); }
CC4_HD const uint32_t* cold_sus(const EnvCold* c, int steps, int b) { return reinterpret_cast<const uint32_t*>(c + 1) + (size_t)b * (size_t)cold_sus_cap(steps); }
CC4_HD uint32_t* cold_povf(EnvCold* c, int steps, int h) {   // one Proc = one word: pid | kind << 16 | flags << 24
  return reinterpret_cast<uint32_t*>(c + 1) + (size_t)NBLUE * (size_t)cold_sus_cap(steps) + (size_t)h * (size_t)cold_povf_cap(steps);
}
CC4_HD const uint32_t* cold_povf(const EnvCold* c, int steps, int h) {
  return reinterpret_cast<const uint32_t*>(c + 1) + (size_t)NBLUE * (size_t)cold_sus_cap(steps) + (size_t)h * (size_t)cold_povf_cap(steps);
}

}  // namespace cc4
