// cc4_export.h -- true-state export of one episode (host code): the packed EnvState row decoded into a JSON document
// with the content of the reference's CybORG.get_true_state() that the simulator tracks (State.get_true_state,
// Simulator/State.py:150-224; consumer: Agents/Wrappers/TrueStateWrapper.py:25-243).
//
//   {"step":t,"phase":p,"done":0/1,"reward":team reward of the last step (BlueRewardMachine + action_cost),"action_cost":its
//    action-cost part (SimulationController.py:303-311),"blocks":[9 masks: bit f of blocks[to] = traffic from subnet f to subnet `to` is blocked],
//    "cidr":[9 third octets: subnet s is 10.0.X.0/24], "n_green":g,
//    "hosts":[{"h":host id (subnet*17+slot, 136 = internet root),"ip":last octet,
//              "procs":[[pid,kind,root]...] (Host.processes order),
//              "svcs":[[kind,active,reliability percent,pid]...] (Host.services order), "ev":event bits,
//              "files":HF_* bits (1 cmd.sh, 2 escalate.sh, 4 escalate.sh appended last),
//              "blue":pid of the blue session process (0 none),"green":pid of the green session process (0 none)}...],
//    "red":[{"active":0/1,"start":host the scenario generator drew for the agent,"sessions":[[id,host,pid,flags]...] (state.sessions[red_agent_r] order; flags = RS_* bits),
//            "obs_success":TernaryEnum of the last step's observations[0] (1 TRUE 2 UNKNOWN 3 FALSE 4 IN_PROGRESS),"obs_action":[RA_* type,host,subnet] of it (11 = none),
//            "rsc_listed":1 if the end-of-turn RedSessionCheck listed every session,"busy":1 while a multi-tick action is queued,
//            "new_session":[host (255 none), id] the session this step's exploit created,
//            "obs":[[host, OE_* flags]...] the keys of the last step's combined observation in insertion order (OE_KEY_IP 1: keyed by ip, else by hostname;
//            OE_SESS 2 / OE_IFACE 4 / OE_SYSHN 8: it holds 'Sessions' / 'Interface' / 'System info'),
//            "known_sessions":[ids],"as_subnet":mask,"as_ip":[5 words],"as_hostname":[5 words] the ActionSpace's True entries}...x6],
//    "green_fail":[3 words: bit g = green_agent_g's action of the last step returned success False (full builds of the step: cc4_step_ex / event log)],
//    "blue":[{"parent":host id of the VelociraptorServer,"busy":1 while a multi-tick action is in progress,"traffic_ok":outcome of the last Block/Allow (1 TRUE, 3 FALSE),"sus":[[host,pid]...]}...x5],
//    "last_blue":[[BA_* type,host or to-subnet,from-subnet]...x5],"last_red":[[RA_* type,host,subnet,executed (0 = dropped by filter_actions)]...x6] (the actions
//    that resolved in the last step, i.e. CybORG.get_last_action),
//    "events":[[order, seq, host, kind (0 network_connections / 1 process_creation / 2 decoy deployed: local_address host = K_* kind),
//               local_address host, local_port,
//               remote_address host (255 none), remote_port, pid, repeat]...] (only with cc4_enable_event_log; 0 = absent field),
//    "green_hosts":[host id of green_agent_g...]}    hosts[].os: OSDistribution 0 UBUNTU / 1 KALI
#pragma once
#include <stdio.h>
#include <string>
#include "cc4_engine.h"

namespace cc4 {

// process i of host h: the hot part of the list, then the cold part (cold_povf)
inline Proc export_proc(const EnvState& s, const EnvCold& c, int h, int i) {
  if (i < PIN) return s.hd[h].procs[i];
  Proc p; __builtin_memcpy(&p, cold_povf(&c, s.steps, h) + (i - PIN), sizeof(Proc));
  return p;
}

inline std::string export_true_state(const EnvState& s, const EnvCold& cold, bool with_log = true) {
  const HostStatic* hs = cold.hs;
  const uint32_t* sus[NBLUE];
  for (int b = 0; b < NBLUE; ++b) sus[b] = cold_sus(&cold, s.steps, b);
  const EvLog* lg = with_log ? &cold.evlog : nullptr;
  std::string o;
  char b[256];
  auto add = [&](const char* fmt, auto... a) { snprintf(b, sizeof(b), fmt, a...); o += b; };
  add("{\"step\":%d,\"phase\":%d,\"done\":%d,\"reward\":%.9g,\"action_cost\":%.9g,\"blocks\":[", s.step_count, s.phase, (int)s.done, (double)s.reward, (double)s.action_cost);
  for (int i = 0; i < NSUB; ++i) add("%s%u", i ? "," : "", (unsigned)s.blocks[i]);
  o += "],\"cidr\":[";
  for (int i = 0; i < NSUB; ++i) add("%s%u", i ? "," : "", (unsigned)s.cidr_octet[i]);
  add("],\"n_green\":%d,\"hosts\":[", (int)s.n_green);
  bool first = true;
  for (int h = 0; h < MAXH; ++h) {
    if (!bit_get(s.exists, h)) continue;
    const HostDyn& d = s.hd[h];
    add("%s{\"h\":%d,\"ip\":%u,\"os\":%u,\"procs\":[", first ? "" : ",", h, (unsigned)hs[h].ip_octet, (unsigned)((hs[h].exists >> 1) & 1));
    first = false;
    unsigned blue_pid = 0, green_pid = 0;   // the session processes Host.add_session created for the blue / green agent
    for (int i = 0; i < d.nproc; ++i) {
      const Proc pr = export_proc(s, cold, h, i);
      add("%s[%u,%u,%u]", i ? "," : "", (unsigned)pr.pid, (unsigned)pr.kind, (unsigned)(pr.flags & PF_ROOT));
      if (pr.kind == K_SESS_BLUE) blue_pid = pr.pid;
      if (pr.kind == K_SESS_GREEN) green_pid = pr.pid;
    }
    o += "],\"svcs\":[";
    for (int i = 0; i < hd_nsvc(d); ++i)
      add("%s[%u,%u,%u,%u]", i ? "," : "", (unsigned)d.svcs[i].kind, (unsigned)((d.svcs[i].st & SV_ACTIVE) ? 1 : 0), (unsigned)(d.svcs[i].st & 0x7F) * 20u, (unsigned)d.svcs[i].pid);
    add("],\"ev\":%u,\"files\":%u,\"blue\":%u,\"green\":%u}", (unsigned)s.hev[h], (unsigned)hd_files(d), blue_pid, green_pid);
  }
  o += "],\"red\":[";
  for (int r = 0; r < NRED; ++r) {
    const RedAgent& A = s.red[r];
    add("%s{\"active\":%u,\"start\":%u,\"sessions\":[", r ? "," : "", (unsigned)A.h.active, (unsigned)A.h.start_host);
    for (int i = 0; i < A.h.nsess; ++i) { const RSess& q = s.spool[A.sord[i]]; add("%s[%u,%u,%u,%u]", i ? "," : "", (unsigned)q.id, (unsigned)q.host, (unsigned)q.pid, (unsigned)q.flags); }
    // the agent's observation of the last step (what CybORG.get_observation('red_agent_r') is built from) and its ActionSpace
    add("],\"obs_success\":%u,\"obs_action\":[%u,%u,%u],\"rsc_listed\":%u,\"busy\":%u,\"new_session\":[%u,%u],\"obs\":[", (unsigned)A.h.obs_success, (unsigned)A.h.obs_act_type,
        (unsigned)A.h.obs_act_host, (unsigned)A.h.obs_act_arg, (unsigned)A.h.rsc_listed, (unsigned)(A.h.queue.busy ? 1 : 0), (unsigned)A.h.new_sess_host, (unsigned)A.h.new_sess_id);
    for (int i = 0; i < A.h.nobs; ++i) add("%s[%u,%u]", i ? "," : "", (unsigned)A.obs[i].host, (unsigned)A.obs[i].flags);
    o += "],\"known_sessions\":[";
    for (int i = 0; i < A.h.nknown; ++i) add("%s%u", i ? "," : "", (unsigned)cold.known_sid[r][i]);
    add("],\"as_subnet\":%u,\"as_ip\":[%u,%u,%u,%u,%u],\"as_hostname\":[%u,%u,%u,%u,%u]", (unsigned)A.h.as_subnet, A.as_ip[0], A.as_ip[1], A.as_ip[2], A.as_ip[3], A.as_ip[4],
        A.as_hn[0], A.as_hn[1], A.as_hn[2], A.as_hn[3], A.as_hn[4]);
    o += "}";
  }
  o += "],\"blue\":[";
  for (int k = 0; k < NBLUE; ++k) {
    add("%s{\"parent\":%u,\"busy\":%u,\"traffic_ok\":%u,\"sus\":[", k ? "," : "", (unsigned)s.blue[k].parent_host, (unsigned)(s.blue[k].queue.busy ? 1 : 0), (unsigned)s.blue[k].last_ok);
    for (int i = 0; i < s.blue[k].nsus; ++i) add("%s[%u,%u]", i ? "," : "", (unsigned)(sus[k][i] >> 16), (unsigned)(sus[k][i] & 0xFFFF));
    o += "]}";
  }
  o += "],\"last_blue\":[";   // self.action[blue_agent_b][0] of the last step: [BA_* type, host (or to-subnet), arg (from-subnet)]
  for (int k = 0; k < NBLUE; ++k) add("%s[%u,%u,%u]", k ? "," : "", (unsigned)s.bexec[k].type, (unsigned)s.bexec[k].host, (unsigned)s.bexec[k].arg);
  o += "],\"last_red\":[";    // self.action[red_agent_r][0]: [RA_* type, host, arg (subnet of DiscoverRemoteSystems), executed]
  for (int r = 0; r < NRED; ++r) add("%s[%u,%u,%u,%u]", r ? "," : "", (unsigned)s.red[r].h.exec_type, (unsigned)s.red[r].h.exec_host, (unsigned)s.rexec[r].arg, (unsigned)(s.rexec[r].type != RA_NONE));
  if (lg && lg->enabled) {   // the HostEvents entries of the last step (cc4_enable_event_log), in append order
    add("],\"events_step\":%u,\"events_total\":%u,\"events\":[", lg->step, lg->n);
    const uint32_t n = lg->n < (uint32_t)MAX_EV ? lg->n : (uint32_t)MAX_EV;
    for (uint32_t i = 0; i < n; ++i) {
      const EvRec& e = lg->rec[i];
      add("%s[%u,%u,%u,%u,%u,%u,%u,%u,%u,%u]", i ? "," : "", (unsigned)e.order, i, (unsigned)e.host, (unsigned)e.kind, (unsigned)e.laddr,
          (unsigned)e.lport, (unsigned)e.raddr, (unsigned)e.rport, (unsigned)e.pid, (unsigned)e.rep);
    }
  }
  add("],\"green_fail\":[%u,%u,%u", cold.gfail[0], cold.gfail[1], cold.gfail[2]);
  o += "],\"green_hosts\":[";
  for (int g = 0; g < s.n_green; ++g) add("%s%u", g ? "," : "", (unsigned)s.green_host[g]);
  o += "]}";
  return o;
}

}  // namespace cc4
