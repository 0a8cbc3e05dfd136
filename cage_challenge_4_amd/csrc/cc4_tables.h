// cc4_tables.h -- constant tables of the CC4 scenario (data, shared by host and device builds).
// Every table cites the reference lines it restates (paths under /root/reference/CybORG).
#pragma once
#include "cc4_state.h"

namespace cc4 {

#if defined(__HIPCC__)
#define CC4_CONST static __device__ __host__ constexpr
#else
#define CC4_CONST static constexpr
#endif

// alphabetical subnet order used by the wrappers (sorted(state.subnet_name_to_cidr.items()),
// Agents/Wrappers/BlueFlatWrapper.py:195, BlueFixedActionWrapper.py:241): admin, contractor, internet, office,
// operational_a, operational_b, public_access, restricted_a, restricted_b
CC4_HD int sorted_subnet(int i) {
  const uint8_t t[NSUB] = {S_ADM, S_CON, S_INT, S_OFF, S_OZA, S_OZB, S_PUB, S_RZA, S_RZB};
  return t[i];
}
CC4_HD int subnet_rank(int s) {
  const uint8_t t[NSUB] = {7, 4, 8, 5, 1, 6, 0, 3, 2};
  return t[s];
}

// blue zones, EnterpriseScenarioGenerator.py:643-649 (allowed_subnets order) -- bitmask + ordered list
CC4_HD int blue_nsub(int b) { return b == 4 ? 3 : 1; }
CC4_HD int blue_subnet_alloc(int b, int i) {  // allowed_subnets order (session creation order)
  const uint8_t t4[3] = {S_PUB, S_ADM, S_OFF};
  const uint8_t t[4] = {S_RZA, S_OZA, S_RZB, S_OZB};
  return b == 4 ? t4[i] : t[b];
}
CC4_HD int blue_subnet_sorted(int b, int i) {  // sorted(subnets): wrappers' obs/action order
  const uint8_t t4[3] = {S_ADM, S_OFF, S_PUB};
  const uint8_t t[4] = {S_RZA, S_OZA, S_RZB, S_OZB};
  return b == 4 ? t4[i] : t[b];
}
CC4_HD int blue_of_subnet(int s) {
  const int8_t t[NSUB] = {0, 1, 2, 3, -1, 4, 4, 4, -1};
  return t[s];
}
// red zones, EnterpriseScenarioGenerator.py:769-776
CC4_HD int red_of_subnet(int s) {
  const int8_t t[NSUB] = {1, 2, 3, 4, 0, 5, 5, 5, -1};
  return t[s];
}
CC4_HD uint32_t red_allowed_mask(int r) {
  const uint16_t t[NRED] = {1u << S_CON, 1u << S_RZA, 1u << S_OZA, 1u << S_RZB, 1u << S_OZB,
                            (1u << S_PUB) | (1u << S_ADM) | (1u << S_OFF)};
  return t[r];
}
CC4_HD int red_nsub(int r) { return r == 5 ? 3 : 1; }
CC4_HD int red_subnet_alloc(int r, int i) {
  const uint8_t t5[3] = {S_PUB, S_ADM, S_OFF};
  const uint8_t t[5] = {S_CON, S_RZA, S_OZA, S_RZB, S_OZB};
  return r == 5 ? t5[i] : t[r];
}

// router tree, EnterpriseScenarioGenerator.py:388-411 : parent subnet of each subnet's router (internet = root)
CC4_HD int router_parent(int s) {
  const uint8_t t[NSUB] = {S_INT, S_RZA, S_INT, S_RZB, S_INT, S_INT, S_PUB, S_PUB, S_INT};
  return t[s];
}
// server_host_0 info links, EnterpriseScenarioGenerator.py:431-462 (bitmask of target subnets)
CC4_HD uint32_t info_links(int s) {
  const uint16_t t[NSUB] = {
      (1u << S_OZA) | (1u << S_CON), (1u << S_RZA), (1u << S_OZB) | (1u << S_CON), (1u << S_RZB),
      (1u << S_RZA) | (1u << S_RZB) | (1u << S_PUB), (1u << S_ADM) | (1u << S_OFF) | (1u << S_CON),
      (1u << S_PUB), (1u << S_PUB), 0};
  return t[s];
}

// green allowed subnets per mission phase, EnterpriseScenarioGenerator.py:281-306 + SimulationController.py:747-765
CC4_HD uint32_t green_allowed_mask(int phase, int s) {
  // undirected pair lists
  const uint8_t common[9][2] = {{S_PUB, S_CON}, {S_ADM, S_CON}, {S_OFF, S_CON}, {S_PUB, S_RZA}, {S_ADM, S_RZA},
                                {S_OFF, S_RZA}, {S_PUB, S_RZB}, {S_ADM, S_RZB}, {S_OFF, S_RZB}};
  const uint8_t p0[5][2] = {{S_RZA, S_CON}, {S_OZA, S_RZA}, {S_RZB, S_CON}, {S_RZB, S_RZA}, {S_OZB, S_RZB}};
  const uint8_t p1[2][2] = {{S_RZB, S_CON}, {S_OZB, S_RZB}};
  const uint8_t p2[2][2] = {{S_RZA, S_CON}, {S_OZA, S_RZA}};
  uint32_t m = 1u << s;
  for (int i = 0; i < 9; ++i) {
    if (common[i][0] == s) m |= 1u << common[i][1];
    if (common[i][1] == s) m |= 1u << common[i][0];
  }
  if (phase == 0) for (int i = 0; i < 5; ++i) { if (p0[i][0] == s) m |= 1u << p0[i][1]; if (p0[i][1] == s) m |= 1u << p0[i][0]; }
  if (phase == 1) for (int i = 0; i < 2; ++i) { if (p1[i][0] == s) m |= 1u << p1[i][1]; if (p1[i][1] == s) m |= 1u << p1[i][0]; }
  if (phase == 2) for (int i = 0; i < 2; ++i) { if (p2[i][0] == s) m |= 1u << p2[i][1]; if (p2[i][1] == s) m |= 1u << p2[i][0]; }
  return m;
}

// BlueRewardMachine.get_phase_rewards, Shared/BlueRewardMachine.py:35-65 : [phase][subnet][LWF, ASF, RIA]
CC4_HD int reward_table(int phase, int s, int what) {
  const int8_t t[3][NSUB][3] = {
      {{-1, -3, -1}, {-1, -1, -1}, {-1, -3, -1}, {-1, -1, -1}, {0, -5, -5}, {-1, -1, -3}, {-1, -1, -3}, {-1, -1, -3}, {0, 0, -1}},
      {{-2, -1, -3}, {-10, 0, -10}, {-1, -1, -1}, {-1, -1, -1}, {0, 0, 0}, {-1, -1, -3}, {-1, -1, -3}, {-1, -1, -3}, {0, 0, 0}},
      {{-1, -3, -3}, {-1, -1, -1}, {-2, -1, -3}, {-10, 0, -10}, {0, 0, 0}, {-1, -1, -3}, {-1, -1, -3}, {-1, -1, -3}, {0, 0, 0}}};
  return t[phase][s][what];
}
enum : int { RW_LWF = 0, RW_ASF = 1, RW_RIA = 2 };

// comms-policy graph, Agents/Wrappers/BlueFlatWrapper.py:267-302: adjacency bitmask over subnets for `s`
CC4_HD uint32_t comms_adjacent(int phase, int s) {
  const uint32_t core = (1u << S_INT) | (1u << S_ADM) | (1u << S_OFF) | (1u << S_PUB) | (1u << S_CON) | (1u << S_RZA) | (1u << S_RZB);
  uint32_t m = 0;
  if (core & (1u << s)) m = core & ~(1u << s);
  if (s == S_RZA) m |= 1u << S_OZA;
  if (s == S_OZA) m |= 1u << S_RZA;
  if (s == S_RZB) m |= 1u << S_OZB;
  if (s == S_OZB) m |= 1u << S_RZB;
  if (phase == 1) {  // MissionA: remove (rza,oza) (rza,con) (rza,rzb) (rza,int)
    const uint32_t cut = (1u << S_OZA) | (1u << S_CON) | (1u << S_RZB) | (1u << S_INT);
    if (s == S_RZA) m &= ~cut;
    if (cut & (1u << s)) m &= ~(1u << S_RZA);
  }
  if (phase == 2) {  // MissionB
    const uint32_t cut = (1u << S_OZB) | (1u << S_CON) | (1u << S_RZA) | (1u << S_INT);
    if (s == S_RZB) m &= ~cut;
    if (cut & (1u << s)) m &= ~(1u << S_RZB);
  }
  return m;
}

// listening port bit of a process kind (EnterpriseScenarioGenerator.py:597-604, Decoy*.py PORT constants;
// VsftpdDecoyFactory.PORT = 80, DecoyVsftpd.py:10)
CC4_HD int kind_port(int kind) {
  const uint8_t t[13] = {PB_22, PB_1, PB_80, PB_3390, PB_25, PB_80, PB_443, PB_25, PB_80, 0, 0, 0, 0};
  return t[kind];
}
CC4_HD bool kind_is_decoy(int kind) { return kind >= K_DEC_APACHE && kind <= K_DEC_VSFTPD; }

// action durations
CC4_HD int blue_duration(int t) {
  const uint8_t d[8] = {1, 1, 2, 3, 5, 2, 1, 1};
  return d[t];
}
CC4_HD int red_duration(int t) {
  const uint8_t d[12] = {1, 1, 3, 2, 4, 2, 2, 2, 1, 1, 1, 1};
  return d[t];
}

}  // namespace cc4
