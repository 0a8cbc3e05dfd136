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
// (small tables are nibble/byte-packed into immediates: a dynamically indexed local array would be a memory load
// on the device, and these sit inside the per-agent loops)
CC4_HD constexpr int sorted_subnet(int i) { return (int)((0x205317846ull >> (4 * i)) & 0xF); }   // {ADM,CON,INT,OFF,OZA,OZB,PUB,RZA,RZB}
CC4_HD int subnet_rank(int s) { return (int)((0x230615847ull >> (4 * s)) & 0xF); }     // inverse permutation

// blue zones, EnterpriseScenarioGenerator.py:643-649 (allowed_subnets order) -- bitmask + ordered list
CC4_HD constexpr int blue_nsub(int b) { return b == 4 ? 3 : 1; }
CC4_HD int blue_subnet_alloc(int b, int i) {  // allowed_subnets order (session creation order): b<4 -> subnet b
  return b == 4 ? (i == 0 ? S_PUB : (i == 1 ? S_ADM : S_OFF)) : b;
}
CC4_HD constexpr int blue_subnet_sorted(int b, int i) {  // sorted(subnets): wrappers' obs/action order
  return b == 4 ? (i == 0 ? S_ADM : (i == 1 ? S_OFF : S_PUB)) : b;
}
CC4_HD int blue_of_subnet(int s) {  // {0,1,2,3,-1,4,4,4,-1}
  int v = (int)((0xf444f3210ull >> (4 * s)) & 0xF);
  return v == 0xF ? -1 : v;
}
// red zones, EnterpriseScenarioGenerator.py:769-776
CC4_HD int red_of_subnet(int s) {  // {1,2,3,4,0,5,5,5,-1}
  int v = (int)((0xf55504321ull >> (4 * s)) & 0xF);
  return v == 0xF ? -1 : v;
}
CC4_HD uint32_t red_allowed_mask(int r) {
  return r == 5 ? ((1u << S_PUB) | (1u << S_ADM) | (1u << S_OFF)) : (r == 0 ? (1u << S_CON) : (1u << (r - 1)));
}
CC4_HD int red_nsub(int r) { return r == 5 ? 3 : 1; }
CC4_HD int red_subnet_alloc(int r, int i) {
  return r == 5 ? (i == 0 ? S_PUB : (i == 1 ? S_ADM : S_OFF)) : (r == 0 ? S_CON : r - 1);
}

// router tree, EnterpriseScenarioGenerator.py:388-411 : parent subnet of each subnet's router (internet = root)
CC4_HD int router_parent(int s) { return (int)((0x855882808ull >> (4 * s)) & 0xF); }  // {INT,RZA,INT,RZB,INT,INT,PUB,PUB,INT}
// server_host_0 info links, EnterpriseScenarioGenerator.py:431-462 (bitmask of target subnets)
CC4_HD uint32_t info_links(int s) {
  // {OZA|CON, RZA, OZB|CON, RZB, RZA|RZB|PUB, ADM|OFF|CON, PUB, PUB, 0} as nine 8-bit masks in one immediate
  // (subnet bits RZA=0 OZA=1 RZB=2 OZB=3 CON=4 PUB=5 ADM=6 OFF=7)
  const uint64_t t = 0x12ull | (0x01ull << 8) | (0x18ull << 16) | (0x04ull << 24) | (0x25ull << 32) | (0xD0ull << 40) | (0x20ull << 48) | (0x20ull << 56);
  return s < 8 ? (uint32_t)((t >> (8 * s)) & 0xFF) : 0u;
}

// green allowed subnets per mission phase, EnterpriseScenarioGenerator.py:281-306 + SimulationController.py:747-765
CC4_HD uint32_t green_allowed_mask(int phase, int s) {
  // own subnet + every partner in the phase's pair list (policy_1/2/3), 9-bit masks packed 7 + 2 per constant:
  //   phase 0: {0xf7,0x03,0xfd,0x0c,0xf5,0x35,0x55,0x95,0x100}  phase 1: {0xe1,0x02,0xfc,0x0c,0xf4,...}  phase 2: {0xf3,0x03,0xe4,0x08,0xf1,...}
  uint64_t lo = phase == 0 ? 0x1546af5063f406f7ull : (phase == 1 ? 0x1546af4063f004e1ull : 0x1546af10439006f3ull);
  uint32_t hi = phase == 0 ? 0x20095u : (phase == 1 ? 0x20095u : 0x20095u);
  return s < 7 ? (uint32_t)((lo >> (9 * s)) & 0x1FF) : ((hi >> (9 * (s - 7))) & 0x1FF);
}

// BlueRewardMachine.get_phase_rewards, Shared/BlueRewardMachine.py:35-65 : [phase][subnet][LWF, ASF, RIA]
CC4_HD constexpr int reward_table_fn(int phase, int s, int what) {
  // phase 0: {{-1,-3,-1},{-1,-1,-1},{-1,-3,-1},{-1,-1,-1},{0,-5,-5},{-1,-1,-3},{-1,-1,-3},{-1,-1,-3},{0,0,-1}}
  // phase 1: {{-2,-1,-3},{-10,0,-10},{-1,-1,-1},{-1,-1,-1},{0,0,0},{-1,-1,-3},{-1,-1,-3},{-1,-1,-3},{0,0,0}}
  // phase 2: {{-1,-3,-3},{-1,-1,-1},{-2,-1,-3},{-10,0,-10},{0,0,0},{-1,-1,-3},{-1,-1,-3},{-1,-1,-3},{0,0,0}}
  // as 27 nibble codes per phase (entry s*3+what; code 0..5 = 0,-1,-2,-3,-5,-10) in immediates: a const array would be a
  // constant-memory load on the device, inside the green-action and end-of-step paths
  const uint64_t lo = phase == 0 ? 0x1440111131111131ull : (phase == 1 ? 0x1000111111505312ull : 0x1000505312111331ull);
  const uint64_t hi = phase == 0 ? 0x10031131131ull : 0x31131131ull;
  const int e = s * 3 + what;
  const int c = (int)(((e < 16 ? lo >> (4 * e) : hi >> (4 * (e - 16)))) & 0xF);
  return c < 4 ? -c : (c == 4 ? -5 : -10);
}
enum : int { RW_LWF = 0, RW_ASF = 1, RW_RIA = 2 };
#if defined(__HIP_DEVICE_COMPILE__)
// the device reads the 81 values from a table the compiler fills from the function above (a byte load instead of two 64-bit selects and shifts and a
// decode: ~20 vector instructions on the green actions' failure paths of nearly every step; the step kernels are bound by vector issue slots)
struct RewardTab { int8_t v[3 * NSUB * 3]; };
constexpr RewardTab make_reward_tab() {
  RewardTab t{};
  for (int p = 0; p < 3; ++p) for (int s = 0; s < NSUB; ++s) for (int w = 0; w < 3; ++w) t.v[(p * NSUB + s) * 3 + w] = (int8_t)reward_table_fn(p, s, w);
  return t;
}
static __device__ const RewardTab reward_tab = make_reward_tab();
__device__ __forceinline__ int reward_table(int phase, int s, int what) { return reward_tab.v[(phase * NSUB + s) * 3 + what]; }
#else
CC4_HD int reward_table(int phase, int s, int what) { return reward_table_fn(phase, s, what); }
#endif

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
CC4_HD int kind_port(int kind) {  // {22,1,80,3390,25, 80,443,25,80, -,-,-,-} as PB_* bits, one byte per kind
  return kind < 8 ? (int)((0x0820020804021001ull >> (8 * kind)) & 0xFF) : (kind == 8 ? PB_80 : 0);
}
CC4_HD bool kind_is_decoy(int kind) { return kind >= K_DEC_APACHE && kind <= K_DEC_VSFTPD; }

// action durations
CC4_HD int blue_duration(int t) { return (int)((0x11253211u >> (4 * t)) & 0xF); }        // {1,1,2,3,5,2,1,1}
CC4_HD int red_duration(int t) { return (int)((0x111122242311ull >> (4 * t)) & 0xF); }    // {1,1,3,2,4,2,2,2,1,1,1,1}

}  // namespace cc4
