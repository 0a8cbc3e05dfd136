/* cc4_debug.h -- debug, experiment and test hooks of libcc4.so.  Not part of the drop-in boundary (include/cc4.h): nothing a user of the reference's
 * API needs; tools/ and tests/ call them.  The experiment cc4_debug_policy_probe exists only in libraries built with -DCC4_POLICY_PROBE
 * (tools/policy_group_probe.py; DESIGN 3.4: a concluded experiment of r05). */
#ifndef CC4_DEBUG_H
#define CC4_DEBUG_H
#include "cc4.h"
#ifdef __cplusplus
extern "C" {
#endif

/* debug: per-episode cycle counters of the step kernels ([N][128] u64: 16 phase slots, 8 per red agent, then (cycles, count) per red action type; see
 * tools/phase_profile.py, tools/tail_profile.py) */
int cc4_debug_profile(cc4_handle* h, int enable, unsigned long long* out);
/* debug / experiment: the red policy phase of every episode with the agents of G = 1 / 2 / 4 / 8 episodes side by side on one wave, on the
 * batch as it stands (nothing is written back).  out[0] = mean launch duration (us), out[1] = mean cycles of a wave in the phase, out[2] = waves per launch. */
int cc4_debug_policy_probe(cc4_handle* h, int32_t G, int32_t reps, double* out);

/* test hooks: keep the gathered rows of the next `steps` steps exchanged from inside a one-launch kernel (0 frees the log); read `count`
 * of them from step `first` as [count][world*N][CC4_OBS_PACKED_BYTES]; returns the number of steps logged so far. */
int cc4_debug_gather_log(cc4_handle* h, int32_t steps);
int cc4_get_gather_log(cc4_handle* h, uint8_t* out, int32_t first, int32_t count);

/* test hook: from now on every all-gather is preceded, on the communication stream, by a kernel that idles for about `us`
 * microseconds (0: off) -- an exchange slower than the steps, which makes the guard of the observation ring actually wait */
int cc4_debug_comm_delay_us(cc4_handle* h, int us);

/* measurement (tools/valu_phases.py): the per-step launches of k_step_philox1 end behind phase `phase` (1..13: csrc/cc4_philox1_body.h CC4_STOP) and write no
 * row back; 14 = whole steps, still of the full build; 0 = off.  Restore the batch (cc4_set_state / cc4_set_cold) after such a step. */
int cc4_debug_stop_phase(cc4_handle* h, int phase);

/* debug: where a rollout stands (see csrc/cc4_api.hip) */
int cc4_debug_rollout_state(cc4_handle* h, int64_t* out /* [22] */);
/* test hook: `bytes` bytes of device memory of this handle's device -- e.g. an action slot of a rollout (cc4_rollout_actions), packed observation rows
 * (cc4_rollout_obs_packed) -- copied to the host, behind everything enqueued on the handle's streams. */
int cc4_debug_copy_from_device(cc4_handle* h, void* host_dst, const void* device_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
