// Kernel argument blocks and launchers shared between mdr_kernels.hip and the C-ABI host side (mdr_api.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/mdr.h"

namespace mdr {

// Episode start (sampling or replay): everything k_sample_* / k_load_* / k_env_max_power need.
struct EpisodeArgs {
  mdr_buffers_t b;
  int E, N, dt;
  int64_t env_offset, house_offset;
  uint32_t k0, k1, episode;
  double temp_ref;
  double init_air, init_mass, target, deadband, Ua, Cm, Ca, Hm, COP, latent;
  double std_start, std_target, f_low, f_high;
  int lockout, lockout_noise, ncaps;
  double caps[MDR_MAX_CAPACITIES];
  int start_random, random_phase;
  int64_t start_epoch;
  double artificial_ratio, ratio_range;
};

// Time tables for rows [j0, j0 + rows)
struct TableArgs {
  float* tab_od;
  float* tab_solar;
  double* tab_signal;
  double* tab_abs_noise;      // optional
  const int64_t* t0;
  const double* phase;
  const double* ratio;
  const double* max_power;
  const double* od_ext;
  int64_t od_ext_rows;
  int64_t j0;
  int rows, E, dt;
  int64_t env_offset;
  uint32_t k0, k1, episode;
  double temp_ref;
  double day_temp, night_temp, temp_std;
  int solar_on;
  double area_shading;
  int64_t n_total;
  double avg_power_per_hvac;
  const double* base_power;   // [E] per-env base power (interpolation mode) or nullptr (constant mode)
  int signal_mode, nb_sin, perlin_octaves;
  double sin_periods[MDR_MAX_SINUSOIDS], sin_ratios[MDR_MAX_SINUSOIDS];
  double steps_amp, steps_period;
  double perlin_amp, perlin_step, perlin_period;
};

// PowerGrid.interpolatePower on the device (k_interp_base)
struct InterpArgs {
  const double* values;
  int dims[MDR_INTERP_AXES];
  double axes[MDR_INTERP_AXES][MDR_INTERP_MAX_AXIS];
  const float *Ta, *Tm, *target, *Ua, *Cm, *Ca, *Hm, *capacity;
  const float* od_now;      // table row of the current time index (OD temp minus temp_ref)
  const int64_t* t0;
  double* base_power;       // [E] out
  int E, N, dt, nb_agents, solar_on;
  int N_total, house_offset;   // sharded houses: this device holds houses [house_offset, house_offset + N) of N_total
  int64_t j;                // time index of the update
  int64_t env_offset;
  uint32_t k0, k1, episode;
  double def_Ua, def_Cm, def_Ca, def_Hm;
};

// One env step
struct StepArgs {
  float *Ta, *Tm;
  int32_t* sso;
  uint8_t* flags;
  const float *k01, *s0, *k10, *s1, *inv_Ua, *Q_hvac, *P_max, *target, *deadband;
  const int32_t* lockout;
  uint8_t* actions;
  float* reward;
  float* obs;
  double *P, *tot_sum, *tot_max, *partials;
  const double* gathered;              // [world][3][E] all-gathered local aggregates (sharded houses), or nullptr
  const double* records;               // [world][E][nblk][3] per-workgroup partial records for the finish kernel to re-sum, or nullptr
  int world;
  const float *od_old, *solar_new;     // table rows for this step: OD temp at time index k-1, solar at k
  const double *sig_old, *sig_new;     // regulation signal at k-1 (reward) and k (observation)
  // graph mode (mdr_buffers_t.cursor): the four pointers above address table row 0 / 1 and the kernels add cursor[0] rows,
  // so that a captured launch keeps stepping through the tables when it is replayed; nullptr = host-computed rows
  const int32_t* cursor;
  int32_t cursor_max;                  // last valid row (table_steps - 1): a graph replayed too often re-reads it instead of running off the tables
  int32_t* cursor_adv;                 // graph mode: the step's LAST kernel moves the cursor on (see cursor_done / the split pair's row notes); else nullptr
  int32_t snap_rd, snap_wr;            // graph mode, split kernels: which row note (0 | 1) the kernel reads / writes
  int32_t cursor_steps;                // graph mode, k_step_finish: steps the cursor moves on by (the steps begun since it last moved)
  float* stash;                        // where k_step_partial leaves each house's own temperature penalty for the finish (`reward`, or mdr_buffers_t.pen_stash)
  int64_t plane;                       // E * N: stride between observation planes
  int E, N, dt, penalty_mode, action_source, nblk;
  float c_temp;                        // alpha_temp / norm_temp_penalty
  float mix_i, mix_c, mix_m;           // mixture weights divided by their sum
  float obs_tshift;                    // temp_ref - 20
  double c_sig;                        // alpha_sig / norm_sig_penalty
  double inv_n_total;                  // 1 / nb_agents
  double inv_obs_norm;                 // 1 / (norm_reg_sig * nb_agents)
};

// utils.normStateDict for all houses (k_obs_vector)
struct ObsArgs {
  const float *Ta, *Tm, *target, *deadband, *P_max, *Ua, *Cm, *Ca, *Hm, *capacity, *COP, *latent;
  const int32_t *sso, *lockout;
  const uint8_t* flags;
  const double* P;          // [E]
  const double* sig_now;    // table row of the current time index
  const float *od_now, *solar_now;
  const int64_t* t0;
  const int32_t* links;     // [N][c] or nullptr (circular neighbours)
  int random_links;         // senders re-drawn per house and step (agents_comm_mode random_sample)
  float* out;
  int64_t plane;            // E * N
  int64_t out_plane;        // stride between output feature planes (>= plane)
  int64_t k;                // steps taken (time index)
  const int32_t* cursor;    // graph mode: {table row, time index} on the device (sig_now / od_now / solar_now address row 0, k unused)
  int32_t cursor_max;
  int E, N, c, F, dt;
  int n_total;              // houses of the whole env (== N unless the houses are sharded): random_sample draws among them
  int f_hour, f_day, f_solar, f_thermal, f_hvac, m_thermal, m_hvac;
  int64_t env_offset, house_offset;
  uint32_t k0, k1, episode;
  float defect_prob;
  // sharded houses: message records [E][ext_entries][mf] = the local houses' messages followed by the halo the caller
  // fetched from the other shards; `links` then holds record slots instead of house ids
  const float* msg_ext_in;
  float* msg_ext_out;
  int64_t ext_entries;
  int mf;                   // floats per message record: 4 + 4 * m_thermal + 3 * m_hvac
  int lds_entries;          // sender entries staged per workgroup by k_obs_planes4 (set by the launcher)
  uint32_t magic_n;         // ((1 << 20) + N - 1) / N: r / N == (r * magic_n) >> 20 for r < 1024 (set by the launcher)
  float obs_tshift;
  double inv_obs_norm;
  float inv_norm_reg, inv_cap, inv_Ua, inv_Cm, inv_Ca, inv_Hm, inv_COP, inv_latent;
};

// Accumulators of the fused multi-step rollout (all optional)
struct RolloutArgs {
  double* power_trace;
  float* reward_sum;
  double* sq_temp_error_sum;
  double* sq_signal_error_sum;
  int nsteps;
  int defer_last_signal_error;   // the last step's reg_signal is not final yet (interpolation update due): the host adds it
};

// Observe -> act without observation rows (mdr_env_actor_sample, csrc/mdr_policy.hip): what the fused policy kernels read
// instead of the rows of mdr_env_obs_vector, for the reference's DEFAULT observation (11 own features + 10 circular
// neighbours x 4 message fields, utils.py:774-878)
struct ObserveArgs {
  const float *Ta, *Tm, *target, *deadband, *capacity, *P_max;
  const int32_t *sso, *lockout;
  const uint8_t* flags;
  const double* P;         // [E] cluster_hvac_power
  const double* sig_now;   // [E] regulation signal of the current time index (graph mode: table row 0, the kernel adds cursor[0] rows)
  const int32_t* cursor;   // graph mode: {table row, time index} on the device, or nullptr
  int32_t cursor_max;
  int E, N;
  float obs_tshift, inv_norm_reg, inv_cap;
  double inv_obs_norm;
  // The extended form (ext != 0): optional STATE columns (utils.py:774-830), any number of circular neighbours with 4-field
  // messages, link defects (env 988-1002).  c neighbours: c / 2 before the house, c - c / 2 after (env 816-828); own = number of
  // own normStateDict features (11 + 5 thermal + 2 hvac + 2 hour + 2 day + 1 solar); a row of the window is
  // [4 c message floats | own features in normStateDict order | zeros | L | 1 / L] = `row` floats.
  int ext, c, before, own;
  int row;                   // floats per staged row (set by the launcher: the forward's reads of a row + L, 1 / L, rounded up to 16 bytes)
  int f_hour, f_day, f_solar, f_thermal, f_hvac;
  const float *Ua, *Cm, *Ca, *Hm, *COP, *latent;       // raw per-house parameters (thermal / hvac columns)
  float inv_Ua, inv_Cm, inv_Ca, inv_Hm, inv_COP, inv_latent;
  // per-env columns, [6][E] floats written by k_observe_env_extras into the handle's tot_sum / tot_max scratch (unsharded handles
  // do not use those between steps): (OD - 20) / 5, sin / cos of the day angle, sin / cos of the hour angle, solar gain / 1000
  float* env_extra_a;        // [4][E]: OD, sin day, cos day, sin hour
  float* env_extra_b;        // [2][E]: cos hour, solar
  const float *od_now, *solar_now;
  const int64_t* t0;
  int64_t k;                 // time index (graph mode: cursor[1])
  int dt;
  float defect_prob;
  int64_t env_offset, house_offset;
  uint32_t k0, k1, episode;
  // Senders by a table instead of the circular neighbours (ext only): links[e * links_env_stride + h * c + m] = house id of slot m's
  // sender within the env - the static table of agents_comm_mode closed_groups / random_fixed / neighbours_2D (stride 0: one table
  // for every env) or this step's random_sample draws (k_comm_draws: stride N c).  msg_rec: every house's SingleHouse.message
  // record [E][N][4] (k_obs_messages of this step), which the staging gathers.
  const int32_t* links;
  int64_t links_env_stride;
  const float* msg_rec;
};

// Persistent sharded rollout (mdr_persist.hip): the mailboxes of every rank, laid out in 8-byte granules as
//   [PERSIST_HDR header | SLOTS x E x world x stride x PERSIST_G record granules | SLOTS x E x PERSIST_TOT totals granules]
constexpr int PERSIST_SLOTS = 16;      // mailbox slots a stream of records / totals cycles through (slot = tag mod SLOTS)
constexpr int PERSIST_MAX_DEPTH = 7;   // steps a house workgroup may run ahead of the totals; a peer rank may be another depth + 1 ahead of this rank's reducer: 2 * depth + 2 <= SLOTS
constexpr int PERSIST_MAX_STEPS = 128;  // steps per launch (their table rows are staged in LDS)
constexpr int PERSIST_G = 5;        // granules per record: power sum lo / hi, penalty sum lo / hi, penalty max
constexpr int PERSIST_TOT = 8;      // granules per totals slot (5 used; 64 bytes)
constexpr int PERSIST_HDR = 16;     // header granules (128 bytes); granule 0 = error word
struct PersistArgs {
  uint64_t* box[MDR_MAX_SHARDS];    // every rank's mailbox as this device sees it (box[rank] = this rank's own)
  int32_t nrec[MDR_MAX_SHARDS];     // records (house workgroups) per env of every rank
  int32_t world, rank, stride;      // stride = record slots per env and rank (>= every nrec)
  uint32_t tag_base;                // tag of this launch's step 0 (steps are counted over the life of the handle)
  uint32_t spin_limit;
  int32_t depth;                    // steps the houses run ahead of the totals, 1 .. PERSIST_MAX_DEPTH
  int32_t reducers;                 // reducer workgroups per env, 1 .. PERSIST_MAX_REDUCERS: reducer j takes the steps s with s % reducers == j
  double* serr_part;                // [reducers][E] squared-signal-error partial sums (reducers > 1; the host adds them up in order)
};
constexpr int PERSIST_MAX_REDUCERS = 4;
int persist_reducers(int64_t records_of_all_ranks);
int64_t persist_mailbox_granules(int E, int world, int stride);
hipError_t persist_resident_blocks(int vec, bool system_scope, int depth, int64_t* blocks);   // workgroups of the kernel the device holds at once
hipError_t launch_rollout_persist(const StepArgs& a, const RolloutArgs& r, const PersistArgs& m, bool system_scope, hipStream_t s);
hipError_t launch_persist_combine(const PersistArgs& m, int E, double* sq_signal_error_sum, hipStream_t s);   // reducers > 1: adds the partial sums

enum StepKind { STEP_FUSED = 0, STEP_GROUP = 1, STEP_SPLIT = 2, STEP_SINGLE = 3, STEP_MULTI = 4, STEP_PACKED = 5 };   // STEP_MULTI: `tiles` envs share a group of `threads` lanes; STEP_PACKED: `tiles` whole envs of `threads` lanes each per wavefront (mdr_multi.hip)
struct StepPlan {
  int kind, vec, threads, tiles;
};

StepPlan plan_step(int N, int64_t E);
StepPlan plan_rollout(int N, int64_t E);
int split_threads(int N, int64_t E);          // workgroup size of the split path for this shape
int64_t split_blocks(int N, int threads);     // workgroups (= partial records) per env

hipError_t launch_sample(const EpisodeArgs& a, hipStream_t s);
hipError_t launch_load(const EpisodeArgs& a, const mdr_episode_t& ep, hipStream_t s);
hipError_t launch_tables(const TableArgs& a, hipStream_t s);
hipError_t launch_interp_base(const InterpArgs& a, hipStream_t s);
hipError_t launch_patch_signal_plane(const StepArgs& a, hipStream_t s);   // obs plane 5 <- sig_old row
hipError_t launch_pack_env(const StepArgs& a, int e, double temp_ref, const double* max_power, const double* ratio,
                           const double* abs_noise_row, double* out, hipStream_t s);
hipError_t launch_cursor_set(int32_t* cursor, int32_t row, int32_t k, hipStream_t s);
hipError_t launch_signal_error(const StepArgs& a, double* sq_signal_error_sum, hipStream_t s);   // += (sig_old - P)^2
hipError_t launch_greedy_myopic(const StepArgs& a, hipStream_t s);   // GreedyMyopic actions of every env -> a.actions (mdr_control.hip); N <= 2048
hipError_t launch_reset_obs(const StepArgs& a, bool zero_reward, hipStream_t s);  // planes of the current state; uses sig_old = the row of the current time index
hipError_t launch_step(const StepArgs& a, const StepPlan& p, hipStream_t s);
int64_t multi_blocks(int64_t E, const StepPlan& p);
hipError_t launch_step_multi(const StepArgs& a, const StepPlan& p, hipStream_t s);
hipError_t launch_rollout_multi(const StepArgs& a, const RolloutArgs& r, const StepPlan& p, hipStream_t s);
hipError_t launch_step_end_begin_split(const StepArgs& finish, const StepArgs& begin, hipStream_t s);   // finish of step k and partial of step k + 1 in ONE launch
bool rollout_fused_supported(const StepPlan& p);
hipError_t launch_rollout_accumulate(const StepArgs& a, const RolloutArgs& ro, hipStream_t s);   // after ONE single step
hipError_t launch_rollout_fused(const StepArgs& a, const RolloutArgs& r, const StepPlan& p, hipStream_t s);
hipError_t launch_obs_vector(const ObsArgs& a, int layout, hipStream_t s);
hipError_t launch_obs_messages(const ObsArgs& a, hipStream_t s);               // msg_ext_out[e][h][:] for the local houses
hipError_t launch_obs_vector_ext(const ObsArgs& a, int layout, hipStream_t s);  // senders read from msg_ext_in through links
hipError_t launch_comm_draws(const ObsArgs& a, int32_t* senders, uint8_t* keep, hipStream_t s);   // per-slot sender ids + keep flags
int obs_message_fields(const mdr_obs_spec_t& s);
// the policy kernels of csrc/mdr_policy.hip on the compact state; actor / step_dev etc. as mdr_actor_sample
int launch_actor_observe(const struct mdr_actor* actor, const ObserveArgs& o, uint64_t seed, uint64_t step, const int32_t* step_dev,
                         uint8_t* action, float* a_prob, float* probs, float* rows_out, hipStream_t s);
int obs_vector_length(const mdr_obs_spec_t& spec);
hipError_t launch_step_begin_split(const StepArgs& a, bool reduce, hipStream_t s);
hipError_t launch_step_end_split(const StepArgs& a, hipStream_t s);

}  // namespace mdr
