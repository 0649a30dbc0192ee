/*
 * mdr.h - C ABI of the MI355X-native batched demand-response environment step.
 *
 * Drop-in boundary for ONE hot path of zhimaerfan/marl-demandresponse-original:
 * MADemandResponseEnv.reset()/step() (env/MA_DemandResponse.py:135-210 of the reference)
 * for E independent environments x N houses at once.  The reference has no FFI of its
 * own (it is pure Python); these entry points are what a ctypes binding inside the
 * reference's env module would call (INTEGRATION.md shows that stub).
 *
 * Conventions
 *   - plain C: pointers, sizes, PODs.  No torch / C++ types cross this boundary.
 *   - every buffer is CALLER-OWNED DEVICE memory (hipMalloc'ed / a torch tensor's data_ptr());
 *     the library allocates nothing on the device and nothing per step.
 *   - every call is asynchronous on the hipStream_t passed as `stream` (void* here so that C
 *     callers need no HIP headers); no hidden synchronisation.
 *   - every function returns an mdr_status (0 = OK, negative = error); nothing throws.
 *   - a handle is not thread-safe; distinct handles are independent.
 *   - temperatures in device buffers are stored RELATIVE to mdr_config.temp_ref (deg C), see DESIGN.md.
 *
 * Layout: all per-house arrays are row-major [nb_envs][nb_houses] (house index fastest).
 */
#ifndef MDR_H
#define MDR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDR_ABI_VERSION 4
#define MDR_MAX_SINUSOIDS 8
#define MDR_MAX_CAPACITIES 16
#define MDR_OBS_COLUMNS 7
#define MDR_INTERP_AXES 10
#define MDR_INTERP_MAX_AXIS 16
#define MDR_MAX_SHARDS 8

typedef struct mdr_env mdr_env_t; /* opaque handle */

typedef enum mdr_status {
  MDR_OK = 0,
  MDR_ERR_INVALID = -1,     /* bad argument / inconsistent config (ValueError in the reference) */
  MDR_ERR_UNBOUND = -2,     /* buffers not bound, or episode not started */
  MDR_ERR_HIP = -3,         /* a HIP runtime call failed; see mdr_last_error() */
  MDR_ERR_UNSUPPORTED = -4  /* shape/mode combination this build has no kernel for */
} mdr_status;

/* PowerGrid.step signal families, env/MA_DemandResponse.py:1257-1310 */
typedef enum mdr_signal_mode {
  MDR_SIGNAL_FLAT = 0,
  MDR_SIGNAL_SINUSOIDALS = 1,
  MDR_SIGNAL_REGULAR_STEPS = 2,
  MDR_SIGNAL_PERLIN = 3
} mdr_signal_mode;

/* compute_temp_penalty modes, env/MA_DemandResponse.py:263-326 */
typedef enum mdr_penalty_mode {
  MDR_PENALTY_INDIVIDUAL_L2 = 0,
  MDR_PENALTY_COMMON_L2 = 1,
  MDR_PENALTY_COMMON_MAX = 2,
  MDR_PENALTY_MIXTURE = 3
} mdr_penalty_mode;

typedef enum mdr_action_source {
  MDR_ACTIONS_EXTERNAL = 0, /* read uint8 actions[E][N] (truthy = ON), ClusterHouses.step 1026-1033 */
  MDR_ACTIONS_BANGBANG = 1, /* on iff house_temp > target (BangBangController, agents/bangbang_controllers.py:41-61),
                               evaluated in-kernel on the pre-step observation; the chosen action is
                               written to actions[E][N] when that pointer is not NULL */
  MDR_ACTIONS_DEADBAND = 2, /* off below target - deadband / 2, on above target + deadband / 2, else what the HVAC is doing
                               (hvac_turned_on): DeadbandBangBangController and BasicController, agents/bangbang_controllers.py:13-38,
                               64-88 (the same rule twice); in-kernel like MDR_ACTIONS_BANGBANG */
  MDR_ACTIONS_ALWAYS_ON = 3 /* AlwaysOnController, agents/bangbang_controllers.py:1-10 (the lockout still applies) */
} mdr_action_source;

/* Flat restatement of the config.py entries the path consumes. */
typedef struct mdr_config {
  uint32_t struct_size;            /* sizeof(mdr_config_t), ABI guard */
  int32_t nb_envs;                 /* E on this device */
  int32_t nb_houses;               /* N on this device (== nb_houses_total unless houses are sharded) */
  int64_t nb_houses_total;         /* nb_agents of the whole env: reward / signal normalisation */
  int64_t env_offset;              /* global index of local env 0: RNG streams are partition invariant */
  int64_t house_offset;            /* global index of local house 0 */
  int32_t time_step;               /* seconds, default_env_prop.time_step */
  int32_t table_steps;             /* K: per-env time tables are (re)built K steps at a time (>=1) */
  double temp_ref;                 /* deg C; device temperatures are stored relative to it */
  /* default_house_prop (config.py:12-26) */
  double init_air_temp, init_mass_temp, target_temp, deadband;
  double Ua, Cm, Ca, Hm;
  double window_area, shading_coeff;
  int32_t solar_gain;              /* solar_gain_bool */
  /* default_hvac_prop (config.py:130-138) */
  int32_t lockout_duration, lockout_noise;
  double COP, cooling_capacity, latent_cooling_fraction;
  /* noise_house_prop / noise_hvac_prop of the selected noise_mode (config.py:27-170) */
  double std_start_temp, std_target_temp, factor_thermo_low, factor_thermo_high;
  int32_t nb_capacities;
  int32_t start_random;            /* start_datetime_mode == "random" */
  double capacity_list[MDR_MAX_CAPACITIES];
  int64_t start_epoch;             /* start_datetime as naive seconds since 1970-01-01 */
  /* cluster_prop.temp_parameters[temp_mode] (config.py:206-291) */
  double day_temp, night_temp, temp_std;
  int32_t random_phase_offset;
  /* power_grid_prop (config.py:324-396); base_power_mode "constant" only */
  int32_t signal_mode;             /* mdr_signal_mode */
  double avg_power_per_hvac;
  int32_t nb_sinusoids;
  int32_t perlin_nb_octaves;
  double sin_periods[MDR_MAX_SINUSOIDS];
  double sin_amplitude_ratios[MDR_MAX_SINUSOIDS];
  double steps_amplitude_per_hvac, steps_period;
  double perlin_amplitude, perlin_octaves_step, perlin_period;
  double artificial_ratio, artificial_signal_ratio_range;
  /* reward_prop (config.py:397-421) */
  double alpha_temp, alpha_sig;
  double norm_temp_penalty;        /* deadbandL2(T0, 0, T0+1), env 346-350 */
  double norm_sig_penalty;         /* deadbandL2(R, 0, 0.75 R), env 352-356 */
  int32_t penalty_mode;            /* mdr_penalty_mode */
  int32_t base_power_mode;         /* 0 = "constant" (avg_power_per_hvac * nb_agents, env 1249);
                                      1 = "interpolation" (env 1250-1255): needs mdr_env_set_interp_grid */
  double mix_ind_L2, mix_common_L2, mix_common_max;
  /* normStateDict: reg_signal and cluster_hvac_power are divided by norm_reg_sig * nb_agents (utils.py:832-841) */
  double obs_power_norm;
} mdr_config_t;

/* Caller-owned device buffers.  [E][N] unless noted. */
typedef struct mdr_buffers {
  uint32_t struct_size;
  uint32_t reserved0;
  /* state, read+written every step */
  float *Ta, *Tm;                  /* indoor air / mass temperature minus temp_ref */
  int32_t *sso;                    /* HVAC seconds_since_off */
  uint8_t *flags;                  /* bit0 turned_on, bit1 lockout */
  /* derived per-episode parameters, read every step (written by reset / load_episode) */
  float *k01, *s0, *k10, *s1;      /* thermal map, difference form (DESIGN.md "Thermal update") */
  float *inv_Ua, *Q_hvac, *P_max;  /* 1/Ua ; -cap/(1+latent) ; cap/COP */
  float *target, *deadband;        /* target minus temp_ref ; deadband */
  int32_t *lockout;                /* lockout_duration incl. noise */
  /* raw per-episode parameters (static observation columns), written by reset / load_episode */
  float *Ua, *Cm, *Ca, *Hm, *capacity, *COP, *latent;
  /* outputs, written every step */
  float *reward;
  float *obs;                      /* [MDR_OBS_COLUMNS][E][N] planes: (Ta-20)/5, (Tm-20)/5, on, lock,
                                      sso/lockout, reg_signal/norm, cluster_hvac_power/norm.
                                      Optional (NULL = the step kernels do not write them: 71 instead of 99 bytes per
                                      house-step) - for callers that observe through mdr_env_actor_sample / mdr_env_obs_vector,
                                      which read the state itself (rollout collection, train_ppo.py:69-72);
                                      mdr_env_refresh_obs brings planes bound later up to date */
  /* per env, [E] */
  int64_t *t0;                     /* episode start, epoch seconds */
  double *phase, *ratio, *max_power;
  double *P;                       /* cluster_hvac_power after the last step */
  double *tot_sum;                 /* [2][E]: local sum of HVAC power, local sum of temperature penalties */
  double *tot_max;                 /* [E]: local max of temperature penalties */
  /* per-env time tables, [(table_steps+1)][E], row r <-> time index j0 + r */
  float *tab_od;                   /* outdoor temperature minus temp_ref */
  float *tab_solar;                /* window_area * shading_coeff * SCL, W */
  double *tab_signal;              /* regulation signal, W */
  /* scratch for the split (multi-workgroup per env) path: [E][mdr_partials_per_env()][3] - all of it: mdr_env_rollout keeps two
   * sets of records in it (the records path takes the stride it is given, records_per_env) */
  double *partials;
  double *base_power;              /* [E] PowerGrid.base_power (written in interpolation mode) */
  /* Optional (NULL = off): graph mode.  int32 [8] = {table row, time index, row note 0 of the split kernels, arrival counter,
   * row note 1, reserved x 3} kept on the device: the step and observation kernels then take their table rows from it instead of from launch arguments, and the
   * step's last kernel advances it (small grids and the split pair; a big one-kernel step is followed by a one-thread launch)
   * - so a captured mdr_env_step / mdr_env_obs_vector (hipGraph, torch.cuda.CUDAGraph) keeps walking through the episode when
   * it is replayed.  See mdr_env_graph_room / mdr_env_graph_replayed. */
  int32_t *cursor;
  /* Optional (NULL = off): [(table_steps+1)][E] what PowerGrid.step adds to cumulated_abs_noise at that time index,
   * |base_power * amplitude * perlin| (env 1301); 0 for the signal families without noise. */
  double *tab_abs_noise;
  /* Optional (NULL = the reward array serves): float [E][N], where the split path's partial kernel leaves each house's own
   * temperature penalty for the finish.  With a buffer of its own the rewards of a step stay readable while the next step is
   * begun - what mdr_env_step_end_begin_records needs. */
  float *pen_stash;
  /* Optional (all NULL = off; tab2_abs_noise iff tab_abs_noise): a SECOND set of time tables, same shapes.  The tables of the next
   * window of table_steps steps are then built ahead on a stream of the library's own while the steps of the current window run
   * on the caller's stream, and the two sets swap roles at the window's end (the caller's stream waits for an event that is long
   * done by then): the O(nb_envs) refill - fp64 transcendentals and Philox draws per env and row - leaves the steps' critical path,
   * which matters for batches of many small envs (209,715 envs x 20 houses: 8 us of every 72 us step).  Not used in
   * base_power_mode "interpolation", in graph mode or inside a capture (refills are built in place there, as without the set).
   * mdr_env_active_tables() says which set the current window reads: 0 = tab_*, 1 = tab2_*. */
  float *tab2_od, *tab2_solar;
  double *tab2_signal, *tab2_abs_noise;
} mdr_buffers_t;

/* Raw episode parameters for mdr_env_load_episode (replay of an episode sampled elsewhere).
 * Device pointers, fp64, [E][N] / [E]; temperatures in deg C (NOT relative). */
typedef struct mdr_episode {
  uint32_t struct_size;
  uint32_t reserved0;
  const double *Ta, *Tm, *target, *deadband, *Ua, *Cm, *Ca, *Hm, *capacity, *COP, *latent;
  const int64_t *lockout;
  const int64_t *t0;
  const double *phase, *ratio;
} mdr_episode_t;

/* What utils.normStateDict (utils.py:740-880) reads from config_dict besides the observation itself. */
typedef enum mdr_obs_layout {
  MDR_OBS_PLANES = 0, /* out[F][E][N]: feature-major planes (coalesced; a GEMM consumes it as X^T) */
  MDR_OBS_ROWS = 1    /* out[E][N][F]: one contiguous normStateDict vector per house */
} mdr_obs_layout;

typedef struct mdr_obs_spec {
  uint32_t struct_size;
  int32_t layout;                  /* mdr_obs_layout */
  /* default_env_prop.state_properties (config.py:311-317) */
  int32_t state_hour, state_day, state_solar_gain, state_thermal, state_hvac;
  /* default_env_prop.message_properties (config.py:319-322) */
  int32_t message_thermal, message_hvac;
  int32_t nb_comm;                 /* messages per house: min(nb_agents_comm, nb_agents - 1), env 808-810 */
  const int32_t *links;            /* device int32 [N][nb_comm] sender ids (ClusterHouses.agent_communicators,
                                      env 806-902), shared by all envs; NULL = circular "neighbours" (816-828) */
  int32_t random_links;            /* agents_comm_mode "random_sample" (env 976-983): every house draws nb_comm distinct
                                      senders among the others anew at every step (`links` is ignored) */
  int32_t reserved0;
  double comm_defect_prob;         /* per-link probability of an all-zero message (env 992-1002) */
  int64_t out_plane_stride;        /* MDR_OBS_PLANES: elements between feature planes; 0 = nb_envs * nb_houses.  A stride
                                      that is not a multiple of 2 MiB keeps the F concurrently written planes off one HBM channel */
  /* normalisation defaults (default_house_prop / default_hvac_prop / reward_prop.norm_reg_sig) */
  double def_Ua, def_Cm, def_Ca, def_Hm, def_COP, def_capacity, def_latent, norm_reg_sig;
} mdr_obs_spec_t;

int mdr_abi_version(void);
const char *mdr_status_string(int status);
/* Last error text of a handle ("" if none); valid until the next call on that handle. */
const char *mdr_last_error(const mdr_env_t *env);

/* Upper bound of the per-env partial records the split path writes for (nb_houses), whatever nb_envs: size `partials` with it.
 * mdr_env_partial_records: the count THIS env writes (one per workgroup of 1024 houses) - what a sharded run all-gathers. */
int64_t mdr_partials_per_env(int32_t nb_houses);
int64_t mdr_env_partial_records(const mdr_env_t *env);

/* MADemandResponseEnv.__init__ (env 73-96) minus build_environment: validates and stores the config. */
int mdr_env_create(const mdr_config_t *config, mdr_env_t **out);
int mdr_env_destroy(mdr_env_t *env);
/* Hands the library the caller's device buffers: they stand where the reference keeps its Python object graph
 * (ClusterHouses.houses[*] / .hvac, env/MA_DemandResponse.py:776-780). */
int mdr_env_bind(mdr_env_t *env, const mdr_buffers_t *buffers);

/* build_environment, part 1 (env 98-123; utils.applyPropertyNoise 573-709): sample every per-house and
 * per-env episode parameter on the device from Philox4x32-10 streams keyed by (seed, episode), derive the
 * kernel coefficients, initialise the state (HVAC off, not locked, sso = lockout: env 432-434) and leave
 * the LOCAL sum of max consumption in max_power[E] (env 796-802). */
int mdr_env_reset(mdr_env_t *env, uint64_t seed, uint32_t episode, void *stream);
/* Same, but from caller-supplied raw parameters instead of sampling. */
int mdr_env_load_episode(mdr_env_t *env, const mdr_episode_t *episode, uint64_t seed, uint32_t episode_index,
                         void *stream);
/* Optional: replace ClusterHouses.compute_OD_temp (env 1057-1081, incl. its random.gauss draw 1079) by a table, double [rows][E] deg C (row = time index);
 * NULL restores the model.  Takes effect at the next mdr_env_begin_episode / table refill; mdr_env_reset (a freshly sampled
 * episode) drops it. */
int mdr_env_set_od_table(mdr_env_t *env, const double *od_table, int64_t rows);
/* The 10-D bang-bang average-power grid of monteCarlo/ (PowerInterpolator, monteCarlo/interpolation.py:21-47).
 * Axis order is the reference's interp_dict_keys.csv: Ua_ratio, Cm_ratio, Ca_ratio, Hm_ratio, air_temp, mass_temp,
 * OD_temp, HVAC_power, hour (seconds), date (days).  `values` is device memory, C order over the axes. */
typedef struct mdr_interp_grid {
  uint32_t struct_size;
  int32_t update_period;           /* interp_update_period, seconds (config.py:347) */
  int32_t nb_agents;               /* interp_nb_agents: at most this many houses are evaluated per update (config.py:348) */
  int32_t reserved0;
  const double *values;
  int32_t dims[MDR_INTERP_AXES];
  double axes[MDR_INTERP_AXES][MDR_INTERP_MAX_AXIS];
} mdr_interp_grid_t;
/* Required before reset/load_episode when config.base_power_mode == 1.  PowerGrid.interpolatePower (env 1195-1234)
 * then runs on the device every ceil(update_period / time_step) steps: per house nearest grid value in the four
 * thermal ratios and HVAC power, 5-D multilinear in (air, mass, OD temperature minus target, hour, date) after
 * clipping (utils.clipInterpolationPoint), summed over all houses (or nb_agents sampled ones, scaled). */
int mdr_env_set_interp_grid(mdr_env_t *env, const mdr_interp_grid_t *grid);

/* build_environment, part 2 (env 125-133): with max_power[E] final (all-reduced by the caller when houses
 * are sharded), build the time tables from time index 0: OD temp (env 793), initial signal (env 133). */
int mdr_env_begin_episode(mdr_env_t *env, void *stream);

/* The seven observation planes of the CURRENT state (what the last step would have written): after steps taken with
 * mdr_buffers_t.obs == NULL and a re-bind with the planes in place. */
int mdr_env_refresh_obs(mdr_env_t *env, void *stream);

/* MADemandResponseEnv.step (env 174-210), whole step on this device. */
int mdr_env_step(mdr_env_t *env, uint8_t *actions, int action_source, void *stream);
/* `nb_steps` consecutive steps without returning to the host: the `for i in range(nb_time_steps)` loop of
 * main-deploy.py:99-104 (one launch per step; see mdr_env_rollout_fused for the in-register form). */
int mdr_env_rollout(mdr_env_t *env, uint8_t *actions, int action_source, int32_t nb_steps, void *stream);

/* Device-resident closed-loop rollout: `nb_steps` consecutive env steps with the bang-bang rule in ONE launch per
 * table chunk.  State and per-episode parameters stay in registers between steps, so per step only the per-env
 * scalars are read; HBM sees the state once per launch instead of once per step.  This is the loop of
 * main-deploy.py:99-148 (BangBangController + metric accumulation) and of monteCarlo.py:193-201.  The bound buffers
 * end in exactly the state `nb_steps` calls of mdr_env_step(..., MDR_ACTIONS_BANGBANG) leave (bit for bit:
 * state, last reward, last observation planes, last actions, P).  Optional accumulators (NULL = skip):  */
typedef struct mdr_rollout_out {
  uint32_t struct_size;
  uint32_t reserved0;
  double *power_trace;             /* [nb_steps][E]   cluster_hvac_power after each step */
  float *reward_sum;               /* [E][N]  += per-agent reward of every step (fp32, in step order) */
  double *sq_temp_error_sum;       /* [E]     += sum over steps and houses of (house_temp - target)^2, main-deploy.py:127,138 */
  double *sq_signal_error_sum;     /* [E]     += sum over steps of (reg_signal - cluster_hvac_power)^2, main-deploy.py:145-152 */
} mdr_rollout_out_t;
/* Shapes without an env-per-workgroup kernel (N > 2048, or N > 512 with N % 4 != 0) are run as single steps with the
 * same accumulators (same results, no fusion).  Returns MDR_ERR_UNSUPPORTED for sharded houses: use mdr_env_rollout. */
int mdr_env_rollout_fused(mdr_env_t *env, uint8_t *actions, int32_t nb_steps, const mdr_rollout_out_t *out, void *stream);

/* The controller of the closed-loop rollouts that take no action_source (mdr_env_rollout_fused, mdr_env_rollout_persistent):
 * MDR_ACTIONS_BANGBANG (the default), MDR_ACTIONS_DEADBAND or MDR_ACTIONS_ALWAYS_ON - the reference's
 * agents/bangbang_controllers.py classes as main-deploy.py:57-104 drives them.  -1 for anything else. */
int mdr_env_set_controller(mdr_env_t *env, int action_source);

/* GreedyMyopic (agents/greedy_myopic_controller.py:6-50): for every env, the houses ranked by -(house_temp - target) - hottest
 * relative to its target first, equal ones in house order - and switched on one after the other while
 *   p + total < reg_signal   or   (|p + total - reg_signal| < |total - reg_signal| and not hvac_lockout),   p = cooling_capacity / COP,
 * a taken house adding p to total.  Writes the uint8 actions[E][N] for the CURRENT observation (reg_signal = the signal of the
 * current time index); follow with mdr_env_step(env, actions, MDR_ACTIONS_EXTERNAL, stream).  One workgroup sorts an env in LDS:
 * at most 2048 houses per env, unsharded houses (MDR_ERR_UNSUPPORTED otherwise). */
int mdr_env_greedy_myopic_actions(mdr_env_t *env, uint8_t *actions, void *stream);

/* Sharded houses (one env spans several devices).  Houses interact only through the cluster power sum (env 1042-1050)
 * and the common penalty sum / max (env 274-321): step_begin updates the local houses and leaves the local
 * reductions in tot_sum/tot_max; the caller all-reduces them (SUM / MAX); step_end writes rewards and the
 * two power observation columns from the reduced values. */
int mdr_env_step_begin(mdr_env_t *env, uint8_t *actions, int action_source, void *stream);
int mdr_env_step_end(mdr_env_t *env, void *stream);
/* One collective per step instead of two: every rank lays its aggregates out as one [3][E] block (tot_max placed
 * right behind tot_sum), the caller ALL-GATHERS the blocks into `gathered` [world][3][E] and this call reduces them on
 * the fly (sum, sum, max over ranks, in rank order) while writing rewards - the all-reduced values are never stored. */
int mdr_env_step_end_gathered(mdr_env_t *env, const double *gathered, int32_t world, void *stream);

/* The same split with one launch less: step_begin_records stops at the per-workgroup partial records - `partials`
 * [E][records_per_env][3] (power sum, penalty sum, penalty max per workgroup; records_per_env >=
 * mdr_env_partial_records(env), the tail stays as the caller left it: zero) - the caller ALL-GATHERS the ranks' `partials`
 * into `records` [world][E][records_per_env][3] (equal records_per_env on every rank: the largest shard's), and every workgroup
 * of step_end_records re-sums its env's world * records_per_env records from L2 in one fixed order while it writes the rewards.
 * Two launches around ONE collective; payload 24 B per 1024 houses instead of 24 B per rank - still latency-bound.
 * Graph mode (mdr_buffers_t.cursor): the pair takes its table rows from the device cursor and step_end_records advances it, so
 * begin - collective - end can be captured in one hipGraph (RCCL collectives are capturable) and replayed mdr_env_graph_room()
 * times between mdr_env_graph_replayed() calls: no host work per step at all.
 * records == NULL: this device's own `partials`, world = 1 (the unsharded split path of mdr_env_step for N > 4096). */
int mdr_env_step_begin_records(mdr_env_t *env, uint8_t *actions, int action_source, int32_t records_per_env, void *stream);
int mdr_env_step_end_records(mdr_env_t *env, const double *records, int32_t world, void *stream);
/* Inside a rollout: step_end_records of the pending step and step_begin_records of the next one in ONE launch (the finish needs the
 * gathered records, the partial only the state the pending step left; a house's two halves run in the same thread).  A rollout of
 * T steps is then begin, (all-gather, end_begin) x (T - 1), all-gather, end: T + 1 launches and T collectives instead of 2 T + T.
 * The next step writes its records into `partials` again (same records_per_env).  Returns MDR_ERR_UNSUPPORTED (-4), nothing
 * launched, where the two halves cannot share a launch - the next step leaves the time tables (un-captured calls: the refill is
 * host work), base_power_mode "interpolation", pen_stash not bound: take step_end_records + step_begin_records there.
 * Capturable like the pair (graph mode): the table row travels in two notes of mdr_buffers_t.cursor, the closing step_end_records
 * moves the cursor on by every step begun. */
int mdr_env_step_end_begin_records(mdr_env_t *env, const double *records, int32_t world, uint8_t *actions, int action_source, void *stream);

/* Sharded houses WITHOUT a kernel boundary or a collective per step: the persistent rollout.  The bang-bang closed loop of
 * main-deploy.py:99-148 for an env whose houses span several workgroups and - sharded - several ranks, ONE launch per time-table
 * window.  The houses stay in registers across steps (as in mdr_env_rollout_fused); what the houses of one env share per step -
 * the cluster power sum (env/MA_DemandResponse.py:1042-1050) and the temperature-penalty sum / max of the common penalty modes
 * (274-321) - travels through a MAILBOX: every house workgroup (1024 houses) pushes its (power sum, penalty sum, penalty max)
 * record - the record mdr_env_step_begin_records writes - into the mailbox of every rank as self-validating 8-byte
 * {step tag, 32 data bits} granules; one reducer workgroup per env and rank re-sums the world * records records in the fixed
 * order of mdr_env_step_end_records (bit-identical totals on every rank and to the records path) and hands the totals to the
 * house workgroups, which run up to seven steps ahead (the state never depends on the totals, only the rewards do).
 *
 * The mailbox is caller-owned device memory of mdr_mailbox_bytes() bytes per rank, ZERO-FILLED ONCE when it is created and
 * from then on written only by these launches (step tags count over the life of the env handle; there is no per-launch
 * re-initialisation, a peer may already be pushing when a launch begins).  `boxes[r]` is rank r's mailbox as THIS device
 * addresses it: the same allocation for shards driven on one device, a peer mapping (hipIpcOpenMemHandle / peer access;
 * mdr_mailbox_alloc / _export / _open below) across devices - then system_scope = 1 and the memory must be fine-grained.
 * All ranks call in lockstep with the same nb_steps (as they would a collective).
 *
 * Guards.  Every workgroup of the launch - and of the `co_resident` launches sharing the device - must be resident at once:
 * the call checks the grid against the occupancy of the kernel and returns MDR_ERR_UNSUPPORTED (nothing launched) otherwise.
 * Every spin in the kernel is bounded by `spin_limit` polls (0 = the default, 2^20: about a second); on expiry the workgroup
 * writes {tag, kind << 28 | workgroup} (kind 1: a house workgroup waiting for totals, 2: a reducer waiting for records) into
 * word 0 of every rank's mailbox and leaves, every other workgroup sees the word and leaves too, and NOTHING is written back:
 * the bound buffers then still hold the state before the launch.  The caller reads word 0 of its mailbox after the stream
 * has drained (0 = ok); a non-zero word means the handle's step count no longer matches its buffers - rebuild both.
 *
 * Results: state, last reward / observation planes / actions, P and the accumulators of mdr_rollout_out_t exactly as
 * nb_steps of mdr_env_step_begin_records / all-gather / mdr_env_step_end_records with MDR_ACTIONS_BANGBANG leave them (state,
 * P, rewards and reward_sum bit for bit; sq_temp_error_sum is summed per workgroup first).  base_power_mode "interpolation"
 * returns MDR_ERR_UNSUPPORTED (its update is host work). */
typedef struct mdr_mailbox {
  uint32_t struct_size;
  int32_t world;                          /* shards of the env (1 = unsharded, or a world of one) */
  int32_t rank;                           /* this shard */
  int32_t records_per_env;                /* record slots per env and rank in every mailbox: >= every records[r] */
  int32_t records[MDR_MAX_SHARDS];        /* mdr_env_partial_records() of every rank's handle */
  int32_t system_scope;                   /* 0: every mailbox is memory of this device; 1: peers are other devices */
  int32_t co_resident;                    /* persistent launches sharing this device at the same time (>= 1) */
  uint32_t spin_limit;                    /* polls before a wait gives up; 0 = default */
  uint32_t reserved0;
  uint64_t *boxes[MDR_MAX_SHARDS];
} mdr_mailbox_t;
int64_t mdr_mailbox_bytes(int32_t nb_envs, int32_t world, int32_t records_per_env);
/* records a handle of `nb_houses` local houses pushes per env and step (what mdr_mailbox_t.records[] holds for that rank) */
int64_t mdr_persist_records(int32_t nb_houses);
int mdr_env_rollout_persistent(mdr_env_t *env, uint8_t *actions, int32_t nb_steps, const mdr_rollout_out_t *out,
                               const mdr_mailbox_t *mailbox, void *stream);
/* Mailbox memory other processes / devices can push into: a zero-filled allocation (fine-grained when asked, so that
 * stores of a peer device become visible inside a running kernel), its 64-byte inter-process handle, and the mapping of a
 * peer's handle into this process.  Free / close with the matching call.  Plain hipMalloc'ed (or torch) memory serves when
 * every shard runs on the one device of one process. */
int mdr_mailbox_alloc(int64_t bytes, int32_t fine_grained, uint64_t **out);
int mdr_mailbox_free(uint64_t *box);
int mdr_mailbox_export(uint64_t *box, uint8_t handle[64]);
int mdr_mailbox_open(const uint8_t handle[64], uint64_t **out);
int mdr_mailbox_close(uint64_t *box);
/* Word 0 of a mailbox (the error word) copied to the host: a SYNCHRONOUS 8-byte copy - call it when the stream has drained. */
int mdr_mailbox_peek(const uint64_t *box, uint64_t *word0);

/* Sharded houses with base_power_mode "interpolation": PowerGrid.interpolatePower (env 1195-1234) averages up to
 * interp_nb_agents houses drawn from the WHOLE env (env 1209-1215), so the update at episode start and every
 * ceil(interp_update_period / time_step) steps (env 1250-1255) needs one more exchange.  After mdr_env_begin_episode /
 * mdr_env_step_end*, while mdr_env_interp_due() is 1:
 *   mdr_env_interp_local  - this shard's part of the new base power -> base_power[E] (every shard walks the same
 *                           draws, which are functions of global indices, and adds the houses it holds);
 *   caller                - SUM all-reduce of base_power over the shards;
 *   mdr_env_interp_apply  - rebuilds the signal rows from the summed base power and re-writes the reg_signal
 *                           observation plane of the step just taken.
 * mdr_env_step_begin refuses to run while an update is due.  Unsharded handles never report 1. */
int mdr_env_interp_due(const mdr_env_t *env);
int mdr_env_interp_local(mdr_env_t *env, void *stream);
int mdr_env_interp_apply(mdr_env_t *env, void *stream);

/* utils.normStateDict (utils.py:740-880) for every house at once, including the neighbour messages
 * (SingleHouse.message env 624-662, gathered through `links`): writes the flat state vector of length
 * mdr_obs_vector_length(spec) per house into `out` in the requested layout.  Reads the post-step (or post-reset)
 * state of the bound buffers; sharded houses need the neighbouring shards' halo and are not supported here. */
int32_t mdr_obs_vector_length(const mdr_obs_spec_t *spec);
int mdr_env_obs_vector(mdr_env_t *env, const mdr_obs_spec_t *spec, float *out, void *stream);

/* Sharded houses: the neighbour messages cross shard edges ("halo exchange").  Three steps per observation:
 *   mdr_env_obs_messages    - SingleHouse.message (env 624-662) of every LOCAL house as a record of
 *                             mdr_obs_message_fields(spec) floats (diff/5, sso, curr/norm, max/norm, [Ua Cm Ca Hm / def],
 *                             [COP latent capacity / def]) -> messages[e][h][:], h < nb_houses, row stride
 *                             entries_per_env records;
 *   caller                  - copies the records of the remote senders its houses listen to behind the local ones
 *                             (slots nb_houses .. entries_per_env-1): peer copies or an all-gather of the exported records;
 *   mdr_env_obs_vector_ext  - as mdr_env_obs_vector, but message slot m of house h is the record
 *                             messages[e][spec->links[h * nb_comm + m]] (links = record slots, not house ids).
 * Own columns, comm defects (drawn per global house index) and layouts are those of mdr_env_obs_vector; an unsharded
 * handle may use the pair too (entries_per_env = nb_houses, links = house ids).  With random_links the senders are drawn
 * among ALL houses of the env, so `messages` must hold every house's record at slot = global house id
 * (entries_per_env >= nb_houses_total; mdr_env_obs_messages then gets `messages + house_offset * fields`). */
int32_t mdr_obs_message_fields(const mdr_obs_spec_t *spec);
int mdr_env_obs_messages(mdr_env_t *env, const mdr_obs_spec_t *spec, float *messages, int64_t entries_per_env, void *stream);
int mdr_env_obs_vector_ext(mdr_env_t *env, const mdr_obs_spec_t *spec, const float *messages, int64_t entries_per_env,
                           float *out, void *stream);

/* The random part of the message gather on its own (env 976-1002): for every local house and message slot m < spec->nb_comm the
 * sender's GLOBAL house id -> senders[e][h][m] (int32; the link-table entry, the circular neighbour, or - random_links - the
 * `random.sample` draw of this step) and whether the link delivers -> keep[e][h][m] (uint8; 0 = `np.random.rand() >
 * comm_defect_prob` failed: an all-zero message).  Same Philox streams and counters as mdr_env_obs_vector at the same time
 * index, so a host that builds the reference's `message` lists itself (the dict adapter) shows exactly what the flat vector holds. */
int mdr_env_comm_draws(mdr_env_t *env, const mdr_obs_spec_t *spec, int32_t *senders, uint8_t *keep, void *stream);

/* Cursor: k = number of steps taken this episode (env.datetime == start_datetime + k * time_step, env 189);
 * j0 = time index of table row 0. */
/* Graph mode (mdr_buffers_t.cursor bound).  The launch-bound regime - a policy in the loop on small batches, where a step is
 * six short kernels - is served by capturing ONE step (observation, policy, mdr_env_step) into a graph and replaying it.
 * The host does not run during a replay, so it is told afterwards:
 *   mdr_env_graph_room      how many replays are allowed now: the steps the time tables still cover, and - interpolation mode -
 *                           only up to the step BEFORE the next interpolatePower update: the step that lands on an update
 *                           must be an ordinary (un-captured) mdr_env_step, which runs the update before anything enqueued
 *                           behind it reads the signal.  0 = take one ordinary step now; capture needs room >= 1.  One
 *                           capture may record several steps (at most `room`: the call that would record one more is
 *                           refused); every replay of that graph then counts as that many steps;
 *   mdr_env_graph_replayed  `n` steps were replayed: the host cursor catches up, refills the tables and runs a due
 *                           interpolation update (launches on `stream`, outside any capture).
 * Ordinary (non-captured) calls keep working in graph mode. */
int64_t mdr_env_graph_room(const mdr_env_t *env);
int mdr_env_graph_replayed(mdr_env_t *env, int64_t n, void *stream);

/* What the reference's dict surface shows of ONE env after reset / step (make_cluster_obs_dict env 904-1003, rewards 330-373),
 * gathered into one fp64 vector on the device so that a host adapter needs a single copy:
 * out[5 N + 7] = house_temp[N] | house_mass_temp[N] (deg C) | seconds_since_off[N] | flags[N] (bit 0 on, bit 1 lockout) |
 *                reward[N] | OD_temp, reg_signal, solar gain of the current time index, cluster_hvac_power, max_power,
 *                artificial_ratio, |base_power * amplitude * perlin| of the current time index (0 without tab_abs_noise). */
int mdr_env_pack(mdr_env_t *env, int32_t env_index, double *out, void *stream);

int mdr_env_cursor(const mdr_env_t *env, int64_t *k, int64_t *j0);
int mdr_env_active_tables(const mdr_env_t *env);
/* Re-create a cursor on a new handle whose buffers were cloned from another env: copy.deepcopy(env) as
 * utils.test_*_agent use it (utils.py:890, 931, 970, 1008). */
int mdr_env_set_cursor(mdr_env_t *env, uint64_t seed, uint32_t episode, int64_t k, int64_t j0);

#ifdef __cplusplus
}
#endif
#endif /* MDR_H */
