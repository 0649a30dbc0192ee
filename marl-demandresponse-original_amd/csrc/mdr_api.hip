// C-ABI host side of libmdr_hip.so (include/mdr.h).  No device allocation, no hidden synchronisation:
// each call validates, fills a kernel argument block and enqueues launches on the caller's stream.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/mdr_policy.h"
#include "mdr_kernels.h"

struct mdr_env {
  mdr_config_t cfg;
  mdr_buffers_t buf;
  bool bound = false;
  bool has_episode = false;   // per-house parameters present
  bool has_tables = false;    // begin_episode done
  bool split_pending = false; // step_begin issued, step_end outstanding
  int64_t captured = 0;       // graph mode: steps recorded into the capture in progress (one graph may hold several; each replay runs them all)
  int32_t snap_slot = 0;      // records path: which row note (mdr_buffers_t.cursor[2] | [4]) holds the pending step's table row
  int32_t pending_adv = 0;    // records path: steps begun since the device cursor last moved (the finish moves it by that many)
  bool interp_due = false;    // sharded houses, interpolation mode: the base power of the current time index awaits its exchange
  int64_t dev_row = -1, dev_k = -1;   // graph mode: what the device-resident cursor holds (as far as the host knows)
  uint64_t seed = 0;
  uint32_t episode = 0;
  int64_t k = 0;              // steps taken this episode
  int64_t j0 = 0;             // time index of table row 0
  const double* od_ext = nullptr;
  int64_t od_ext_rows = 0;
  mdr::StepPlan plan;
  mdr::StepPlan rollout_plan;
  int64_t nblk = 1;
  int64_t records_stride = 0;   // > 0 between mdr_env_step_begin_records and mdr_env_step_end_records: record stride of `partials`
  mdr_interp_grid_t interp{};   // base_power_mode == 1
  bool has_interp = false;
  int64_t interp_steps = 0;     // U: env steps between two interpolatePower calls = ceil(update_period / time_step)
  // Table prefetch (mdr_buffers_t.tab2_*): the tables of the NEXT window are built on a side stream while the current window's
  // steps run, into the set of table buffers the steps do not read; at the window's end the sets swap roles.
  struct TableSet { float* od; float* solar; double* signal; double* abs_noise; } tabs[2] = {};
  int active = 0;               // tabs[active] is what buf.tab_* point at
  bool has_alt = false;
  bool prefetched = false;      // tabs[active ^ 1] holds (or is being filled with) the tables of time index prefetch_j0
  int64_t prefetch_j0 = -1;
  hipStream_t side = nullptr;
  hipEvent_t ev_fill = nullptr, ev_free = nullptr;
  int controller = MDR_ACTIONS_BANGBANG;   // the rule the rollouts without an action_source apply (mdr_env_set_controller)
  uint32_t mailbox_tag = 1;     // persistent rollout: tag of the next step pushed through the mailbox (counts over the handle's life; 0 = never written)
  std::string err;
};

namespace {

int fail(mdr_env* env, int code, const std::string& msg) {
  if (env) env->err = msg;
  return code;
}

int hip_fail(mdr_env* env, hipError_t e, const char* what) {
  return fail(env, MDR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

bool finite_pos(double v) { return std::isfinite(v) && v > 0.0; }

std::string validate(const mdr_config_t& c) {
  if (c.struct_size != sizeof(mdr_config_t)) return "mdr_config_t size mismatch (ABI)";
  if (c.nb_envs < 1 || c.nb_houses < 1) return "nb_envs and nb_houses must be >= 1";
  if (c.nb_houses_total < c.nb_houses) return "nb_houses_total must be >= nb_houses";
  if (c.env_offset < 0 || c.house_offset < 0) return "offsets must be >= 0";
  if (c.house_offset + c.nb_houses > c.nb_houses_total) return "house shard exceeds nb_houses_total";
  if (c.time_step < 1) return "time_step must be >= 1 s";
  if (c.table_steps < 1 || c.table_steps > (1 << 20)) return "table_steps out of range";
  if (!finite_pos(c.Ua) || !finite_pos(c.Cm) || !finite_pos(c.Ca) || !finite_pos(c.Hm))
    return "Ua, Cm, Ca, Hm must be positive";
  if (c.factor_thermo_low <= 0.0 || c.factor_thermo_high < c.factor_thermo_low) return "bad thermal noise factors";
  // HVAC.__init__ validation, env/MA_DemandResponse.py:438-461
  if (c.latent_cooling_fraction > 1.0 || c.latent_cooling_fraction < 0.0)
    return "Latent cooling fraction must be between 0 and 1";
  if (c.lockout_noise < 0) return "lockout_noise must be >= 0";
  if (c.lockout_duration - c.lockout_noise < 0) return "Lockout duration must be positive";
  if (c.COP <= 0.0) return "Coefficient of performance (COP) must be positive";
  if (c.nb_capacities < 1 || c.nb_capacities > MDR_MAX_CAPACITIES) return "nb_capacities out of range";
  for (int i = 0; i < c.nb_capacities; ++i)
    if (!(c.capacity_list[i] >= 0.0)) return "Cooling capacity must be positive";
  if (c.signal_mode < MDR_SIGNAL_FLAT || c.signal_mode > MDR_SIGNAL_PERLIN) return "Invalid power grid signal mode";
  if (c.nb_sinusoids < 0 || c.nb_sinusoids > MDR_MAX_SINUSOIDS) return "too many sinusoids";
  if (c.signal_mode == MDR_SIGNAL_SINUSOIDALS)
    for (int i = 0; i < c.nb_sinusoids; ++i)
      if (!(c.sin_periods[i] > 0.0)) return "sinusoid periods must be positive";
  if (c.signal_mode == MDR_SIGNAL_REGULAR_STEPS && !(c.steps_amplitude_per_hvac > 0.0 && c.steps_period > 0.0))
    return "regular_steps needs positive amplitude_per_hvac and period";
  if (c.signal_mode == MDR_SIGNAL_PERLIN &&
      !(c.perlin_nb_octaves >= 1 && c.perlin_nb_octaves <= 16 && c.perlin_period > 0.0 && c.perlin_octaves_step > 0.0))
    return "perlin needs 1..16 octaves, positive period and octaves_step";
  if (!(c.artificial_signal_ratio_range > 0.0)) return "artificial_signal_ratio_range must be positive";
  if (c.penalty_mode < MDR_PENALTY_INDIVIDUAL_L2 || c.penalty_mode > MDR_PENALTY_MIXTURE)
    return "Unknown temperature penalty mode";
  if (c.penalty_mode == MDR_PENALTY_MIXTURE && !(c.mix_ind_L2 + c.mix_common_L2 + c.mix_common_max != 0.0))
    return "mixture weights sum to zero";
  if (c.base_power_mode != 0 && c.base_power_mode != 1) return "base_power_mode can only be 0 (constant) or 1 (interpolation)";
  if (!(c.norm_temp_penalty > 0.0) || !(c.norm_sig_penalty > 0.0) || !(c.obs_power_norm > 0.0))
    return "normalisation constants must be positive";
  if ((int64_t)c.nb_envs * (int64_t)c.nb_houses > (int64_t)1 << 40) return "E * N too large";
  return "";
}

std::string check_buffers(const mdr_buffers_t& b, bool need_partials) {
  if (b.struct_size != sizeof(mdr_buffers_t)) return "mdr_buffers_t size mismatch (ABI)";
#define MDR_NEED(p) \
  if (!b.p) return "buffer '" #p "' is NULL"
  MDR_NEED(Ta); MDR_NEED(Tm); MDR_NEED(sso); MDR_NEED(flags);
  MDR_NEED(k01); MDR_NEED(s0); MDR_NEED(k10); MDR_NEED(s1); MDR_NEED(inv_Ua); MDR_NEED(Q_hvac); MDR_NEED(P_max);
  MDR_NEED(target); MDR_NEED(deadband); MDR_NEED(lockout);
  MDR_NEED(Ua); MDR_NEED(Cm); MDR_NEED(Ca); MDR_NEED(Hm); MDR_NEED(capacity); MDR_NEED(COP); MDR_NEED(latent);
  MDR_NEED(reward);   /* obs is optional: NULL = the step kernels do not write the seven planes */
  MDR_NEED(t0); MDR_NEED(phase); MDR_NEED(ratio); MDR_NEED(max_power); MDR_NEED(P); MDR_NEED(tot_sum); MDR_NEED(tot_max);
  MDR_NEED(tab_od); MDR_NEED(tab_solar); MDR_NEED(tab_signal);
#undef MDR_NEED
  if (need_partials && !b.partials) return "buffer 'partials' is NULL (needed by the split path)";
  // the vector kernels use 16-byte accesses on the float/int arrays
  const void* aligned16[] = {b.Ta, b.Tm, b.sso, b.k01, b.s0, b.k10, b.s1, b.inv_Ua, b.Q_hvac, b.P_max,
                             b.target, b.deadband, b.lockout, b.reward, b.obs};
  for (const void* p : aligned16)
    if (((uintptr_t)p & 15u) != 0) return "per-house float/int buffers must be 16-byte aligned";
  if (((uintptr_t)b.flags & 3u) != 0) return "flags must be 4-byte aligned";
  if (((uintptr_t)b.pen_stash & 15u) != 0) return "pen_stash must be 16-byte aligned";
  return "";
}

mdr::EpisodeArgs episode_args(const mdr_env& env) {
  const mdr_config_t& c = env.cfg;
  mdr::EpisodeArgs a{};
  a.b = env.buf;
  a.E = c.nb_envs;
  a.N = c.nb_houses;
  a.dt = c.time_step;
  a.env_offset = c.env_offset;
  a.house_offset = c.house_offset;
  a.k0 = (uint32_t)(env.seed & 0xFFFFFFFFull);
  a.k1 = (uint32_t)(env.seed >> 32);
  a.episode = env.episode;
  a.temp_ref = c.temp_ref;
  a.init_air = c.init_air_temp;
  a.init_mass = c.init_mass_temp;
  a.target = c.target_temp;
  a.deadband = c.deadband;
  a.Ua = c.Ua; a.Cm = c.Cm; a.Ca = c.Ca; a.Hm = c.Hm;
  a.COP = c.COP;
  a.latent = c.latent_cooling_fraction;
  a.std_start = c.std_start_temp;
  a.std_target = c.std_target_temp;
  a.f_low = c.factor_thermo_low;
  a.f_high = c.factor_thermo_high;
  a.lockout = c.lockout_duration;
  a.lockout_noise = c.lockout_noise;
  a.ncaps = c.nb_capacities;
  for (int i = 0; i < MDR_MAX_CAPACITIES; ++i) a.caps[i] = c.capacity_list[i];
  a.start_random = c.start_random;
  a.random_phase = c.random_phase_offset;
  a.start_epoch = c.start_epoch;
  a.artificial_ratio = c.artificial_ratio;
  a.ratio_range = c.artificial_signal_ratio_range;
  return a;
}

void use_tables(mdr_env* env, int which) {
  env->active = which;
  env->buf.tab_od = env->tabs[which].od;
  env->buf.tab_solar = env->tabs[which].solar;
  env->buf.tab_signal = env->tabs[which].signal;
  env->buf.tab_abs_noise = env->tabs[which].abs_noise;
}

// `into` == nullptr: the tables the steps read (the active set) from time index j0, on the caller's stream; else a prefetch into
// the other set (the host cursor does not move)
int fill_tables(mdr_env* env, int64_t j0, hipStream_t s, const mdr_env::TableSet* into = nullptr) {
  const mdr_config_t& c = env->cfg;
  mdr::TableArgs t{};
  t.tab_od = into ? into->od : env->buf.tab_od;
  t.tab_solar = into ? into->solar : env->buf.tab_solar;
  t.tab_signal = into ? into->signal : env->buf.tab_signal;
  t.tab_abs_noise = into ? into->abs_noise : env->buf.tab_abs_noise;
  t.t0 = env->buf.t0;
  t.phase = env->buf.phase;
  t.ratio = env->buf.ratio;
  t.max_power = env->buf.max_power;
  t.od_ext = env->od_ext;
  t.od_ext_rows = env->od_ext_rows;
  t.j0 = j0;
  t.rows = c.table_steps + 1;
  t.E = c.nb_envs;
  t.dt = c.time_step;
  t.env_offset = c.env_offset;
  t.k0 = (uint32_t)(env->seed & 0xFFFFFFFFull);
  t.k1 = (uint32_t)(env->seed >> 32);
  t.episode = env->episode;
  t.temp_ref = c.temp_ref;
  t.day_temp = c.day_temp;
  t.night_temp = c.night_temp;
  t.temp_std = c.temp_std;
  t.solar_on = c.solar_gain;
  t.area_shading = c.window_area * c.shading_coeff;
  t.n_total = c.nb_houses_total;
  t.avg_power_per_hvac = c.avg_power_per_hvac;
  t.base_power = c.base_power_mode == 1 ? env->buf.base_power : nullptr;
  t.signal_mode = c.signal_mode;
  t.nb_sin = c.nb_sinusoids;
  t.perlin_octaves = c.perlin_nb_octaves;
  for (int i = 0; i < MDR_MAX_SINUSOIDS; ++i) {
    t.sin_periods[i] = c.sin_periods[i];
    t.sin_ratios[i] = c.sin_amplitude_ratios[i];
  }
  t.steps_amp = c.steps_amplitude_per_hvac;
  t.steps_period = c.steps_period;
  t.perlin_amp = c.perlin_amplitude;
  t.perlin_step = c.perlin_octaves_step;
  t.perlin_period = c.perlin_period;
  hipError_t e = mdr::launch_tables(t, s);
  if (e != hipSuccess) return hip_fail(env, e, "fill_tables");
  if (into == nullptr) {
    env->j0 = j0;
    env->prefetched = false;   // whatever the other set holds was built for another window
  }
  return MDR_OK;
}


// PowerGrid.step in interpolation mode (env 1250-1255) at time index j: the outdoor temperature row has to exist
// before interpolatePower can read it, and the signal rows need the new base power, hence fill - interpolate - fill.
int interp_local(mdr_env* env, int64_t j, hipStream_t s) {
  const mdr_config_t& c = env->cfg;
  int rc = fill_tables(env, j, s);
  if (rc != MDR_OK) return rc;
  mdr::InterpArgs a{};
  a.values = env->interp.values;
  for (int d = 0; d < MDR_INTERP_AXES; ++d) {
    a.dims[d] = env->interp.dims[d];
    for (int i = 0; i < MDR_INTERP_MAX_AXIS; ++i) a.axes[d][i] = env->interp.axes[d][i];
  }
  const mdr_buffers_t& b = env->buf;
  a.Ta = b.Ta; a.Tm = b.Tm; a.target = b.target; a.Ua = b.Ua; a.Cm = b.Cm; a.Ca = b.Ca; a.Hm = b.Hm; a.capacity = b.capacity;
  a.od_now = b.tab_od;   // row 0 == time index j
  a.t0 = b.t0;
  a.base_power = b.base_power;
  a.E = c.nb_envs; a.N = c.nb_houses; a.dt = c.time_step; a.nb_agents = env->interp.nb_agents; a.solar_on = c.solar_gain;
  a.N_total = c.nb_houses_total; a.house_offset = c.house_offset;
  a.j = j;
  a.env_offset = c.env_offset;
  a.k0 = (uint32_t)(env->seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(env->seed >> 32); a.episode = env->episode;
  a.def_Ua = c.Ua; a.def_Cm = c.Cm; a.def_Ca = c.Ca; a.def_Hm = c.Hm;
  hipError_t e = mdr::launch_interp_base(a, s);
  if (e != hipSuccess) return hip_fail(env, e, "interp_base");
  return MDR_OK;
}

int refresh_interp(mdr_env* env, int64_t j, hipStream_t s) {
  int rc = interp_local(env, j, s);
  if (rc != MDR_OK) return rc;
  return fill_tables(env, j, s);
}

bool interp_mode(const mdr_env* env) { return env->cfg.base_power_mode == 1; }
bool sharded(const mdr_env* env) { return env->cfg.nb_houses_total != env->cfg.nb_houses; }
bool graph_mode(const mdr_env* env) { return env->bound && env->buf.cursor != nullptr; }

bool capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// The tables of the window after the current one, built on the side stream while the current window's steps run on `main`.
// Off in interpolation mode (the tables depend on state at every update), in graph mode (captured launches carry the pointers of
// one table set) and inside a capture.
int issue_prefetch(mdr_env* env, hipStream_t main) {
  if (!env->has_alt || interp_mode(env) || graph_mode(env) || capturing(main)) return MDR_OK;
  hipError_t e = hipSuccess;
  if (env->side == nullptr) {
    // MDR_PREFETCH_PRIORITY=low|high: experiment knob (default: the priority of an ordinary stream)
    static const int prio = [] { const char* t = getenv("MDR_PREFETCH_PRIORITY"); return !t ? 0 : (t[0] == 'l' ? 1 : (t[0] == 'h' ? 2 : 0)); }();
    if (prio != 0) {
      int least = 0, greatest = 0;
      e = hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (e == hipSuccess) e = hipStreamCreateWithPriority(&env->side, hipStreamNonBlocking, prio == 1 ? least : greatest);
    } else {
      e = hipStreamCreateWithFlags(&env->side, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&env->ev_fill, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&env->ev_free, hipEventDisableTiming);
    if (e != hipSuccess) return hip_fail(env, e, "table prefetch: side stream");
  }
  // everything enqueued on `main` so far - the previous window's steps read the set about to be overwritten - comes first
  e = hipEventRecord(env->ev_free, main);
  if (e == hipSuccess) e = hipStreamWaitEvent(env->side, env->ev_free, 0);
  if (e != hipSuccess) return hip_fail(env, e, "table prefetch: ordering");
  const int64_t next = env->j0 + env->cfg.table_steps;
  int rc = fill_tables(env, next, env->side, &env->tabs[env->active ^ 1]);
  if (rc != MDR_OK) return rc;
  e = hipEventRecord(env->ev_fill, env->side);
  if (e != hipSuccess) return hip_fail(env, e, "table prefetch: event");
  env->prefetched = true;
  env->prefetch_j0 = next;
  return MDR_OK;
}

// Before anything on `main` rewrites what a prefetch in flight reads (t0, phase, ratio, max_power at an episode start): wait for it
// and forget it
int settle_prefetch(mdr_env* env, hipStream_t main) {
  if (env->prefetched && env->ev_fill != nullptr) {
    hipError_t e = hipStreamWaitEvent(main, env->ev_fill, 0);
    if (e != hipSuccess) return hip_fail(env, e, "table prefetch: settle");
  }
  env->prefetched = false;
  return MDR_OK;
}

// Graph mode: make the device cursor say (k - j0, k).  Never inside a capture - a captured reset would rewind every replay.
int sync_cursor(mdr_env* env, hipStream_t s) {
  if (!graph_mode(env)) return MDR_OK;
  const int64_t row = env->k - env->j0;
  if (env->dev_row == row && env->dev_k == env->k) return MDR_OK;
  if (capturing(s)) return fail(env, MDR_ERR_INVALID, "graph mode: the device cursor is stale; make one un-captured call (or mdr_env_graph_replayed) before capturing");
  hipError_t e = mdr::launch_cursor_set(env->buf.cursor, (int32_t)row, (int32_t)env->k, s);
  if (e != hipSuccess) return hip_fail(env, e, "cursor_set");
  env->dev_row = row;
  env->dev_k = env->k;
  return MDR_OK;
}

// Graph mode: the launch carries table row 0 / 1 and the kernels add the device cursor (StepArgs.cursor_adv: the last one moves it on).
void graph_rows(const mdr_env* env, mdr::StepArgs* a) {
  const mdr_buffers_t& b = env->buf;
  a->od_old = b.tab_od;
  a->solar_new = b.tab_solar + env->cfg.nb_envs;
  a->sig_old = b.tab_signal;
  a->sig_new = b.tab_signal + env->cfg.nb_envs;
  a->cursor = b.cursor;
  a->cursor_max = env->cfg.table_steps - 1;
}


// Called after a step that brought the cursor onto an interpolation update: new base power, new tables from the
// current time index, and the reg_signal observation plane of the step just taken re-written with the final signal.
int interp_boundary(mdr_env* env, hipStream_t s, double* sq_signal_error_sum);

// Builds the argument block of step k -> k+1, refilling the time tables when the cursor leaves them.
int step_args(mdr_env* env, uint8_t* actions, int action_source, hipStream_t s, mdr::StepArgs* out) {
  const mdr_config_t& c = env->cfg;
  if (!env->bound) return fail(env, MDR_ERR_UNBOUND, "buffers not bound");
  if (!env->has_tables) return fail(env, MDR_ERR_UNBOUND, "no episode: call reset/load_episode and begin_episode first");
  if (action_source < MDR_ACTIONS_EXTERNAL || action_source > MDR_ACTIONS_ALWAYS_ON)
    return fail(env, MDR_ERR_INVALID, "unknown action_source");
  if (action_source == MDR_ACTIONS_EXTERNAL && !actions) return fail(env, MDR_ERR_INVALID, "actions is NULL");
  if (actions && c.nb_houses % 4 == 0 && ((uintptr_t)actions & 3u) != 0)  // uchar4 accesses when N % 4 == 0
    return fail(env, MDR_ERR_INVALID, "actions must be 4-byte aligned when nb_houses is a multiple of 4");
  if (actions && c.nb_houses % 2 == 0 && ((uintptr_t)actions & 1u) != 0)  // uchar2 accesses when N is even
    return fail(env, MDR_ERR_INVALID, "actions must be 2-byte aligned when nb_houses is even");
  if (env->k + 1 - env->j0 > c.table_steps) {
    if (capturing(s)) return fail(env, MDR_ERR_INVALID, "the time tables end here: a refill cannot be captured (mdr_env_graph_room() is 0)");
    int rc;
    if (env->prefetched && env->prefetch_j0 == env->k) {   // built meanwhile on the side stream: wait for it (long done) and swap the sets
      hipError_t e = hipStreamWaitEvent(s, env->ev_fill, 0);
      if (e != hipSuccess) return hip_fail(env, e, "table prefetch: wait");
      use_tables(env, env->active ^ 1);
      env->j0 = env->k;
      env->prefetched = false;
    } else {
      rc = fill_tables(env, env->k, s);
      if (rc != MDR_OK) return rc;
    }
    rc = issue_prefetch(env, s);
    if (rc != MDR_OK) return rc;
  }
  const int64_t r0 = env->k - env->j0, r1 = r0 + 1;
  const mdr_buffers_t& b = env->buf;
  mdr::StepArgs a{};
  a.Ta = b.Ta; a.Tm = b.Tm; a.sso = b.sso; a.flags = b.flags;
  a.k01 = b.k01; a.s0 = b.s0; a.k10 = b.k10; a.s1 = b.s1; a.inv_Ua = b.inv_Ua; a.Q_hvac = b.Q_hvac; a.P_max = b.P_max;
  a.target = b.target; a.deadband = b.deadband; a.lockout = b.lockout;
  a.actions = actions;
  a.reward = b.reward; a.obs = b.obs;
  a.P = b.P; a.tot_sum = b.tot_sum; a.tot_max = b.tot_max; a.partials = b.partials;
  a.od_old = b.tab_od + r0 * c.nb_envs;
  a.solar_new = b.tab_solar + r1 * c.nb_envs;
  a.sig_old = b.tab_signal + r0 * c.nb_envs;
  a.sig_new = b.tab_signal + r1 * c.nb_envs;
  a.plane = (int64_t)c.nb_envs * c.nb_houses;
  a.E = c.nb_envs; a.N = c.nb_houses; a.dt = c.time_step;
  a.penalty_mode = c.penalty_mode;
  a.action_source = action_source;
  a.nblk = (int)env->nblk;
  a.c_temp = (float)(c.alpha_temp / c.norm_temp_penalty);
  const double ms = c.mix_ind_L2 + c.mix_common_L2 + c.mix_common_max;
  if (c.penalty_mode == MDR_PENALTY_MIXTURE) {
    a.mix_i = (float)(c.mix_ind_L2 / ms); a.mix_c = (float)(c.mix_common_L2 / ms); a.mix_m = (float)(c.mix_common_max / ms);
  }
  a.obs_tshift = (float)(c.temp_ref - 20.0);
  a.c_sig = c.alpha_sig / c.norm_sig_penalty;
  a.inv_n_total = 1.0 / (double)c.nb_houses_total;
  a.inv_obs_norm = 1.0 / c.obs_power_norm;
  a.stash = b.pen_stash ? b.pen_stash : b.reward;   // split path: each house's own penalty between the partial and the finish kernel
  *out = a;
  return MDR_OK;
}

int interp_boundary(mdr_env* env, hipStream_t s, double* sq_signal_error_sum) {
  if (!interp_mode(env) || env->k % env->interp_steps != 0) return MDR_OK;
  int rc = refresh_interp(env, env->k, s);
  if (rc != MDR_OK) return rc;
  mdr::StepArgs a;
  rc = step_args(env, nullptr, MDR_ACTIONS_BANGBANG, s, &a);   // sig_old == row 0 == the signal of the current time index
  if (rc != MDR_OK) return rc;
  hipError_t e = mdr::launch_patch_signal_plane(a, s);
  if (e != hipSuccess) return hip_fail(env, e, "patch_signal_plane");
  if (sq_signal_error_sum) {
    e = mdr::launch_signal_error(a, sq_signal_error_sum, s);
    if (e != hipSuccess) return hip_fail(env, e, "signal_error");
  }
  return MDR_OK;
}

}  // namespace

extern "C" {

int mdr_abi_version(void) { return MDR_ABI_VERSION; }

const char* mdr_status_string(int status) {
  switch (status) {
    case MDR_OK: return "ok";
    case MDR_ERR_INVALID: return "invalid argument";
    case MDR_ERR_UNBOUND: return "buffers not bound or episode not started";
    case MDR_ERR_HIP: return "HIP runtime error";
    case MDR_ERR_UNSUPPORTED: return "unsupported shape or mode";
    default: return "unknown status";
  }
}

const char* mdr_last_error(const mdr_env_t* env) { return env ? env->err.c_str() : ""; }

// upper bound over every batch size (the 64-thread form): what `partials` must be able to hold
int64_t mdr_partials_per_env(int32_t nb_houses) {
  return nb_houses < 1 ? 0 : mdr::split_blocks(nb_houses, nb_houses % 4 == 0 ? 64 : 256);
}

int64_t mdr_env_partial_records(const mdr_env_t* env) { return env ? env->nblk : 0; }

int mdr_env_create(const mdr_config_t* config, mdr_env_t** out) {
  if (!config || !out) return MDR_ERR_INVALID;
  *out = nullptr;
  mdr_env* env = new (std::nothrow) mdr_env();
  if (!env) return MDR_ERR_INVALID;
  const std::string msg = validate(*config);
  env->cfg = *config;
  env->err = msg;
  *out = env;  // returned even on failure so that mdr_last_error() can explain; caller destroys it
  if (!msg.empty()) return MDR_ERR_INVALID;
  env->plan = mdr::plan_step(config->nb_houses, config->nb_envs);
  if ((env->plan.kind == mdr::STEP_SPLIT || config->nb_houses_total != config->nb_houses) && config->nb_envs > 65535) {
    env->err = "the split / sharded step path puts the env index on grid.y: nb_envs must be <= 65535 there";
    return MDR_ERR_UNSUPPORTED;
  }
  env->rollout_plan = mdr::plan_rollout(config->nb_houses, config->nb_envs);
  env->nblk = mdr::split_blocks(config->nb_houses, mdr::split_threads(config->nb_houses, config->nb_envs));
  return MDR_OK;
}

int mdr_env_destroy(mdr_env_t* env) {
  if (env) {
    if (env->side) (void)hipStreamDestroy(env->side);
    if (env->ev_fill) (void)hipEventDestroy(env->ev_fill);
    if (env->ev_free) (void)hipEventDestroy(env->ev_free);
  }
  delete env;
  return MDR_OK;
}

int mdr_env_active_tables(const mdr_env_t* env) { return env ? env->active : 0; }

int mdr_env_bind(mdr_env_t* env, const mdr_buffers_t* buffers) {
  if (!env || !buffers) return MDR_ERR_INVALID;
  const bool sharded = env->cfg.nb_houses_total != env->cfg.nb_houses;
  const std::string msg = check_buffers(*buffers, sharded || env->plan.kind == mdr::STEP_SPLIT);
  if (!msg.empty()) return fail(env, MDR_ERR_INVALID, msg);
  const bool any2 = buffers->tab2_od || buffers->tab2_solar || buffers->tab2_signal || buffers->tab2_abs_noise;
  const bool all2 = buffers->tab2_od && buffers->tab2_solar && buffers->tab2_signal && ((buffers->tab2_abs_noise != nullptr) == (buffers->tab_abs_noise != nullptr));
  if (any2 && !all2) return fail(env, MDR_ERR_INVALID, "tab2_od / tab2_solar / tab2_signal (and tab2_abs_noise iff tab_abs_noise) go together");
  const mdr_env::TableSet t0{buffers->tab_od, buffers->tab_solar, buffers->tab_signal, buffers->tab_abs_noise};
  const mdr_env::TableSet t1{buffers->tab2_od, buffers->tab2_solar, buffers->tab2_signal, buffers->tab2_abs_noise};
  // a re-bind that leaves the table buffers where they are (another optional buffer came or went mid-episode) keeps the window
  const bool same_tables = env->bound && memcmp(&t0, &env->tabs[0], sizeof t0) == 0 && memcmp(&t1, &env->tabs[1], sizeof t1) == 0;
  if (!same_tables && env->prefetched && env->ev_fill) (void)hipEventSynchronize(env->ev_fill);   // a prefetch into the old buffers: let it land
  env->buf = *buffers;
  env->tabs[0] = t0;
  env->tabs[1] = t1;
  env->has_alt = all2;
  if (same_tables) {
    use_tables(env, env->active);
  } else {
    env->active = 0;
    env->prefetched = false;
  }
  env->bound = true;
  env->dev_row = env->dev_k = -1;
  env->err.clear();
  return MDR_OK;
}

int mdr_env_reset(mdr_env_t* env, uint64_t seed, uint32_t episode, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!env->bound) return fail(env, MDR_ERR_UNBOUND, "buffers not bound");
  env->seed = seed;
  env->episode = episode;
  env->has_tables = false;
  env->split_pending = false;
  env->records_stride = 0;
  env->interp_due = false;
  env->od_ext = nullptr;      // a recorded outdoor-temperature sequence belongs to the episode it was loaded with
  env->od_ext_rows = 0;
  if (settle_prefetch(env, (hipStream_t)stream) != MDR_OK) return MDR_ERR_HIP;
  hipError_t e = mdr::launch_sample(episode_args(*env), (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "reset");
  env->has_episode = true;
  env->k = 0;
  return MDR_OK;
}

int mdr_env_load_episode(mdr_env_t* env, const mdr_episode_t* ep, uint64_t seed, uint32_t episode_index, void* stream) {
  if (!env || !ep) return MDR_ERR_INVALID;
  if (!env->bound) return fail(env, MDR_ERR_UNBOUND, "buffers not bound");
  if (ep->struct_size != sizeof(mdr_episode_t)) return fail(env, MDR_ERR_INVALID, "mdr_episode_t size mismatch (ABI)");
  if (!ep->Ta || !ep->Tm || !ep->target || !ep->deadband || !ep->Ua || !ep->Cm || !ep->Ca || !ep->Hm || !ep->capacity ||
      !ep->COP || !ep->latent || !ep->lockout || !ep->t0)
    return fail(env, MDR_ERR_INVALID, "mdr_episode_t has NULL arrays");
  env->seed = seed;
  env->episode = episode_index;
  env->has_tables = false;
  env->split_pending = false;
  env->records_stride = 0;
  env->interp_due = false;
  if (settle_prefetch(env, (hipStream_t)stream) != MDR_OK) return MDR_ERR_HIP;
  hipError_t e = mdr::launch_load(episode_args(*env), *ep, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "load_episode");
  env->has_episode = true;
  env->k = 0;
  return MDR_OK;
}

int mdr_env_set_od_table(mdr_env_t* env, const double* od_table, int64_t rows) {
  if (!env) return MDR_ERR_INVALID;
  if (od_table && rows < 1) return fail(env, MDR_ERR_INVALID, "od table needs rows >= 1");
  env->od_ext = od_table;
  env->od_ext_rows = od_table ? rows : 0;
  env->prefetched = false;   // a window built ahead used the old sequence: the next refill is built in place
  return MDR_OK;
}

int mdr_env_set_interp_grid(mdr_env_t* env, const mdr_interp_grid_t* grid) {
  if (!env || !grid) return MDR_ERR_INVALID;
  if (grid->struct_size != sizeof(mdr_interp_grid_t)) return fail(env, MDR_ERR_INVALID, "mdr_interp_grid_t size mismatch (ABI)");
  if (!grid->values) return fail(env, MDR_ERR_INVALID, "interpolation grid values is NULL");
  if (grid->update_period < 1 || grid->nb_agents < 1) return fail(env, MDR_ERR_INVALID, "interp_update_period and interp_nb_agents must be >= 1");
  for (int d = 0; d < MDR_INTERP_AXES; ++d) {
    const bool linear = !(d < 4 || d == 7);
    if (grid->dims[d] < (linear ? 2 : 1) || grid->dims[d] > MDR_INTERP_MAX_AXIS) return fail(env, MDR_ERR_INVALID, "interpolation axis length out of range");
    for (int i = 1; i < grid->dims[d]; ++i)
      if (!(grid->axes[d][i] > grid->axes[d][i - 1])) return fail(env, MDR_ERR_INVALID, "interpolation axes must be strictly ascending");
  }
  env->interp = *grid;
  env->has_interp = true;
  env->interp_steps = (grid->update_period + env->cfg.time_step - 1) / env->cfg.time_step;
  if (env->cfg.table_steps < env->interp_steps)
    return fail(env, MDR_ERR_INVALID, "table_steps must be >= ceil(interp_update_period / time_step) in interpolation mode");
  return MDR_OK;
}

int mdr_env_begin_episode(mdr_env_t* env, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!env->bound || !env->has_episode) return fail(env, MDR_ERR_UNBOUND, "call reset or load_episode first");
  if (interp_mode(env)) {
    if (!env->has_interp) return fail(env, MDR_ERR_UNBOUND, "base_power_mode interpolation: call mdr_env_set_interp_grid first");
    if (!env->buf.base_power) return fail(env, MDR_ERR_UNBOUND, "buffer 'base_power' is NULL");
  }
  env->k = 0;
  env->interp_due = false;
  int rc = settle_prefetch(env, (hipStream_t)stream);
  if (rc != MDR_OK) return rc;
  if (interp_mode(env) && sharded(env)) {   // the signal rows stay provisional until mdr_env_interp_apply
    rc = fill_tables(env, 0, (hipStream_t)stream);
    env->interp_due = true;
  } else {
    rc = interp_mode(env) ? refresh_interp(env, 0, (hipStream_t)stream) : fill_tables(env, 0, (hipStream_t)stream);
  }
  if (rc != MDR_OK) return rc;
  env->has_tables = true;
  mdr::StepArgs a;
  rc = step_args(env, nullptr, MDR_ACTIONS_BANGBANG, (hipStream_t)stream, &a);  // row 0 of the fresh tables
  if (rc != MDR_OK) return rc;
  hipError_t e = mdr::launch_reset_obs(a, true, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "reset_obs");
  return issue_prefetch(env, (hipStream_t)stream);   // the second window's tables, built while the first is stepped through
}

int mdr_env_refresh_obs(mdr_env_t* env, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!env->bound || !env->has_tables) return fail(env, MDR_ERR_UNBOUND, "no episode: call reset/load_episode and begin_episode first");
  if (!env->buf.obs) return fail(env, MDR_ERR_UNBOUND, "buffer 'obs' is NULL");
  if (env->split_pending) return fail(env, MDR_ERR_INVALID, "step_begin without step_end");
  if (capturing((hipStream_t)stream) && graph_mode(env)) return fail(env, MDR_ERR_INVALID, "mdr_env_refresh_obs takes its table row from the host cursor: not inside a capture");
  mdr::StepArgs a;
  int rc = step_args(env, nullptr, MDR_ACTIONS_BANGBANG, (hipStream_t)stream, &a);   // sig_old = the row of the current time index
  if (rc != MDR_OK) return rc;
  hipError_t e = mdr::launch_reset_obs(a, false, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "refresh_obs");
  return MDR_OK;
}

int mdr_env_step(mdr_env_t* env, uint8_t* actions, int action_source, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (env->cfg.nb_houses_total != env->cfg.nb_houses)
    return fail(env, MDR_ERR_INVALID, "houses are sharded: use mdr_env_step_begin / all-reduce / mdr_env_step_end");
  if (env->split_pending) return fail(env, MDR_ERR_INVALID, "step_begin without step_end");
  mdr::StepArgs a;
  int rc = step_args(env, actions, action_source, (hipStream_t)stream, &a);
  if (rc != MDR_OK) return rc;
  const bool graph = graph_mode(env);
  if (graph) {
    rc = sync_cursor(env, (hipStream_t)stream);
    if (rc != MDR_OK) return rc;
    graph_rows(env, &a);
  }
  const bool recording = graph && capturing((hipStream_t)stream);   // launches are recorded, not run: the host state must not move
  if (recording && mdr_env_graph_room(env) - env->captured < 1)
    return fail(env, MDR_ERR_INVALID, "graph mode: the next step lands on an interpolatePower update or past the time tables (mdr_env_graph_room() steps were recorded): run it un-captured");
  if (!recording) env->captured = 0;
  if (graph) {
    a.cursor_adv = env->buf.cursor;   // the step's last kernel moves the cursor on
    a.cursor_steps = 1;
  }
  hipError_t e = mdr::launch_step(a, env->plan, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "step");
  if (graph) {
    if (recording) { env->captured += 1; return MDR_OK; }   // mdr_env_graph_replayed accounts for every replay (and runs a due interpolation update)
    env->dev_row += 1;
    env->dev_k += 1;
  }
  env->k += 1;
  return interp_boundary(env, (hipStream_t)stream, nullptr);
}

int mdr_env_pack(mdr_env_t* env, int32_t env_index, double* out, void* stream) {
  if (!env || !out) return MDR_ERR_INVALID;
  if (!env->bound || !env->has_tables) return fail(env, MDR_ERR_UNBOUND, "no episode: call reset/load_episode and begin_episode first");
  if (env_index < 0 || env_index >= env->cfg.nb_envs) return fail(env, MDR_ERR_INVALID, "env_index out of range");
  if (env->k - env->j0 > env->cfg.table_steps) return fail(env, MDR_ERR_INVALID, "cursor outside the tables");
  const mdr_buffers_t& b = env->buf;
  const int64_t row = env->k - env->j0;
  mdr::StepArgs a{};
  a.Ta = b.Ta; a.Tm = b.Tm; a.sso = b.sso; a.flags = b.flags; a.reward = b.reward; a.P = b.P;
  a.N = env->cfg.nb_houses; a.E = env->cfg.nb_envs;
  a.od_old = b.tab_od + row * env->cfg.nb_envs;          // rows of the current time index
  a.solar_new = b.tab_solar + row * env->cfg.nb_envs;
  a.sig_old = b.tab_signal + row * env->cfg.nb_envs;
  hipError_t e = mdr::launch_pack_env(a, env_index, env->cfg.temp_ref, b.max_power, b.ratio,
                                      b.tab_abs_noise ? b.tab_abs_noise + row * env->cfg.nb_envs : nullptr, out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "pack_env");
  return MDR_OK;
}

int64_t mdr_env_graph_room(const mdr_env_t* env) {
  if (!env || !env->bound || !env->has_tables) return 0;
  int64_t room = env->cfg.table_steps - (env->k - env->j0);
  // interpolation mode: the step that LANDS on an interpolatePower update is never replayed - the update is host work, and
  // whatever a graph enqueues behind the step (observation, metrics) would still read the signal built from the old base power
  if (interp_mode(env)) room = std::min<int64_t>(room, env->interp_steps - (env->k % env->interp_steps) - 1);
  if (interp_mode(env) && sharded(env)) room = 0;   // the base-power exchange of sharded houses is host work at every update: no replays
  return room < 0 ? 0 : room;
}

int mdr_env_graph_replayed(mdr_env_t* env, int64_t n, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!graph_mode(env)) return fail(env, MDR_ERR_INVALID, "graph mode is off: bind mdr_buffers_t.cursor");
  if (!env->has_tables) return fail(env, MDR_ERR_UNBOUND, "no episode: call reset/load_episode and begin_episode first");
  if (n < 0 || n > mdr_env_graph_room(env)) return fail(env, MDR_ERR_INVALID, "more steps replayed than mdr_env_graph_room() allowed");
  if (capturing((hipStream_t)stream)) return fail(env, MDR_ERR_INVALID, "mdr_env_graph_replayed inside a capture");
  env->captured = 0;
  env->k += n;                       // the device advanced itself n times
  env->dev_row += n;
  env->dev_k += n;
  int rc = sharded(env) ? MDR_OK : interp_boundary(env, (hipStream_t)stream, nullptr);   // a due interpolatePower update (rebuilds the tables)
  if (rc != MDR_OK) return rc;
  if (env->k - env->j0 >= env->cfg.table_steps) {                // the tables are used up: refill from the current time index
    rc = fill_tables(env, env->k, (hipStream_t)stream);
    if (rc != MDR_OK) return rc;
  }
  return sync_cursor(env, (hipStream_t)stream);
}

// Unsharded envs on the split path (more than 4096 houses): inside a rollout the finish of step k and the partial of step k + 1
// share a launch (k_step_finish_partial), as they do for sharded houses - n + 1 launches instead of 2 n.  The finish re-sums the
// records of step k while the same launch writes those of step k + 1, so the two alternate between the halves of `partials`
// (its documented size, [E][mdr_partials_per_env()][3], holds two sets of records whenever nb_houses % 4 == 0).
static int rollout_split(mdr_env_t* env, uint8_t* actions, int action_source, int32_t nb_steps, hipStream_t s) {
  const int64_t half = (int64_t)env->cfg.nb_envs * env->nblk * 3;
  double* slot[2] = {env->buf.partials, env->buf.partials + half};
  mdr::StepArgs f, p;
  int q = 0;
  int rc = step_args(env, actions, action_source, s, &p);   // refills used-up tables
  if (rc != MDR_OK) return rc;
  p.partials = slot[q];
  hipError_t e = mdr::launch_step_begin_split(p, false, s);
  if (e != hipSuccess) return hip_fail(env, e, "rollout (begin)");
  for (int32_t i = 1; i < nb_steps; ++i) {
    rc = step_args(env, nullptr, MDR_ACTIONS_BANGBANG, s, &f);   // rows of the step begun last
    if (rc != MDR_OK) return rc;
    f.records = slot[q];
    f.world = 1;
    if (env->k + 1 - env->j0 >= env->cfg.table_steps) {   // the next step leaves the time tables: finish, refill, begin
      e = mdr::launch_step_end_split(f, s);
      if (e != hipSuccess) return hip_fail(env, e, "rollout (end)");
      env->k += 1;
      rc = step_args(env, actions, action_source, s, &p);
      if (rc != MDR_OK) return rc;
      p.partials = slot[q];
      e = mdr::launch_step_begin_split(p, false, s);
      if (e != hipSuccess) return hip_fail(env, e, "rollout (begin)");
      continue;
    }
    env->k += 1;
    rc = step_args(env, actions, action_source, s, &p);
    if (rc != MDR_OK) return rc;
    p.partials = slot[1 - q];
    e = mdr::launch_step_end_begin_split(f, p, s);
    if (e != hipSuccess) return hip_fail(env, e, "rollout (end + begin)");
    q = 1 - q;
  }
  rc = step_args(env, nullptr, MDR_ACTIONS_BANGBANG, s, &f);
  if (rc != MDR_OK) return rc;
  f.records = slot[q];
  f.world = 1;
  e = mdr::launch_step_end_split(f, s);
  if (e != hipSuccess) return hip_fail(env, e, "rollout (end)");
  env->k += 1;
  return MDR_OK;
}

int mdr_env_rollout(mdr_env_t* env, uint8_t* actions, int action_source, int32_t nb_steps, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (nb_steps < 0) return fail(env, MDR_ERR_INVALID, "nb_steps must be >= 0");
  static const bool fuse_split = [] { const char* t = getenv("MDR_ROLLOUT_SPLIT_FUSED"); return !(t && t[0] == '0'); }();   // experiment knob
  if (fuse_split && nb_steps > 1 && env->bound && env->has_tables && env->plan.kind == mdr::STEP_SPLIT && env->buf.pen_stash && env->buf.partials &&
      !sharded(env) && !env->split_pending && !interp_mode(env) && !graph_mode(env) &&
      mdr_partials_per_env(env->cfg.nb_houses) >= 2 * env->nblk)
    return rollout_split(env, actions, action_source, nb_steps, (hipStream_t)stream);
  for (int32_t i = 0; i < nb_steps; ++i) {
    int rc = mdr_env_step(env, actions, action_source, stream);
    if (rc != MDR_OK) return rc;
  }
  return MDR_OK;
}

int mdr_env_rollout_fused(mdr_env_t* env, uint8_t* actions, int32_t nb_steps, const mdr_rollout_out_t* out, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (nb_steps < 0) return fail(env, MDR_ERR_INVALID, "nb_steps must be >= 0");
  if (out && out->struct_size != sizeof(mdr_rollout_out_t)) return fail(env, MDR_ERR_INVALID, "mdr_rollout_out_t size mismatch (ABI)");
  if (env->cfg.nb_houses_total != env->cfg.nb_houses)
    return fail(env, MDR_ERR_UNSUPPORTED, "fused rollout needs unsharded houses");
  if (env->split_pending) return fail(env, MDR_ERR_INVALID, "step_begin without step_end");
  const int E = env->cfg.nb_envs;
  if (!mdr::rollout_fused_supported(env->rollout_plan)) {
    // no env-per-workgroup kernel for this shape (N > 2048, or N > 512 with N % 4 != 0): single steps, same accumulators
    for (int32_t i = 0; i < nb_steps; ++i) {
      int rc = mdr_env_step(env, actions, env->controller, stream);
      if (rc != MDR_OK) return rc;
      if (!out) continue;
      mdr::StepArgs a;
      rc = step_args(env, actions, env->controller, (hipStream_t)stream, &a);   // rows of the new time index
      if (rc != MDR_OK) return rc;
      mdr::RolloutArgs r{};
      r.nsteps = 1;
      r.power_trace = out->power_trace ? out->power_trace + (int64_t)i * E : nullptr;
      r.reward_sum = out->reward_sum;
      r.sq_temp_error_sum = out->sq_temp_error_sum;
      r.sq_signal_error_sum = out->sq_signal_error_sum;
      hipError_t e = mdr::launch_rollout_accumulate(a, r, (hipStream_t)stream);
      if (e != hipSuccess) return hip_fail(env, e, "rollout_accumulate");
    }
    return MDR_OK;
  }
  int32_t done = 0;
  while (done < nb_steps) {
    mdr::StepArgs a;
    int rc = step_args(env, actions, env->controller, (hipStream_t)stream, &a);   // refills the tables if the cursor left them
    if (rc != MDR_OK) return rc;
    int64_t room = env->cfg.table_steps - (env->k - env->j0);   // steps the current tables still cover
    if (interp_mode(env)) room = std::min<int64_t>(room, env->interp_steps - (env->k - env->j0));   // stop at the next update
    mdr::RolloutArgs r{};
    r.nsteps = (int)std::min<int64_t>(room, nb_steps - done);
    r.defer_last_signal_error = interp_mode(env) && ((env->k + r.nsteps) % env->interp_steps == 0);
    if (out) {
      r.power_trace = out->power_trace ? out->power_trace + (int64_t)done * E : nullptr;
      r.reward_sum = out->reward_sum;
      r.sq_temp_error_sum = out->sq_temp_error_sum;
      r.sq_signal_error_sum = out->sq_signal_error_sum;
    }
    hipError_t e = mdr::launch_rollout_fused(a, r, env->rollout_plan, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(env, e, "rollout_fused");
    env->k += r.nsteps;
    done += r.nsteps;
    rc = interp_boundary(env, (hipStream_t)stream, out ? out->sq_signal_error_sum : nullptr);
    if (rc != MDR_OK) return rc;
  }
  return MDR_OK;
}

int mdr_env_greedy_myopic_actions(mdr_env_t* env, uint8_t* actions, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!actions) return fail(env, MDR_ERR_INVALID, "actions is NULL");
  if (!env->bound || !env->has_tables) return fail(env, MDR_ERR_UNBOUND, "no episode: call reset/load_episode and begin_episode first");
  if (sharded(env)) return fail(env, MDR_ERR_UNSUPPORTED, "GreedyMyopic ranks ALL houses of an env: not over sharded houses");
  if (env->cfg.nb_houses > 2048) return fail(env, MDR_ERR_UNSUPPORTED, "GreedyMyopic: at most 2048 houses per env (one workgroup sorts an env)");
  if (env->split_pending) return fail(env, MDR_ERR_INVALID, "step_begin without step_end");
  mdr::StepArgs a;
  int rc = step_args(env, actions, MDR_ACTIONS_EXTERNAL, (hipStream_t)stream, &a);   // sig_old = the signal of the current time index
  if (rc != MDR_OK) return rc;
  const hipError_t e = mdr::launch_greedy_myopic(a, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "greedy_myopic");
  return MDR_OK;
}

int mdr_env_set_controller(mdr_env_t* env, int action_source) {
  if (!env) return MDR_ERR_INVALID;
  if (action_source != MDR_ACTIONS_BANGBANG && action_source != MDR_ACTIONS_DEADBAND && action_source != MDR_ACTIONS_ALWAYS_ON)
    return fail(env, MDR_ERR_INVALID, "controller: MDR_ACTIONS_BANGBANG, MDR_ACTIONS_DEADBAND or MDR_ACTIONS_ALWAYS_ON");
  env->controller = action_source;
  return MDR_OK;
}

int64_t mdr_mailbox_bytes(int32_t nb_envs, int32_t world, int32_t records_per_env) {
  if (nb_envs < 1 || world < 1 || world > MDR_MAX_SHARDS || records_per_env < 1) return 0;
  return mdr::persist_mailbox_granules(nb_envs, world, records_per_env) * 8;
}

int64_t mdr_persist_records(int32_t nb_houses) { return nb_houses < 1 ? 0 : mdr::split_blocks(nb_houses, 256); }

int mdr_env_rollout_persistent(mdr_env_t* env, uint8_t* actions, int32_t nb_steps, const mdr_rollout_out_t* out, const mdr_mailbox_t* mb,
                               void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!mb) return fail(env, MDR_ERR_INVALID, "mailbox is NULL");
  if (mb->struct_size != sizeof(mdr_mailbox_t)) return fail(env, MDR_ERR_INVALID, "mdr_mailbox_t size mismatch (ABI)");
  if (out && out->struct_size != sizeof(mdr_rollout_out_t)) return fail(env, MDR_ERR_INVALID, "mdr_rollout_out_t size mismatch (ABI)");
  if (nb_steps < 0) return fail(env, MDR_ERR_INVALID, "nb_steps must be >= 0");
  if (!env->bound || !env->has_tables) return fail(env, MDR_ERR_UNBOUND, "no episode: call reset/load_episode and begin_episode first");
  if (env->split_pending) return fail(env, MDR_ERR_INVALID, "step_begin without step_end");
  if (interp_mode(env)) return fail(env, MDR_ERR_UNSUPPORTED, "interpolated base power: the update is host work between steps");
  const mdr_config_t& c = env->cfg;
  if (c.nb_envs > 65535) return fail(env, MDR_ERR_UNSUPPORTED, "the persistent rollout puts the env index on grid.y: nb_envs must be <= 65535");
  if (mb->world < 1 || mb->world > MDR_MAX_SHARDS || mb->rank < 0 || mb->rank >= mb->world)
    return fail(env, MDR_ERR_INVALID, "mailbox: need 0 <= rank < world <= MDR_MAX_SHARDS");
  if (mb->world == 1 && sharded(env)) return fail(env, MDR_ERR_INVALID, "mailbox: a shard of the env needs its peers (world > 1)");
  if (mb->co_resident < 1) return fail(env, MDR_ERR_INVALID, "mailbox: co_resident must be >= 1");
  const int64_t mine = mdr::split_blocks(c.nb_houses, 256);   // one record per 256-thread workgroup: 1024 houses, 256 when nb_houses % 4 != 0
  if (mb->records[mb->rank] != mine) return fail(env, MDR_ERR_INVALID, "mailbox: records[rank] is not this handle's workgroup count");
  for (int r = 0; r < mb->world; ++r) {
    if (mb->records[r] < 1 || mb->records[r] > mb->records_per_env) return fail(env, MDR_ERR_INVALID, "mailbox: records[r] must be in [1, records_per_env]");
    if (!mb->boxes[r] || ((uintptr_t)mb->boxes[r] & 7u) != 0) return fail(env, MDR_ERR_INVALID, "mailbox: boxes[r] is NULL or not 8-byte aligned");
  }
  if (nb_steps == 0) return MDR_OK;
  hipStream_t s = (hipStream_t)stream;
  if (capturing(s)) return fail(env, MDR_ERR_INVALID, "the persistent rollout counts its steps on the host: it cannot be captured");
  const bool sys = mb->system_scope != 0;
  // how far the houses run ahead of the totals: deeper hides more of the exchange latency (each level costs 4 KB of LDS per workgroup)
  static const int depth = [] { const char* t = getenv("MDR_PERSIST_DEPTH"); const int v = t ? atoi(t) : mdr::PERSIST_MAX_DEPTH; return std::max(1, std::min(v, mdr::PERSIST_MAX_DEPTH)); }();
  int64_t resident = 0;
  hipError_t e = mdr::persist_resident_blocks(c.nb_houses % 4 == 0 ? 4 : 1, sys, depth, &resident);
  if (e != hipSuccess) return hip_fail(env, e, "occupancy query");
  int64_t all_records = 0;
  for (int r = 0; r < mb->world; ++r) all_records += mb->records[r];
  int reducers = std::min(mdr::persist_reducers(all_records), depth);
  // (their partial sums of the squared signal error go through `partials`, the split path's scratch: [E][...][3] doubles, idle here)
  if (reducers > 1 && (!env->buf.partials || (int64_t)reducers > mdr_partials_per_env(c.nb_houses) * 3)) reducers = 1;
  const int64_t grid = (mine + reducers) * c.nb_envs;
  if (grid * mb->co_resident > resident) {
    char msg[200];
    snprintf(msg, sizeof msg, "persistent rollout: %lld workgroups x %d co-resident launches exceed the %lld the device holds at once",
             (long long)grid, (int)mb->co_resident, (long long)resident);
    return fail(env, MDR_ERR_UNSUPPORTED, msg);
  }
  int32_t done = 0;
  while (done < nb_steps) {
    mdr::StepArgs a;
    int rc = step_args(env, actions, env->controller, s, &a);   // refills the tables if the cursor left them
    if (rc != MDR_OK) return rc;
    const int64_t room = c.table_steps - (env->k - env->j0);
    mdr::RolloutArgs r{};
    r.nsteps = (int)std::min<int64_t>(std::min<int64_t>(room, mdr::PERSIST_MAX_STEPS), nb_steps - done);
    if (out) {
      r.power_trace = out->power_trace ? out->power_trace + (int64_t)done * c.nb_envs : nullptr;
      r.reward_sum = out->reward_sum;
      r.sq_temp_error_sum = out->sq_temp_error_sum;
      r.sq_signal_error_sum = out->sq_signal_error_sum;
    }
    mdr::PersistArgs m{};
    for (int q = 0; q < mb->world; ++q) {
      m.box[q] = mb->boxes[q];
      m.nrec[q] = mb->records[q];
    }
    m.world = mb->world;
    m.rank = mb->rank;
    m.stride = mb->records_per_env;
    m.tag_base = env->mailbox_tag;
    m.spin_limit = mb->spin_limit ? mb->spin_limit : (1u << 20);
    m.depth = depth;
    m.reducers = reducers;
    m.serr_part = reducers > 1 ? env->buf.partials : nullptr;   // [reducers][E] doubles of the split path's scratch (idle here)
    e = mdr::launch_rollout_persist(a, r, m, sys, s);
    if (e != hipSuccess) return hip_fail(env, e, "rollout_persist");
    e = mdr::launch_persist_combine(m, c.nb_envs, r.sq_signal_error_sum, s);
    if (e != hipSuccess) return hip_fail(env, e, "persist_combine");
    env->mailbox_tag += (uint32_t)r.nsteps + 1u;   // + the pseudo-step that carries the squared temperature errors
    env->k += r.nsteps;
    done += r.nsteps;
  }
  env->dev_row = env->dev_k = -1;   // graph mode: the device cursor did not move
  return MDR_OK;
}

int mdr_mailbox_alloc(int64_t bytes, int32_t fine_grained, uint64_t** out) {
  if (!out || bytes < 8) return MDR_ERR_INVALID;
  void* p = nullptr;
  hipError_t e = fine_grained ? hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, (size_t)bytes);
  if (e != hipSuccess) return MDR_ERR_HIP;
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p);
    return MDR_ERR_HIP;
  }
  *out = (uint64_t*)p;
  return MDR_OK;
}

int mdr_mailbox_free(uint64_t* box) { return (!box || hipFree(box) == hipSuccess) ? MDR_OK : MDR_ERR_HIP; }

int mdr_mailbox_export(uint64_t* box, uint8_t handle[64]) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  if (!box || !handle) return MDR_ERR_INVALID;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, box) != hipSuccess) return MDR_ERR_HIP;
  memcpy(handle, &h, 64);
  return MDR_OK;
}

int mdr_mailbox_open(const uint8_t handle[64], uint64_t** out) {
  if (!handle || !out) return MDR_ERR_INVALID;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return MDR_ERR_HIP;
  *out = (uint64_t*)p;
  return MDR_OK;
}

int mdr_mailbox_peek(const uint64_t* box, uint64_t* word0) {
  if (!box || !word0) return MDR_ERR_INVALID;
  return hipMemcpy(word0, box, 8, hipMemcpyDeviceToHost) == hipSuccess ? MDR_OK : MDR_ERR_HIP;
}

int mdr_mailbox_close(uint64_t* box) { return (!box || hipIpcCloseMemHandle(box) == hipSuccess) ? MDR_OK : MDR_ERR_HIP; }

int mdr_env_step_begin(mdr_env_t* env, uint8_t* actions, int action_source, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (env->split_pending) return fail(env, MDR_ERR_INVALID, "step_begin called twice");
  if (env->interp_due)
    return fail(env, MDR_ERR_INVALID, "base power update pending: mdr_env_interp_local, SUM all-reduce of base_power, mdr_env_interp_apply");
  if (env->bound && !env->buf.partials) return fail(env, MDR_ERR_UNBOUND, "buffer 'partials' is NULL");
  mdr::StepArgs a;
  int rc = step_args(env, actions, action_source, (hipStream_t)stream, &a);
  if (rc != MDR_OK) return rc;
  hipError_t e = mdr::launch_step_begin_split(a, true, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "step_begin");
  env->split_pending = true;
  return MDR_OK;
}

int mdr_env_step_begin_records(mdr_env_t* env, uint8_t* actions, int action_source, int32_t records_per_env, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (env->split_pending) return fail(env, MDR_ERR_INVALID, "step_begin called twice");
  if (env->interp_due)
    return fail(env, MDR_ERR_INVALID, "base power update pending: mdr_env_interp_local, SUM all-reduce of base_power, mdr_env_interp_apply");
  if (env->bound && !env->buf.partials) return fail(env, MDR_ERR_UNBOUND, "buffer 'partials' is NULL");
  if (records_per_env < env->nblk) return fail(env, MDR_ERR_INVALID, "records_per_env smaller than mdr_env_partial_records()");
  mdr::StepArgs a;
  int rc = step_args(env, actions, action_source, (hipStream_t)stream, &a);
  if (rc != MDR_OK) return rc;
  a.nblk = records_per_env;
  if (graph_mode(env)) {   // graph mode: begin - collective - end can be captured as ONE hipGraph node sequence and replayed
    if (!capturing((hipStream_t)stream)) env->captured = 0;
    else if (mdr_env_graph_room(env) - env->captured < 1)
      return fail(env, MDR_ERR_INVALID, "graph mode: no room to replay a step here (mdr_env_graph_room() steps were recorded): run it un-captured");
    rc = sync_cursor(env, (hipStream_t)stream);
    if (rc != MDR_OK) return rc;
    graph_rows(env, &a);
    a.cursor_adv = env->buf.cursor;   // k_step_partial notes the row for the finish, which moves the cursor on
    a.snap_wr = 0;
    if (capturing((hipStream_t)stream)) env->captured += 1;   // a step recorded: mdr_env_graph_replayed accounts for every replay
  }
  hipError_t e = mdr::launch_step_begin_split(a, false, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "step_begin_records");
  env->split_pending = true;
  env->records_stride = records_per_env;
  env->snap_slot = 0;
  env->pending_adv = 1;
  return MDR_OK;
}

// Finish of the pending step and partial of the next one in ONE launch: between two exchanges of a rollout there is then one
// launch instead of two.  MDR_ERR_UNSUPPORTED where the two halves cannot share a launch (the next step leaves the time tables,
// interpolated base power, no pen_stash bound): the caller then takes mdr_env_step_end_records + mdr_env_step_begin_records.
int mdr_env_step_end_begin_records(mdr_env_t* env, const double* records, int32_t world, uint8_t* actions, int action_source, void* stream) {
  if (!env || !records) return MDR_ERR_INVALID;
  if (!env->split_pending || env->records_stride <= 0) return fail(env, MDR_ERR_INVALID, "no mdr_env_step_begin_records is pending");
  if (world < 1) return fail(env, MDR_ERR_INVALID, "world must be >= 1");
  if (!env->buf.pen_stash) return fail(env, MDR_ERR_UNSUPPORTED, "buffer 'pen_stash' is not bound");
  if (interp_mode(env)) return fail(env, MDR_ERR_UNSUPPORTED, "interpolated base power: every step may end in an update");
  const bool graph = graph_mode(env);
  const bool recording = graph && capturing((hipStream_t)stream);
  if (recording) {
    if (mdr_env_graph_room(env) - env->captured < 1)
      return fail(env, MDR_ERR_INVALID, "graph mode: no room to replay a step here (mdr_env_graph_room() steps were recorded): run it un-captured");
  } else if (env->k + 1 - env->j0 >= env->cfg.table_steps) {
    return fail(env, MDR_ERR_UNSUPPORTED, "the next step leaves the time tables: finish, then begin");
  }
  mdr::StepArgs f, p;
  int rc = step_args(env, nullptr, MDR_ACTIONS_BANGBANG, (hipStream_t)stream, &f);   // rows of the pending step k
  if (rc != MDR_OK) return rc;
  f.records = records;
  f.nblk = (int)env->records_stride;
  f.world = world;
  if (!recording) env->k += 1;                                                          // step k is finished by this launch
  rc = step_args(env, actions, action_source, (hipStream_t)stream, &p);                  // rows of step k + 1 (no refill: checked above)
  if (rc != MDR_OK) {
    if (!recording) env->k -= 1;   // nothing was launched: the step stays pending
    return rc;
  }
  p.nblk = (int)env->records_stride;
  if (graph) {   // rows from the notes on the device (k_step_finish_partial)
    graph_rows(env, &f);
    graph_rows(env, &p);
    p.cursor_adv = env->buf.cursor;
    p.snap_rd = env->snap_slot;
    p.snap_wr = 1 - env->snap_slot;
    if (recording) env->captured += 1;
    else env->captured = 0;
  }
  hipError_t e = mdr::launch_step_end_begin_split(f, p, (hipStream_t)stream);
  if (e != hipSuccess) {
    if (!recording) env->k -= 1;
    return hip_fail(env, e, "step_end_begin_records");
  }
  env->snap_slot = 1 - env->snap_slot;
  env->pending_adv += 1;
  return MDR_OK;
}

static int step_end_impl(mdr_env_t* env, const double* gathered, const double* records, int32_t world, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!env->split_pending) return fail(env, MDR_ERR_INVALID, "step_end without step_begin");
  if ((gathered || records) && world < 1) return fail(env, MDR_ERR_INVALID, "world must be >= 1");
  if ((records != nullptr) != (env->records_stride > 0))
    return fail(env, MDR_ERR_INVALID, "mdr_env_step_begin pairs with mdr_env_step_end / _gathered, mdr_env_step_begin_records with mdr_env_step_end_records");
  mdr::StepArgs a;
  int rc = step_args(env, nullptr, MDR_ACTIONS_BANGBANG, (hipStream_t)stream, &a);  // actions unused here
  if (rc != MDR_OK) return rc;
  a.gathered = gathered;
  a.records = records;
  if (records) a.nblk = (int)env->records_stride;
  env->records_stride = 0;
  a.world = world;
  const bool graph = graph_mode(env) && records != nullptr;      // the records pair is the graph-capable one
  const bool recording = graph && capturing((hipStream_t)stream);
  const int adv = env->pending_adv > 0 ? env->pending_adv : 1;
  if (graph) {
    graph_rows(env, &a);
    a.cursor_adv = env->buf.cursor;   // k_step_finish moves the cursor on: by every step begun since it last moved
    a.snap_rd = env->snap_slot;
    a.cursor_steps = adv;
  }
  hipError_t e = mdr::launch_step_end_split(a, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "step_end");
  env->split_pending = false;
  env->pending_adv = 0;
  if (graph) {
    if (recording) return MDR_OK;   // recorded, not run: mdr_env_graph_replayed accounts for every replay
    env->dev_row += adv;
    env->dev_k += adv;
  }
  env->k += 1;
  if (interp_mode(env) && env->k % env->interp_steps == 0) {
    if (!sharded(env)) return interp_boundary(env, (hipStream_t)stream, nullptr);
    env->interp_due = true;   // the caller runs the exchange: interp_local - all-reduce - interp_apply
  }
  return MDR_OK;
}

int mdr_env_interp_due(const mdr_env_t* env) { return (env && env->interp_due) ? 1 : 0; }

int mdr_env_interp_local(mdr_env_t* env, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!env->bound || !env->has_tables) return fail(env, MDR_ERR_UNBOUND, "no episode: call reset/load_episode and begin_episode first");
  if (!env->interp_due) return fail(env, MDR_ERR_INVALID, "no base power update is due");
  return interp_local(env, env->k, (hipStream_t)stream);
}

int mdr_env_interp_apply(mdr_env_t* env, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!env->bound || !env->has_tables) return fail(env, MDR_ERR_UNBOUND, "no episode: call reset/load_episode and begin_episode first");
  if (!env->interp_due) return fail(env, MDR_ERR_INVALID, "no base power update is due");
  int rc = fill_tables(env, env->k, (hipStream_t)stream);   // signal rows from the summed base power
  if (rc != MDR_OK) return rc;
  mdr::StepArgs a;
  rc = step_args(env, nullptr, MDR_ACTIONS_BANGBANG, (hipStream_t)stream, &a);
  if (rc != MDR_OK) return rc;
  hipError_t e = mdr::launch_patch_signal_plane(a, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "patch_signal_plane");
  env->interp_due = false;
  return MDR_OK;
}


int32_t mdr_obs_vector_length(const mdr_obs_spec_t* spec) {
  if (!spec || spec->struct_size != sizeof(mdr_obs_spec_t) || spec->nb_comm < 0) return -1;
  return mdr::obs_vector_length(*spec);
}

// Checks shared by the observation entry points and the argument block they all start from.
static int obs_args(mdr_env_t* env, const mdr_obs_spec_t* spec, bool need_layout, mdr::ObsArgs* out_args) {
  if (spec->struct_size != sizeof(mdr_obs_spec_t)) return fail(env, MDR_ERR_INVALID, "mdr_obs_spec_t size mismatch (ABI)");
  if (!env->bound || !env->has_tables) return fail(env, MDR_ERR_UNBOUND, "no episode: call reset/load_episode and begin_episode first");
  const mdr_config_t& c = env->cfg;
  if (need_layout && spec->layout != MDR_OBS_PLANES && spec->layout != MDR_OBS_ROWS) return fail(env, MDR_ERR_INVALID, "unknown obs layout");
  const int n = c.nb_houses_total;
  if (spec->nb_comm < 0 || spec->nb_comm > n - 1 + (n == 1 ? 1 : 0) || (n == 1 && spec->nb_comm != 0))
    return fail(env, MDR_ERR_INVALID, "nb_comm must be in [0, nb_houses - 1]");
  if (spec->random_links && spec->nb_comm > 16) return fail(env, MDR_ERR_UNSUPPORTED, "random_sample links support nb_comm <= 16");
  if (!(spec->comm_defect_prob >= 0.0 && spec->comm_defect_prob <= 1.0)) return fail(env, MDR_ERR_INVALID, "comm_defect_prob outside [0, 1]");
  if (!(spec->def_Ua > 0 && spec->def_Cm > 0 && spec->def_Ca > 0 && spec->def_Hm > 0 && spec->def_COP > 0 && spec->def_capacity > 0 &&
        spec->def_latent > 0 && spec->norm_reg_sig > 0))
    return fail(env, MDR_ERR_INVALID, "normalisation defaults must be positive");
  const mdr_buffers_t& b = env->buf;
  const int64_t row = env->k - env->j0;
  mdr::ObsArgs a{};
  a.Ta = b.Ta; a.Tm = b.Tm; a.target = b.target; a.deadband = b.deadband; a.P_max = b.P_max;
  a.Ua = b.Ua; a.Cm = b.Cm; a.Ca = b.Ca; a.Hm = b.Hm; a.capacity = b.capacity; a.COP = b.COP; a.latent = b.latent;
  a.sso = b.sso; a.lockout = b.lockout; a.flags = b.flags;
  a.P = b.P;
  a.sig_now = b.tab_signal + row * c.nb_envs;
  a.od_now = b.tab_od + row * c.nb_envs;
  a.solar_now = b.tab_solar + row * c.nb_envs;
  if (graph_mode(env)) {   // rows and time index from the device cursor (see mdr_buffers_t.cursor)
    a.sig_now = b.tab_signal; a.od_now = b.tab_od; a.solar_now = b.tab_solar;
    a.cursor = b.cursor;
    a.cursor_max = c.table_steps - 1;
  }
  a.t0 = b.t0;
  a.links = spec->random_links ? nullptr : spec->links;
  a.random_links = spec->random_links ? 1 : 0;
  a.plane = (int64_t)c.nb_envs * c.nb_houses;
  a.out_plane = spec->out_plane_stride > 0 ? spec->out_plane_stride : a.plane;
  if (a.out_plane < a.plane) return fail(env, MDR_ERR_INVALID, "out_plane_stride smaller than nb_envs * nb_houses");
  a.k = env->k;
  a.E = c.nb_envs; a.N = c.nb_houses; a.c = spec->nb_comm; a.F = mdr::obs_vector_length(*spec); a.dt = c.time_step;
  a.n_total = c.nb_houses_total;
  a.f_hour = spec->state_hour; a.f_day = spec->state_day; a.f_solar = spec->state_solar_gain;
  a.f_thermal = spec->state_thermal; a.f_hvac = spec->state_hvac;
  a.m_thermal = spec->message_thermal; a.m_hvac = spec->message_hvac;
  a.mf = mdr::obs_message_fields(*spec);
  a.env_offset = c.env_offset; a.house_offset = c.house_offset;
  a.k0 = (uint32_t)(env->seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(env->seed >> 32); a.episode = env->episode;
  a.defect_prob = (float)spec->comm_defect_prob;
  a.obs_tshift = (float)(c.temp_ref - 20.0);
  a.inv_obs_norm = 1.0 / c.obs_power_norm;
  a.inv_norm_reg = (float)(1.0 / spec->norm_reg_sig);
  a.inv_cap = (float)(1.0 / spec->def_capacity); a.inv_Ua = (float)(1.0 / spec->def_Ua); a.inv_Cm = (float)(1.0 / spec->def_Cm);
  a.inv_Ca = (float)(1.0 / spec->def_Ca); a.inv_Hm = (float)(1.0 / spec->def_Hm); a.inv_COP = (float)(1.0 / spec->def_COP);
  a.inv_latent = (float)(1.0 / spec->def_latent);
  *out_args = a;
  return MDR_OK;
}

int mdr_env_obs_vector(mdr_env_t* env, const mdr_obs_spec_t* spec, float* out, void* stream) {
  if (!env || !spec || !out) return MDR_ERR_INVALID;
  if (env->cfg.nb_houses_total != env->cfg.nb_houses)
    return fail(env, MDR_ERR_UNSUPPORTED, "sharded houses: mdr_env_obs_messages, halo exchange, mdr_env_obs_vector_ext");
  mdr::ObsArgs a;
  int rc = obs_args(env, spec, true, &a);
  if (rc == MDR_OK) rc = sync_cursor(env, (hipStream_t)stream);
  if (rc != MDR_OK) return rc;
  a.out = out;
  hipError_t e = mdr::launch_obs_vector(a, spec->layout, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "obs_vector");
  return MDR_OK;
}

int32_t mdr_obs_message_fields(const mdr_obs_spec_t* spec) {
  if (!spec || spec->struct_size != sizeof(mdr_obs_spec_t)) return -1;
  return mdr::obs_message_fields(*spec);
}

int mdr_env_obs_messages(mdr_env_t* env, const mdr_obs_spec_t* spec, float* messages, int64_t entries_per_env, void* stream) {
  if (!env || !spec || !messages) return MDR_ERR_INVALID;
  mdr::ObsArgs a;
  int rc = obs_args(env, spec, false, &a);
  if (rc == MDR_OK) rc = sync_cursor(env, (hipStream_t)stream);
  if (rc != MDR_OK) return rc;
  if (entries_per_env < env->cfg.nb_houses) return fail(env, MDR_ERR_INVALID, "entries_per_env smaller than nb_houses");
  a.msg_ext_out = messages;
  a.ext_entries = entries_per_env;
  hipError_t e = mdr::launch_obs_messages(a, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "obs_messages");
  return MDR_OK;
}

int mdr_env_obs_vector_ext(mdr_env_t* env, const mdr_obs_spec_t* spec, const float* messages, int64_t entries_per_env, float* out,
                           void* stream) {
  if (!env || !spec || !out) return MDR_ERR_INVALID;
  mdr::ObsArgs a;
  int rc = obs_args(env, spec, true, &a);
  if (rc == MDR_OK) rc = sync_cursor(env, (hipStream_t)stream);
  if (rc != MDR_OK) return rc;
  if (spec->random_links && entries_per_env < env->cfg.nb_houses_total)
    return fail(env, MDR_ERR_INVALID, "random_sample links draw among all houses of the env: messages must hold every house's record, slot = house id");
  if (spec->nb_comm > 0 && (!messages || (!spec->links && !spec->random_links)))
    return fail(env, MDR_ERR_INVALID, "messages and a link table of record slots are required");
  if (entries_per_env < 0) return fail(env, MDR_ERR_INVALID, "entries_per_env must be >= 0");
  a.out = out;
  a.msg_ext_in = messages;
  a.ext_entries = entries_per_env;
  hipError_t e = mdr::launch_obs_vector_ext(a, spec->layout, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "obs_vector_ext");
  return MDR_OK;
}

// msg_scratch / senders_scratch: the caller's scratch of mdr_env_actor_sample_links, NULL for mdr_env_actor_sample
static int actor_sample_impl(mdr_env_t* env, const mdr_obs_spec_t* spec, const mdr_actor_t* actor, float* msg_scratch, int32_t* senders_scratch,
                             uint64_t seed, uint64_t step, const int32_t* step_dev, uint8_t* action, float* a_prob, float* probs, float* rows_out,
                             void* stream) {
  if (!env || !spec || !actor || !action) return MDR_ERR_INVALID;
  if (env->cfg.nb_houses_total != env->cfg.nb_houses) return fail(env, MDR_ERR_UNSUPPORTED, "observe -> act needs unsharded houses");
  mdr::ObsArgs a;
  int rc = obs_args(env, spec, false, &a);
  if (rc == MDR_OK) rc = sync_cursor(env, (hipStream_t)stream);
  if (rc != MDR_OK) return rc;
  // circular neighbours with 4-field messages (utils.py:843-878 without the optional message columns); the optional STATE columns
  // (774-830), any neighbour count and link defects (env 988-1002) take the extended kernels
  if (spec->message_thermal || spec->message_hvac)
    return fail(env, MDR_ERR_UNSUPPORTED, "observe -> act covers 4-field messages: the optional message columns go through mdr_env_obs_vector + mdr_actor_sample");
  const bool table = spec->nb_comm > 0 && (spec->links != nullptr || spec->random_links);
  if (table && !msg_scratch)
    return fail(env, MDR_ERR_UNSUPPORTED, "observe -> act with a link table or random_sample senders gathers message records: call mdr_env_actor_sample_links with its scratch");
  if (table && spec->random_links && !senders_scratch) return fail(env, MDR_ERR_INVALID, "random_sample: senders_scratch is NULL");
  mdr::ObserveArgs o{};
  o.Ta = a.Ta; o.Tm = a.Tm; o.target = a.target; o.deadband = a.deadband; o.capacity = a.capacity; o.P_max = a.P_max;
  o.sso = a.sso; o.lockout = a.lockout; o.flags = a.flags; o.P = a.P; o.sig_now = a.sig_now;
  o.cursor = a.cursor; o.cursor_max = a.cursor_max;
  o.E = a.E; o.N = a.N;
  o.obs_tshift = a.obs_tshift; o.inv_norm_reg = a.inv_norm_reg; o.inv_cap = a.inv_cap; o.inv_obs_norm = a.inv_obs_norm;
  static const bool force_ext = [] { const char* t = getenv("MDR_OBSERVE_EXT"); return t && t[0] == '1'; }();   // experiment knob: the extended kernels on the default shape
  const bool ext = spec->state_hour || spec->state_day || spec->state_solar_gain || spec->state_thermal || spec->state_hvac ||
                   spec->nb_comm != 10 || spec->comm_defect_prob > 0.0 || force_ext || table;
  if (ext) {
    if (env->split_pending) return fail(env, MDR_ERR_INVALID, "step_begin without step_end");
    o.ext = 1;
    o.c = spec->nb_comm;
    o.before = spec->nb_comm / 2;
    o.f_hour = spec->state_hour != 0; o.f_day = spec->state_day != 0; o.f_solar = spec->state_solar_gain != 0;
    o.f_thermal = spec->state_thermal != 0; o.f_hvac = spec->state_hvac != 0;
    o.own = 11 + 2 * o.f_hour + 2 * o.f_day + o.f_solar + 5 * o.f_thermal + 2 * o.f_hvac;
    o.Ua = a.Ua; o.Cm = a.Cm; o.Ca = a.Ca; o.Hm = a.Hm; o.COP = a.COP; o.latent = a.latent;
    o.inv_Ua = a.inv_Ua; o.inv_Cm = a.inv_Cm; o.inv_Ca = a.inv_Ca; o.inv_Hm = a.inv_Hm; o.inv_COP = a.inv_COP; o.inv_latent = a.inv_latent;
    // per-env columns: the handle's local-aggregate scratch (tot_sum [2][E], tot_max [E] doubles = 6 E floats), idle between the
    // steps of an unsharded handle
    o.env_extra_a = reinterpret_cast<float*>(env->buf.tot_sum);
    o.env_extra_b = reinterpret_cast<float*>(env->buf.tot_max);
    o.od_now = a.od_now; o.solar_now = a.solar_now; o.t0 = a.t0; o.k = a.k; o.dt = a.dt;
    o.defect_prob = a.defect_prob;
    o.env_offset = a.env_offset; o.house_offset = a.house_offset;
    o.k0 = a.k0; o.k1 = a.k1; o.episode = a.episode;
  }
  if (table) {   // every house's message record of this step, and - random_sample - this step's senders, then the gather inside the actor kernel
    mdr::ObsArgs w = a;
    w.msg_ext_out = msg_scratch;
    w.ext_entries = env->cfg.nb_houses;
    hipError_t e = mdr::launch_obs_messages(w, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(env, e, "obs_messages");
    o.msg_rec = msg_scratch;
    o.links = spec->links;
    o.links_env_stride = 0;
    if (spec->random_links) {
      e = mdr::launch_comm_draws(a, senders_scratch, nullptr, (hipStream_t)stream);
      if (e != hipSuccess) return hip_fail(env, e, "comm_draws");
      o.links = senders_scratch;
      o.links_env_stride = (int64_t)env->cfg.nb_houses * spec->nb_comm;
    }
  }
  rc = mdr::launch_actor_observe(actor, o, seed, step, step_dev, action, a_prob, probs, rows_out, (hipStream_t)stream);
  if (rc == MDR_ERR_UNSUPPORTED) return fail(env, rc, "observe -> act: shape or actor layout without a kernel (nb_houses > nb_comm <= 13, at most 64 features, FRAG16 / BF16X3 packed in MDR_FEATURES_OBSERVE order for this nb_comm)");
  if (rc != MDR_OK) return fail(env, rc, "actor_observe launch failed");
  return MDR_OK;
}

int mdr_env_actor_sample(mdr_env_t* env, const mdr_obs_spec_t* spec, const mdr_actor_t* actor, uint64_t seed, uint64_t step,
                         const int32_t* step_dev, uint8_t* action, float* a_prob, float* probs, float* rows_out, void* stream) {
  return actor_sample_impl(env, spec, actor, nullptr, nullptr, seed, step, step_dev, action, a_prob, probs, rows_out, stream);
}

int mdr_env_actor_sample_links(mdr_env_t* env, const mdr_obs_spec_t* spec, const mdr_actor_t* actor, float* msg_scratch, int32_t* senders_scratch,
                               uint64_t seed, uint64_t step, const int32_t* step_dev, uint8_t* action, float* a_prob, float* probs,
                               float* rows_out, void* stream) {
  if (!msg_scratch || ((uintptr_t)msg_scratch & 15u) != 0) return env ? fail(env, MDR_ERR_INVALID, "msg_scratch must be a 16-byte aligned device buffer of nb_envs * nb_houses * 4 floats") : MDR_ERR_INVALID;
  return actor_sample_impl(env, spec, actor, msg_scratch, senders_scratch, seed, step, step_dev, action, a_prob, probs, rows_out, stream);
}

int mdr_env_comm_draws(mdr_env_t* env, const mdr_obs_spec_t* spec, int32_t* senders, uint8_t* keep, void* stream) {
  if (!env || !spec || !senders || !keep) return MDR_ERR_INVALID;
  mdr::ObsArgs a;
  int rc = obs_args(env, spec, false, &a);
  if (rc == MDR_OK) rc = sync_cursor(env, (hipStream_t)stream);
  if (rc != MDR_OK) return rc;
  hipError_t e = mdr::launch_comm_draws(a, senders, keep, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(env, e, "comm_draws");
  return MDR_OK;
}

int mdr_env_step_end(mdr_env_t* env, void* stream) { return step_end_impl(env, nullptr, nullptr, 0, stream); }

int mdr_env_step_end_gathered(mdr_env_t* env, const double* gathered, int32_t world, void* stream) {
  if (!gathered) return env ? fail(env, MDR_ERR_INVALID, "gathered is NULL") : MDR_ERR_INVALID;
  return step_end_impl(env, gathered, nullptr, world, stream);
}

int mdr_env_step_end_records(mdr_env_t* env, const double* records, int32_t world, void* stream) {
  if (!env) return MDR_ERR_INVALID;
  if (!records) {          // this device's own records: a world of one
    records = env->bound ? env->buf.partials : nullptr;
    world = 1;
  }
  if (!records) return fail(env, MDR_ERR_UNBOUND, "buffer 'partials' is NULL");
  return step_end_impl(env, nullptr, records, world, stream);
}

int mdr_env_cursor(const mdr_env_t* env, int64_t* k, int64_t* j0) {
  if (!env) return MDR_ERR_INVALID;
  if (k) *k = env->k;
  if (j0) *j0 = env->j0;
  return MDR_OK;
}

int mdr_env_set_cursor(mdr_env_t* env, uint64_t seed, uint32_t episode, int64_t k, int64_t j0) {
  if (!env) return MDR_ERR_INVALID;
  if (!env->bound) return fail(env, MDR_ERR_UNBOUND, "buffers not bound");
  if (k < 0 || j0 < 0 || k < j0 || k - j0 > env->cfg.table_steps) return fail(env, MDR_ERR_INVALID, "cursor outside the tables");
  env->seed = seed;
  env->episode = episode;
  env->k = k;
  env->j0 = j0;
  env->dev_row = env->dev_k = -1;   // the caller replaced the buffers (incl. the device cursor): nothing is known about it
  if (env->prefetched && env->ev_fill) (void)hipEventSynchronize(env->ev_fill);
  env->prefetched = false;
  use_tables(env, 0);               // ... and put the tables of (k, j0) into the first table set
  env->has_episode = true;
  env->has_tables = true;
  env->split_pending = false;
  env->records_stride = 0;
  env->interp_due = false;
  return MDR_OK;
}

}  // extern "C"
