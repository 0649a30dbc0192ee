// HIP kernels of the batched demand-response step path for MI355X (gfx950, wave64).
//
// Bound: HBM bandwidth (element-wise fp32, ~0.4 flop/B) - no MFMA.  One fused launch per env step:
// every per-house array is read/written once with 16-byte accesses, the per-env reductions stay in
// registers + LDS, per-env scalars (weather, solar, signal) come from small pre-built time tables.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "mdr_device.h"
#include "mdr_kernels.h"
#include "mdr_step_common.h"

namespace mdr {

// =================================================================================================
// Episode start
// =================================================================================================

// Writes every derived / raw per-house buffer from fp64 raw values (shared by sampling and replay).
__device__ __forceinline__ void store_house(const EpisodeArgs& a, int64_t i, double Ta, double Tm, double target,
                                            double deadband, double Ua, double Cm, double Ca, double Hm, double cap,
                                            double COP, double latent, int64_t lockout) {
  const ThermalMap m = thermal_map(Ua, Cm, Ca, Hm, (double)a.dt);
  a.b.Ta[i] = (float)(Ta - a.temp_ref);
  a.b.Tm[i] = (float)(Tm - a.temp_ref);
  a.b.sso[i] = (int32_t)lockout;  // env 434
  a.b.flags[i] = 0;               // env 432-433
  a.b.k01[i] = (float)m.k01;
  a.b.s0[i] = (float)m.s0;
  a.b.k10[i] = (float)m.k10;
  a.b.s1[i] = (float)m.s1;
  a.b.inv_Ua[i] = (float)(1.0 / Ua);
  a.b.Q_hvac[i] = (float)(-cap / (1.0 + latent));  // env 505
  a.b.P_max[i] = (float)(cap / COP);               // env 436
  a.b.target[i] = (float)(target - a.temp_ref);
  a.b.deadband[i] = (float)deadband;
  a.b.lockout[i] = (int32_t)lockout;
  a.b.Ua[i] = (float)Ua;
  a.b.Cm[i] = (float)Cm;
  a.b.Ca[i] = (float)Ca;
  a.b.Hm[i] = (float)Hm;
  a.b.capacity[i] = (float)cap;
  a.b.COP[i] = (float)COP;
  a.b.latent[i] = (float)latent;
}

// utils.apply_house_noise / apply_hvac_noise (utils.py:623-676) + HVAC.__init__ lockout noise (env 430)
__global__ __launch_bounds__(256) void k_sample_houses(EpisodeArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.E * a.N) return;
  const uint32_t e = (uint32_t)(i / a.N + a.env_offset);
  const uint32_t h = (uint32_t)(i % a.N + a.house_offset);
  const u32x4 A = philox4x32_10(e, h, a.episode, TAG_HOUSE_TEMPS, a.k0, a.k1);
  const u32x4 B = philox4x32_10(e, h, a.episode, TAG_HOUSE_HVAC, a.k0, a.k1);
  const u32x4 C = philox4x32_10(e, h, a.episode, TAG_HOUSE_THERMO, a.k0, a.k1);
  const double Ta = a.init_air + fabs(a.std_start * gauss01(A.x, A.y));
  const double Tm = a.init_mass + fabs(a.std_start * gauss01(A.z, A.w));
  const double target = a.target + fabs(a.std_target * gauss01(B.x, B.y));
  const double cap = a.caps[mulhi_pick(B.z, (uint32_t)a.ncaps)];
  const int64_t lockout = a.lockout + (-(int64_t)a.lockout_noise + mulhi_pick(B.w, (uint32_t)(2 * a.lockout_noise + 1)));
  const double Ua = a.Ua * triangular_mode1(u01(C.x), a.f_low, a.f_high);
  const double Cm = a.Cm * triangular_mode1(u01(C.y), a.f_low, a.f_high);
  const double Ca = a.Ca * triangular_mode1(u01(C.z), a.f_low, a.f_high);
  const double Hm = a.Hm * triangular_mode1(u01(C.w), a.f_low, a.f_high);
  store_house(a, i, Ta, Tm, target, a.deadband, Ua, Cm, Ca, Hm, cap, a.COP, a.latent, lockout);
}

// start datetime (utils.get_random_date_time 701-709), phase (env 789-792), artificial ratio (env 1116)
__global__ __launch_bounds__(256) void k_sample_envs(EpisodeArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.E) return;
  const u32x4 D = philox4x32_10((uint32_t)(e + a.env_offset), ENV_LEVEL, a.episode, TAG_ENV_START, a.k0, a.k1);
  int64_t t0 = a.start_epoch;
  if (a.start_random) t0 += mulhi_pick(D.x, 364u) * 86400 + mulhi_pick(D.y, 86400u);
  a.b.t0[e] = t0;
  a.b.phase[e] = a.random_phase ? u01(D.z) * 24.0 : 0.0;
  a.b.ratio[e] = a.artificial_ratio * pow(a.ratio_range, 2.0 * u01(D.w) - 1.0);
}

__global__ __launch_bounds__(256) void k_load_houses(EpisodeArgs a, mdr_episode_t ep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.E * a.N) return;
  store_house(a, i, ep.Ta[i], ep.Tm[i], ep.target[i], ep.deadband[i], ep.Ua[i], ep.Cm[i], ep.Ca[i], ep.Hm[i],
              ep.capacity[i], ep.COP[i], ep.latent[i], ep.lockout[i]);
}

__global__ __launch_bounds__(256) void k_load_envs(EpisodeArgs a, mdr_episode_t ep) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.E) return;
  a.b.t0[e] = ep.t0[e];
  a.b.phase[e] = ep.phase ? ep.phase[e] : 0.0;
  a.b.ratio[e] = ep.ratio ? ep.ratio[e] : a.artificial_ratio;
}

// Local sum of max consumption per env (ClusterHouses.__init__, env 796-802); P <- 0.  The sum is EXACT in fp64 whatever the order
// (fp32 addends within a factor of 8 of each other: 24 + log2(N) significant bits), so an env of more than 65,536 houses is summed by
// several workgroups through atomic adds - one workgroup took 0.98 ms for 1,000,000 houses at every episode start.
constexpr int MAX_POWER_CHUNK = 65536;
__global__ __launch_bounds__(256) void k_env_max_power(EpisodeArgs a, int chunks) {
  __shared__ double lds[3 * 4];
  const int e = blockIdx.x / chunks, c = blockIdx.x - e * chunks;
  const float* p = a.b.P_max + (int64_t)e * a.N;
  const int lo = c * MAX_POWER_CHUNK, hi = chunks == 1 ? a.N : min(a.N, lo + MAX_POWER_CHUNK);
  Red3 v{0.0, 0.0, 0.0f};
  for (int i = lo + threadIdx.x; i < hi; i += 256) v.sum_p += (double)p[i];
  v = block_reduce<256>(v, lds);
  if (threadIdx.x == 0) {
    if (chunks == 1) a.b.max_power[e] = v.sum_p;
    else atomicAdd(&a.b.max_power[e], v.sum_p);   // (zeroed by the launcher)
    a.b.P[e] = 0.0;
  }
}

static hipError_t launch_max_power(const EpisodeArgs& a, hipStream_t s) {
  const int chunks = a.N > MAX_POWER_CHUNK ? (a.N + MAX_POWER_CHUNK - 1) / MAX_POWER_CHUNK : 1;
  if (chunks > 1) {
    const hipError_t e = hipMemsetAsync(a.b.max_power, 0, (size_t)a.E * sizeof(double), s);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_env_max_power, dim3((unsigned)((int64_t)a.E * chunks)), dim3(256), 0, s, a, chunks);
  return hipGetLastError();
}

// Observation planes right after reset (MADemandResponseEnv.reset, env 163-170): every HVAC is off
// (env 796-801) so cluster_hvac_power = 0; reg_signal is the initial signal (table row 0); rewards <- 0.
// Also mdr_env_refresh_obs (zero_reward = 0): the seven planes of the CURRENT state, e.g. after steps taken without planes.
__global__ __launch_bounds__(256) void k_reset_obs(StepArgs a, int zero_reward) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.plane) return;
  if (zero_reward) a.reward[i] = 0.0f;
  if (a.obs == nullptr) return;
  const int e = (int)(i / a.N);
  a.obs[0 * a.plane + i] = (a.Ta[i] + a.obs_tshift) * 0.2f;
  a.obs[1 * a.plane + i] = (a.Tm[i] + a.obs_tshift) * 0.2f;
  a.obs[2 * a.plane + i] = (a.flags[i] & 1u) ? 1.0f : 0.0f;
  a.obs[3 * a.plane + i] = (a.flags[i] & 2u) ? 1.0f : 0.0f;
  a.obs[4 * a.plane + i] = (float)a.sso[i] / (float)a.lockout[i];
  a.obs[5 * a.plane + i] = (float)(a.sig_old[e] * a.inv_obs_norm);
  a.obs[6 * a.plane + i] = (float)(a.P[e] * a.inv_obs_norm);
}

// =================================================================================================
// Interpolated base power: PowerGrid.interpolatePower (env 1195-1234) + PowerInterpolator.interpolateGridFast
// (monteCarlo/interpolation.py:113-142).  One workgroup per env; thread q evaluates sampled house q.
// =================================================================================================
__device__ __forceinline__ int nearest_index(const double* ax, int n, double v) {   // np.argmin(|ax - v|): first minimum
  int best = 0;
  double bd = fabs(ax[0] - v);
  for (int i = 1; i < n; ++i) {
    const double d = fabs(ax[i] - v);
    if (d < bd) {
      bd = d;
      best = i;
    }
  }
  return best;
}

__global__ __launch_bounds__(128) void k_interp_base(InterpArgs a) {
  __shared__ double lds[3 * 2];
  const int e = blockIdx.x;
  const int64_t base = (int64_t)e * a.N;
  // sharded houses: every shard walks the same sample slots (the draws are functions of global indices) and adds the
  // houses it holds; the caller sums base_power over the shards
  const bool all = a.N_total <= a.nb_agents;
  const int count = all ? a.N_total : a.nb_agents;
  const Civil c = civil_from_epoch(a.t0[e] + a.j * (int64_t)a.dt);
  // env 1198-1207: tm_yday and seconds since midnight, or (0, 0) when the solar gain is not modelled
  const double date = a.solar_on ? (double)c.yday : 0.0;
  const double hour = a.solar_on ? (double)c.sod : 0.0;
  // strides of the C-ordered grid
  int64_t stride[MDR_INTERP_AXES];
  stride[MDR_INTERP_AXES - 1] = 1;
  for (int d = MDR_INTERP_AXES - 2; d >= 0; --d) stride[d] = stride[d + 1] * a.dims[d + 1];
  double sum = 0.0;
  for (int q = threadIdx.x; q < count; q += 128) {
    int h = q;
    if (!all) {   // random.choices(all_ids, k = interp_nb_agents), env 1214
      const u32x4 r = philox4x32_10((uint32_t)(e + a.env_offset), (uint32_t)q, (uint32_t)a.j, TAG_INTERP | (a.episode << 8), a.k0, a.k1);
      h = (int)mulhi_pick(r.x, (uint32_t)a.N_total);
    }
    h -= a.house_offset;
    if (h < 0 || h >= a.N) continue;
    const int64_t i = base + h;
    const double tgt = (double)a.target[i];   // all temperatures here are relative to temp_ref: differences are unaffected
    double p[MDR_INTERP_AXES];
    p[0] = (double)a.Ua[i] / a.def_Ua;
    p[1] = (double)a.Cm[i] / a.def_Cm;
    p[2] = (double)a.Ca[i] / a.def_Ca;
    p[3] = (double)a.Hm[i] / a.def_Hm;
    p[4] = (double)a.Ta[i] - tgt;
    p[5] = (double)a.Tm[i] - tgt;
    p[6] = (double)a.od_now[e] - tgt;
    p[7] = (double)a.capacity[i];
    p[8] = hour;
    p[9] = date;
    int64_t off = 0;
    int lo[5];
    double w[5];
    int nl = 0;
#pragma unroll
    for (int d = 0; d < MDR_INTERP_AXES; ++d) {
      const double* ax = a.axes[d];
      const int n = a.dims[d];
      const double v = fmin(fmax(p[d], ax[0]), ax[n - 1]);   // utils.clipInterpolationPoint (axes are ascending)
      if (d < 4 || d == 7) {
        off += (int64_t)nearest_index(ax, n, v) * stride[d];
      } else {   // interval index as scipy.interpolate.interpn: searchsorted(right) - 1 clipped to [0, n - 2]
        int k = 0;
        while (k < n - 2 && v >= ax[k + 1]) ++k;
        lo[nl] = k;
        w[nl] = (v - ax[k]) / (ax[k + 1] - ax[k]);
        ++nl;
      }
    }
    const int ld[5] = {4, 5, 6, 8, 9};
    double acc = 0.0;
    for (int corner = 0; corner < 32; ++corner) {
      double weight = 1.0;
      int64_t o = off;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int up = (corner >> b) & 1;
        weight *= up ? w[b] : 1.0 - w[b];
        o += (int64_t)(lo[b] + up) * stride[ld[b]];
      }
      acc += weight * a.values[o];
    }
    sum += acc;
  }
  Red3 r{sum, 0.0, 0.0f};
  r = block_reduce<128>(r, lds);
  if (threadIdx.x == 0) a.base_power[e] = r.sum_p * (all ? 1.0 : (double)a.N_total / (double)a.nb_agents);
}

// obs plane 5 (reg_signal / norm) <- the freshly computed signal of the current time index (sig_old row)
__global__ __launch_bounds__(256) void k_patch_signal_plane(StepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.plane) return;
  a.obs[5 * a.plane + i] = (float)(a.sig_old[i / a.N] * a.inv_obs_norm);
}

// =================================================================================================
// Per-env time tables: outdoor temperature, solar gain, regulation signal for K+1 consecutive time
// indices, in fp64, one thread per (row, env).  These are functions of (env, time) only - the step
// kernel reads them as wave-uniform scalars.
// =================================================================================================
__global__ __launch_bounds__(256) void k_fill_tables(TableArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.rows * a.E) return;
  const int r = (int)(i / a.E);
  const int e = (int)(i % a.E);
  const int64_t j = a.j0 + r;
  const uint32_t eg = (uint32_t)(e + a.env_offset);
  const Civil c = civil_from_epoch(a.t0[e] + j * (int64_t)a.dt);

  // ClusterHouses.compute_OD_temp (env 1057-1081): minute resolution, coldest at 06:00 + phase
  double od;
  if (a.od_ext != nullptr && j < a.od_ext_rows) {
    od = a.od_ext[j * a.E + e];
  } else {
    const double amp = 0.5 * (a.day_temp - a.night_temp), bias = 0.5 * (a.day_temp + a.night_temp);
    const double tday = (double)c.hour + (double)c.minute / 60.0;
    od = amp * sin(6.283185307179586476925286766559 * (tday + (-6.0 + a.phase[e])) / 24.0) + bias;   // amp == 0: exactly bias
    if (a.temp_std != 0.0) {   // random.gauss(0, 0) adds exactly 0 in the reference; skipping the draw changes nothing
      const u32x4 g = philox4x32_10(eg, (uint32_t)j, a.episode, TAG_OD_NOISE, a.k0, a.k1);
      od += a.temp_std * gauss01(g.x, g.y);
    }
  }
  a.tab_od[i] = (float)(od - a.temp_ref);

  // utils.house_solar_gain (utils.py:1277-1350); identical for every house of the env
  a.tab_solar[i] = a.solar_on ? (float)(a.area_shading * solar_cooling_load(c.hour, c.minute, c.month, c.day)) : 0.0f;

  // PowerGrid.step (env 1236-1316): constant base power (env 1249) or the last interpolated one (env 1250-1255)
  const double base = a.base_power ? a.base_power[e] : a.avg_power_per_hvac * (double)a.n_total;
  const double sod = (double)c.sod;
  double sig, abs_noise = 0.0;
  if (a.signal_mode == MDR_SIGNAL_FLAT) {
    sig = base;
  } else if (a.signal_mode == MDR_SIGNAL_SINUSOIDALS) {
    sig = base;
    for (int q = 0; q < a.nb_sin; ++q)
      sig += base * a.sin_ratios[q] * sin(6.283185307179586476925286766559 * sod / a.sin_periods[q]);
  } else if (a.signal_mode == MDR_SIGNAL_REGULAR_STEPS) {
    const double ampl = a.steps_amp * (double)a.n_total;
    const double duty = base / ampl;
    // np.heaviside(x, 1) is discontinuous at x == 0: the product must be rounded on its own, as NumPy does.  (HIP's
    // __dmul_rn is a plain `*` that the compiler may still contract into an FMA, hence the pragma.)
    {
#pragma clang fp contract(off)
      const double edge = (1.0 - duty) * a.steps_period;
      const double x = fmod(sod, a.steps_period) - edge;
      sig = x >= 0.0 ? ampl : 0.0;
    }
  } else {  // Perlin family; mktime(t) % 86400 == seconds of day for naive/UTC time (env 1297)
    const double x = sod / a.perlin_period;
    double n = 0.0;
    for (int q = 0; q < a.perlin_octaves; ++q) {
      const double f = ldexp(a.perlin_step, q);
      const double w = (q < a.perlin_octaves - 1) ? ldexp(1.0, -q) : 1.0 / (ldexp(1.0, a.perlin_octaves) - 1.0);
      n += w * lattice_noise_1d(x * f, eg, a.episode, a.k0, a.k1);
    }
    sig = fmax(0.0, base + base * a.perlin_amp * n);
    abs_noise = fabs(base * a.perlin_amp * n);   // PowerGrid.cumulated_abs_noise += |signal * amplitude * perlin| (env 1301)
  }
  if (a.tab_abs_noise != nullptr) a.tab_abs_noise[i] = abs_noise;
  sig *= a.ratio[e];                        // env 1312
  a.tab_signal[i] = fmin(sig, a.max_power[e]);  // env 1314
}

// The same tables with one thread per (env, run of `chunk` consecutive rows): what depends on the MINUTE only - the calendar,
// the outdoor sinusoid (minute stairs, env 1066-1068), the solar polynomial - is evaluated when the minute changes, and a Perlin
// octave's two lattice gradients (two Philox calls) when the octave's lattice cell changes (0.05 .. 0.8 cells per 4-s step for the
// default five octaves) instead of at every row.  For batches of small envs the tables are O(E) fp64 + Philox work per step
// (209,715 envs x 65 rows per refill at 20 houses per env): per row this leaves the OD Gaussian, the fade polynomials and ~1.5
// Philox calls of the ten.  Every value is produced by the very expressions of k_fill_tables from the very inputs: bit-identical
// tables (tests/test_gpu_tables.py).  Thread id = run * E + env: a wave writes consecutive envs of one row.
constexpr int TABLE_RUN_OCTAVES = 8;

#ifndef MDR_TABLE_RUN_WAVES
#define MDR_TABLE_RUN_WAVES 3
#endif
__global__ __launch_bounds__(256, MDR_TABLE_RUN_WAVES) void k_fill_tables_runs(TableArgs a, int chunk) {
  // a thread's lattice cells and gradients per octave: in LDS, so that the octave loop stays a loop (unrolled over registers it
  // cost 206 of them and two waves per SIMD)
  __shared__ uint32_t s_cell[TABLE_RUN_OCTAVES][256];
  __shared__ double2 s_grad[TABLE_RUN_OCTAVES][256];
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int runs = (a.rows + chunk - 1) / chunk;
  if (id >= (int64_t)runs * a.E) return;
  const int run = (int)(id / a.E);
  const int e = (int)(id % a.E);
  const int r0 = run * chunk, r1 = min(a.rows, r0 + chunk);
  const uint32_t eg = (uint32_t)(e + a.env_offset);
  const int64_t t0 = a.t0[e];
  const double phase = a.phase[e], ratio = a.ratio[e], max_power = a.max_power[e];
  const double base = a.base_power ? a.base_power[e] : a.avg_power_per_hvac * (double)a.n_total;
  const double amp = 0.5 * (a.day_temp - a.night_temp), bias = 0.5 * (a.day_temp + a.night_temp);
  int64_t minute = INT64_MIN;   // floor(t / 60) of the cached calendar
  Civil c{};
  int64_t t_civil = 0;          // the time the cached calendar was taken at (its seconds-of-day belong to it)
  double od_base = 0.0;
  float solar = 0.0f;
  uint32_t have = 0u;           // bit q: octave q's cell is cached
  const double w_last = 1.0 / (ldexp(1.0, a.perlin_octaves) - 1.0);
  for (int r = r0; r < r1; ++r) {
    const int64_t i = (int64_t)r * a.E + e;
    const int64_t j = a.j0 + r;
    const int64_t t = t0 + j * (int64_t)a.dt;
    int64_t m = t / 60;
    if (t - m * 60 < 0) m -= 1;
    if (m != minute) {
      minute = m;
      c = civil_from_epoch(t);
      t_civil = t;
      const double tday = (double)c.hour + (double)c.minute / 60.0;
      od_base = amp * sin(6.283185307179586476925286766559 * (tday + (-6.0 + phase)) / 24.0) + bias;
      solar = a.solar_on ? (float)(a.area_shading * solar_cooling_load(c.hour, c.minute, c.month, c.day)) : 0.0f;
    }
    double od;
    if (a.od_ext != nullptr && j < a.od_ext_rows) {
      od = a.od_ext[j * a.E + e];
    } else {
      od = od_base;
      if (a.temp_std != 0.0) {
        const u32x4 g = philox4x32_10(eg, (uint32_t)j, a.episode, TAG_OD_NOISE, a.k0, a.k1);
        od += a.temp_std * gauss01(g.x, g.y);
      }
    }
    a.tab_od[i] = (float)(od - a.temp_ref);
    a.tab_solar[i] = solar;
    const double sod = (double)(c.sod + (int)(t - t_civil));   // same minute, hence same day: the calendar's seconds-of-day moved on
    double sig, abs_noise = 0.0;
    if (a.signal_mode == MDR_SIGNAL_FLAT) {
      sig = base;
    } else if (a.signal_mode == MDR_SIGNAL_SINUSOIDALS) {
      sig = base;
      for (int q = 0; q < a.nb_sin; ++q)
        sig += base * a.sin_ratios[q] * sin(6.283185307179586476925286766559 * sod / a.sin_periods[q]);
    } else if (a.signal_mode == MDR_SIGNAL_REGULAR_STEPS) {
      const double ampl = a.steps_amp * (double)a.n_total;
      const double duty = base / ampl;
      {
#pragma clang fp contract(off)
        const double edge = (1.0 - duty) * a.steps_period;
        const double x = fmod(sod, a.steps_period) - edge;
        sig = x >= 0.0 ? ampl : 0.0;
      }
    } else {
      const double x = sod / a.perlin_period;
      double n = 0.0;
      for (int q = 0; q < a.perlin_octaves; ++q) {
        const double f = ldexp(a.perlin_step, q);
        const double w = (q < a.perlin_octaves - 1) ? ldexp(1.0, -q) : w_last;
        // lattice_noise_1d(x * f, ...) with the cell's two gradients kept while the cell stays
        const double xx = x * f;
        const double l0 = floor(xx);
        const double d0 = xx - l0;
        const double d1 = d0 - 1.0;
        const uint32_t li = (uint32_t)(int64_t)l0;
        const bool cached = (have >> q) & 1u;
        const uint32_t was = s_cell[q][threadIdx.x];
        double2 g = s_grad[q][threadIdx.x];
        if (!cached || li != was) {
          g.x = (cached && li == was + 1u) ? g.y : 2.0 * u01(philox4x32_10(eg, li, a.episode, TAG_PERLIN, a.k0, a.k1).x) - 1.0;
          g.y = 2.0 * u01(philox4x32_10(eg, li + 1u, a.episode, TAG_PERLIN, a.k0, a.k1).x) - 1.0;
          s_cell[q][threadIdx.x] = li;
          s_grad[q][threadIdx.x] = g;
          have |= 1u << q;
        }
        n += w * (fade5(1.0 - d0) * g.x * d0 + fade5(1.0 + d1) * g.y * d1);
      }
      sig = fmax(0.0, base + base * a.perlin_amp * n);
      abs_noise = fabs(base * a.perlin_amp * n);
    }
    if (a.tab_abs_noise != nullptr) a.tab_abs_noise[i] = abs_noise;
    sig *= ratio;
    a.tab_signal[i] = fmin(sig, max_power);
  }
}

// The tables of a TILE of 64 envs by one workgroup, what rows share computed exactly once.  In the runs kernel above a wave holds
// 64 envs whose start times differ: whenever ANY lane enters a new minute (probability 0.99 per row at dt = 4 s) or a new lattice
// cell (~1 per octave and row) the whole wave walks the calendar + solar polynomial / the Philox rounds - the caches save little.
// Here the shared work is split evenly over the lanes first:
//   phase 0  per (env, minute of the window): calendar, outdoor sinusoid, solar polynomial, seconds-of-day   -> LDS
//   phase 1  per (env, octave): the lattice cell of row 0                                          (Perlin)   -> LDS
//   phase 2  per (env, gradient slot): one Philox call - the window needs ceil(span_q) + 2 gradients of octave q
//            (6 + 9 + 15 + 28 + 54 for the default five octaves and 65 rows, against 650 calls by entry)         -> LDS
//   phase 3  per (env, row): OD Gaussian, the signal, the stores; lane = env (coalesced rows), wave = a quarter of the rows
// A row whose cell lies outside the slots (a window across midnight: seconds-of-day start again) draws its gradients directly.
// Same expressions on the same inputs as k_fill_tables: bit-identical tables (tests/test_gpu_tables.py).
constexpr int TABLE_TILE_ENVS = 64;
constexpr int TABLE_TILE_MAX_MINUTES = 8;
constexpr int TABLE_TILE_MAX_SLOTS = 320;

struct TableTile {
  int base[TABLE_RUN_OCTAVES + 1];   // gradient slots of octave q: [base[q], base[q + 1])
  int minutes;                       // minute slots per env
};

#ifndef MDR_TABLE_TILE_WAVES
#define MDR_TABLE_TILE_WAVES 4
#endif
__global__ __launch_bounds__(256, MDR_TABLE_TILE_WAVES) void k_fill_tables_tile(TableArgs a, TableTile tl) {
  extern __shared__ uint32_t s_u[];                               // [slots][64] Philox words of the gradients
  __shared__ uint32_t s_first[TABLE_RUN_OCTAVES][TABLE_TILE_ENVS];
  __shared__ double s_odb[TABLE_TILE_MAX_MINUTES][TABLE_TILE_ENVS];
  __shared__ float s_solar[TABLE_TILE_MAX_MINUTES][TABLE_TILE_ENVS];
  __shared__ int s_sod[TABLE_TILE_MAX_MINUTES][TABLE_TILE_ENVS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int e_raw = blockIdx.x * TABLE_TILE_ENVS + lane;
  const bool live = e_raw < a.E;
  const int e = live ? e_raw : a.E - 1;
  const uint32_t eg = (uint32_t)(e + a.env_offset);
  const int64_t t0e = a.t0[e];
  const int64_t t_first = t0e + a.j0 * (int64_t)a.dt;
  int64_t m_first = t_first / 60;
  if (t_first - m_first * 60 < 0) m_first -= 1;
  const int sec_first = (int)(t_first - m_first * 60);     // seconds into the first minute
  const double amp = 0.5 * (a.day_temp - a.night_temp), bias = 0.5 * (a.day_temp + a.night_temp);
  const bool perlin = a.signal_mode == MDR_SIGNAL_PERLIN;

  {   // phase 0
    const double phase = a.phase[e];
    for (int m = w; m < tl.minutes; m += 4) {
      const Civil c = civil_from_epoch((m_first + m) * 60);
      const double tday = (double)c.hour + (double)c.minute / 60.0;
      s_odb[m][lane] = amp * sin(6.283185307179586476925286766559 * (tday + (-6.0 + phase)) / 24.0) + bias;
      s_solar[m][lane] = a.solar_on ? (float)(a.area_shading * solar_cooling_load(c.hour, c.minute, c.month, c.day)) : 0.0f;
      s_sod[m][lane] = c.sod;
    }
  }
  __syncthreads();
  if (perlin) {
    for (int q = w; q < a.perlin_octaves; q += 4) {   // phase 1
      const double sod = (double)(s_sod[0][lane] + sec_first);
      const double xx = (sod / a.perlin_period) * ldexp(a.perlin_step, q);
      s_first[q][lane] = (uint32_t)(int64_t)floor(xx);
    }
    __syncthreads();
    const int slots = tl.base[a.perlin_octaves];      // phase 2
    int q = 0;
    for (int sl = w; sl < slots; sl += 4) {
      while (sl >= tl.base[q + 1]) ++q;
      const uint32_t li = s_first[q][lane] + (uint32_t)(sl - tl.base[q]);
      s_u[sl * TABLE_TILE_ENVS + lane] = philox4x32_10(eg, li, a.episode, TAG_PERLIN, a.k0, a.k1).x;
    }
    __syncthreads();
  }
  if (!live) return;

  // phase 3
  const int per_wave = (a.rows + 3) / 4;
  const int r0 = w * per_wave, r1 = min(a.rows, r0 + per_wave);
  const double ratio = a.ratio[e], max_power = a.max_power[e];
  const double base = a.base_power ? a.base_power[e] : a.avg_power_per_hvac * (double)a.n_total;
  const double w_last = 1.0 / (ldexp(1.0, a.perlin_octaves) - 1.0);
  // minute slot and seconds into it, carried from row to row (the same integers as floor(t / 60) and t - 60 floor(t / 60))
  int64_t since = (int64_t)sec_first + (int64_t)r0 * a.dt;
  int ms = (int)(since / 60);
  int sec = (int)(since - (int64_t)ms * 60);
  for (int r = r0; r < r1; ++r) {
    const int64_t i = (int64_t)r * a.E + e;
    const int64_t j = a.j0 + r;
    double od;
    if (a.od_ext != nullptr && j < a.od_ext_rows) {
      od = a.od_ext[j * a.E + e];
    } else {
      od = s_odb[ms][lane];
      if (a.temp_std != 0.0) {
        const u32x4 g = philox4x32_10(eg, (uint32_t)j, a.episode, TAG_OD_NOISE, a.k0, a.k1);
        od += a.temp_std * gauss01(g.x, g.y);
      }
    }
    a.tab_od[i] = (float)(od - a.temp_ref);
    a.tab_solar[i] = s_solar[ms][lane];
    const double sod = (double)(s_sod[ms][lane] + sec);
    double sig, abs_noise = 0.0;
    if (a.signal_mode == MDR_SIGNAL_FLAT) {
      sig = base;
    } else if (a.signal_mode == MDR_SIGNAL_SINUSOIDALS) {
      sig = base;
      for (int q = 0; q < a.nb_sin; ++q)
        sig += base * a.sin_ratios[q] * sin(6.283185307179586476925286766559 * sod / a.sin_periods[q]);
    } else if (a.signal_mode == MDR_SIGNAL_REGULAR_STEPS) {
      const double ampl = a.steps_amp * (double)a.n_total;
      const double duty = base / ampl;
      {
#pragma clang fp contract(off)
        const double edge = (1.0 - duty) * a.steps_period;
        const double x = fmod(sod, a.steps_period) - edge;
        sig = x >= 0.0 ? ampl : 0.0;
      }
    } else {
      const double x = sod / a.perlin_period;
      double n = 0.0;
      for (int q = 0; q < a.perlin_octaves; ++q) {
        const double f = ldexp(a.perlin_step, q);
        const double wq = (q < a.perlin_octaves - 1) ? ldexp(1.0, -q) : w_last;
        const double xx = x * f;
        const double l0 = floor(xx);
        const double d0 = xx - l0;
        const double d1 = d0 - 1.0;
        const uint32_t li = (uint32_t)(int64_t)l0;
        const uint32_t idx = li - s_first[q][lane];
        uint32_t u0, u1;
        if (idx + 1u < (uint32_t)(tl.base[q + 1] - tl.base[q])) {
          const int sl = tl.base[q] + (int)idx;
          u0 = s_u[sl * TABLE_TILE_ENVS + lane];
          u1 = s_u[(sl + 1) * TABLE_TILE_ENVS + lane];
        } else {
          u0 = philox4x32_10(eg, li, a.episode, TAG_PERLIN, a.k0, a.k1).x;
          u1 = philox4x32_10(eg, li + 1u, a.episode, TAG_PERLIN, a.k0, a.k1).x;
        }
        const double g0 = 2.0 * u01(u0) - 1.0, g1 = 2.0 * u01(u1) - 1.0;
        n += wq * (fade5(1.0 - d0) * g0 * d0 + fade5(1.0 + d1) * g1 * d1);
      }
      sig = fmax(0.0, base + base * a.perlin_amp * n);
      abs_noise = fabs(base * a.perlin_amp * n);
    }
    if (a.tab_abs_noise != nullptr) a.tab_abs_noise[i] = abs_noise;
    sig *= ratio;
    a.tab_signal[i] = fmin(sig, max_power);
    sec += a.dt;
    while (sec >= 60) {
      sec -= 60;
      ms += 1;
    }
  }
}

// =================================================================================================
// Step kernels
// =================================================================================================
// Graph mode: a captured launch carries the pointers of table row 0; the device-resident cursor says where the episode is.
__device__ __forceinline__ void rebase(ObsArgs& a) {
  if (a.cursor == nullptr) return;
  const int64_t off = (int64_t)min(a.cursor[0], a.cursor_max + 1) * a.E;
  a.sig_now += off;
  a.od_now += off;
  a.solar_now += off;
  a.k = a.cursor[1];
}

// The split kernels need no counting: a partial kernel notes its table row in a slot (cursor[2] | cursor[4]) that nobody in its
// launch reads, the finish takes the row from there - so nobody in the finish launch reads cursor[0] / [1] and one of its
// threads moves them on.  k_step_finish_partial (finish of step k + partial of step k + 1) reads one slot and writes the other.
__device__ __forceinline__ int snap_index(int slot) { return slot ? 4 : 2; }

__device__ __forceinline__ void rebase_finish(StepArgs& a) {
  if (a.cursor == nullptr) return;
  const int64_t off = (int64_t)min(a.cursor_adv != nullptr ? a.cursor[snap_index(a.snap_rd)] : a.cursor[0], a.cursor_max) * a.E;
  a.sig_old += off;
  a.sig_new += off;
}

__global__ void k_cursor_advance(int32_t* cursor) {
  cursor[0] += 1;
  cursor[1] += 1;
}

__global__ void k_cursor_set(int32_t* cursor, int32_t row, int32_t k) {
  cursor[0] = row;
  cursor[1] = k;
  cursor[2] = row;   // the split kernels' row notes
  cursor[3] = 0;     // arrival counter of cursor_done
  cursor[4] = row;
}

// Everything the dict adapter shows of env e after a step, as one fp64 vector (ONE launch + ONE device->host copy):
// Ta[N] | Tm[N] (deg C) | sso[N] | flags[N] | reward[N] | OD temp, reg signal, solar gain, cluster power, max power, ratio, |noise|
__global__ __launch_bounds__(256) void k_pack_env(StepArgs a, int e, double temp_ref, const double* max_power, const double* ratio,
                                                  const double* abs_noise_row, double* out) {
  const int64_t base = (int64_t)e * a.N;
  for (int h = threadIdx.x; h < a.N; h += 256) {
    out[h] = (double)a.Ta[base + h] + temp_ref;
    out[a.N + h] = (double)a.Tm[base + h] + temp_ref;
    out[2 * a.N + h] = (double)a.sso[base + h];
    out[3 * a.N + h] = (double)a.flags[base + h];
    out[4 * a.N + h] = (double)a.reward[base + h];
  }
  if (threadIdx.x == 0) {
    double* s = out + 5 * (int64_t)a.N;
    s[0] = (double)a.od_old[e] + temp_ref;    // od_old / solar_new / sig_old: the rows of the CURRENT time index here
    s[1] = a.sig_old[e];
    s[2] = (double)a.solar_new[e];
    s[3] = a.P[e];
    s[4] = max_power[e];
    s[5] = ratio[e];
    s[6] = abs_noise_row ? abs_noise_row[e] : 0.0;
  }
}

hipError_t launch_pack_env(const StepArgs& a, int e, double temp_ref, const double* max_power, const double* ratio,
                           const double* abs_noise_row, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_env, dim3(1), dim3(256), 0, s, a, e, temp_ref, max_power, ratio, abs_noise_row, out);
  return hipGetLastError();
}

hipError_t launch_cursor_set(int32_t* cursor, int32_t row, int32_t k, hipStream_t s) {
  hipLaunchKernelGGL(k_cursor_set, dim3(1), dim3(1), 0, s, cursor, row, k);
  return hipGetLastError();
}

// temp_penalty / reward_value / signal_term, load_vec / store_vec / store_out, step_vec, store_obs_local, store_reward_power:
// mdr_step_common.h (shared with mdr_persist.hip)

template <int VEC, int TILES, int THREADS>
__global__ __launch_bounds__(THREADS) void k_step_fused(StepArgs a) {
  rebase(a);
  __shared__ double lds[3 * (THREADS / 64)];
  const int e = blockIdx.x;
  const float od_old = a.od_old[e];
  const float solar = a.solar_new[e];
  const int64_t base = (int64_t)e * a.N;
  HouseOut o[TILES][VEC];
  int lockout[TILES][VEC];
  Red3 acc{0.0, 0.0, 0.0f};
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const int h = (t * THREADS + (int)threadIdx.x) * VEC;
    if (h < a.N) {
      step_vec<VEC>(a, base + h, od_old, solar, o[t], lockout[t]);
      float p = 0.0f, ps = 0.0f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        p += o[t][v].power;
        ps += o[t][v].pen;
        acc.max_pen = fmaxf(acc.max_pen, o[t][v].pen);
      }
      acc.sum_p += (double)p;
      acc.sum_pen += (double)ps;
      store_obs_local<VEC>(a, base + h, o[t], lockout[t]);
    }
  }
  const Red3 tot = block_reduce<THREADS>(acc, lds);
  const float sig_term = signal_term(a, tot.sum_p, a.sig_old[e]);
  const float o_sig = (float)(a.sig_new[e] * a.inv_obs_norm);
  const float o_pow = (float)(tot.sum_p * a.inv_obs_norm);
  if (threadIdx.x == 0) a.P[e] = tot.sum_p;
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const int h = (t * THREADS + (int)threadIdx.x) * VEC;
    if (h < a.N) {
      float pen[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) pen[v] = o[t][v].pen;
      store_reward_power<VEC>(a, base + h, pen, tot.sum_pen, tot.max_pen, sig_term, o_sig, o_pow);
    }
  }
  cursor_done(a);
}

// Per-env table rows for the multi-step kernels: a window of W consecutive steps lives in W lanes (lane l holds the
// row of step s0 + l) and is read back with a shuffle; the next window is loaded one window ahead, so the dependent
// global-load latency is paid once per W steps and overlaps W steps of arithmetic.
struct EnvRow {
  float od, solar;
  double sig_old, sig_new;
};

__device__ __forceinline__ EnvRow window_load(const StepArgs& a, int e, int s_first, int nsteps, int lane_in_window) {
  const int sr = min(s_first + lane_in_window, nsteps - 1);
  const int64_t row = (int64_t)sr * a.E + e;
  return EnvRow{a.od_old[row], a.solar_new[row], a.sig_old[row], a.sig_new[row]};
}

__device__ __forceinline__ EnvRow window_get(const EnvRow& w, int src_lane) {
  return EnvRow{__shfl(w.od, src_lane, 64), __shfl(w.solar, src_lane, 64), __shfl(w.sig_old, src_lane, 64),
                __shfl(w.sig_new, src_lane, 64)};
}

// ---- (1b) multi-step closed loop: the same workgroup-per-env mapping, but the houses stay in registers for
// `nsteps` steps (bang-bang rule in-kernel).  Per step only the env's table scalars are read; the state is read
// once and written once per launch.  One barrier per step (LDS partials are double-buffered by step parity).
// (BB: the bang-bang rule compiled in - the default controller; the other rules go through controller_cmds)
template <int VEC, int TILES, int THREADS, bool WINDOW, bool BB>
__global__ __launch_bounds__(THREADS) void k_rollout_fused(StepArgs a, RolloutArgs ro) {
  __shared__ double lds[2][3 * (THREADS / 64)];
  const bool need_pen = a.penalty_mode != MDR_PENALTY_INDIVIDUAL_L2;   // common penalty modes need the env's penalty sum / max
  const bool want_rsum = ro.reward_sum != nullptr, want_terr = ro.sq_temp_error_sum != nullptr;   // accumulators nobody asked for are not computed
  const int e = blockIdx.x;
  const int64_t base = (int64_t)e * a.N;
  HouseIn hs[TILES][VEC];
  float rsum[TILES][VEC], pen[TILES][VEC];
  // The HVAC's on / lockout bits and the latest command as 64-bit LANE MASKS (house_advance_m); bytes are formed once, at the end.
  uint64_t on_m[TILES][VEC], lock_m[TILES][VEC], cmd_m[TILES][VEC];
  bool live[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const int h = (t * THREADS + (int)threadIdx.x) * VEC;
    live[t] = h < a.N;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      hs[t][v] = HouseIn{};
      rsum[t][v] = 0.0f;
      pen[t][v] = 0.0f;
    }
    if (live[t]) {
      const int64_t i = base + h;
      float Ta[VEC], Tm[VEC], k01[VEC], s0[VEC], k10[VEC], s1[VEC], iu[VEC], q[VEC], pm[VEC], tg[VEC], db[VEC];
      int sso[VEC], lk[VEC];
      unsigned fl[VEC];
      load_vec<VEC>(a.Ta, i, Ta);
      load_vec<VEC>(a.Tm, i, Tm);
      load_vec<VEC>(a.sso, i, sso);
      load_bytes<VEC>(a.flags, i, fl);
      load_vec<VEC>(a.k01, i, k01);
      load_vec<VEC>(a.s0, i, s0);
      load_vec<VEC>(a.k10, i, k10);
      load_vec<VEC>(a.s1, i, s1);
      load_vec<VEC>(a.inv_Ua, i, iu);
      load_vec<VEC>(a.Q_hvac, i, q);
      load_vec<VEC>(a.P_max, i, pm);
      load_vec<VEC>(a.target, i, tg);
      load_vec<VEC>(a.deadband, i, db);
      load_vec<VEC>(a.lockout, i, lk);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        hs[t][v] = HouseIn{Ta[v], Tm[v], sso[v], fl[v], k01[v], s0[v], k10[v], s1[v], iu[v], q[v], pm[v], tg[v], db[v], lk[v]};
      }
      if (ro.reward_sum) load_vec<VEC>(ro.reward_sum, i, rsum[t]);   // continue the caller's running sum in step order
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {   // (outside the branch: a ballot is a wave-wide value)
      on_m[t][v] = __builtin_amdgcn_ballot_w64(live[t] && (hs[t][v].flags & 1u) != 0u);
      lock_m[t][v] = __builtin_amdgcn_ballot_w64(live[t] && (hs[t][v].flags & 2u) != 0u);
      cmd_m[t][v] = 0;
    }
  }
  double terr = 0.0, serr = 0.0;
  Red3 tot{0.0, 0.0, 0.0f};
  float sig_term = 0.0f;
  // WINDOW: few workgroups in flight (latency-bound) -> shuffle-window prefetch; otherwise the rows are plain
  // wave-uniform (scalar) loads whose latency the other resident waves hide, at no VALU cost.
  const int wl = threadIdx.x & 63;
  EnvRow cur{}, nxt{}, er{};
  if (WINDOW) {
    cur = window_load(a, e, 0, ro.nsteps, wl);
    nxt = window_load(a, e, 64, ro.nsteps, wl);
  }
  for (int s = 0; s < ro.nsteps; ++s) {
    const int64_t row = (int64_t)s * a.E + e;
    if (WINDOW) {
      if ((s & 63) == 0 && s > 0) {
        cur = nxt;
        nxt = window_load(a, e, s + 64, ro.nsteps, wl);
      }
      er = window_get(cur, s & 63);
    } else {
      er = EnvRow{a.od_old[row], a.solar_new[row], a.sig_old[row], a.sig_new[row]};
    }
    const float od_old = er.od;
    const float solar = er.solar;
    Red3 acc{0.0, 0.0, 0.0f};
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      {   // every lane, also those past the env's end: their blank houses (all parameters 0) stay at 0 degrees, draw no power and earn no
          // penalty - exact zeros in every sum - and with no divergent branch around them the ballots below see the whole wave
        float p = 0.0f, ps = 0.0f, te = 0.0f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          cmd_m[t][v] = BB ? __builtin_amdgcn_ballot_w64(hs[t][v].Ta > hs[t][v].target)   // agents/bangbang_controllers.py:49-59
                           : controller_cmd_m(a.action_source, hs[t][v].Ta, hs[t][v].target, hs[t][v].deadband, on_m[t][v]);
          const HouseNextM n = house_advance_m(hs[t][v], on_m[t][v], cmd_m[t][v], od_old, solar, a.dt);
          hs[t][v].Ta = n.Ta;
          hs[t][v].Tm = n.Tm;
          hs[t][v].sso = live[t] ? n.sso : 0;
          on_m[t][v] = n.on;
          lock_m[t][v] = n.lock;
          pen[t][v] = n.pen;
          p += n.power;
          if (want_terr) {
            const float d = n.Ta - hs[t][v].target;
            te = fmaf(d, d, te);
          }
        }
        if (need_pen) {   // individual_L2 reduces the cluster power alone
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            ps += pen[t][v];
            acc.max_pen = fmaxf(acc.max_pen, pen[t][v]);
          }
        }
        acc.sum_p += (double)p;
        acc.sum_pen += (double)ps;
        terr += (double)te;
      }
    }
    tot = block_reduce<THREADS>(acc, lds[s & 1], need_pen);
    sig_term = signal_term(a, tot.sum_p, er.sig_old);
    if (want_rsum) {   // (the last step's reward is written by the epilogue either way)
#pragma unroll
      for (int t = 0; t < TILES; ++t)
        if (live[t]) {
          add_rewards<VEC>(a, need_pen, pen[t], tot.sum_pen, tot.max_pen, sig_term, rsum[t]);
        }
    }
    if (threadIdx.x == 0) {
      if (ro.power_trace) ro.power_trace[row] = tot.sum_p;
      const double d = er.sig_new - tot.sum_p;
      if (!(ro.defer_last_signal_error && s == ro.nsteps - 1)) serr += d * d;
    }
  }
  if (ro.nsteps <= 0) return;
  // final state, and the last step's outputs exactly as k_step_fused leaves them
  const float o_sig = (float)(er.sig_new * a.inv_obs_norm);
  const float o_pow = (float)(tot.sum_p * a.inv_obs_norm);
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    if (!live[t]) continue;
    const int64_t i = base + (t * THREADS + (int)threadIdx.x) * VEC;
    float nTa[VEC], nTm[VEC];
    int nsso[VEC], lk[VEC];
    unsigned nfl[VEC], act[VEC];
    HouseOut o[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      nTa[v] = hs[t][v].Ta;
      nTm[v] = hs[t][v].Tm;
      nsso[v] = hs[t][v].sso;
      nfl[v] = house_flags(__builtin_amdgcn_inverse_ballot_w64(on_m[t][v]), __builtin_amdgcn_inverse_ballot_w64(lock_m[t][v]));
      lk[v] = hs[t][v].lockout;
      act[v] = __builtin_amdgcn_inverse_ballot_w64(cmd_m[t][v]) ? 1u : 0u;
      o[v] = HouseOut{nTa[v], nTm[v], nsso[v], nfl[v], pen[t][v], 0.0f};
    }
    store_vec<VEC>(a.Ta, i, nTa);
    store_vec<VEC>(a.Tm, i, nTm);
    store_vec<VEC>(a.sso, i, nsso);
    store_bytes<VEC>(a.flags, i, nfl);
    if (a.actions != nullptr) store_bytes<VEC>(a.actions, i, act);
    store_obs_local<VEC>(a, i, o, lk);
    store_reward_power<VEC>(a, i, pen[t], tot.sum_pen, tot.max_pen, sig_term, o_sig, o_pow);
    if (ro.reward_sum) store_vec<VEC>(ro.reward_sum, i, rsum[t]);
  }
  if (ro.sq_temp_error_sum) {   // one more reduction; the parity of the LDS buffer continues the step sequence
    Red3 r{terr, 0.0, 0.0f};
    r = block_reduce<THREADS>(r, lds[ro.nsteps & 1]);
    if (threadIdx.x == 0) ro.sq_temp_error_sum[e] += r.sum_p;
  }
  if (threadIdx.x == 0) {
    a.P[e] = tot.sum_p;
    if (ro.sq_signal_error_sum) ro.sq_signal_error_sum[e] += serr;
  }
}

// ---- (2) small envs: GROUP lanes per env, VEC consecutive houses per lane (N <= GROUP * VEC, N % VEC == 0), several
// envs per wavefront; reductions by sub-wave DPP exchanges only (no LDS, no barrier).  VEC = 4 / 2 keeps the accesses
// 16 / 8 bytes wide for the agent counts the reference actually trains with (cli.py: 20 and 50 houses).
template <int GROUP, int VEC>
__global__ __launch_bounds__(256) void k_step_group(StepArgs a) {
  rebase(a);
  const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / GROUP;
  const int lane = threadIdx.x % GROUP;
  const bool env_ok = gid < a.E;
  const int e = env_ok ? (int)gid : a.E - 1;
  const bool active = env_ok && lane * VEC < a.N;
  const int64_t i = (int64_t)e * a.N + lane * VEC;
  HouseOut o[VEC];
  int lockout[VEC];
  float pen[VEC];
  Red3 acc{0.0, 0.0, 0.0f};
  // (fetching the env's four table values by the group's first lane only and handing them round by shuffles was measured: no gain)
  const float od_e = a.od_old[e], so_e = a.solar_new[e];
  const double sg_old = a.sig_old[e], sg_new = a.sig_new[e];
  if (active) {
    step_vec<VEC>(a, i, od_e, so_e, o, lockout);
    float p = 0.0f, ps = 0.0f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      p += o[v].power;
      ps += o[v].pen;
      pen[v] = o[v].pen;
      acc.max_pen = fmaxf(acc.max_pen, o[v].pen);
    }
    acc.sum_p = (double)p;
    acc.sum_pen = (double)ps;
    store_obs_local<VEC>(a, i, o, lockout);
  }
  const Red3 tot = lanes_reduce<GROUP>(acc);
  if (active) {
    const float sig_term = signal_term(a, tot.sum_p, sg_old);
    if (lane == 0) a.P[e] = tot.sum_p;
    store_reward_power<VEC>(a, i, pen, tot.sum_pen, tot.max_pen, sig_term, (float)(sg_new * a.inv_obs_norm),
                            (float)(tot.sum_p * a.inv_obs_norm));
  }
  cursor_done(a);
}

// ---- (2a) single-house envs (config.py's literal default nb_agents = 1; the Monte-Carlo grid): the "env" axis is the
// vector axis - 4 consecutive envs per thread, every per-house array AND the per-env table rows read 16 bytes wide.
// No reduction: the cluster is the house.
__global__ __launch_bounds__(256) void k_step_single_house(StepArgs a) {
  rebase(a);
  const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e0 >= a.E) return;   // E % 4 == 0
  float Ta[4], Tm[4], k01[4], s0[4], k10[4], s1[4], iu[4], q[4], pm[4], tg[4], db[4], od[4], solar[4];
  int sso[4], lk[4];
  unsigned fl[4], act[4];
  load_vec<4>(a.Ta, e0, Ta);
  load_vec<4>(a.Tm, e0, Tm);
  load_vec<4>(a.sso, e0, sso);
  load_bytes<4>(a.flags, e0, fl);
  if (a.action_source == MDR_ACTIONS_EXTERNAL) load_bytes<4>(a.actions, e0, act);
  load_param<4>(a.k01, e0, k01);
  load_param<4>(a.s0, e0, s0);
  load_param<4>(a.k10, e0, k10);
  load_param<4>(a.s1, e0, s1);
  load_param<4>(a.inv_Ua, e0, iu);
  load_param<4>(a.Q_hvac, e0, q);
  load_param<4>(a.P_max, e0, pm);
  load_param<4>(a.target, e0, tg);
  load_param<4>(a.deadband, e0, db);
  load_vec<4>(a.lockout, e0, lk);
  load_vec<4>(a.od_old, e0, od);
  load_vec<4>(a.solar_new, e0, solar);
  HouseOut o[4];
  float nTa[4], nTm[4], rew[4], c5[4], c6[4];
  int nsso[4];
  unsigned nfl[4];
  bool cmds[4];   // three wave-uniform arms (see step_vec_rows)
  if (a.action_source == MDR_ACTIONS_EXTERNAL) {
#pragma unroll
    for (int v = 0; v < 4; ++v) cmds[v] = act[v] != 0u;
  } else if (a.action_source == MDR_ACTIONS_BANGBANG) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      cmds[v] = Ta[v] > tg[v];
      act[v] = cmds[v] ? 1u : 0u;
    }
  } else {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      cmds[v] = controller_cmd(a.action_source, Ta[v], tg[v], db[v], fl[v]);
      act[v] = cmds[v] ? 1u : 0u;
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    HouseIn h{Ta[v], Tm[v], sso[v], fl[v], k01[v], s0[v], k10[v], s1[v], iu[v], q[v], pm[v], tg[v], db[v], lk[v]};
    const bool cmd = cmds[v];
    o[v] = house_step(h, cmd, od[v], solar[v], a.dt);
    nTa[v] = o[v].Ta;
    nTm[v] = o[v].Tm;
    nsso[v] = o[v].sso;
    nfl[v] = o[v].flags;
    const double P = (double)o[v].power;
    const float sig_term = signal_term(a, P, a.sig_old[e0 + v]);
    rew[v] = reward_value(a, o[v].pen, (double)o[v].pen, o[v].pen, sig_term);   // one house: common == individual
    c5[v] = (float)(a.sig_new[e0 + v] * a.inv_obs_norm);
    c6[v] = (float)(P * a.inv_obs_norm);
    a.P[e0 + v] = P;
  }
  store_vec<4>(a.Ta, e0, nTa);
  store_vec<4>(a.Tm, e0, nTm);
  store_vec<4>(a.sso, e0, nsso);
  store_bytes<4>(a.flags, e0, nfl);
  if (a.action_source != MDR_ACTIONS_EXTERNAL && a.actions != nullptr) store_bytes<4>(a.actions, e0, act);
  store_obs_local<4>(a, e0, o, lk);
  store_out<4>(a.reward, e0, rew);
  if (a.obs != nullptr) {
    store_out<4>(a.obs + 5 * a.plane, e0, c5);
    store_out<4>(a.obs + 6 * a.plane, e0, c6);
  }
  cursor_done(a);
}

// ---- (2b) multi-step closed loop for small envs: GROUP lanes per env, VEC houses per lane (the same mapping and the
// same reduction tree as k_step_group, so both end bit for bit in the same state), no LDS, no barrier
template <int GROUP, int VEC, bool WINDOW, bool BB>
__global__ __launch_bounds__(256) void k_rollout_group(StepArgs a, RolloutArgs ro) {
  const bool need_pen = a.penalty_mode != MDR_PENALTY_INDIVIDUAL_L2;
  const bool want_rsum = ro.reward_sum != nullptr;
  const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / GROUP;
  const int lane = threadIdx.x % GROUP;
  const bool env_ok = gid < a.E;
  const int e = env_ok ? (int)gid : a.E - 1;
  const bool active = env_ok && lane * VEC < a.N;
  const int64_t i = (int64_t)e * a.N + lane * VEC;
  HouseIn hs[VEC];
  int lockout[VEC];
  float rsum[VEC], pen[VEC];
  uint64_t on_m[VEC], lock_m[VEC], cmd_m[VEC];   // the HVAC bits and the latest command as lane masks (house_advance_m)
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    hs[v] = HouseIn{};
    lockout[v] = 1;
    rsum[v] = 0.0f;
    pen[v] = 0.0f;
  }
  if (active) {
    float Ta[VEC], Tm[VEC], k01[VEC], s0[VEC], k10[VEC], s1[VEC], iu[VEC], q[VEC], pm[VEC], tg[VEC], db[VEC];
    int sso[VEC];
    unsigned fl[VEC];
    load_vec<VEC>(a.Ta, i, Ta);
    load_vec<VEC>(a.Tm, i, Tm);
    load_vec<VEC>(a.sso, i, sso);
    load_bytes<VEC>(a.flags, i, fl);
    load_vec<VEC>(a.k01, i, k01);
    load_vec<VEC>(a.s0, i, s0);
    load_vec<VEC>(a.k10, i, k10);
    load_vec<VEC>(a.s1, i, s1);
    load_vec<VEC>(a.inv_Ua, i, iu);
    load_vec<VEC>(a.Q_hvac, i, q);
    load_vec<VEC>(a.P_max, i, pm);
    load_vec<VEC>(a.target, i, tg);
    load_vec<VEC>(a.deadband, i, db);
    load_vec<VEC>(a.lockout, i, lockout);
#pragma unroll
    for (int v = 0; v < VEC; ++v)
      hs[v] = HouseIn{Ta[v], Tm[v], sso[v], fl[v], k01[v], s0[v], k10[v], s1[v], iu[v], q[v], pm[v], tg[v], db[v], lockout[v]};
    if (ro.reward_sum) load_vec<VEC>(ro.reward_sum, i, rsum);   // continue the caller's running sum in step order
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    on_m[v] = __builtin_amdgcn_ballot_w64((hs[v].flags & 1u) != 0u);
    lock_m[v] = __builtin_amdgcn_ballot_w64((hs[v].flags & 2u) != 0u);
    cmd_m[v] = 0;
  }
  float sig_term = 0.0f;
  double terr = 0.0, serr = 0.0;
  Red3 tot{0.0, 0.0, 0.0f};
  const int gbase = (threadIdx.x & 63) & ~(GROUP - 1);   // first lane of this env's group inside the wavefront
  EnvRow cur{}, nxt{}, er{};
  if (WINDOW) {
    cur = window_load(a, e, 0, ro.nsteps, lane);
    nxt = window_load(a, e, GROUP, ro.nsteps, lane);
  }
  for (int s = 0; s < ro.nsteps; ++s) {
    const int64_t row = (int64_t)s * a.E + e;
    if (WINDOW) {
      if ((s & (GROUP - 1)) == 0 && s > 0) {
        cur = nxt;
        nxt = window_load(a, e, s + GROUP, ro.nsteps, lane);
      }
      er = window_get(cur, gbase + (s & (GROUP - 1)));
    } else {
      er = EnvRow{a.od_old[row], a.solar_new[row], a.sig_old[row], a.sig_new[row]};
    }
    Red3 acc{0.0, 0.0, 0.0f};
    {   // every lane, idle ones too: their blank houses (all parameters 0) stay at 0 degrees, draw no power and earn no penalty - exact
        // zeros in every sum - and the lane masks are formed in wave-uniform control flow
      float p = 0.0f, ps = 0.0f, te = 0.0f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        cmd_m[v] = BB ? __builtin_amdgcn_ballot_w64(hs[v].Ta > hs[v].target)   // agents/bangbang_controllers.py:49-59
                      : controller_cmd_m(a.action_source, hs[v].Ta, hs[v].target, hs[v].deadband, on_m[v]);
        const HouseNextM n = house_advance_m(hs[v], on_m[v], cmd_m[v], er.od, er.solar, a.dt);
        hs[v].Ta = n.Ta;
        hs[v].Tm = n.Tm;
        hs[v].sso = active ? n.sso : 0;
        on_m[v] = n.on;
        lock_m[v] = n.lock;
        pen[v] = n.pen;
        p += n.power;
        const float d = n.Ta - hs[v].target;
        te = fmaf(d, d, te);
      }
      if (need_pen) {   // individual_L2 reduces the cluster power alone
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          ps += pen[v];
          acc.max_pen = fmaxf(acc.max_pen, pen[v]);
        }
      }
      acc.sum_p = (double)p;
      acc.sum_pen = (double)ps;
      terr += (double)te;
    }
    tot = lanes_reduce<GROUP>(acc, need_pen);
    sig_term = signal_term(a, tot.sum_p, er.sig_old);
    if (active) {
      if (want_rsum) {
        add_rewards<VEC>(a, need_pen, pen, tot.sum_pen, tot.max_pen, sig_term, rsum);
      }
      if (lane == 0) {
        if (ro.power_trace) ro.power_trace[row] = tot.sum_p;
        const double d = er.sig_new - tot.sum_p;
        if (!(ro.defer_last_signal_error && s == ro.nsteps - 1)) serr += d * d;
      }
    }
  }
  if (ro.nsteps <= 0) return;
  Red3 tr{terr, 0.0, 0.0f};
  tr = lanes_reduce<GROUP>(tr);
  if (!active) return;
  float nTa[VEC], nTm[VEC];
  int nsso[VEC];
  unsigned nfl[VEC], act[VEC];
  HouseOut o[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    nTa[v] = hs[v].Ta;
    nTm[v] = hs[v].Tm;
    nsso[v] = hs[v].sso;
    nfl[v] = house_flags(lane_bit(on_m[v]), lane_bit(lock_m[v]));
    act[v] = lane_bit(cmd_m[v]) ? 1u : 0u;
    o[v] = HouseOut{nTa[v], nTm[v], nsso[v], nfl[v], pen[v], 0.0f};
  }
  store_vec<VEC>(a.Ta, i, nTa);
  store_vec<VEC>(a.Tm, i, nTm);
  store_vec<VEC>(a.sso, i, nsso);
  store_bytes<VEC>(a.flags, i, nfl);
  if (a.actions != nullptr) store_bytes<VEC>(a.actions, i, act);
  store_obs_local<VEC>(a, i, o, lockout);
  store_reward_power<VEC>(a, i, pen, tot.sum_pen, tot.max_pen, sig_term, (float)(er.sig_new * a.inv_obs_norm),
                          (float)(tot.sum_p * a.inv_obs_norm));
  if (ro.reward_sum) store_vec<VEC>(ro.reward_sum, i, rsum);
  if (lane == 0) {
    a.P[e] = tot.sum_p;
    if (ro.sq_temp_error_sum) ro.sq_temp_error_sum[e] += tr.sum_p;
    if (ro.sq_signal_error_sum) ro.sq_signal_error_sum[e] += serr;
  }
}

// ---- (3) split path: any N, several workgroups per env, and the sharded-houses case -------------
// grid = (split_blocks, E); workgroup b of env e owns houses [b * THREADS * VEC, (b + 1) * THREADS * VEC).  THREADS = 64 when
// the launch would otherwise have fewer workgroups than CUs to spread over (a 125,000-house shard is 123 workgroups of 1024
// houses: half the CUs idle and each busy one limited by what a single CU can stream).
template <int VEC, int THREADS>
__device__ __forceinline__ void partial_block(const StepArgs& a, double* lds) {
  const int e = blockIdx.y;
  const int h = ((int)blockIdx.x * THREADS + (int)threadIdx.x) * VEC;
  const int64_t base = (int64_t)e * a.N;
  Red3 acc{0.0, 0.0, 0.0f};
  if (h < a.N) {
    HouseOut o[VEC];
    int lockout[VEC];
    step_vec<VEC>(a, base + h, a.od_old[e], a.solar_new[e], o, lockout);
    float p = 0.0f, ps = 0.0f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      p += o[v].power;
      ps += o[v].pen;
      acc.max_pen = fmaxf(acc.max_pen, o[v].pen);
    }
    acc.sum_p = (double)p;
    acc.sum_pen = (double)ps;
    store_obs_local<VEC>(a, base + h, o, lockout);
    // the house's own temperature penalty waits for the finish kernel (which then needs neither the temperature nor the
    // target / deadband again: 4 B written + 4 B read per house instead of 12 B re-read) - in the reward array, or in pen_stash
    float pen[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) pen[v] = o[v].pen;
    store_vec<VEC>(a.stash, base + h, pen);
  }
  const Red3 tot = block_reduce<THREADS>(acc, lds);
  if (threadIdx.x == 0) {
    double* rec = a.partials + ((int64_t)e * a.nblk + blockIdx.x) * 3;
    rec[0] = tot.sum_p;
    rec[1] = tot.sum_pen;
    rec[2] = (double)tot.max_pen;
  }
}

template <int VEC, int THREADS>
__global__ __launch_bounds__(THREADS) void k_step_partial(StepArgs a) {
  if (a.cursor_adv != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.cursor_adv[snap_index(a.snap_wr)] = a.cursor[0];
  rebase(a);
  __shared__ double lds[3 * 4];
  partial_block<VEC, THREADS>(a, lds);
}

// one workgroup per env: fixed-order sum of the per-workgroup partial records -> tot_sum / tot_max
__global__ __launch_bounds__(256) void k_reduce_partials(StepArgs a) {
  __shared__ double lds[3 * 4];
  const int e = blockIdx.x;
  Red3 acc{0.0, 0.0, 0.0f};
  for (int b = threadIdx.x; b < a.nblk; b += 256) {
    const double* rec = a.partials + ((int64_t)e * a.nblk + b) * 3;
    acc.sum_p += rec[0];
    acc.sum_pen += rec[1];
    acc.max_pen = fmaxf(acc.max_pen, (float)rec[2]);
  }
  const Red3 tot = block_reduce<256>(acc, lds);
  if (threadIdx.x == 0) {
    a.tot_sum[e] = tot.sum_p;
    a.tot_sum[a.E + e] = tot.sum_pen;
    a.tot_max[e] = (double)tot.max_pen;
  }
}

// rewards and the two power columns from the env's totals and the per-house penalties k_step_partial left in `reward`.  Where the totals come from:
//   records != nullptr  the per-workgroup partial records themselves, [world][E][nblk][3] (world = 1: this device's own
//                       `partials`; world > 1: every rank's records, all-gathered): EVERY finish workgroup re-sums its env's
//                       world * nblk records from L2 in one fixed order (thread t: ranks in order, records t, t + THREADS, ... ;
//                       then the workgroup tree) - identical totals in every workgroup, no reduction launch in between;
//   gathered != nullptr [world][3][E] per-rank totals (mdr_env_step_end_gathered): summed in rank order;
//   else                tot_sum / tot_max as the caller left them (mdr_env_step_end).
template <int VEC, int THREADS>
__device__ __forceinline__ void finish_block(const StepArgs& a, double* lds) {
  const int e = blockIdx.y;
  const int h = ((int)blockIdx.x * THREADS + (int)threadIdx.x) * VEC;
  double P, sum_pen;
  float max_pen;
  if (a.records != nullptr) {
    Red3 acc{0.0, 0.0, 0.0f};
    for (int r = 0; r < a.world; ++r) {
      const double* rec = a.records + ((int64_t)r * a.E + e) * a.nblk * 3;
      for (int b = threadIdx.x; b < a.nblk; b += THREADS) {
        acc.sum_p += rec[3 * b];
        acc.sum_pen += rec[3 * b + 1];
        acc.max_pen = fmaxf(acc.max_pen, (float)rec[3 * b + 2]);
      }
    }
    const Red3 tot = block_reduce<THREADS>(acc, lds);
    P = tot.sum_p;
    sum_pen = tot.sum_pen;
    max_pen = tot.max_pen;
  } else if (a.gathered != nullptr) {   // [world][3][E] local aggregates of every rank: reduce here, in rank order
    P = 0.0;
    sum_pen = 0.0;
    max_pen = 0.0f;
    for (int r = 0; r < a.world; ++r) {
      const double* g = a.gathered + (int64_t)r * 3 * a.E;
      P += g[e];
      sum_pen += g[a.E + e];
      max_pen = fmaxf(max_pen, (float)g[2 * a.E + e]);
    }
  } else {
    P = a.tot_sum[e];
    sum_pen = a.tot_sum[a.E + e];
    max_pen = (float)a.tot_max[e];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.P[e] = P;
  if (h < a.N) {
    const int64_t i = (int64_t)e * a.N + h;
    float pen[VEC];
    load_vec<VEC>(a.stash, i, pen);   // left there by the partial kernel
    store_reward_power<VEC>(a, i, pen, sum_pen, max_pen, signal_term(a, P, a.sig_old[e]),
                            (float)(a.sig_new[e] * a.inv_obs_norm), (float)(P * a.inv_obs_norm));
  }
}

template <int VEC, int THREADS>
__global__ __launch_bounds__(THREADS) void k_step_finish(StepArgs a) {
  rebase_finish(a);
  __shared__ double lds[3 * 4];
  finish_block<VEC, THREADS>(a, lds);
  if (a.cursor_adv != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // see rebase_finish
    a.cursor_adv[0] = a.cursor_adv[snap_index(a.snap_rd)] + 1;
    a.cursor_adv[1] += a.cursor_steps;
  }
}

// Finish of step k and partial of step k + 1 in ONE launch (mdr_env_step_end_begin_records): between two exchanges of a rollout
// there is then one launch, not two - the finish needs the gathered records of step k, the partial nothing but the state step k
// left, and a house's two halves run in the same thread (its stashed penalty is read before the next one is written).
template <int VEC, int THREADS>
__global__ __launch_bounds__(THREADS) void k_step_finish_partial(StepArgs f, StepArgs p) {
  if (p.cursor != nullptr) {   // graph mode: the row of step k is in one note, this launch leaves the row of step k + 1 in the other
    const int row = p.cursor[snap_index(p.snap_rd)];
    const int64_t off_f = (int64_t)min(row, p.cursor_max) * p.E, off_p = (int64_t)min(row + 1, p.cursor_max) * p.E;
    f.sig_old += off_f;
    f.sig_new += off_f;
    p.od_old += off_p;
    p.solar_new += off_p;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.cursor_adv[snap_index(p.snap_wr)] = row + 1;
  }
  __shared__ double lds[3 * 4];
  finish_block<VEC, THREADS>(f, lds);
  __syncthreads();
  partial_block<VEC, THREADS>(p, lds);
}

// =================================================================================================
// Full flat observation: utils.normStateDict (utils.py:740-880) for every house, messages included.
// Feature order = the dict insertion order of normStateDict:
//   (Ta-20)/5, (Tm-20)/5, (target-20)/5, [(OD-20)/5], deadband, [sin,cos day], [sin,cos hour], [solar/1000],
//   capacity/def, [Ua,Cm,Ca,Hm/def], [COP,latent/def], on, lock, sso/L, 1, S/norm, P/norm,
//   then per sender j: diff_j/5, sso_j/L_own, curr_j/norm_reg, max_j/norm_reg, [Ua..Hm_j/def], [COP,latent,cap_j/def]
// =================================================================================================
struct MsgFields {  // SingleHouse.message (env 624-662), already normalised except sso (needs the receiver's lockout)
  float diff, sso, curr, pmax, Ua, Cm, Ca, Hm, COP, latent, cap;
};

__device__ __forceinline__ MsgFields sender_from_global(const ObsArgs& a, int64_t s) {
  MsgFields m;
  const float pmax = a.P_max[s];
  m.diff = (a.Ta[s] - a.target[s]) * 0.2f;
  m.sso = (float)a.sso[s];
  m.curr = ((a.flags[s] & 1u) ? pmax : 0.0f) * a.inv_norm_reg;
  m.pmax = pmax * a.inv_norm_reg;
  if (a.m_thermal) {
    m.Ua = a.Ua[s] * a.inv_Ua;
    m.Cm = a.Cm[s] * a.inv_Cm;
    m.Ca = a.Ca[s] * a.inv_Ca;
    m.Hm = a.Hm[s] * a.inv_Hm;
  }
  if (a.m_hvac) {
    m.COP = a.COP[s] * a.inv_COP;
    m.latent = a.latent[s] * a.inv_latent;
    m.cap = a.capacity[s] * a.inv_cap;
  }
  return m;
}

// agents_comm_mode "random_sample" (env 976-983): random.sample(others, k = nb_comm) per house and step, i.e. an
// ordered draw without replacement.  Exactly that, by rejection: slot m takes uniform draws among the N - 1 other
// houses (TAG_LINKS Philox stream of (env, house, time index), 4 draws per block) until one is not among the m senders
// already chosen.  The chosen list lives in registers (nb_comm <= 16), all indexing is static.
constexpr int MAX_RANDOM_LINKS = 16;

struct LinkSampler {
  int prev[MAX_RANDOM_LINKS];
  int count = 0, draws = 0;
  u32x4 buf{0, 0, 0, 0};

  __device__ __forceinline__ int next(const ObsArgs& a, int e, int h) {
    const uint32_t D = (uint32_t)(a.n_total - 1);   // >= nb_comm >= 1; the whole env's other houses
    const int hg = h + (int)a.house_offset;         // global id of the receiving house
    int pick;
    bool taken;
    do {
      if ((draws & 3) == 0)
        buf = philox4x32_10((uint32_t)(e + a.env_offset), (uint32_t)(h + a.house_offset), (uint32_t)a.k,
                            TAG_LINKS | ((uint32_t)(draws >> 2) << 8), a.k0, a.k1 ^ (a.episode * 0x85EBCA6Bu));
      const uint32_t x = (draws & 3) == 0 ? buf.x : (draws & 3) == 1 ? buf.y : (draws & 3) == 2 ? buf.z : buf.w;
      ++draws;
      pick = (int)mulhi_pick(x, D);
      taken = false;
#pragma unroll
      for (int t = 0; t < MAX_RANDOM_LINKS; ++t) taken = taken || (t < count && prev[t] == pick);
    } while (taken);
#pragma unroll
    for (int t = 0; t < MAX_RANDOM_LINKS; ++t)
      if (t == count) prev[t] = pick;
    ++count;
    return pick < hg ? pick : pick + 1;   // index among the OTHER houses -> (global) house id
  }
};

struct NoSampler {   // stands in for LinkSampler in the instantiations without random links: no registers, never called
  __device__ __forceinline__ int next(const ObsArgs&, int, int) { return 0; }
};

// sender id of message slot m of house h: link table, or circular neighbours (env 816-828); random_sample: LinkSampler
__device__ __forceinline__ int sender_id(const ObsArgs& a, int h, int m) {
  if (a.links != nullptr) return a.links[(int64_t)h * a.c + m];
  const int before = a.c / 2;
  int j = (m < before ? h - before + m : h + 1 + (m - before)) % a.N;
  return j < 0 ? j + a.N : j;
}

// Emits the F features of houses h .. h+VEC-1 of env e through put(v[VEC]) in normStateDict order;
// sender(m, q) yields message slot m of house h+q.
template <int VEC, bool WITH_MSG = true, typename Put, typename Sender>
__device__ __forceinline__ void obs_features(const ObsArgs& a, int e, int h, int64_t i, Put&& put, Sender&& sender) {
  float v[VEC], w[VEC], L[VEC];
  int li[VEC];
  unsigned fl[VEC];
  auto scaled = [&](const float* __restrict__ p, float shift, float scale) {
    load_vec<VEC>(p, i, w);
#pragma unroll
    for (int q = 0; q < VEC; ++q) v[q] = (w[q] + shift) * scale;
    put(v);
  };
  auto all = [&](float x) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) v[q] = x;
    put(v);
  };
  load_vec<VEC>(a.lockout, i, li);
#pragma unroll
  for (int q = 0; q < VEC; ++q) L[q] = (float)li[q];
  scaled(a.Ta, a.obs_tshift, 0.2f);
  scaled(a.Tm, a.obs_tshift, 0.2f);
  scaled(a.target, a.obs_tshift, 0.2f);
  if (a.f_thermal) all((a.od_now[e] + a.obs_tshift) * 0.2f);
  scaled(a.deadband, 0.0f, 1.0f);
  if (a.f_day || a.f_hour) {
    const Civil c = civil_from_epoch(a.t0[e] + a.k * (int64_t)a.dt);
    if (a.f_day) {   // utils.py:806-809: tm_yday * 2 pi / 365
      const double ang = (double)c.yday * 6.283185307179586476925286766559 / 365.0;
      all((float)sin(ang));
      all((float)cos(ang));
    }
    if (a.f_hour) {  // utils.py:810-813: integer hour * 2 pi / 24
      const double ang = (double)c.hour * 6.283185307179586476925286766559 / 24.0;
      all((float)sin(ang));
      all((float)cos(ang));
    }
  }
  if (a.f_solar) all(a.k > 0 ? a.solar_now[e] * 1e-3f : 0.0f);   // current_solar_gain is 0 until the first step (env 573)
  scaled(a.capacity, 0.0f, a.inv_cap);
  if (a.f_thermal) {
    scaled(a.Ua, 0.0f, a.inv_Ua);
    scaled(a.Cm, 0.0f, a.inv_Cm);
    scaled(a.Ca, 0.0f, a.inv_Ca);
    scaled(a.Hm, 0.0f, a.inv_Hm);
  }
  if (a.f_hvac) {
    scaled(a.COP, 0.0f, a.inv_COP);
    scaled(a.latent, 0.0f, a.inv_latent);
  }
  load_bytes<VEC>(a.flags, i, fl);
#pragma unroll
  for (int q = 0; q < VEC; ++q) v[q] = (fl[q] & 1u) ? 1.0f : 0.0f;
  put(v);
#pragma unroll
  for (int q = 0; q < VEC; ++q) v[q] = (fl[q] & 2u) ? 1.0f : 0.0f;
  put(v);
  load_vec<VEC>(a.sso, i, li);
#pragma unroll
  for (int q = 0; q < VEC; ++q) v[q] = (float)li[q] / L[q];
  put(v);
#pragma unroll
  for (int q = 0; q < VEC; ++q) v[q] = L[q] / L[q];
  put(v);
  all((float)(a.sig_now[e] * a.inv_obs_norm));
  all((float)(a.P[e] * a.inv_obs_norm));
  if (!WITH_MSG) return;
  u32x4 rnd[VEC];
  for (int m = 0; m < a.c; ++m) {
    float z[VEC];
    MsgFields s[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      z[q] = 1.0f;
      if (a.defect_prob > 0.0f) {   // np.random.rand() > comm_defect_prob keeps the message (env 992)
        if ((m & 3) == 0)
          rnd[q] = philox4x32_10((uint32_t)(e + a.env_offset), (uint32_t)(h + q + a.house_offset), (uint32_t)a.k,
                                 TAG_COMM | ((uint32_t)(m >> 2) << 8), a.k0, a.k1 ^ (a.episode * 0x85EBCA6Bu));
        const uint32_t x = (m & 3) == 0 ? rnd[q].x : (m & 3) == 1 ? rnd[q].y : (m & 3) == 2 ? rnd[q].z : rnd[q].w;
        z[q] = (float)u01(x) > a.defect_prob ? 1.0f : 0.0f;
      }
      s[q] = sender(m, q);
    }
#define MDR_MSG(expr)                               \
  {                                                 \
    _Pragma("unroll") for (int q = 0; q < VEC; ++q) v[q] = z[q] * (expr); \
    put(v);                                         \
  }
    MDR_MSG(s[q].diff)
    MDR_MSG(s[q].sso / L[q])
    MDR_MSG(s[q].curr)
    MDR_MSG(s[q].pmax)
    if (a.m_thermal) {
      MDR_MSG(s[q].Ua)
      MDR_MSG(s[q].Cm)
      MDR_MSG(s[q].Ca)
      MDR_MSG(s[q].Hm)
    }
    if (a.m_hvac) {
      MDR_MSG(s[q].COP)
      MDR_MSG(s[q].latent)
      MDR_MSG(s[q].cap)
    }
#undef MDR_MSG
  }
}

// message fields of `count` consecutive (circular) houses starting at `first` -> LDS, field-major [nf][span]
__device__ __forceinline__ void stage_senders(const ObsArgs& a, int64_t base, int first, int count, int span, float* msg,
                                              int tid, int nthreads) {
  for (int idx = tid; idx < count; idx += nthreads) {
    int j = (first + idx) % a.N;
    if (j < 0) j += a.N;
    const MsgFields m = sender_from_global(a, base + j);
    msg[0 * span + idx] = m.diff;
    msg[1 * span + idx] = m.sso;
    msg[2 * span + idx] = m.curr;
    msg[3 * span + idx] = m.pmax;
    int q = 4;
    if (a.m_thermal) {
      msg[(q + 0) * span + idx] = m.Ua;
      msg[(q + 1) * span + idx] = m.Cm;
      msg[(q + 2) * span + idx] = m.Ca;
      msg[(q + 3) * span + idx] = m.Hm;
      q += 4;
    }
    if (a.m_hvac) {
      msg[(q + 0) * span + idx] = m.COP;
      msg[(q + 1) * span + idx] = m.latent;
      msg[(q + 2) * span + idx] = m.cap;
    }
  }
}

__device__ __forceinline__ MsgFields sender_from_lds(const ObsArgs& a, const float* msg, int span, int idx) {
  MsgFields s;
  s.diff = msg[0 * span + idx];
  s.sso = msg[1 * span + idx];
  s.curr = msg[2 * span + idx];
  s.pmax = msg[3 * span + idx];
  int q = 4;
  if (a.m_thermal) {
    s.Ua = msg[(q + 0) * span + idx];
    s.Cm = msg[(q + 1) * span + idx];
    s.Ca = msg[(q + 2) * span + idx];
    s.Hm = msg[(q + 3) * span + idx];
    q += 4;
  }
  if (a.m_hvac) {
    s.COP = msg[(q + 0) * span + idx];
    s.latent = msg[(q + 1) * span + idx];
    s.cap = msg[(q + 2) * span + idx];
  }
  return s;
}

// message record of a sender slot (sharded houses: local messages followed by the halo), fields in MsgFields order
__device__ __forceinline__ MsgFields sender_from_ext(const ObsArgs& a, int e, int slot) {
  const float* r = a.msg_ext_in + ((int64_t)e * a.ext_entries + slot) * a.mf;
  MsgFields s;
  s.diff = r[0];
  s.sso = r[1];
  s.curr = r[2];
  s.pmax = r[3];
  int q = 4;
  if (a.m_thermal) {
    s.Ua = r[q]; s.Cm = r[q + 1]; s.Ca = r[q + 2]; s.Hm = r[q + 3];
    q += 4;
  }
  if (a.m_hvac) {
    s.COP = r[q]; s.latent = r[q + 1]; s.cap = r[q + 2];
  }
  return s;
}

// SingleHouse.message (env 624-662) of every local house as one record of mf floats: what a shard exports to its peers
__global__ __launch_bounds__(256) void k_obs_messages(ObsArgs a) {
  rebase(a);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.plane) return;
  const int e = (int)(i / a.N);
  const int h = (int)(i - (int64_t)e * a.N);
  const MsgFields m = sender_from_global(a, i);
  float* r = a.msg_ext_out + ((int64_t)e * a.ext_entries + h) * a.mf;
  r[0] = m.diff;
  r[1] = m.sso;
  r[2] = m.curr;
  r[3] = m.pmax;
  int q = 4;
  if (a.m_thermal) {
    r[q] = m.Ua; r[q + 1] = m.Cm; r[q + 2] = m.Ca; r[q + 3] = m.Hm;
    q += 4;
  }
  if (a.m_hvac) {
    r[q] = m.COP; r[q + 1] = m.latent; r[q + 2] = m.cap;
  }
}

// ---- the random draws of the message gather on their own: sender id and keep flag of every message slot of every local
// house at the current time index - exactly what the observation kernels use (same Philox streams, same counters), for
// hosts that build the reference's `message` lists themselves (the dict adapter) and for draw-by-draw parity tests.
// senders: global house ids (link table entry / circular neighbour among nb_houses_total / random_sample draw).
template <bool RANDOM>
__global__ __launch_bounds__(256) void k_comm_draws(ObsArgs a, int32_t* __restrict__ senders, uint8_t* __restrict__ keep) {
  rebase(a);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.plane) return;
  const int e = (int)(i / a.N);
  const int h = (int)(i - (int64_t)e * a.N);
  const int hg = h + (int)a.house_offset;
  typename std::conditional<RANDOM, LinkSampler, NoSampler>::type smp;
  u32x4 rnd{0, 0, 0, 0};
  const int before = a.c / 2;
  for (int m = 0; m < a.c; ++m) {
    int sid;
    if (RANDOM) {
      sid = smp.next(a, e, h);
    } else if (a.links != nullptr) {
      sid = a.links[(int64_t)h * a.c + m];
    } else {   // circular neighbours of the whole env (env 816-828)
      sid = (m < before ? hg - before + m : hg + 1 + (m - before)) % a.n_total;
      if (sid < 0) sid += a.n_total;
    }
    unsigned ok = 1u;
    if (a.defect_prob > 0.0f) {   // np.random.rand() > comm_defect_prob keeps the message (env 992)
      if ((m & 3) == 0)
        rnd = philox4x32_10((uint32_t)(e + a.env_offset), (uint32_t)hg, (uint32_t)a.k, TAG_COMM | ((uint32_t)(m >> 2) << 8), a.k0,
                            a.k1 ^ (a.episode * 0x85EBCA6Bu));
      const uint32_t x = (m & 3) == 0 ? rnd.x : (m & 3) == 1 ? rnd.y : (m & 3) == 2 ? rnd.z : rnd.w;
      ok = (float)u01(x) > a.defect_prob ? 1u : 0u;
    }
    senders[i * a.c + m] = sid;
    if (keep != nullptr) keep[i * a.c + m] = (uint8_t)ok;
  }
}

hipError_t launch_comm_draws(const ObsArgs& a, int32_t* senders, uint8_t* keep, hipStream_t s) {
  const dim3 g((unsigned)((a.plane + 255) / 256)), b(256);
  if (a.random_links) hipLaunchKernelGGL(k_comm_draws<true>, g, b, 0, s, a, senders, keep);
  else hipLaunchKernelGGL(k_comm_draws<false>, g, b, 0, s, a, senders, keep);
  return hipGetLastError();
}

// ---- simple form: one thread per house, direct (4-byte) stores; fallback for shapes the other kernels cannot hold.
// EXT: the senders are message records (sender_from_ext) addressed through the link table - the sharded-houses form.
template <int LAYOUT, bool RANDOM, bool EXT = false>
__global__ __launch_bounds__(256) void k_obs_vector(ObsArgs a) {
  rebase(a);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.plane) return;
  const int e = (int)(i / a.N);
  const int h = (int)(i - (int64_t)e * a.N);
  const int64_t base = (int64_t)e * a.N;
  int f = 0;
  typename std::conditional<RANDOM, LinkSampler, NoSampler>::type smp;
  obs_features<1>(a, e, h, i,
                  [&](const float* v) {
                    if (LAYOUT == MDR_OBS_PLANES) a.out[(int64_t)f * a.out_plane + i] = v[0];
                    else a.out[i * a.F + f] = v[0];
                    ++f;
                  },
                  [&](int m, int) {
                    if (EXT) return sender_from_ext(a, e, RANDOM ? smp.next(a, e, h) : a.links[(int64_t)h * a.c + m]);
                    return sender_from_global(a, base + (RANDOM ? smp.next(a, e, h) : sender_id(a, h, m)));
                  });
}

// A workgroup's share of the houses ("tile"): a TILE-house slice of one env when N >= TILE, or floor(TILE / N) WHOLE envs
// when N < TILE (the agent counts the reference trains with are 20 and 50) so that small envs still fill the workgroup.
// Senders are staged per env segment of (houses + c) entries - the segment's houses preceded / followed by their circular
// neighbours - hence message slot m of tile row r always sits at  r + env_l * c + m + (m >= c/2)  with env_l = r / N.
struct ObsTile {
  int e0, nenv, h0, nh, rows;   // first env, envs in the tile, first house / houses per env segment, total rows
};

template <int TILE>
__device__ __forceinline__ ObsTile obs_tile(const ObsArgs& a, int64_t id) {
  ObsTile t;
  if (a.N < TILE) {
    const int epb = TILE / a.N;
    t.e0 = (int)(id * epb);
    t.nenv = min(epb, a.E - t.e0);
    t.h0 = 0;
    t.nh = a.N;
    t.rows = t.nenv * a.N;
  } else {
    const int tpe = (a.N + TILE - 1) / TILE;
    t.e0 = (int)(id / tpe);
    t.h0 = (int)(id - (int64_t)t.e0 * tpe) * TILE;
    t.nenv = 1;
    t.nh = min(TILE, a.N - t.h0);
    t.rows = t.nh;
  }
  return t;
}

template <int TILE>
static int64_t obs_tile_count(int64_t E, int N) {
  if (N < TILE) {
    const int epb = TILE / N;
    return (E + epb - 1) / epb;
  }
  return E * ((N + TILE - 1) / TILE);
}

// message fields of every env segment of the tile -> LDS.  FIELD_MAJOR: msg[field][entry] (planes kernel) or
// entry-major msg[entry][field] (rows kernels).
template <bool FIELD_MAJOR>
__device__ __forceinline__ void stage_tile_senders(const ObsArgs& a, const ObsTile& t, float* msg, int entries, int mf, int tid,
                                                   int nthreads) {
  const int seg = t.nh + a.c;
  const int before = a.c / 2;
  for (int idx = tid; idx < t.nenv * seg; idx += nthreads) {
    const int el = idx / seg;
    const int p = idx - el * seg;
    int j = (t.h0 - before + p) % a.N;
    if (j < 0) j += a.N;
    const MsgFields m = sender_from_global(a, (int64_t)(t.e0 + el) * a.N + j);
    float v[11] = {m.diff, m.sso, m.curr, m.pmax, 0, 0, 0, 0, 0, 0, 0};
    int q = 4;
    if (a.m_thermal) {
      v[q] = m.Ua; v[q + 1] = m.Cm; v[q + 2] = m.Ca; v[q + 3] = m.Hm;
      q += 4;
    }
    if (a.m_hvac) {
      v[q] = m.COP; v[q + 1] = m.latent; v[q + 2] = m.cap;
    }
    for (int k = 0; k < mf; ++k) msg[FIELD_MAJOR ? k * entries + idx : idx * mf + k] = v[k];
  }
}

// ---- planes layout: VEC houses per thread (N % VEC == 0), up to 256 * VEC houses per workgroup.  The senders' message
// fields are staged once in LDS, so HBM sees each state array once and the 10x message fan-out happens in LDS; every
// feature plane is written with 4 * VEC-byte non-temporal stores.
template <int VEC, bool RANDOM>
__global__ __launch_bounds__(256) void k_obs_planes(ObsArgs a) {
  rebase(a);
  constexpr int OBS_PTILE = 256 * VEC;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const ObsTile t = obs_tile<OBS_PTILE>(a, blockIdx.x);
  const int before = a.c / 2;
  const int mf = 4 + (a.m_thermal ? 4 : 0) + (a.m_hvac ? 3 : 0);
  const int entries = a.lds_entries;   // nenv * (nh + c), computed by the launcher
  const bool staged = (a.links == nullptr) && !a.random_links;
  if (staged) {
    stage_tile_senders<true>(a, t, lds, entries, mf, tid, 256);
    __syncthreads();
  }
  const int r = tid * VEC;   // first of this thread's VEC tile rows (same env: N % VEC == 0)
  if (r >= t.rows) return;
  const int el = (t.nenv > 1) ? (int)(((uint32_t)r * a.magic_n) >> 20) : 0;
  const int e = t.e0 + el;
  const int h = t.h0 + r - el * a.N;
  const int64_t i = (int64_t)e * a.N + h;
  int f = 0;
  typename std::conditional<RANDOM, LinkSampler, NoSampler>::type smp[VEC];
  obs_features<VEC>(a, e, h, i,
                  [&](const float* v) {
                    store_out<VEC>(a.out + (int64_t)f * a.out_plane, i, v);
                    ++f;
                  },
                  [&](int m, int q) {
                    if (!staged) return sender_from_global(a, (int64_t)e * a.N + (RANDOM ? smp[q].next(a, e, h + q) : sender_id(a, h + q, m)));
                    return sender_from_lds(a, lds, entries, r + q + el * a.c + m + (m >= before ? 1 : 0));
                  });
}

// ---- rows layout (and planes for N % 4 != 0): one workgroup per TILE consecutive houses of one env.
//  (1) sender fields staged in LDS as above; (2) every thread builds its house's F features into an LDS chunk laid
//  out like the OUTPUT tile; (3) the chunk is streamed out with 16-byte stores: rows -> one contiguous TILE*F-float
//  block, planes -> F rows of TILE floats.  TILE = 64 keeps a workgroup at one wavefront (LDS ~13 KB at F = 51).
template <int LAYOUT, int TILE, bool RANDOM>
__global__ __launch_bounds__(TILE) void k_obs_tiled(ObsArgs a) {
  rebase(a);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int tpe = (a.N + TILE - 1) / TILE;              // 1-D grid: tile id -> (env, first house)
  const int e = (int)(blockIdx.x / (unsigned)tpe);
  const int h0 = (int)(blockIdx.x - (unsigned)e * (unsigned)tpe) * TILE;
  const int h = h0 + tid;
  const int nh = min(TILE, a.N - h0);
  const int64_t base = (int64_t)e * a.N;
  const int Fp = (LAYOUT == MDR_OBS_ROWS) ? (a.F | 1) : TILE;   // odd row stride: conflict-free ds_write
  float* chunk = lds;
  float* msg = lds + ((LAYOUT == MDR_OBS_ROWS) ? TILE * Fp : a.F * TILE);
  const int span = TILE + a.c;
  const int before = a.c / 2;
  const bool staged = (a.links == nullptr) && !a.random_links;
  if (staged) {
    stage_senders(a, base, h0 - before, span, span, msg, tid, TILE);
    __syncthreads();
  }
  if (h < a.N) {
    int f = 0;
    typename std::conditional<RANDOM, LinkSampler, NoSampler>::type smp;
    obs_features<1>(a, e, h, base + h,
                    [&](const float* v) {
                      chunk[(LAYOUT == MDR_OBS_ROWS) ? tid * Fp + f : f * TILE + tid] = v[0];
                      ++f;
                    },
                    [&](int m, int) {
                      if (!staged) return sender_from_global(a, base + (RANDOM ? smp.next(a, e, h) : sender_id(a, h, m)));
                      return sender_from_lds(a, msg, span, tid + m + (m >= before ? 1 : 0));
                    });
  }
  __syncthreads();
  if (LAYOUT == MDR_OBS_ROWS) {
    float* dst = a.out + (base + h0) * a.F;          // nh * F contiguous floats
    const int total = nh * a.F;
    const bool vec = (((uintptr_t)dst) & 15u) == 0;
    // element o of the tile <-> (row r, feature f) with o = r * F + f, advanced incrementally (no division in the loop)
    int r = (tid * 4) / a.F, f = tid * 4 - r * a.F;
    const int dr = (TILE * 4) / a.F, df = TILE * 4 - dr * a.F;
    for (int o = tid * 4; o < total; o += TILE * 4) {
      float v[4];
      int rr = r, ff = f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = chunk[min(rr, nh - 1) * Fp + ff];
        if (++ff == a.F) {
          ff = 0;
          ++rr;
        }
      }
      if (vec && o + 3 < total) {
        store_out<4>(dst, o, v);
      } else {
        for (int q = 0; q < 4 && o + q < total; ++q) dst[o + q] = v[q];
      }
      r += dr;
      f += df;
      if (f >= a.F) {
        f -= a.F;
        ++r;
      }
    }
  } else {
    for (int idx = tid; idx < a.F * TILE; idx += TILE) {   // N % 4 != 0 here: plain 4-byte stores
      const int f = idx / TILE;
      const int c = idx - f * TILE;
      if (c < nh) a.out[(int64_t)f * a.out_plane + base + h0 + c] = chunk[idx];
    }
  }
}

// ---- rows layout, circular neighbours: one workgroup per TILE consecutive houses of one env - or, for envs of fewer than TILE
// houses, per TILE / N whole envs (the reference trains with 20 houses and deploys with 50: one env per 256-thread workgroup left
// four threads in five idle and took 3x the time of the default-shape kernel).  Only the COMPACT data is staged in LDS - each
// house's own features [TILE][own], the senders' message fields of every env segment [(houses + c)][mf] and the receivers'
// lockout - and the output rows are then generated directly in output order (element o = r * F + f) and streamed with 16-byte
// stores.  HBM sees each state array once while the 10x message fan-out is served from LDS.
template <int TILE>
__global__ __launch_bounds__(TILE) void k_obs_rows(ObsArgs a) {
  rebase(a);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const ObsTile t = obs_tile<TILE>(a, blockIdx.x);
  const int mf = 4 + (a.m_thermal ? 4 : 0) + (a.m_hvac ? 3 : 0);
  const int own = a.F - a.c * mf;
  const int ownp = own | 1;
  const int entries = t.nenv * (t.nh + a.c);             // staged senders: per env segment its houses and the c around them
  const int before = a.c / 2;
  const bool multi = t.nenv > 1;
  float* ownbuf = lds;                                   // [TILE][ownp]
  float* msg = ownbuf + TILE * ownp;                     // [a.lds_entries][mf]
  float* lock_f = msg + a.lds_entries * mf;              // [TILE] the receiver's lockout L ...
  float* lock_r = lock_f + TILE;                         // [TILE] ... and RN(1 / L)
  uint32_t* dead = reinterpret_cast<uint32_t*>(lock_r + TILE);   // [TILE] bit m: message slot m is defective
  uint32_t* desc = dead + TILE;                          // [F] where element f of a row comes from
  const int msg_base = TILE * ownp;                      // index of msg[] inside lds[]
  constexpr uint32_t D_MSG = 1u << 31, D_SSO = 1u << 30;
  // feature descriptor: own feature f -> offset f in the house's own row; message feature -> offset of (sender slot,
  // field) relative to the house's first sender in msg[], flagged when it is the sender's seconds_since_off
  for (int f = tid; f < a.F; f += TILE) {
    uint32_t d;
    if (f < own) {
      d = (uint32_t)f;
    } else {
      const int g = f - own, m = g / mf, k = g - m * mf;
      d = D_MSG | (k == 1 ? D_SSO : 0u) | ((uint32_t)m << 16) | (uint32_t)((m + (m >= before ? 1 : 0)) * mf + k);
    }
    desc[f] = d;
  }
  stage_tile_senders<false>(a, t, msg, entries, mf, tid, TILE);
  if (tid < t.rows) {
    const int el = multi ? (int)(((uint32_t)tid * a.magic_n) >> 20) : 0;
    const int e = t.e0 + el, h = t.h0 + tid - el * a.N;
    const int64_t i = (int64_t)e * a.N + h;
    int f = 0;
    obs_features<1, false>(a, e, h, i, [&](const float* v) { ownbuf[tid * ownp + f] = v[0]; ++f; },
                           [&](int, int) { return MsgFields{}; });
    lock_f[tid] = (float)a.lockout[i];
    lock_r[tid] = 1.0f / lock_f[tid];
    uint32_t mask = 0;
    if (a.defect_prob > 0.0f) {
      u32x4 rnd{0, 0, 0, 0};
      for (int m = 0; m < a.c; ++m) {
        if ((m & 3) == 0)
          rnd = philox4x32_10((uint32_t)(e + a.env_offset), (uint32_t)(h + a.house_offset), (uint32_t)a.k,
                              TAG_COMM | ((uint32_t)(m >> 2) << 8), a.k0, a.k1 ^ (a.episode * 0x85EBCA6Bu));
        const uint32_t x = (m & 3) == 0 ? rnd.x : (m & 3) == 1 ? rnd.y : (m & 3) == 2 ? rnd.z : rnd.w;
        if (!((float)u01(x) > a.defect_prob)) mask |= 1u << m;
      }
    }
    dead[tid] = mask;
  }
  __syncthreads();
  float* dst = a.out + ((int64_t)t.e0 * a.N + t.h0) * a.F;          // rows * F contiguous floats
  const int nh = t.rows;
  const int total = nh * a.F;
  // the tile's first byte is only 4-byte aligned in general (odd F): the 16-byte store grid starts at the next boundary, the
  // up-to-3 leading floats go out singly (thread 0)
  const int lead = (int)(((16u - ((uint32_t)(uintptr_t)dst & 15u)) & 15u) >> 2);
  const bool defects = a.defect_prob > 0.0f;
  auto element = [&](int rc, int ff) {
    const uint32_t d = desc[ff];
    const bool is_msg = (d & D_MSG) != 0u;
    const int el = multi ? (int)(((uint32_t)rc * a.magic_n) >> 20) : 0;   // the row's env segment: its senders start c * el entries further
    float val = lds[(is_msg ? msg_base + (rc + a.c * el) * mf : rc * ownp) + (int)(d & 0xFFFFu)];
    if (d & D_SSO) val = div_by_lockout(val, lock_f[rc], lock_r[rc]);   // sender's seconds_since_off over the RECEIVER's lockout (utils.py:849-851)
    if (defects && is_msg && ((dead[rc] >> ((d >> 16) & 63u)) & 1u)) val = 0.0f;
    return val;
  };
  if (tid == 0)
    for (int q = 0; q < lead && q < total; ++q) dst[q] = element(q / a.F, q % a.F);
  const int o0 = lead + tid * 4;
  int r = o0 / a.F, f = o0 - r * a.F;
  const int dr = (TILE * 4) / a.F, df = TILE * 4 - dr * a.F;
  const bool vec = true;
  for (int o = o0; o < total; o += TILE * 4) {
    float v[4];
    int rr = r, ff = f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rc = min(rr, nh - 1);
      v[q] = element(rc, ff);
      if (++ff == a.F) {
        ff = 0;
        ++rr;
      }
    }
    if (vec && o + 3 < total) {
      store_out<4>(dst, o, v);   // non-temporal: the rows are written once and read by another kernel
    } else {
      for (int q = 0; q < 4 && o + q < total; ++q) dst[o + q] = v[q];
    }
    r += dr;
    f += df;
    if (f >= a.F) {
      f -= a.F;
      ++r;
    }
  }
}

// ---- rows layout, the reference's DEFAULT observation (state_properties / message_properties all False, 10 circular
// neighbours, no link defects): F = 11 + 10 * 4 = 51 is a compile-time constant, so element o of a tile maps to its LDS
// source with a handful of integer ops: row r = o / 51, f = o % 51; own feature -> own[r][f]; message feature
// g = f - 11 -> msg[4 (r + 10 env_l + slot) + k] = msg[4 (r + 10 env_l) + g + (g >= 20 ? 4 : 0)] (the sender list skips
// the house itself after slot 4).  Only compact data is staged in LDS (own features, sender fields, sso / lockout pairs), the
// output rows are generated directly in output order and streamed with 16-byte non-temporal stores.
template <int TILE>
__global__ __launch_bounds__(TILE) void k_obs_rows_default(ObsArgs a) {
  rebase(a);
  constexpr int OWN = 11, MF = 4, C = 10, F = OWN + C * MF, OWNP = 11;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const ObsTile t = obs_tile<TILE>(a, blockIdx.x);
  float* ownbuf = lds;                 // [TILE][OWNP]
  float* quot = lds + TILE * OWNP;     // [TILE][C]  sender's seconds_since_off over the RECEIVER's lockout (utils.py:849-851)
  float* msg = quot + TILE * C;        // [nenv * (nh + C)][MF]
  stage_tile_senders<false>(a, t, msg, 0, MF, tid, TILE);
  const int el_own = (tid < t.rows && t.nenv > 1) ? (int)(((uint32_t)tid * a.magic_n) >> 20) : 0;
  float L = 1.0f;
  if (tid < t.rows) {
    const int e = t.e0 + el_own;
    const int h = t.h0 + tid - el_own * a.N;
    const int64_t i = (int64_t)e * a.N + h;
    int f = 0;
    obs_features<1, false>(a, e, h, i, [&](const float* v) { ownbuf[tid * OWNP + f] = v[0]; ++f; },
                           [&](int, int) { return MsgFields{}; });
    L = (float)a.lockout[i];
  }
  __syncthreads();
  if (tid < t.rows) {                  // one quotient per (receiver, slot), rounded as the `/` of the other kernels rounds it
    const float y = 1.0f / L;
#pragma unroll
    for (int m = 0; m < C; ++m)
      quot[tid * C + m] = div_by_lockout(msg[4 * (tid + C * el_own + m + (m >= C / 2 ? 1 : 0)) + 1], L, y);
  }
  __syncthreads();
  float* dst = a.out + ((int64_t)t.e0 * a.N + t.h0) * F;   // rows * F contiguous floats
  const int total = t.rows * F;
  // the tile's first byte is only 4-byte aligned in general (F is odd): start the 16-byte store grid at the next
  // boundary; the up-to-3 leading floats are covered by the group at o = lead - 4 through the scalar branch below
  const int lead = (int)(((16u - ((uint32_t)(uintptr_t)dst & 15u)) & 15u) >> 2);
  const bool multi = t.nenv > 1;
  // element (row r, feature f) of the tile; a thread's groups of four lie TILE * 4 floats apart = 20 rows and 4 features (1024 = 20 * 51
  // + 4), so (r, f) of a group's first float is carried from group to group instead of divided out of its offset each time
  auto element_rf = [&](int r, int f) {
    if (f < OWN) return ownbuf[r * OWNP + f];
    const int g = f - OWN;
    const int el = multi ? (int)(((uint32_t)r * a.magic_n) >> 20) : 0;
    if ((g & 3) == 1) return quot[r * C + (g >> 2)];
    return msg[4 * (r + C * el) + g + (g >= 20 ? 4 : 0)];
  };
  static_assert(TILE * 4 == 20 * F + 4, "the (row, feature) step of a thread's store grid");
  const int last = total - 1;
  const int rl = (int)(((uint32_t)last * 20561u) >> 20), fl = last - rl * F;   // the tile's last element (clamp target of the edge groups)
  auto group = [&](int r, int f, float* v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int rq = r, fq = f + q;
      if (fq >= F) {
        fq -= F;
        ++rq;
      }
      const bool past = rq > rl || (rq == rl && fq > fl);
      v[q] = element_rf(past ? rl : rq, past ? fl : fq);
    }
  };
  if (lead == 0) {   // aligned tile (always the case when N * F % 4 == 0): no edge handling inside the loop
    int o = tid * 4;
    int r = (int)(((uint32_t)o * 20561u) >> 20), f = o - r * F;   // == o / 51, o % 51 for every o < 13107 (checked exhaustively)
    for (; o < total; o += TILE * 4) {
      float v[4];
      group(r, f, v);
      if (o + 3 < total) {
        store_out<4>(dst, o, v);
      } else {
        for (int q = 0; q < 4 && o + q < total; ++q) dst[o + q] = v[q];
      }
      r += 20;
      f += 4;
      if (f >= F) {
        f -= F;
        ++r;
      }
    }
  } else {
    int o = lead - 4 + tid * 4;
    if (o < 0) {   // the leading partial group (thread 0 only): its floats at o + q >= 0
      for (int q = 0; q < 4; ++q)
        if (o + q >= 0 && o + q < total) dst[o + q] = element_rf(0, o + q);   // (lead <= 3 floats: row 0, features 0 .. 2)
      o += TILE * 4;
    }
    int r = (int)(((uint32_t)o * 20561u) >> 20), f = o - r * F;
    for (; o < total; o += TILE * 4) {
      float v[4];
      group(r, f, v);
      if (o + 3 < total) {
        store_out<4>(dst, o, v);
      } else {
        for (int q = 0; q < 4; ++q)
          if (o + q < total) dst[o + q] = v[q];
      }
      r += 20;
      f += 4;
      if (f >= F) {
        f -= F;
        ++r;
      }
    }
  }
}

int obs_message_fields(const mdr_obs_spec_t& s) { return 4 + (s.message_thermal ? 4 : 0) + (s.message_hvac ? 3 : 0); }

hipError_t launch_obs_messages(const ObsArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_obs_messages, dim3((unsigned)((a.plane + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_obs_vector_ext(const ObsArgs& a, int layout, hipStream_t s) {
  const dim3 g((unsigned)((a.plane + 255) / 256)), b(256);
  if (a.random_links) {   // record slots are global house ids (the caller gathered every house's record)
    if (layout == MDR_OBS_PLANES) hipLaunchKernelGGL((k_obs_vector<MDR_OBS_PLANES, true, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_obs_vector<MDR_OBS_ROWS, true, true>), g, b, 0, s, a);
  } else if (layout == MDR_OBS_PLANES) {
    hipLaunchKernelGGL((k_obs_vector<MDR_OBS_PLANES, false, true>), g, b, 0, s, a);
  } else {
    hipLaunchKernelGGL((k_obs_vector<MDR_OBS_ROWS, false, true>), g, b, 0, s, a);
  }
  return hipGetLastError();
}

int obs_vector_length(const mdr_obs_spec_t& s) {
  int own = 11 + (s.state_thermal ? 5 : 0) + (s.state_day ? 2 : 0) + (s.state_hour ? 2 : 0) + (s.state_solar_gain ? 1 : 0) +
            (s.state_hvac ? 2 : 0);
  int msg = 4 + (s.message_thermal ? 4 : 0) + (s.message_hvac ? 3 : 0);
  return own + s.nb_comm * msg;
}

template <typename K>
static hipError_t launch_with_lds(K kernel, dim3 g, dim3 b, size_t lds_bytes, hipStream_t s, const ObsArgs& a) {
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, g, b, lds_bytes, s, a);
  return hipGetLastError();
}

hipError_t launch_obs_vector(const ObsArgs& a_in, int layout, hipStream_t s) {
  ObsArgs a = a_in;
  a.magic_n = (uint32_t)(((1u << 20) + (uint32_t)a.N - 1u) / (uint32_t)a.N);   // r / N == (r * magic) >> 20 for r < 1024, N <= 1024
  const size_t nf = 4 + (a.m_thermal ? 4 : 0) + (a.m_hvac ? 3 : 0);
  const size_t lds_cap = 160 * 1024;
  const bool circular = a.links == nullptr && !a.random_links && a.N > a.c;   // every sender is a distinct circular neighbour
  if (layout == MDR_OBS_PLANES && (((uintptr_t)a.out) & 15u) == 0 && a.out_plane % 4 == 0) {
    const int vec = a.N % 4 == 0 ? 4 : (a.N % 2 == 0 ? 2 : 1);
    const int ptile = 256 * vec;
    a.lds_entries = a.N < ptile ? (ptile / a.N) * (a.N + a.c) : ptile + a.c;
    const size_t lds_bytes = (a.links == nullptr && !a.random_links) ? nf * (size_t)a.lds_entries * sizeof(float) : 16;
    const int64_t tiles = vec == 4 ? obs_tile_count<1024>(a.E, a.N) : (vec == 2 ? obs_tile_count<512>(a.E, a.N) : obs_tile_count<256>(a.E, a.N));
    if (lds_bytes <= lds_cap && tiles < (int64_t)1 << 31) {
      const dim3 g((unsigned)tiles), b(256);
      if (a.random_links) {
        if (vec == 4) return launch_with_lds(k_obs_planes<4, true>, g, b, lds_bytes, s, a);
        if (vec == 2) return launch_with_lds(k_obs_planes<2, true>, g, b, lds_bytes, s, a);
        return launch_with_lds(k_obs_planes<1, true>, g, b, lds_bytes, s, a);
      }
      if (vec == 4) return launch_with_lds(k_obs_planes<4, false>, g, b, lds_bytes, s, a);
      if (vec == 2) return launch_with_lds(k_obs_planes<2, false>, g, b, lds_bytes, s, a);
      return launch_with_lds(k_obs_planes<1, false>, g, b, lds_bytes, s, a);
    }
  }
  if (layout == MDR_OBS_ROWS && circular && a.c == 10 && a.F == 51 && a.defect_prob <= 0.0f) {
    constexpr int RT = 256;   // 256 * 51 = 13056 < 13107: the kernel's mul-shift division is exact
    const size_t lds_bytes = (RT * (11 + 10) + (RT + (RT / 11) * 10) * 4) * sizeof(float);   // N >= 11: at most RT / 11 env segments
    const int64_t tiles = obs_tile_count<RT>(a.E, a.N);
    if (tiles < (int64_t)1 << 31)
      return launch_with_lds(k_obs_rows_default<RT>, dim3((unsigned)tiles), dim3(RT), lds_bytes, s, a);
  }
  if (layout == MDR_OBS_ROWS && circular && a.c <= 32 && (a.c + 1) * (int)nf < 65536) {
    constexpr int RT = 256;
    const size_t own = (size_t)(a.F - a.c * (int)nf) | 1;
    a.lds_entries = a.N < RT ? (RT / a.N) * (a.N + a.c) : RT + a.c;   // senders staged per tile: whole small envs, or a chunk of a big one
    const size_t lds_bytes = (RT * own + (size_t)a.lds_entries * nf + 3 * RT + a.F) * sizeof(float);
    const int64_t tiles = obs_tile_count<RT>(a.E, a.N);
    if (lds_bytes <= lds_cap && tiles < (int64_t)1 << 31)
      return launch_with_lds(k_obs_rows<RT>, dim3((unsigned)tiles), dim3(RT), lds_bytes, s, a);
  }
  {
    constexpr int TILE = 64;
    const size_t chunk = (layout == MDR_OBS_ROWS) ? (size_t)TILE * (a.F | 1) : (size_t)a.F * TILE;
    const size_t lds_bytes = (chunk + nf * (TILE + a.c)) * sizeof(float);
    const int64_t tiles = a.E * (int64_t)((a.N + TILE - 1) / TILE);
    if (lds_bytes <= lds_cap && tiles < (int64_t)1 << 31) {
      const dim3 g((unsigned)tiles), b(TILE);
      if (a.random_links) {
        if (layout == MDR_OBS_ROWS) return launch_with_lds(k_obs_tiled<MDR_OBS_ROWS, TILE, true>, g, b, lds_bytes, s, a);
        return launch_with_lds(k_obs_tiled<MDR_OBS_PLANES, TILE, true>, g, b, lds_bytes, s, a);
      }
      if (layout == MDR_OBS_ROWS) return launch_with_lds(k_obs_tiled<MDR_OBS_ROWS, TILE, false>, g, b, lds_bytes, s, a);
      return launch_with_lds(k_obs_tiled<MDR_OBS_PLANES, TILE, false>, g, b, lds_bytes, s, a);
    }
  }
  const dim3 g((unsigned)((a.plane + 255) / 256)), b(256);
  if (a.random_links) {
    if (layout == MDR_OBS_PLANES) hipLaunchKernelGGL((k_obs_vector<MDR_OBS_PLANES, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_obs_vector<MDR_OBS_ROWS, true>), g, b, 0, s, a);
  } else {
    if (layout == MDR_OBS_PLANES) hipLaunchKernelGGL((k_obs_vector<MDR_OBS_PLANES, false>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_obs_vector<MDR_OBS_ROWS, false>), g, b, 0, s, a);
  }
  return hipGetLastError();
}

// =================================================================================================
// Launchers (host)
// =================================================================================================
hipError_t launch_sample(const EpisodeArgs& a, hipStream_t s) {
  const int64_t n = (int64_t)a.E * a.N;
  hipLaunchKernelGGL(k_sample_houses, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_sample_envs, dim3((unsigned)((a.E + 255) / 256)), dim3(256), 0, s, a);
  return launch_max_power(a, s);
}

hipError_t launch_load(const EpisodeArgs& a, const mdr_episode_t& ep, hipStream_t s) {
  const int64_t n = (int64_t)a.E * a.N;
  hipLaunchKernelGGL(k_load_houses, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, ep);
  hipLaunchKernelGGL(k_load_envs, dim3((unsigned)((a.E + 255) / 256)), dim3(256), 0, s, a, ep);
  return launch_max_power(a, s);
}

hipError_t launch_interp_base(const InterpArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_interp_base, dim3((unsigned)a.E), dim3(128), 0, s, a);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_signal_error(StepArgs a, double* acc) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.E) return;
  const double d = a.sig_old[e] - a.P[e];
  acc[e] += d * d;
}

hipError_t launch_signal_error(const StepArgs& a, double* acc, hipStream_t s) {
  hipLaunchKernelGGL(k_signal_error, dim3((unsigned)((a.E + 255) / 256)), dim3(256), 0, s, a, acc);
  return hipGetLastError();
}

// Stepwise stand-in for the accumulators of the fused rollout kernels (shapes they do not cover: N > 2048, or N > 512
// with N % 4 != 0): run after each single step; a.sig_old is the row of the NEW time index, i.e. the step's new signal.
__global__ __launch_bounds__(256) void k_rollout_accumulate(StepArgs a, RolloutArgs ro) {
  __shared__ double lds[3 * 4];
  const int e = blockIdx.x;
  const int64_t base = (int64_t)e * a.N;
  double terr = 0.0;
  for (int h = threadIdx.x; h < a.N; h += 256) {
    const int64_t i = base + h;
    if (ro.reward_sum) ro.reward_sum[i] = ro.reward_sum[i] + a.reward[i];
    const float d = a.Ta[i] - a.target[i];
    terr += (double)(d * d);
  }
  Red3 r{terr, 0.0, 0.0f};
  r = block_reduce<256>(r, lds);
  if (threadIdx.x == 0) {
    if (ro.sq_temp_error_sum) ro.sq_temp_error_sum[e] += r.sum_p;
    const double d = a.sig_old[e] - a.P[e];
    if (ro.sq_signal_error_sum) ro.sq_signal_error_sum[e] += d * d;
    if (ro.power_trace) ro.power_trace[e] = a.P[e];
  }
}

hipError_t launch_rollout_accumulate(const StepArgs& a, const RolloutArgs& ro, hipStream_t s) {
  hipLaunchKernelGGL(k_rollout_accumulate, dim3((unsigned)a.E), dim3(256), 0, s, a, ro);
  return hipGetLastError();
}

hipError_t launch_patch_signal_plane(const StepArgs& a, hipStream_t s) {
  if (a.obs == nullptr) return hipSuccess;
  hipLaunchKernelGGL(k_patch_signal_plane, dim3((unsigned)((a.plane + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_reset_obs(const StepArgs& a, bool zero_reward, hipStream_t s) {
  hipLaunchKernelGGL(k_reset_obs, dim3((unsigned)((a.plane + 255) / 256)), dim3(256), 0, s, a, zero_reward ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_tables(const TableArgs& a, hipStream_t s) {
  const int64_t n = (int64_t)a.rows * a.E;
  // Runs of rows per thread once the batch has envs to spare: as long as a run per thread still puts ~256 K threads on the device
  // (MDR_TABLE_RUN: 0 = always one thread per entry, n = runs of n rows whatever the batch)
  static const int knob = [] { const char* t = getenv("MDR_TABLE_RUN"); return t ? atoi(t) : -1; }();
  int chunk = 1;
  if (a.perlin_octaves <= TABLE_RUN_OCTAVES || a.signal_mode != MDR_SIGNAL_PERLIN) {
    if (knob >= 0) chunk = knob;
    else if (n >= (int64_t)1 << 19) chunk = (int)std::min<int64_t>(a.rows, n >> 18);
  }
  // A batch of many envs: a workgroup per 64 envs, calendar per minute and Perlin gradient once (MDR_TABLE_TILE=0: the runs kernel)
  static const int tile_knob = [] { const char* t = getenv("MDR_TABLE_TILE"); return t ? atoi(t) : 1; }();
  const bool perlin = a.signal_mode == MDR_SIGNAL_PERLIN;
  if (chunk > 1 && tile_knob && a.E >= 4096 && a.dt > 0 &&
      (!perlin || (a.perlin_octaves <= TABLE_RUN_OCTAVES && a.perlin_period > 0.0 && a.perlin_step > 0.0))) {
    TableTile tl{};
    const double window = (double)(a.rows - 1) * (double)a.dt;
    const double minutes = floor(window / 60.0) + 2.0;
    bool fits = minutes <= (double)TABLE_TILE_MAX_MINUTES;
    for (int q = 0; perlin && q < a.perlin_octaves && fits; ++q) {
      const double span = window / a.perlin_period * std::ldexp(a.perlin_step, q);
      fits = span < (double)TABLE_TILE_MAX_SLOTS;
      if (fits) tl.base[q + 1] = tl.base[q] + (int)std::ceil(span) + 2;
    }
    fits = fits && (!perlin || tl.base[a.perlin_octaves] <= TABLE_TILE_MAX_SLOTS);
    if (fits) {
      tl.minutes = (int)minutes;
      const int slots = perlin ? tl.base[a.perlin_octaves] : 0;
      const size_t lds = (size_t)slots * TABLE_TILE_ENVS * sizeof(uint32_t);
      // (beyond 64 KB of dynamic LDS the limit is raised per launch: the attribute belongs to the current device)
      const bool raised = lds <= 65536 || hipFuncSetAttribute((const void*)k_fill_tables_tile, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                TABLE_TILE_MAX_SLOTS * TABLE_TILE_ENVS * (int)sizeof(uint32_t)) == hipSuccess;
      if (raised) {
        hipLaunchKernelGGL(k_fill_tables_tile, dim3((unsigned)((a.E + TABLE_TILE_ENVS - 1) / TABLE_TILE_ENVS)), dim3(256), lds, s, a, tl);
        return hipGetLastError();
      }
    }
  }
  if (chunk > 1) {
    const int64_t threads = (int64_t)((a.rows + chunk - 1) / chunk) * a.E;
    hipLaunchKernelGGL(k_fill_tables_runs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, chunk);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_fill_tables, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

template <int VEC, int THREADS>
static void launch_fused_tiles(const StepArgs& a, int tiles, hipStream_t s) {
  const dim3 g((unsigned)a.E), b(THREADS);
  switch (tiles) {
    case 1: hipLaunchKernelGGL((k_step_fused<VEC, 1, THREADS>), g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_step_fused<VEC, 2, THREADS>), g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_step_fused<VEC, 3, THREADS>), g, b, 0, s, a); break;
    default: hipLaunchKernelGGL((k_step_fused<VEC, 4, THREADS>), g, b, 0, s, a); break;
  }
}

static int pow2_at_least(int n) {
  int g = 1;
  while (g < n) g <<= 1;
  return g;
}

StepPlan plan_step(int N, int64_t E) {
  StepPlan p{};
  int vec = (N % 4 == 0) ? 4 : ((N % 2 == 0) ? 2 : 1);
  // Small batches (less than ~16 wavefronts per CU even at one house per lane) are latency-bound: one house per lane
  // exposes the most parallelism; wide accesses only pay once the device is full.
  if (N <= 64 && E * N < 262144) vec = 1;
  const int lanes = (N + vec - 1) / vec;
  // Device-filling batches of small envs with N % 4 == 2 (the reference deploys with 50 houses): TWO envs per lane group keep every
  // lane on 16-byte accesses where one env per group allows 8 bytes (mdr_multi.hip; measured at 4.19 M houses: 50 houses 66.9 ->
  // 63.9 us, 10 houses 68.7 -> 65.9 us.  Three or four envs per group - 20 houses at 15 of 16 lanes, odd N - were measured too and
  // lose: the per-env reductions and selects outweigh the fuller lanes, e.g. 20 houses 64.1 -> 69.6 us, 27 houses 70.1 -> 98.6 us)
  static const bool multi_ok = [] { const char* t = getenv("MDR_PLAN_MULTI"); return !(t && t[0] == '0'); }();
  if (multi_ok && N % 4 == 2 && N >= 6 && N <= 126 && E >= 2 && E * N >= 262144) {
    p.kind = STEP_MULTI;
    p.vec = 4;
    p.threads = pow2_at_least(N / 2);   // lanes per group of two envs
    p.tiles = 2;                        // envs per group
    return p;
  }
  // Device-filling batches with N % 4 == 0 whose N / 4 lanes are not a power of two: whole envs packed into the wavefront
  // (k_step_packed) where rounding each env up to a power of two would leave more than 35 % of the lanes idle - 36 houses: 9 lanes
  // of 16 -> 63 of 64, 67.9 -> 64.4 us at 4.19 M houses.  20 and 40 houses (5 of 8, 10 of 16 lanes) gain nothing as a STEP - it is
  // bound by its memory instructions per wave, not by idle lanes - but their multi-step rollouts, which share the step's lane mapping
  // bit for bit and are vector-bound, run 11-12 % faster packed (20 houses: 14.2 -> 12.7 us per step of 4.19 M houses); 12 and 24
  // houses (3 of 4, 6 of 8 lanes) lose 2-3 % as steps and stay unpacked.
  static const bool packed_ok = [] { const char* t = getenv("MDR_PLAN_PACKED"); return !(t && t[0] == '0'); }();
  if (packed_ok && N % 4 == 0 && N >= 12 && N <= 128 && E * N >= 262144) {
    const int L = N / 4, per_wave = 64 / L;
    static const int fill_pm = [] { const char* t = getenv("MDR_PLAN_PACKED_FILL"); return t ? atoi(t) : 650; }();   // experiment knob: pack below this lane fill (per mille)
    if ((L & (L - 1)) != 0 && 1000 * L < fill_pm * pow2_at_least(L) && per_wave * pow2_at_least(L) > 64) {
      p.kind = STEP_PACKED;
      p.vec = 4;
      p.threads = L;
      p.tiles = per_wave;
      return p;
    }
  }
  if (N == 1 && E % 4 == 0 && E >= 262144) {   // single-house envs in bulk: vectorise over the env axis
    p.kind = STEP_SINGLE;
    p.vec = 4;
    p.threads = 256;
    p.tiles = 1;
  } else if (lanes <= 32 || (vec < 4 && lanes <= 64)) {   // at least two envs per wavefront, or no wider form exists
    p.kind = STEP_GROUP;
    p.vec = vec;
    p.threads = pow2_at_least(lanes);   // lanes per env
    p.tiles = 1;
  } else if (N % 4 == 0 && N <= 4096) {
    p.kind = STEP_FUSED;
    p.vec = 4;
    p.threads = N <= 256 ? 64 : (N <= 512 ? 128 : 256);
    if (const char* t = getenv("MDR_PLAN_THREADS")) {   // experiment knob: workgroup size of the fused kernel (64/128/256)
      const int v = atoi(t);
      if ((v == 64 || v == 128 || v == 256) && (N + v * 4 - 1) / (v * 4) <= 4) p.threads = v;
    }
    p.tiles = (N + p.threads * 4 - 1) / (p.threads * 4);
  } else if (N <= 1024) {
    p.kind = STEP_FUSED;
    p.vec = 1;
    p.threads = 256;
    p.tiles = (N + 255) / 256;
  } else {
    p.kind = STEP_SPLIT;
    p.vec = (N % 4 == 0) ? 4 : 1;
    p.threads = 256;
    p.tiles = 1;
  }
  return p;
}

// The multi-step kernels keep one house per lane (sub-wave groups) or VEC x TILES houses per thread (workgroup per env)
StepPlan plan_rollout(int N, int64_t E) {
  StepPlan p = plan_step(N, E);
  if (p.kind == STEP_GROUP || p.kind == STEP_PACKED) return p;   // same lane mapping as the single-step kernel
  if (p.kind == STEP_MULTI) {           // k_step_multi forms the totals of one env per group, a pair of houses per lane (mdr_multi.hip)
    p.kind = STEP_GROUP;
    p.vec = 2;
    p.tiles = 1;                        // p.threads = pow2_at_least(N / 2) already
    return p;
  }
  if (p.kind == STEP_SINGLE) {          // one lane per env; nothing to reduce, so any mapping gives the same bits
    p.kind = STEP_GROUP;
    p.vec = 1;
    p.threads = 1;
    return p;
  }
  p = StepPlan{};
  if (N % 4 == 0 && N <= 2048) {
    p.kind = STEP_FUSED;
    p.vec = 4;
    p.threads = N <= 256 ? 64 : (N <= 512 ? 128 : 256);
    p.tiles = (N + p.threads * 4 - 1) / (p.threads * 4);
  } else if (N <= 512) {
    p.kind = STEP_FUSED;
    p.vec = 1;
    p.threads = 256;
    p.tiles = (N + 255) / 256;
  } else {
    p.kind = STEP_SPLIT;
  }
  return p;
}

// Workgroup size of the split path: 256 threads (1024 houses with 16-byte accesses; N % 4 != 0: one house per lane).
// MDR_SPLIT_THREADS=64 selects 256-house workgroups for small launches (experiment knob): measured SLOWER on a 125,000-house
// shard (10.4 us against 8.7 us for the two launches) - four times the records for every finish workgroup to re-sum buy
// nothing while both launches sit on the dispatch-latency floor.
int split_threads(int N, int64_t E) {
  if (N % 4 != 0) return 256;
  static const int knob = [] { const char* t = getenv("MDR_SPLIT_THREADS"); return t ? atoi(t) : 0; }();
  if (knob == 64 && E * (((int64_t)N + 1023) / 1024) < 512) return 64;
  return 256;
}

int64_t split_blocks(int N, int threads) {
  const int vec = (N % 4 == 0) ? 4 : 1;
  return ((int64_t)N + threads * vec - 1) / (threads * vec);
}

// `reduce`: also sum the records into tot_sum / tot_max (mdr_env_step_begin's contract); the two-launch forms leave that to
// the finish kernel.  a.nblk is the record stride of `partials`; the grid covers the split_blocks(N) workgroups that exist.
hipError_t launch_step_begin_split(const StepArgs& a, bool reduce, hipStream_t s) {
  const int th = split_threads(a.N, a.E);
  const dim3 g((unsigned)split_blocks(a.N, th), (unsigned)a.E), b(th);
  if (a.N % 4 != 0)
    hipLaunchKernelGGL((k_step_partial<1, 256>), g, b, 0, s, a);
  else if (th == 64)
    hipLaunchKernelGGL((k_step_partial<4, 64>), g, b, 0, s, a);
  else
    hipLaunchKernelGGL((k_step_partial<4, 256>), g, b, 0, s, a);
  if (reduce) hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)a.E), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_step_end_split(const StepArgs& a, hipStream_t s) {
  const int th = split_threads(a.N, a.E);
  const dim3 g((unsigned)split_blocks(a.N, th), (unsigned)a.E), b(th);
  if (a.N % 4 != 0)
    hipLaunchKernelGGL((k_step_finish<1, 256>), g, b, 0, s, a);
  else if (th == 64)
    hipLaunchKernelGGL((k_step_finish<4, 64>), g, b, 0, s, a);
  else
    hipLaunchKernelGGL((k_step_finish<4, 256>), g, b, 0, s, a);
  return hipGetLastError();
}

hipError_t launch_step_end_begin_split(const StepArgs& f, const StepArgs& p, hipStream_t s) {
  const int th = split_threads(p.N, p.E);
  const dim3 g((unsigned)split_blocks(p.N, th), (unsigned)p.E), b(th);
  if (p.N % 4 != 0)
    hipLaunchKernelGGL((k_step_finish_partial<1, 256>), g, b, 0, s, f, p);
  else if (th == 64)
    hipLaunchKernelGGL((k_step_finish_partial<4, 64>), g, b, 0, s, f, p);
  else
    hipLaunchKernelGGL((k_step_finish_partial<4, 256>), g, b, 0, s, f, p);
  return hipGetLastError();
}

bool rollout_fused_supported(const StepPlan& p) { return p.kind == STEP_GROUP || p.kind == STEP_FUSED || p.kind == STEP_PACKED; }

hipError_t launch_rollout_fused(const StepArgs& a, const RolloutArgs& r, const StepPlan& p, hipStream_t s) {
  if (!rollout_fused_supported(p)) return hipErrorInvalidValue;
  if (p.kind == STEP_PACKED) return launch_rollout_multi(a, r, p, s);
  const bool bb = a.action_source == MDR_ACTIONS_BANGBANG;
  if (p.kind == STEP_GROUP) {
    const int64_t lanes = (int64_t)a.E * p.threads;
    const dim3 gg((unsigned)((lanes + 255) / 256)), b(256);
    const bool win = lanes < (int64_t)64 * 8 * 256;   // fewer than ~8 wavefronts per CU: latency-bound
#define MDR_ROLLGV(G, V)                                                                    \
  if (bb) {                                                                                 \
    if (win) hipLaunchKernelGGL((k_rollout_group<G, V, true, true>), gg, b, 0, s, a, r);    \
    else hipLaunchKernelGGL((k_rollout_group<G, V, false, true>), gg, b, 0, s, a, r);       \
  } else {                                                                                  \
    if (win) hipLaunchKernelGGL((k_rollout_group<G, V, true, false>), gg, b, 0, s, a, r);   \
    else hipLaunchKernelGGL((k_rollout_group<G, V, false, false>), gg, b, 0, s, a, r);      \
  }
#define MDR_ROLLG(G)                      \
  if (p.vec == 4) { MDR_ROLLGV(G, 4) }    \
  else if (p.vec == 2) { MDR_ROLLGV(G, 2) } \
  else { MDR_ROLLGV(G, 1) }               \
  break;
    switch (p.threads) {
      case 1: MDR_ROLLG(1)
      case 2: MDR_ROLLG(2)
      case 4: MDR_ROLLG(4)
      case 8: MDR_ROLLG(8)
      case 16: MDR_ROLLG(16)
      case 32: MDR_ROLLG(32)
      default: MDR_ROLLG(64)
    }
#undef MDR_ROLLG
#undef MDR_ROLLGV
    return hipGetLastError();
  }
  const dim3 g((unsigned)a.E);
  const bool win = (int64_t)a.E * p.threads < (int64_t)64 * 8 * 256;
#define MDR_ROLL(V, T, TH)                                                                  \
  do {                                                                                                  \
    if (bb) {                                                                                           \
      if (win) hipLaunchKernelGGL((k_rollout_fused<V, T, TH, true, true>), g, dim3(TH), 0, s, a, r);    \
      else hipLaunchKernelGGL((k_rollout_fused<V, T, TH, false, true>), g, dim3(TH), 0, s, a, r);       \
    } else {                                                                                            \
      if (win) hipLaunchKernelGGL((k_rollout_fused<V, T, TH, true, false>), g, dim3(TH), 0, s, a, r);   \
      else hipLaunchKernelGGL((k_rollout_fused<V, T, TH, false, false>), g, dim3(TH), 0, s, a, r);      \
    }                                                                                                   \
  } while (0)
  if (p.vec == 4) {
    if (p.threads == 64) { if (p.tiles == 1) MDR_ROLL(4, 1, 64); else MDR_ROLL(4, 2, 64); }
    else if (p.threads == 128) { if (p.tiles == 1) MDR_ROLL(4, 1, 128); else MDR_ROLL(4, 2, 128); }
    else { if (p.tiles == 1) MDR_ROLL(4, 1, 256); else MDR_ROLL(4, 2, 256); }
  } else {
    if (p.tiles == 1) MDR_ROLL(1, 1, 256); else MDR_ROLL(1, 2, 256);
  }
#undef MDR_ROLL
  return hipGetLastError();
}

// Graph mode, one-kernel steps: the last workgroup moves the cursor on (cursor_done) while the grid is small - the arrivals are
// same-address device-scope atomics, ~9 ns apiece and serialised, so a big grid pays more for them than the one-thread launch
// they replace costs (measured: +8.7 us at 977 workgroups, +0.3 us at 123).
static int cursor_atomic_blocks() {
  static const int v = [] {
    const char* e = getenv("MDR_CURSOR_ATOMIC_BLOCKS");
    return e ? atoi(e) : 128;
  }();
  return v;
}

static int64_t step_blocks(const StepArgs& a, const StepPlan& p) {
  if (p.kind == STEP_SINGLE) return (a.E / 4 + 255) / 256;
  if (p.kind == STEP_FUSED) return a.E;
  if (p.kind == STEP_MULTI || p.kind == STEP_PACKED) return multi_blocks(a.E, p);
  return ((int64_t)a.E * p.threads + 255) / 256;
}

hipError_t launch_step(const StepArgs& args, const StepPlan& p, hipStream_t s) {
  StepArgs a = args;
  if (p.kind != STEP_SPLIT && a.cursor_adv != nullptr && step_blocks(a, p) > cursor_atomic_blocks()) {
    a.cursor_adv = nullptr;
    const hipError_t err = launch_step(a, p, s);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_cursor_advance, dim3(1), dim3(1), 0, s, args.cursor_adv);
    return hipGetLastError();
  }
  if (p.kind == STEP_MULTI || p.kind == STEP_PACKED) return launch_step_multi(a, p, s);
  if (p.kind == STEP_SINGLE) {
    hipLaunchKernelGGL(k_step_single_house, dim3((unsigned)((a.E / 4 + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  if (p.kind == STEP_FUSED) {
    if (p.vec == 4) {
      if (p.threads == 64) launch_fused_tiles<4, 64>(a, p.tiles, s);
      else if (p.threads == 128) launch_fused_tiles<4, 128>(a, p.tiles, s);
      else launch_fused_tiles<4, 256>(a, p.tiles, s);
    } else {
      launch_fused_tiles<1, 256>(a, p.tiles, s);
    }
    return hipGetLastError();
  }
  if (p.kind == STEP_GROUP) {
    const int64_t lanes = (int64_t)a.E * p.threads;
    const dim3 g((unsigned)((lanes + 255) / 256)), b(256);
#define MDR_GROUP(G)                                                                   \
  if (p.vec == 4) hipLaunchKernelGGL((k_step_group<G, 4>), g, b, 0, s, a);             \
  else if (p.vec == 2) hipLaunchKernelGGL((k_step_group<G, 2>), g, b, 0, s, a);        \
  else hipLaunchKernelGGL((k_step_group<G, 1>), g, b, 0, s, a);                        \
  break;
    switch (p.threads) {
      case 1: MDR_GROUP(1)
      case 2: MDR_GROUP(2)
      case 4: MDR_GROUP(4)
      case 8: MDR_GROUP(8)
      case 16: MDR_GROUP(16)
      case 32: MDR_GROUP(32)
      default: MDR_GROUP(64)
    }
#undef MDR_GROUP
    return hipGetLastError();
  }
  // split path on one device: partial records, then the finish kernel re-sums them itself (two launches)
  hipError_t err = launch_step_begin_split(a, false, s);
  if (err != hipSuccess) return err;
  StepArgs f = a;
  f.records = a.partials;
  f.world = 1;
  return launch_step_end_split(f, s);
}

}  // namespace mdr
