/*
 * A torch-free client of the C ABI (include/mdr.h): plain C + the HIP runtime for memory only.
 *
 *   hipcc -x c tools/c_abi_client.c -Iinclude -Lmarl-demandresponse-original_amd/csrc -lmdr_hip \
 *         -Wl,-rpath,$PWD/marl-demandresponse-original_amd/csrc -o /tmp/c_abi_client
 *   /tmp/c_abi_client <nb_envs> <nb_houses> <seed> <steps> [records]
 *
 * Allocates every buffer with hipMalloc, samples an episode on the device, takes `steps` bang-bang steps and
 * prints checksums of the final state.  With `records` the steps go through the sharded-houses entry points as a rank
 * of a world of one would drive them - mdr_env_step_begin_records, then per step a device copy of the partial records
 * (what the all-gather delivers) and mdr_env_step_end_begin_records (or, where the library asks for it, the separate
 * end and begin), mdr_env_step_end_records at the end - and must land on the same numbers.  tests/test_gpu_c_client.py runs it and compares the numbers with the
 * Python host's run of the same configuration and seed: the boundary carries no Python / torch state.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mdr.h"

#define CHECK_HIP(x)                                                             \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
      return 2;                                                                  \
    }                                                                            \
  } while (0)
#define CHECK_MDR(x)                                                             \
  do {                                                                           \
    int rc_ = (x);                                                               \
    if (rc_ != MDR_OK) {                                                         \
      fprintf(stderr, "%s: %s - %s\n", #x, mdr_status_string(rc_), mdr_last_error(env)); \
      return 3;                                                                  \
    }                                                                            \
  } while (0)

static void *dev_alloc(size_t bytes) {
  void *p = NULL;
  if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return NULL;
  hipMemset(p, 0, bytes ? bytes : 16);
  return p;
}

int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 4;
  const int N = argc > 2 ? atoi(argv[2]) : 64;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 10) : 7;
  const int steps = argc > 4 ? atoi(argv[4]) : 100;
  const int use_records = argc > 5 && strcmp(argv[5], "records") == 0;

  /* the reference's defaults (config.py), with noise_house_prop "big_noise", noise_hvac_prop "big_noise",
   * signal_mode "sinusoidals", base_power_mode "constant" - the same dict tests/test_gpu_c_client.py builds */
  mdr_config_t c;
  memset(&c, 0, sizeof c);
  c.struct_size = sizeof c;
  c.nb_envs = E; c.nb_houses = N; c.nb_houses_total = N;
  c.time_step = 4; c.table_steps = 64; c.temp_ref = 20.0;
  c.init_air_temp = 20; c.init_mass_temp = 20; c.target_temp = 20; c.deadband = 0;
  c.Ua = 2.18e02; c.Cm = 3.45e06; c.Ca = 9.08e05; c.Hm = 2.84e03;
  c.window_area = 7.175; c.shading_coeff = 0.67; c.solar_gain = 1;
  c.lockout_duration = 40; c.lockout_noise = 0; c.COP = 2.5; c.cooling_capacity = 15000; c.latent_cooling_fraction = 0.35;
  c.std_start_temp = 5; c.std_target_temp = 2; c.factor_thermo_low = 0.8; c.factor_thermo_high = 1.2;
  c.nb_capacities = 5;
  { const double caps[5] = {10000, 12500, 15000, 17500, 20000}; memcpy(c.capacity_list, caps, sizeof caps); }
  c.start_random = 1; c.start_epoch = 1609459200; /* 2021-01-01 00:00:00 */
  c.day_temp = 34; c.night_temp = 28; c.temp_std = 0.5; c.random_phase_offset = 0;
  c.signal_mode = MDR_SIGNAL_SINUSOIDALS; c.avg_power_per_hvac = 4200;
  c.nb_sinusoids = 2; c.sin_periods[0] = 400; c.sin_periods[1] = 1200; c.sin_amplitude_ratios[0] = 0.1; c.sin_amplitude_ratios[1] = 0.3;
  c.artificial_ratio = 1.0; c.artificial_signal_ratio_range = 1.0;
  c.alpha_temp = 1; c.alpha_sig = 1; c.norm_temp_penalty = 1.0; c.norm_sig_penalty = 3515625.0;
  c.penalty_mode = MDR_PENALTY_INDIVIDUAL_L2; c.base_power_mode = 0;
  c.mix_ind_L2 = 1; c.mix_common_L2 = 1; c.mix_common_max = 0;
  c.obs_power_norm = 7500.0 * N;

  mdr_env_t *env = NULL;
  int rc = mdr_env_create(&c, &env);
  if (rc != MDR_OK) {
    fprintf(stderr, "mdr_env_create: %s - %s\n", mdr_status_string(rc), mdr_last_error(env));
    return 3;
  }

  const size_t H = (size_t)E * N, K1 = (size_t)c.table_steps + 1;
  mdr_buffers_t b;
  memset(&b, 0, sizeof b);
  b.struct_size = sizeof b;
#define F32(name) b.name = (float *)dev_alloc(H * 4)
  F32(Ta); F32(Tm); F32(k01); F32(s0); F32(k10); F32(s1); F32(inv_Ua); F32(Q_hvac); F32(P_max); F32(target); F32(deadband);
  F32(Ua); F32(Cm); F32(Ca); F32(Hm); F32(capacity); F32(COP); F32(latent); F32(reward);
#undef F32
  b.sso = (int32_t *)dev_alloc(H * 4); b.lockout = (int32_t *)dev_alloc(H * 4); b.flags = (uint8_t *)dev_alloc(H);
  b.obs = (float *)dev_alloc(H * 4 * MDR_OBS_COLUMNS);
  b.t0 = (int64_t *)dev_alloc((size_t)E * 8);
  b.phase = (double *)dev_alloc((size_t)E * 8); b.ratio = (double *)dev_alloc((size_t)E * 8);
  b.max_power = (double *)dev_alloc((size_t)E * 8); b.P = (double *)dev_alloc((size_t)E * 8);
  b.base_power = (double *)dev_alloc((size_t)E * 8);
  b.tot_sum = (double *)dev_alloc((size_t)E * 8 * 3); b.tot_max = b.tot_sum + 2 * (size_t)E;
  b.tab_od = (float *)dev_alloc(K1 * E * 4); b.tab_solar = (float *)dev_alloc(K1 * E * 4); b.tab_signal = (double *)dev_alloc(K1 * E * 8);
  b.partials = (double *)dev_alloc((size_t)E * (size_t)mdr_partials_per_env(N) * 3 * 8);
  b.pen_stash = (float *)dev_alloc(H * 4);   /* optional: lets rollouts of the split path (and the records path) take one launch per step */
  uint8_t *actions = (uint8_t *)dev_alloc(H);

  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_MDR(mdr_env_bind(env, &b));
  CHECK_MDR(mdr_env_reset(env, seed, 0, stream));
  CHECK_MDR(mdr_env_begin_episode(env, stream));
  int fused = 0;
  if (use_records && steps > 0) {
    const int32_t R = (int32_t)mdr_env_partial_records(env);
    const size_t bytes = (size_t)E * (size_t)R * 3 * sizeof(double);
    double *records = (double *)dev_alloc(bytes);   /* [world = 1][E][R][3] */
    CHECK_MDR(mdr_env_step_begin_records(env, actions, MDR_ACTIONS_BANGBANG, R, stream));
    for (int i = 1; i < steps; ++i) {
      CHECK_HIP(hipMemcpyAsync(records, b.partials, bytes, hipMemcpyDeviceToDevice, stream));
      rc = mdr_env_step_end_begin_records(env, records, 1, actions, MDR_ACTIONS_BANGBANG, stream);
      if (rc == MDR_ERR_UNSUPPORTED) {   /* the next step leaves the time tables: the refill is host work */
        CHECK_MDR(mdr_env_step_end_records(env, records, 1, stream));
        CHECK_MDR(mdr_env_step_begin_records(env, actions, MDR_ACTIONS_BANGBANG, R, stream));
      } else {
        CHECK_MDR(rc);
        ++fused;
      }
    }
    CHECK_HIP(hipMemcpyAsync(records, b.partials, bytes, hipMemcpyDeviceToDevice, stream));
    CHECK_MDR(mdr_env_step_end_records(env, records, 1, stream));
  } else {
    CHECK_MDR(mdr_env_rollout(env, actions, MDR_ACTIONS_BANGBANG, steps, stream));
  }
  CHECK_HIP(hipStreamSynchronize(stream));

  float *Ta = (float *)malloc(H * 4), *reward = (float *)malloc(H * 4);
  int32_t *sso = (int32_t *)malloc(H * 4);
  double *P = (double *)malloc((size_t)E * 8);
  CHECK_HIP(hipMemcpy(Ta, b.Ta, H * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(reward, b.reward, H * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(sso, b.sso, H * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(P, b.P, (size_t)E * 8, hipMemcpyDeviceToHost));
  double sTa = 0, sR = 0, sP = 0;
  long long sS = 0;
  for (size_t i = 0; i < H; ++i) { sTa += Ta[i]; sR += reward[i]; sS += sso[i]; }
  for (int e = 0; e < E; ++e) sP += P[e];
  int64_t k = 0, j0 = 0;
  mdr_env_cursor(env, &k, &j0);
  printf("{\"fused\": %d, \"steps\": %lld, \"sum_Ta\": %.9e, \"sum_reward\": %.9e, \"sum_sso\": %lld, \"sum_P\": %.9e, \"Ta0\": %.9e, \"TaLast\": %.9e}\n",
         fused, (long long)k, sTa, sR, sS, sP, (double)Ta[0], (double)Ta[H - 1]);
  mdr_env_destroy(env);
  return 0;
}
