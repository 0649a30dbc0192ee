// Several small envs per lane group (STEP_MULTI): the reference's own cluster sizes - 20 houses in training, 50 in deployment
// (cli.py:53, 629) - in batches that fill the device.  env/MA_DemandResponse.py:1005-1055 (ClusterHouses.step), 234-373 (rewards).
#include <cstdlib>

#include "mdr_device.h"
#include "mdr_kernels.h"
#include "mdr_step_common.h"

namespace mdr {

// ---- (2c) several small envs per lane group, four FLAT houses per lane.  The reference trains with 20 houses and deploys with
// 50 (cli.py:53, 629).  One env per lane group leaves lanes idle (20 houses = 5 lanes of 8) or - N % 4 != 0 - forces narrow
// accesses (50 houses: rows start every 200 bytes, so 8-byte accesses at most).  Here ENVS consecutive envs share a group of GROUP
// lanes, ENVS * N a multiple of 4: lane l holds the four houses 4 l .. 4 l + 3 of the group's ENVS * N, always on a 16-byte
// boundary of the [E][N] arrays and in at most two envs (N >= 4).  20 houses: 3 envs = 15 lanes of 16; 50 houses: 2 envs = 25
// lanes of 32 with 16-byte accesses.  (Only ENVS = 2 is instantiated: three and four envs per group measured slower than one, see plan_step.)  Every env's totals come out of the same sub-wave DPP tree, once per env of the group (a
// lane contributes zeros to the envs it holds nothing of), so k_step_multi and k_rollout_multi end bit for bit alike.
template <int ENVS>
struct MultiLane {
  int nv;           // houses of this lane that exist (4, or fewer at the very end of the batch; 0: idle lane)
  int q[4];         // env of each house within the group
  int64_t i;        // flat index of the first house
  int64_t ea, eb;   // the (at most two) envs the lane's houses belong to: first and last house's
  __device__ __forceinline__ void init(const StepArgs& a, int64_t group, int lane) {
    const int k0 = lane * 4;
    const int64_t e0 = group * ENVS;
    i = e0 * a.N + k0;
    const int64_t total = (int64_t)a.E * a.N;
    nv = (k0 < ENVS * a.N && i < total) ? (int)(total - i < 4 ? total - i : 4) : 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = k0 + v;
      q[v] = (k >= a.N ? 1 : 0) + (ENVS > 2 && k >= 2 * a.N ? 1 : 0) + (ENVS > 3 && k >= 3 * a.N ? 1 : 0);
    }
    const int64_t last = (int64_t)a.E - 1;
    ea = e0 + q[0] < last ? e0 + q[0] : last;
    eb = e0 + q[3] < last ? e0 + q[3] : last;
  }
};

template <int ENVS>
__device__ __forceinline__ Red3 pick_env(const Red3* t, int q) {
  Red3 r = t[0];
#pragma unroll
  for (int k = 1; k < ENVS; ++k)
    if (q == k) r = t[k];
  return r;
}

template <int GROUP, int ENVS>
__global__ __launch_bounds__(256) void k_step_multi(StepArgs a) {
  rebase(a);
  const bool need_pen = a.penalty_mode != MDR_PENALTY_INDIVIDUAL_L2;
  const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) / GROUP;
  const int lane = threadIdx.x % GROUP;
  MultiLane<ENVS> m;
  m.init(a, group, lane);
  HouseOut o[4];
  int lockout[4];
  Red3 acc[ENVS];
#pragma unroll
  for (int k = 0; k < ENVS; ++k) acc[k] = Red3{0.0, 0.0, 0.0f};
  if (m.nv > 0) {
    const float od_a = a.od_old[m.ea], od_b = a.od_old[m.eb], so_a = a.solar_new[m.ea], so_b = a.solar_new[m.eb];
    float od[4], so[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      od[v] = m.q[v] == m.q[0] ? od_a : od_b;
      so[v] = m.q[v] == m.q[0] ? so_a : so_b;
    }
    if (m.nv == 4) {
      step_vec_rows<4>(a, m.i, od, so, o, lockout);
      store_obs_local<4>(a, m.i, o, lockout);
    } else {   // the last lane of the whole batch when E * N is not a multiple of 4
#pragma unroll
      for (int v = 0; v < 3; ++v)
        if (v < m.nv) {
          step_vec<1>(a, m.i + v, od[v], so[v], o + v, lockout + v);
          store_obs_local<1>(a, m.i + v, o + v, lockout + v);
        }
    }
#pragma unroll
    for (int k = 0; k < ENVS; ++k) {
      float p = 0.0f, ps = 0.0f;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (v < m.nv && m.q[v] == k) {
          p += o[v].power;
          ps += o[v].pen;
          acc[k].max_pen = fmaxf(acc[k].max_pen, o[v].pen);
        }
      acc[k].sum_p = (double)p;
      acc[k].sum_pen = (double)ps;
    }
  }
  Red3 tot[ENVS];
#pragma unroll
  for (int k = 0; k < ENVS; ++k) tot[k] = lanes_reduce<GROUP>(acc[k], need_pen);
  if (m.nv > 0) {
    const Red3 ta = pick_env<ENVS>(tot, m.q[0]), tb = pick_env<ENVS>(tot, m.q[3]);
    const float st_a = signal_term(a, ta.sum_p, a.sig_old[m.ea]), st_b = signal_term(a, tb.sum_p, a.sig_old[m.eb]);
    const float os_a = (float)(a.sig_new[m.ea] * a.inv_obs_norm), os_b = (float)(a.sig_new[m.eb] * a.inv_obs_norm);
    const float op_a = (float)(ta.sum_p * a.inv_obs_norm), op_b = (float)(tb.sum_p * a.inv_obs_norm);
    float r[4], c5[4], c6[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bool first = m.q[v] == m.q[0];
      r[v] = first ? reward_value(a, o[v].pen, ta.sum_pen, ta.max_pen, st_a) : reward_value(a, o[v].pen, tb.sum_pen, tb.max_pen, st_b);
      c5[v] = first ? os_a : os_b;
      c6[v] = first ? op_a : op_b;
    }
    if (m.nv == 4) {
      store_out<4>(a.reward, m.i, r);
      if (a.obs != nullptr) {
        store_out<4>(a.obs + 5 * a.plane, m.i, c5);
        store_out<4>(a.obs + 6 * a.plane, m.i, c6);
      }
    } else {
#pragma unroll
      for (int v = 0; v < 3; ++v)
        if (v < m.nv) {
          store_out<1>(a.reward, m.i + v, r + v);
          if (a.obs != nullptr) {
            store_out<1>(a.obs + 5 * a.plane, m.i + v, c5 + v);
            store_out<1>(a.obs + 6 * a.plane, m.i + v, c6 + v);
          }
        }
    }
  }
  if (lane < ENVS && group * ENVS + lane < a.E) a.P[group * ENVS + lane] = pick_env<ENVS>(tot, lane).sum_p;
  cursor_done(a);
}

// The closed loop on the same mapping: houses in registers for nsteps steps (k_rollout_group's counterpart).
template <int GROUP, int ENVS, bool BB>   // BB: the bang-bang rule compiled in (mdr_kernels.hip k_rollout_fused)
__global__ __launch_bounds__(256) void k_rollout_multi(StepArgs a, RolloutArgs ro) {
  const bool need_pen = a.penalty_mode != MDR_PENALTY_INDIVIDUAL_L2;
  const bool want_rsum = ro.reward_sum != nullptr;
  const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) / GROUP;
  const int lane = threadIdx.x % GROUP;
  MultiLane<ENVS> m;
  m.init(a, group, lane);
  HouseIn hs[4];
  HouseOut o[4];
  float rsum[4];
  unsigned act[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    hs[v] = HouseIn{};
    hs[v].lockout = 1;
    o[v] = HouseOut{};
    rsum[v] = 0.0f;
    act[v] = 0u;
  }
  if (m.nv == 4) {
    float Ta[4], Tm[4], k01[4], s0[4], k10[4], s1[4], iu[4], q[4], pm[4], tg[4], db[4];
    int sso[4], lk[4];
    unsigned fl[4];
    load_vec<4>(a.Ta, m.i, Ta);
    load_vec<4>(a.Tm, m.i, Tm);
    load_vec<4>(a.sso, m.i, sso);
    load_bytes<4>(a.flags, m.i, fl);
    load_vec<4>(a.k01, m.i, k01);
    load_vec<4>(a.s0, m.i, s0);
    load_vec<4>(a.k10, m.i, k10);
    load_vec<4>(a.s1, m.i, s1);
    load_vec<4>(a.inv_Ua, m.i, iu);
    load_vec<4>(a.Q_hvac, m.i, q);
    load_vec<4>(a.P_max, m.i, pm);
    load_vec<4>(a.target, m.i, tg);
    load_vec<4>(a.deadband, m.i, db);
    load_vec<4>(a.lockout, m.i, lk);
#pragma unroll
    for (int v = 0; v < 4; ++v) hs[v] = HouseIn{Ta[v], Tm[v], sso[v], fl[v], k01[v], s0[v], k10[v], s1[v], iu[v], q[v], pm[v], tg[v], db[v], lk[v]};
    if (ro.reward_sum) load_vec<4>(ro.reward_sum, m.i, rsum);
  } else {
#pragma unroll
    for (int v = 0; v < 3; ++v)
      if (v < m.nv) {
        const int64_t iv = m.i + v;
        hs[v] = HouseIn{a.Ta[iv], a.Tm[iv], a.sso[iv], a.flags[iv], a.k01[iv], a.s0[iv], a.k10[iv], a.s1[iv], a.inv_Ua[iv], a.Q_hvac[iv],
                        a.P_max[iv], a.target[iv], a.deadband[iv], a.lockout[iv]};
        if (ro.reward_sum) rsum[v] = ro.reward_sum[iv];
      }
  }
  double terr[ENVS], serr = 0.0;   // serr: lane k < ENVS keeps env k's
#pragma unroll
  for (int k = 0; k < ENVS; ++k) terr[k] = 0.0;
  Red3 tot[ENVS];
#pragma unroll
  for (int k = 0; k < ENVS; ++k) tot[k] = Red3{0.0, 0.0, 0.0f};
  Red3 ta{0.0, 0.0, 0.0f}, tb{0.0, 0.0, 0.0f};
  float st_a = 0.0f, st_b = 0.0f;
  double sn_a = 0.0, sn_b = 0.0;
  const int64_t my_env = group * ENVS + lane;   // lane k < ENVS also keeps env k's per-step scalars
  const bool env_lane = lane < ENVS && my_env < a.E;
  for (int s = 0; s < ro.nsteps; ++s) {
    const int64_t base = (int64_t)s * a.E;
    Red3 acc[ENVS];
#pragma unroll
    for (int k = 0; k < ENVS; ++k) acc[k] = Red3{0.0, 0.0, 0.0f};
    if (m.nv > 0) {
      const float od_a = a.od_old[base + m.ea], od_b = a.od_old[base + m.eb], so_a = a.solar_new[base + m.ea], so_b = a.solar_new[base + m.eb];
      float te[ENVS];
#pragma unroll
      for (int k = 0; k < ENVS; ++k) te[k] = 0.0f;
      bool cmds[4];
      controller_cmds<4>(BB ? MDR_ACTIONS_BANGBANG : a.action_source, hs, cmds);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (v >= m.nv) continue;
        const bool first = m.q[v] == m.q[0];
        const bool cmd = cmds[v];
        act[v] = cmd ? 1u : 0u;
        o[v] = house_step(hs[v], cmd, first ? od_a : od_b, first ? so_a : so_b, a.dt);
        hs[v].Ta = o[v].Ta;
        hs[v].Tm = o[v].Tm;
        hs[v].sso = o[v].sso;
        hs[v].flags = o[v].flags;
      }
#pragma unroll
      for (int k = 0; k < ENVS; ++k) {
        float p = 0.0f, ps = 0.0f;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (v < m.nv && m.q[v] == k) {
            p += o[v].power;
            ps += o[v].pen;
            acc[k].max_pen = fmaxf(acc[k].max_pen, o[v].pen);
            const float d = o[v].Ta - hs[v].target;
            te[k] = fmaf(d, d, te[k]);
          }
        acc[k].sum_p = (double)p;
        acc[k].sum_pen = (double)ps;
        terr[k] += (double)te[k];
      }
    }
#pragma unroll
    for (int k = 0; k < ENVS; ++k) tot[k] = lanes_reduce<GROUP>(acc[k], need_pen);
    if (m.nv > 0) {
      ta = pick_env<ENVS>(tot, m.q[0]);
      tb = pick_env<ENVS>(tot, m.q[3]);
      st_a = signal_term(a, ta.sum_p, a.sig_old[base + m.ea]);
      st_b = signal_term(a, tb.sum_p, a.sig_old[base + m.eb]);
      if (want_rsum) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (v >= m.nv) continue;
          const bool first = m.q[v] == m.q[0];
          rsum[v] = __fadd_rn(rsum[v], first ? reward_value(a, o[v].pen, ta.sum_pen, ta.max_pen, st_a) : reward_value(a, o[v].pen, tb.sum_pen, tb.max_pen, st_b));
        }
      }
    }
    if (env_lane) {
      const double P = pick_env<ENVS>(tot, lane).sum_p;
      if (ro.power_trace) ro.power_trace[base + my_env] = P;
      const double d = a.sig_new[base + my_env] - P;
      if (!(ro.defer_last_signal_error && s == ro.nsteps - 1)) serr += d * d;
    }
  }
  if (ro.nsteps <= 0) return;
  Red3 tr[ENVS];
#pragma unroll
  for (int k = 0; k < ENVS; ++k) {
    tr[k] = Red3{terr[k], 0.0, 0.0f};
    tr[k] = lanes_reduce<GROUP>(tr[k], false);
  }
  if (env_lane) {
    a.P[my_env] = pick_env<ENVS>(tot, lane).sum_p;
    if (ro.sq_temp_error_sum) ro.sq_temp_error_sum[my_env] += pick_env<ENVS>(tr, lane).sum_p;
    if (ro.sq_signal_error_sum) ro.sq_signal_error_sum[my_env] += serr;
  }
  if (m.nv == 0) return;
  const int64_t last = (int64_t)(ro.nsteps - 1) * a.E;
  sn_a = a.sig_new[last + m.ea];
  sn_b = a.sig_new[last + m.eb];
  const float os_a = (float)(sn_a * a.inv_obs_norm), os_b = (float)(sn_b * a.inv_obs_norm);
  const float op_a = (float)(ta.sum_p * a.inv_obs_norm), op_b = (float)(tb.sum_p * a.inv_obs_norm);
  float nTa[4], nTm[4], r[4], c5[4], c6[4];
  int nsso[4], lk[4];
  unsigned nfl[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const bool first = m.q[v] == m.q[0];
    nTa[v] = hs[v].Ta;
    nTm[v] = hs[v].Tm;
    nsso[v] = hs[v].sso;
    nfl[v] = hs[v].flags;
    lk[v] = hs[v].lockout;
    r[v] = first ? reward_value(a, o[v].pen, ta.sum_pen, ta.max_pen, st_a) : reward_value(a, o[v].pen, tb.sum_pen, tb.max_pen, st_b);
    c5[v] = first ? os_a : os_b;
    c6[v] = first ? op_a : op_b;
  }
  if (m.nv == 4) {
    store_vec<4>(a.Ta, m.i, nTa);
    store_vec<4>(a.Tm, m.i, nTm);
    store_vec<4>(a.sso, m.i, nsso);
    store_bytes<4>(a.flags, m.i, nfl);
    if (a.actions != nullptr) store_bytes<4>(a.actions, m.i, act);
    store_obs_local<4>(a, m.i, o, lk);
    store_out<4>(a.reward, m.i, r);
    if (a.obs != nullptr) {
      store_out<4>(a.obs + 5 * a.plane, m.i, c5);
      store_out<4>(a.obs + 6 * a.plane, m.i, c6);
    }
    if (ro.reward_sum) store_vec<4>(ro.reward_sum, m.i, rsum);
  } else {
#pragma unroll
    for (int v = 0; v < 3; ++v)
      if (v < m.nv) {
        const int64_t iv = m.i + v;
        a.Ta[iv] = nTa[v];
        a.Tm[iv] = nTm[v];
        a.sso[iv] = nsso[v];
        a.flags[iv] = (uint8_t)nfl[v];
        if (a.actions != nullptr) a.actions[iv] = (uint8_t)act[v];
        store_obs_local<1>(a, iv, o + v, lk + v);
        store_out<1>(a.reward, iv, r + v);
        if (a.obs != nullptr) {
          store_out<1>(a.obs + 5 * a.plane, iv, c5 + v);
          store_out<1>(a.obs + 6 * a.plane, iv, c6 + v);
        }
        if (ro.reward_sum) ro.reward_sum[iv] = rsum[v];
      }
  }
}

// ---- (2d) whole envs packed into a wavefront without rounding their lane count up to a power of two.  N % 4 == 0, L = N / 4
// lanes per env, floor(64 / L) envs per wavefront: 20 houses (the reference's training size, cli.py:53) fill 60 of 64 lanes
// instead of 5 of every 8.  The env's totals come from a segmented reduction over its L consecutive lanes: ceil(log2 L) shuffle
// steps that add the value d lanes up while that lane is still in the env - lane 0 of the env then holds
// ((a0 + a1) + (a2 + a3)) + ... - and one broadcast.  Same function in the step and the rollout kernel: bit for bit alike.
struct PackedLane {
  int L, lane, first;   // lanes per env; this lane's position in its env; wavefront lane of the env's first lane
  int64_t env;          // env index (>= E: idle lane)
  __device__ __forceinline__ void init(const StepArgs& a) {
    L = a.N >> 2;
    const int l64 = threadIdx.x & 63;
    const int per_wave = 64 / L;
    const int k = (l64 * ((65536 + L - 1) / L)) >> 16;   // l64 / L for l64 < 64, L <= 32
    lane = l64 - k * L;
    first = k * L;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    env = k < per_wave ? wave * per_wave + k : INT64_MAX;
  }
};

__device__ __forceinline__ Red3 packed_reduce(Red3 v, const PackedLane& m, bool pen) {
  for (int d = 1; d < m.L; d <<= 1) {
    const bool in = m.lane + d < m.L;
    const double p = __shfl_down(v.sum_p, d, 64);
    v.sum_p += in ? p : 0.0;
    if (pen) {
      const double q = __shfl_down(v.sum_pen, d, 64);
      const float x = __shfl_down(v.max_pen, d, 64);
      v.sum_pen += in ? q : 0.0;
      v.max_pen = fmaxf(v.max_pen, in ? x : 0.0f);
    }
  }
  v.sum_p = __shfl(v.sum_p, m.first, 64);
  if (pen) {
    v.sum_pen = __shfl(v.sum_pen, m.first, 64);
    v.max_pen = __shfl(v.max_pen, m.first, 64);
  }
  return v;
}

__global__ __launch_bounds__(256) void k_step_packed(StepArgs a) {
  rebase(a);
  const bool need_pen = a.penalty_mode != MDR_PENALTY_INDIVIDUAL_L2;
  PackedLane m;
  m.init(a);
  const bool active = m.env < a.E;
  const int e = active ? (int)m.env : a.E - 1;
  const int64_t i = (int64_t)e * a.N + m.lane * 4;
  HouseOut o[4];
  int lockout[4];
  float pen[4];
  Red3 acc{0.0, 0.0, 0.0f};
  if (active) {
    step_vec<4>(a, i, a.od_old[e], a.solar_new[e], o, lockout);
    float p = 0.0f, ps = 0.0f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      p += o[v].power;
      ps += o[v].pen;
      pen[v] = o[v].pen;
      acc.max_pen = fmaxf(acc.max_pen, o[v].pen);
    }
    acc.sum_p = (double)p;
    acc.sum_pen = (double)ps;
    store_obs_local<4>(a, i, o, lockout);
  }
  const Red3 tot = packed_reduce(acc, m, need_pen);
  if (active) {
    const float sig_term = signal_term(a, tot.sum_p, a.sig_old[e]);
    if (m.lane == 0) a.P[e] = tot.sum_p;
    store_reward_power<4>(a, i, pen, tot.sum_pen, tot.max_pen, sig_term, (float)(a.sig_new[e] * a.inv_obs_norm), (float)(tot.sum_p * a.inv_obs_norm));
  }
  cursor_done(a);
}

template <bool BB>
__global__ __launch_bounds__(256) void k_rollout_packed(StepArgs a, RolloutArgs ro) {
  const bool need_pen = a.penalty_mode != MDR_PENALTY_INDIVIDUAL_L2;
  const bool want_rsum = ro.reward_sum != nullptr;
  PackedLane m;
  m.init(a);
  const bool active = m.env < a.E;
  const int e = active ? (int)m.env : a.E - 1;
  const int64_t i = (int64_t)e * a.N + m.lane * 4;
  HouseIn hs[4];
  HouseOut o[4];
  int lockout[4];
  float rsum[4], pen[4];
  unsigned act[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    hs[v] = HouseIn{};
    o[v] = HouseOut{};
    lockout[v] = 1;
    rsum[v] = 0.0f;
    act[v] = 0;
  }
  if (active) {
    float Ta[4], Tm[4], k01[4], s0[4], k10[4], s1[4], iu[4], q[4], pm[4], tg[4], db[4];
    int sso[4];
    unsigned fl[4];
    load_vec<4>(a.Ta, i, Ta);
    load_vec<4>(a.Tm, i, Tm);
    load_vec<4>(a.sso, i, sso);
    load_bytes<4>(a.flags, i, fl);
    load_vec<4>(a.k01, i, k01);
    load_vec<4>(a.s0, i, s0);
    load_vec<4>(a.k10, i, k10);
    load_vec<4>(a.s1, i, s1);
    load_vec<4>(a.inv_Ua, i, iu);
    load_vec<4>(a.Q_hvac, i, q);
    load_vec<4>(a.P_max, i, pm);
    load_vec<4>(a.target, i, tg);
    load_vec<4>(a.deadband, i, db);
    load_vec<4>(a.lockout, i, lockout);
#pragma unroll
    for (int v = 0; v < 4; ++v) hs[v] = HouseIn{Ta[v], Tm[v], sso[v], fl[v], k01[v], s0[v], k10[v], s1[v], iu[v], q[v], pm[v], tg[v], db[v], lockout[v]};
    if (ro.reward_sum) load_vec<4>(ro.reward_sum, i, rsum);
  }
  float sig_term = 0.0f;
  double terr = 0.0, serr = 0.0, sig_new = 0.0;
  Red3 tot{0.0, 0.0, 0.0f};
  for (int s = 0; s < ro.nsteps; ++s) {
    const int64_t row = (int64_t)s * a.E + e;
    const float od = a.od_old[row], solar = a.solar_new[row];
    const double sig_old = a.sig_old[row];
    sig_new = a.sig_new[row];
    Red3 acc{0.0, 0.0, 0.0f};
    if (active) {
      float p = 0.0f, ps = 0.0f, te = 0.0f;
      bool cmds[4];
      controller_cmds<4>(BB ? MDR_ACTIONS_BANGBANG : a.action_source, hs, cmds);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const bool cmd = cmds[v];
        act[v] = cmd ? 1u : 0u;
        o[v] = house_step(hs[v], cmd, od, solar, a.dt);
        hs[v].Ta = o[v].Ta;
        hs[v].Tm = o[v].Tm;
        hs[v].sso = o[v].sso;
        hs[v].flags = o[v].flags;
        p += o[v].power;
        ps += o[v].pen;
        acc.max_pen = fmaxf(acc.max_pen, o[v].pen);
        const float d = o[v].Ta - hs[v].target;
        te = fmaf(d, d, te);
      }
      acc.sum_p = (double)p;
      acc.sum_pen = (double)ps;
      terr += (double)te;
    }
    tot = packed_reduce(acc, m, need_pen);
    sig_term = signal_term(a, tot.sum_p, sig_old);
    if (active) {
      if (want_rsum) {
#pragma unroll
        for (int v = 0; v < 4; ++v) rsum[v] = __fadd_rn(rsum[v], reward_value(a, o[v].pen, tot.sum_pen, tot.max_pen, sig_term));
      }
      if (m.lane == 0) {
        if (ro.power_trace) ro.power_trace[row] = tot.sum_p;
        const double d = sig_new - tot.sum_p;
        if (!(ro.defer_last_signal_error && s == ro.nsteps - 1)) serr += d * d;
      }
    }
  }
  if (ro.nsteps <= 0) return;
  Red3 tr{terr, 0.0, 0.0f};
  tr = packed_reduce(tr, m, false);
  if (!active) return;
  float nTa[4], nTm[4];
  int nsso[4];
  unsigned nfl[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    nTa[v] = hs[v].Ta;
    nTm[v] = hs[v].Tm;
    nsso[v] = hs[v].sso;
    nfl[v] = hs[v].flags;
    pen[v] = o[v].pen;
  }
  store_vec<4>(a.Ta, i, nTa);
  store_vec<4>(a.Tm, i, nTm);
  store_vec<4>(a.sso, i, nsso);
  store_bytes<4>(a.flags, i, nfl);
  if (a.actions != nullptr) store_bytes<4>(a.actions, i, act);
  store_obs_local<4>(a, i, o, lockout);
  store_reward_power<4>(a, i, pen, tot.sum_pen, tot.max_pen, sig_term, (float)(sig_new * a.inv_obs_norm), (float)(tot.sum_p * a.inv_obs_norm));
  if (ro.reward_sum) store_vec<4>(ro.reward_sum, i, rsum);
  if (m.lane == 0) {
    a.P[e] = tot.sum_p;
    if (ro.sq_temp_error_sum) ro.sq_temp_error_sum[e] += tr.sum_p;
    if (ro.sq_signal_error_sum) ro.sq_signal_error_sum[e] += serr;
  }
}

int64_t multi_blocks(int64_t E, const StepPlan& p) {
  if (p.kind == STEP_PACKED) {   // p.tiles envs per wavefront
    const int64_t waves = (E + p.tiles - 1) / p.tiles;
    return (waves + 3) / 4;
  }
  const int64_t groups = (E + p.tiles - 1) / p.tiles;
  return (groups * p.threads + 255) / 256;
}

#define MDR_MULTI_DISPATCH(KERNEL, ...)                                              \
  switch (p.tiles * 100 + p.threads) {                                               \
    case 204: hipLaunchKernelGGL((KERNEL<4, 2>), g, b, 0, s, __VA_ARGS__); break;    \
    case 208: hipLaunchKernelGGL((KERNEL<8, 2>), g, b, 0, s, __VA_ARGS__); break;    \
    case 216: hipLaunchKernelGGL((KERNEL<16, 2>), g, b, 0, s, __VA_ARGS__); break;   \
    case 232: hipLaunchKernelGGL((KERNEL<32, 2>), g, b, 0, s, __VA_ARGS__); break;   \
    case 264: hipLaunchKernelGGL((KERNEL<64, 2>), g, b, 0, s, __VA_ARGS__); break;   \
    default: return hipErrorInvalidValue;                                            \
  }

hipError_t launch_step_multi(const StepArgs& a, const StepPlan& p, hipStream_t s) {
  const dim3 g((unsigned)multi_blocks(a.E, p)), b(256);
  if (p.kind == STEP_PACKED) {
    hipLaunchKernelGGL(k_step_packed, g, b, 0, s, a);
    return hipGetLastError();
  }
  MDR_MULTI_DISPATCH(k_step_multi, a)
  return hipGetLastError();
}

#undef MDR_MULTI_DISPATCH

#define MDR_MULTI_ROLLOUT(BBV)                                                                             \
  switch (p.tiles * 100 + p.threads) {                                                                     \
    case 204: hipLaunchKernelGGL((k_rollout_multi<4, 2, BBV>), g, b, 0, s, a, r); break;                   \
    case 208: hipLaunchKernelGGL((k_rollout_multi<8, 2, BBV>), g, b, 0, s, a, r); break;                   \
    case 216: hipLaunchKernelGGL((k_rollout_multi<16, 2, BBV>), g, b, 0, s, a, r); break;                  \
    case 232: hipLaunchKernelGGL((k_rollout_multi<32, 2, BBV>), g, b, 0, s, a, r); break;                  \
    case 264: hipLaunchKernelGGL((k_rollout_multi<64, 2, BBV>), g, b, 0, s, a, r); break;                  \
    default: return hipErrorInvalidValue;                                                                  \
  }

hipError_t launch_rollout_multi(const StepArgs& a, const RolloutArgs& r, const StepPlan& p, hipStream_t s) {
  const dim3 g((unsigned)multi_blocks(a.E, p)), b(256);
  const bool bb = a.action_source == MDR_ACTIONS_BANGBANG;
  if (p.kind == STEP_PACKED) {
    if (bb) hipLaunchKernelGGL(k_rollout_packed<true>, g, b, 0, s, a, r);
    else hipLaunchKernelGGL(k_rollout_packed<false>, g, b, 0, s, a, r);
    return hipGetLastError();
  }
  if (bb) { MDR_MULTI_ROLLOUT(true) }
  else { MDR_MULTI_ROLLOUT(false) }
  return hipGetLastError();
}
#undef MDR_MULTI_ROLLOUT

}  // namespace mdr
