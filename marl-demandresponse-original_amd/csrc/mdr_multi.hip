// Several small envs per lane group (STEP_MULTI): the reference's own cluster sizes - 20 houses in training, 50 in deployment
// (cli.py:53, 629) - in batches that fill the device.  env/MA_DemandResponse.py:1005-1055 (ClusterHouses.step), 234-373 (rewards).
#include <cstdlib>

#include "mdr_device.h"
#include "mdr_kernels.h"
#include "mdr_step_common.h"

namespace mdr {
// ---- (2c) two small envs per lane group, four FLAT houses per lane.  The reference deploys with 50 houses (cli.py:629).  N % 4 == 2:
// one env per lane group forces narrow accesses (rows start every 200 bytes, so 8-byte accesses at most).  Here two consecutive
// envs share a group of GROUP lanes, 2 N a multiple of 4: lane l holds the four houses 4 l .. 4 l + 3 of the group's 2 N, always on
// a 16-byte boundary of the [E][N] arrays and in at most two envs: 50 houses = 25 lanes of 32 with 16-byte accesses.  (Three and
// four envs per group - 20 houses at 15 of 16 lanes, odd N - measured slower than one env per group, see plan_step.)
//
// The totals are those of the one-env-per-group mapping (k_step_group<GROUP, 2>: GROUP lanes per env, one PAIR of houses per lane)
// bit for bit, so the multi-step kernel of that mapping (k_rollout_group<GROUP, 2>) is this step's closed loop: a rollout is bound
// by vector instructions and by the waves a SIMD holds, not by access width, and four houses of two envs per lane cost it twice the
// registers and a reduction per env (measured at 4.19 M houses of 50: 20.5 us per step on this mapping, 10.8 us on that one).
// An env begins and ends on a pair (N even): with h = N / 2 pairs per env the lane's pair A (houses 0, 1) and pair B (2, 3) are
// leaves 2 l and 2 l + 1 of env 0's tree for l <= (h - 1) / 2 = L0, and env 1's leaves 2 n, 2 n + 1 are pair B of lane L0 + n and
// pair A of lane L0 + n + 1.  So after the first level - fp32 pair sums widened to fp64 and added, own pair + neighbour's as over
// there - env 0's nodes sit in lanes 0 .. and env 1's in lanes L0 ..; those move up to lanes GROUP / 2 .., and ONE tree over the two
// half groups (lanes_reduce<GROUP / 2>) finishes both envs at once.
struct MultiLane {
  int nv;           // houses of this lane that exist: 4, or 2 / 0 at the end of the batch or of the group (N even: whole pairs)
  int q[4];         // env of each house within the group
  int64_t i;        // flat index of the first house
  int64_t ea, eb;   // the (at most two) envs the lane's houses belong to: first and last house's
  __device__ __forceinline__ void init(const StepArgs& a, int64_t group, int lane) {
    const int k0 = lane * 4;
    const int64_t e0 = group * 2;
    i = e0 * a.N + k0;
    const int64_t total = (int64_t)a.E * a.N;
    nv = (k0 < 2 * a.N && i < total) ? (int)(total - i < 4 ? total - i : 4) : 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) q[v] = k0 + v >= a.N ? 1 : 0;
    const int64_t last = (int64_t)a.E - 1;
    ea = e0 + q[0] < last ? e0 + q[0] : last;
    eb = e0 + q[3] < last ? e0 + q[3] : last;
  }
};

template <int GROUP>
__global__ __launch_bounds__(256) void k_step_multi(StepArgs a) {
  constexpr int HALF = GROUP / 2;
  rebase(a);
  const bool need_pen = a.penalty_mode != MDR_PENALTY_INDIVIDUAL_L2;
  const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) / GROUP;
  const int lane = threadIdx.x % GROUP;
  const int wbase = (threadIdx.x & 63) & ~(GROUP - 1);   // first lane of this group inside the wavefront
  MultiLane m;
  m.init(a, group, lane);
  HouseOut o[4];
  int lockout[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) o[v] = HouseOut{};
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
    } else {   // the batch's last lane when E is odd: one pair
      step_vec_rows<2>(a, m.i, od, so, o, lockout);
      store_obs_local<2>(a, m.i, o, lockout);
    }
  }
  // level 1: the lane's pairs as k_step_group<GROUP, 2>'s lanes hold them (fp32 sum of two houses, widened), own + neighbour
  const bool a_in = m.nv >= 2, b_in = m.nv == 4;
  const bool a0 = a_in && m.q[0] == 0, a1 = a_in && m.q[0] == 1, b0 = b_in && m.q[2] == 0, b1 = b_in && m.q[2] == 1;
  const double pa = (double)(o[0].power + o[1].power), pb = (double)(o[2].power + o[3].power);
  const bool has_next = lane < GROUP - 1;   // the next lane's pair A follows this lane's pair B
  const double pa_next = __shfl_down(a1 ? pa : 0.0, 1, 64);
  Red3 n0{(a0 ? pa : 0.0) + (b0 ? pb : 0.0), 0.0, 0.0f};
  Red3 n1{(b1 ? pb : 0.0) + (has_next ? pa_next : 0.0), 0.0, 0.0f};
  if (need_pen) {
    const float sa = o[0].pen + o[1].pen, sb = o[2].pen + o[3].pen, xa = fmaxf(o[0].pen, o[1].pen), xb = fmaxf(o[2].pen, o[3].pen);
    const double sa_next = __shfl_down(a1 ? (double)sa : 0.0, 1, 64);
    const float xa_next = __shfl_down(a1 ? xa : 0.0f, 1, 64);
    n0.sum_pen = (a0 ? (double)sa : 0.0) + (b0 ? (double)sb : 0.0);
    n0.max_pen = fmaxf(a0 ? xa : 0.0f, b0 ? xb : 0.0f);
    n1.sum_pen = (b1 ? (double)sb : 0.0) + (has_next ? sa_next : 0.0);
    n1.max_pen = fmaxf(b1 ? xb : 0.0f, has_next ? xa_next : 0.0f);
  }
  // env 1's nodes from lanes L0 .. up to lanes HALF ..; then one tree over each half group
  const int L0 = (a.N - 2) >> 2;
  const int src = wbase + (lane >= HALF ? lane - HALF + L0 : lane);
  Red3 node;
  node.sum_p = __shfl(n1.sum_p, src, 64);
  node.sum_pen = need_pen ? __shfl(n1.sum_pen, src, 64) : 0.0;
  node.max_pen = need_pen ? __shfl(n1.max_pen, src, 64) : 0.0f;
  if (lane < HALF) node = n0;
  node = lanes_reduce<HALF>(node, need_pen);
  Red3 t0, t1;   // both envs' totals in every lane
  t0.sum_p = __shfl(node.sum_p, wbase, 64);
  t1.sum_p = __shfl(node.sum_p, wbase + HALF, 64);
  t0.sum_pen = need_pen ? __shfl(node.sum_pen, wbase, 64) : 0.0;
  t1.sum_pen = need_pen ? __shfl(node.sum_pen, wbase + HALF, 64) : 0.0;
  t0.max_pen = need_pen ? __shfl(node.max_pen, wbase, 64) : 0.0f;
  t1.max_pen = need_pen ? __shfl(node.max_pen, wbase + HALF, 64) : 0.0f;
  if (m.nv > 0) {
    const Red3 ta = m.q[0] ? t1 : t0, tb = m.q[3] ? t1 : t0;
    const float st_a = signal_term(a, ta.sum_p, a.sig_old[m.ea]), st_b = signal_term(a, tb.sum_p, a.sig_old[m.eb]);
    const float os_a = (float)(a.sig_new[m.ea] * a.inv_obs_norm), os_b = (float)(a.sig_new[m.eb] * a.inv_obs_norm);
    const float op_a = (float)(ta.sum_p * a.inv_obs_norm), op_b = (float)(tb.sum_p * a.inv_obs_norm);
    float r[4], c5[4], c6[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bool first = m.q[v] == m.q[0];
      r[v] = reward_value(a, o[v].pen, first ? ta.sum_pen : tb.sum_pen, first ? ta.max_pen : tb.max_pen, first ? st_a : st_b);
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
      store_out<2>(a.reward, m.i, r);
      if (a.obs != nullptr) {
        store_out<2>(a.obs + 5 * a.plane, m.i, c5);
        store_out<2>(a.obs + 6 * a.plane, m.i, c6);
      }
    }
  }
  if (lane < 2 && group * 2 + lane < a.E) a.P[group * 2 + lane] = lane ? t1.sum_p : t0.sum_p;
  cursor_done(a);
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
  int lockout[4];
  float rsum[4], pen[4];
  uint64_t on_m[4], lock_m[4], cmd_m[4];   // the HVAC bits and the latest command as lane masks (house_advance_m)
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    hs[v] = HouseIn{};
    lockout[v] = 1;
    rsum[v] = 0.0f;
    pen[v] = 0.0f;
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
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    on_m[v] = __builtin_amdgcn_ballot_w64((hs[v].flags & 1u) != 0u);
    lock_m[v] = __builtin_amdgcn_ballot_w64((hs[v].flags & 2u) != 0u);
    cmd_m[v] = 0;
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
    {   // every lane, idle ones too (blank houses: exact zeros in every sum): the lane masks need wave-uniform control flow
      float p = 0.0f, ps = 0.0f, te = 0.0f;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        cmd_m[v] = BB ? __builtin_amdgcn_ballot_w64(hs[v].Ta > hs[v].target)   // agents/bangbang_controllers.py:49-59
                      : controller_cmd_m(a.action_source, hs[v].Ta, hs[v].target, hs[v].deadband, on_m[v]);
        const HouseNextM n = house_advance_m(hs[v], on_m[v], cmd_m[v], od, solar, a.dt);
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
        for (int v = 0; v < 4; ++v) {
          ps += pen[v];
          acc.max_pen = fmaxf(acc.max_pen, pen[v]);
        }
      }
      acc.sum_p = (double)p;
      acc.sum_pen = (double)ps;
      terr += (double)te;
    }
    tot = packed_reduce(acc, m, need_pen);
    sig_term = signal_term(a, tot.sum_p, sig_old);
    if (active) {
      if (want_rsum) {
        add_rewards<4>(a, need_pen, pen, tot.sum_pen, tot.max_pen, sig_term, rsum);
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
  unsigned nfl[4], act[4];
  HouseOut o[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    nTa[v] = hs[v].Ta;
    nTm[v] = hs[v].Tm;
    nsso[v] = hs[v].sso;
    nfl[v] = house_flags(lane_bit(on_m[v]), lane_bit(lock_m[v]));
    act[v] = lane_bit(cmd_m[v]) ? 1u : 0u;
    o[v] = HouseOut{nTa[v], nTm[v], nsso[v], nfl[v], pen[v], 0.0f};
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

hipError_t launch_step_multi(const StepArgs& a, const StepPlan& p, hipStream_t s) {
  const dim3 g((unsigned)multi_blocks(a.E, p)), b(256);
  if (p.kind == STEP_PACKED) {
    hipLaunchKernelGGL(k_step_packed, g, b, 0, s, a);
    return hipGetLastError();
  }
  if (p.tiles != 2 || a.N % 4 != 2) return hipErrorInvalidValue;   // k_step_multi: two envs per group, envs made of whole pairs
  switch (p.threads) {
    case 4: hipLaunchKernelGGL(k_step_multi<4>, g, b, 0, s, a); break;
    case 8: hipLaunchKernelGGL(k_step_multi<8>, g, b, 0, s, a); break;
    case 16: hipLaunchKernelGGL(k_step_multi<16>, g, b, 0, s, a); break;
    case 32: hipLaunchKernelGGL(k_step_multi<32>, g, b, 0, s, a); break;
    case 64: hipLaunchKernelGGL(k_step_multi<64>, g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// STEP_PACKED only: a STEP_MULTI plan's multi-step kernel is k_rollout_group (plan_rollout)
hipError_t launch_rollout_multi(const StepArgs& a, const RolloutArgs& r, const StepPlan& p, hipStream_t s) {
  if (p.kind != STEP_PACKED) return hipErrorInvalidValue;
  const dim3 g((unsigned)multi_blocks(a.E, p)), b(256);
  if (a.action_source == MDR_ACTIONS_BANGBANG) hipLaunchKernelGGL(k_rollout_packed<true>, g, b, 0, s, a, r);
  else hipLaunchKernelGGL(k_rollout_packed<false>, g, b, 0, s, a, r);
  return hipGetLastError();
}

}  // namespace mdr
