// Per-house step pieces shared by the single-step kernels (mdr_kernels.hip) and the persistent sharded rollout
// (mdr_persist.hip): reward arithmetic, vector loads / stores, the observation planes.  gfx950 only.
#pragma once

#include "mdr_device.h"
#include "mdr_kernels.h"

namespace mdr {

// Selects on the (wave-uniform) mode, not branches: written as an if-chain this cost the multi-step kernels a dozen scalar branches
// per HOUSE and step; the env-wide parts (common, the inner fma) are the same for every house of a lane and are formed once.
__device__ __forceinline__ float temp_penalty(const StepArgs& a, float pen, double sum_pen, float max_pen) {
  const float common = (float)(sum_pen * a.inv_n_total);
  const float mixture = __fmaf_rn(a.mix_i, pen, __fmaf_rn(a.mix_c, common, a.mix_m * max_pen));   // explicit: no context-dependent contraction
  const float env_wide = a.penalty_mode == MDR_PENALTY_COMMON_L2 ? common : max_pen;
  const float single = a.penalty_mode == MDR_PENALTY_INDIVIDUAL_L2 ? pen : env_wide;
  return (a.penalty_mode == MDR_PENALTY_INDIVIDUAL_L2 || a.penalty_mode == MDR_PENALTY_COMMON_L2 || a.penalty_mode == MDR_PENALTY_COMMON_MAX) ? single : mixture;
}

// r_i = -(alpha_temp * pen_i / norm_T + alpha_sig * sig / norm_S)  (env 364-372); one explicit fma so that every
// kernel rounds it identically
__device__ __forceinline__ float reward_value(const StepArgs& a, float pen, double sum_pen, float max_pen, float sig_term) {
  return -__fmaf_rn(a.c_temp, temp_penalty(a, pen, sum_pen, max_pen), sig_term);
}

// The running reward sum of a multi-step kernel.  individual_L2 (the default mode; `common` false, wave-uniform) has its own arm: its
// reward is -(c_temp pen + sig) - what reward_value() returns there, bit for bit - without forming the mixture and selecting it away.
template <int VEC>
__device__ __forceinline__ void add_rewards(const StepArgs& a, bool common, const float* pen, double sum_pen, float max_pen, float sig_term, float* rsum) {
  if (!common) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) rsum[v] = __fadd_rn(rsum[v], -__fmaf_rn(a.c_temp, pen[v], sig_term));
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) rsum[v] = __fadd_rn(rsum[v], reward_value(a, pen[v], sum_pen, max_pen, sig_term));
  }
}

// signal part of the reward with the OLD signal (env 196, 234-251), fp64 per env
__device__ __forceinline__ float signal_term(const StepArgs& a, double P, double S_old) {
  const double d = (P - S_old) * a.inv_n_total;
  return (float)(a.c_sig * d * d);
}

// ---- (1) one workgroup per env, VEC houses per thread per tile ---------------------------------
template <typename T, typename V>
__device__ __forceinline__ void unpack(const V& v, T* out);
template <>
__device__ __forceinline__ void unpack<float, float4>(const float4& v, float* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <>
__device__ __forceinline__ void unpack<int, int4>(const int4& v, int* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <>
__device__ __forceinline__ void unpack<unsigned, uchar4>(const uchar4& v, unsigned* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <>
__device__ __forceinline__ void unpack<float, float>(const float& v, float* o) { o[0] = v; }
template <>
__device__ __forceinline__ void unpack<int, int>(const int& v, int* o) { o[0] = v; }
template <>
__device__ __forceinline__ void unpack<unsigned, unsigned char>(const unsigned char& v, unsigned* o) { o[0] = v; }

template <>
__device__ __forceinline__ void unpack<float, float2>(const float2& v, float* o) { o[0] = v.x; o[1] = v.y; }
template <>
__device__ __forceinline__ void unpack<int, int2>(const int2& v, int* o) { o[0] = v.x; o[1] = v.y; }
template <>
__device__ __forceinline__ void unpack<unsigned, uchar2>(const uchar2& v, unsigned* o) { o[0] = v.x; o[1] = v.y; }
__device__ __forceinline__ float2 pack2(const float* v) { return make_float2(v[0], v[1]); }
__device__ __forceinline__ int2 pack2(const int* v) { return make_int2(v[0], v[1]); }
__device__ __forceinline__ uchar2 pack2(const unsigned* v) { return make_uchar2((unsigned char)v[0], (unsigned char)v[1]); }

__device__ __forceinline__ float4 pack4(const float* v) { return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ int4 pack4(const int* v) { return make_int4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ uchar4 pack4(const unsigned* v) {
  return make_uchar4((unsigned char)v[0], (unsigned char)v[1], (unsigned char)v[2], (unsigned char)v[3]);
}

template <int VEC, typename T>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, int64_t i, T* out) {
  if constexpr (VEC == 4) {
    using V = typename std::conditional<std::is_same<T, float>::value, float4, int4>::type;
    unpack<T, V>(*reinterpret_cast<const V*>(p + i), out);
  } else if constexpr (VEC == 2) {
    using V = typename std::conditional<std::is_same<T, float>::value, float2, int2>::type;
    unpack<T, V>(*reinterpret_cast<const V*>(p + i), out);
  } else {
    out[0] = p[i];
  }
}
template <int VEC>
__device__ __forceinline__ void load_bytes(const uint8_t* __restrict__ p, int64_t i, unsigned* out) {
  if constexpr (VEC == 4) {
    unpack<unsigned, uchar4>(*reinterpret_cast<const uchar4*>(p + i), out);
  } else if constexpr (VEC == 2) {
    unpack<unsigned, uchar2>(*reinterpret_cast<const uchar2*>(p + i), out);
  } else {
    out[0] = p[i];
  }
}
template <int VEC, typename T>
__device__ __forceinline__ void store_vec(T* __restrict__ p, int64_t i, const T* v) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<decltype(pack4(v))*>(p + i) = pack4(v);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<decltype(pack2(v))*>(p + i) = pack2(v);
  } else {
    p[i] = v[0];
  }
}
// Output-only streams (reward, observation planes) are never re-read by this library: non-temporal stores keep them
// from displacing the state/parameter lines in L2/MALL (measured at C3: 66.0 -> 60.3 us per step).  Non-temporal
// LOADS of the parameters (+3.5 % alone, no gain on top of the stores) and non-temporal STATE stores (-3 %) lose:
// build with -DMDR_NT_STORES=0 / -DMDR_NT_LOADS=1 / -DMDR_NT_STATE=1 to reproduce.
#ifndef MDR_NT_STORES
#define MDR_NT_STORES 1
#endif
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <int VEC>
__device__ __forceinline__ void store_out(float* __restrict__ p, int64_t i, const float* v) {
#if defined(MDR_NT_STORES) && MDR_NT_STORES
  if constexpr (VEC == 4) {
    v4f_t x = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(x, reinterpret_cast<v4f_t*>(p + i));
  } else if constexpr (VEC == 2) {
    v2f_t x = {v[0], v[1]};
    __builtin_nontemporal_store(x, reinterpret_cast<v2f_t*>(p + i));
  } else {
    __builtin_nontemporal_store(v[0], p + i);
  }
#else
  store_vec<VEC>(p, i, v);
#endif
}
template <int VEC>
__device__ __forceinline__ void load_param(const float* __restrict__ p, int64_t i, float* out) {
#if defined(MDR_NT_LOADS) && MDR_NT_LOADS
  if constexpr (VEC == 4) {
    const v4f_t x = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p + i));
    out[0] = x.x; out[1] = x.y; out[2] = x.z; out[3] = x.w;
  } else if constexpr (VEC == 2) {
    const v2f_t x = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(p + i));
    out[0] = x.x; out[1] = x.y;
  } else {
    out[0] = __builtin_nontemporal_load(p + i);
  }
#else
  load_vec<VEC>(p, i, out);
#endif
}
template <int VEC>
__device__ __forceinline__ void store_bytes(uint8_t* __restrict__ p, int64_t i, const unsigned* v) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<uchar4*>(p + i) = pack4(v);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<uchar2*>(p + i) = pack2(v);
  } else {
    p[i] = (uint8_t)v[0];
  }
}

// Loads VEC houses starting at flat index i, steps them, stores the new state, returns outputs.  od_old[v] / solar[v]: the table
// values of house v's env (one env per call in most kernels; k_step_multi's lanes may hold the end of one env and the start of the next).
// The reference's rule-based controllers on the pre-step observation (agents/bangbang_controllers.py): BangBangController 41-61,
// DeadbandBangBangController 13-38 == BasicController 64-88, AlwaysOnController 1-10.  `src` is wave-uniform.
__device__ __forceinline__ bool controller_cmd(int src, float Ta, float target, float deadband, bool on) {
#if defined(MDR_CONTROLLER_BANGBANG_ONLY) && MDR_CONTROLLER_BANGBANG_ONLY   // experiment build: what the other two rules cost the bang-bang loop
  return Ta > target;
#endif
  // one straight-line form for the three rules (the rollout kernels run this in their latency-bound inner loop): the band is
  // [target - h, target + h] with h = deadband / 2 for the deadband rule and 0 otherwise; inside the band (edges included) the
  // deadband rule keeps what the HVAC is doing, bang-bang says off (its band is the single point Ta == target)
  const float h = (src == MDR_ACTIONS_DEADBAND ? 0.5f : 0.0f) * deadband;
  const bool above = Ta > target + h, below = Ta < target - h;
  const bool keep = src == MDR_ACTIONS_DEADBAND && on;
  return above || (!below && keep) || src == MDR_ACTIONS_ALWAYS_ON;
}

__device__ __forceinline__ bool controller_cmd(int src, float Ta, float target, float deadband, unsigned flags) {   // flags: bit 0 = on
  return controller_cmd(src, Ta, target, deadband, (flags & 1u) != 0u);
}

// The commands of a lane's VEC houses.  The bang-bang rule - the default of every closed loop - sits behind a wave-uniform BRANCH
// of its own: computed through the general form it cost the vector-bound rollout kernels 7-9 % (r03: fused rollout 6.6 -> 7.2 us).
template <int VEC>
__device__ __forceinline__ void controller_cmds(int src, const HouseIn* hs, bool* cmd) {
  if (src == MDR_ACTIONS_BANGBANG) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) cmd[v] = hs[v].Ta > hs[v].target;   // agents/bangbang_controllers.py:49-59
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) cmd[v] = controller_cmd(src, hs[v].Ta, hs[v].target, hs[v].deadband, (hs[v].flags & 1u) != 0u);
  }
}

// The same rules on lane masks (house_advance_m): the commands of all 64 lanes' house v as one mask
__device__ __forceinline__ uint64_t controller_cmd_m(int src, float Ta, float target, float deadband, uint64_t on_m) {
  const float h = (src == MDR_ACTIONS_DEADBAND ? 0.5f : 0.0f) * deadband;
  const uint64_t above = __builtin_amdgcn_ballot_w64(Ta > target + h), below = __builtin_amdgcn_ballot_w64(Ta < target - h);
  const uint64_t keep = src == MDR_ACTIONS_DEADBAND ? on_m : 0ull;
  return above | (~below & keep) | (src == MDR_ACTIONS_ALWAYS_ON ? ~0ull : 0ull);
}

// ... with the on bits carried as booleans (house_advance)
template <int VEC>
__device__ __forceinline__ void controller_cmds(int src, const HouseIn* hs, const bool* on, bool* cmd) {
  if (src == MDR_ACTIONS_BANGBANG) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) cmd[v] = hs[v].Ta > hs[v].target;
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) cmd[v] = controller_cmd(src, hs[v].Ta, hs[v].target, hs[v].deadband, on[v]);
  }
}

template <int VEC>
__device__ __forceinline__ void step_vec_rows(const StepArgs& a, int64_t i, const float* od_old, const float* solar, HouseOut* out, int* lockout) {
  float Ta[VEC], Tm[VEC], k01[VEC], s0[VEC], k10[VEC], s1[VEC], iu[VEC], q[VEC], pm[VEC], tg[VEC], db[VEC];
  int sso[VEC];
  unsigned fl[VEC], act[VEC];
  load_vec<VEC>(a.Ta, i, Ta);
  load_vec<VEC>(a.Tm, i, Tm);
  load_vec<VEC>(a.sso, i, sso);
  load_bytes<VEC>(a.flags, i, fl);
  if (a.action_source == MDR_ACTIONS_EXTERNAL) load_bytes<VEC>(a.actions, i, act);
  load_param<VEC>(a.k01, i, k01);
  load_param<VEC>(a.s0, i, s0);
  load_param<VEC>(a.k10, i, k10);
  load_param<VEC>(a.s1, i, s1);
  load_param<VEC>(a.inv_Ua, i, iu);
  load_param<VEC>(a.Q_hvac, i, q);
  load_param<VEC>(a.P_max, i, pm);
  load_param<VEC>(a.target, i, tg);
  load_param<VEC>(a.deadband, i, db);
  load_vec<VEC>(a.lockout, i, lockout);
  float nTa[VEC], nTm[VEC];
  int nsso[VEC];
  unsigned nfl[VEC];
  // the in-kernel controllers act on the pre-step observation (agents/bangbang_controllers.py); three wave-uniform arms
  bool cmds[VEC];
  if (a.action_source == MDR_ACTIONS_EXTERNAL) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) cmds[v] = act[v] != 0u;
  } else if (a.action_source == MDR_ACTIONS_BANGBANG) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      cmds[v] = Ta[v] > tg[v];
      act[v] = cmds[v] ? 1u : 0u;
    }
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      cmds[v] = controller_cmd(a.action_source, Ta[v], tg[v], db[v], (fl[v] & 1u) != 0u);
      act[v] = cmds[v] ? 1u : 0u;
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    HouseIn h{Ta[v], Tm[v], sso[v], fl[v], k01[v], s0[v], k10[v], s1[v], iu[v], q[v], pm[v], tg[v], db[v], lockout[v]};
    const bool cmd = cmds[v];
    out[v] = house_step(h, cmd, od_old[v], solar[v], a.dt);
    nTa[v] = out[v].Ta;
    nTm[v] = out[v].Tm;
    nsso[v] = out[v].sso;
    nfl[v] = out[v].flags;
  }
#if defined(MDR_NT_STATE) && MDR_NT_STATE
  store_out<VEC>(a.Ta, i, nTa);
  store_out<VEC>(a.Tm, i, nTm);
#else
  store_vec<VEC>(a.Ta, i, nTa);
  store_vec<VEC>(a.Tm, i, nTm);
#endif
  store_vec<VEC>(a.sso, i, nsso);
  store_bytes<VEC>(a.flags, i, nfl);
  if (a.action_source != MDR_ACTIONS_EXTERNAL && a.actions != nullptr) store_bytes<VEC>(a.actions, i, act);
}

template <int VEC>
__device__ __forceinline__ void step_vec(const StepArgs& a, int64_t i, float od_old, float solar, HouseOut* out, int* lockout) {
  float od[VEC], so[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    od[v] = od_old;
    so[v] = solar;
  }
  step_vec_rows<VEC>(a, i, od, so, out, lockout);
}

// The five observation columns that do not depend on the env-wide reductions.
template <int VEC>
__device__ __forceinline__ void store_obs_local(const StepArgs& a, int64_t i, const HouseOut* o, const int* lockout) {
  if (a.obs == nullptr) return;   // mdr_buffers_t.obs == NULL: nobody reads the planes (observe -> act rollouts): 28 B per house-step not written
  float c0[VEC], c1[VEC], c2[VEC], c3[VEC], c4[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    c0[v] = (o[v].Ta + a.obs_tshift) * 0.2f;  // (house_temp - 20) / 5, utils.py:800-802
    c1[v] = (o[v].Tm + a.obs_tshift) * 0.2f;
    c2[v] = (o[v].flags & 1u) ? 1.0f : 0.0f;  // utils.py:823
    c3[v] = (o[v].flags & 2u) ? 1.0f : 0.0f;  // utils.py:824
    c4[v] = (float)o[v].sso / (float)lockout[v];  // utils.py:826-828
  }
  store_out<VEC>(a.obs + 0 * a.plane, i, c0);
  store_out<VEC>(a.obs + 1 * a.plane, i, c1);
  store_out<VEC>(a.obs + 2 * a.plane, i, c2);
  store_out<VEC>(a.obs + 3 * a.plane, i, c3);
  store_out<VEC>(a.obs + 4 * a.plane, i, c4);
}

template <int VEC>
__device__ __forceinline__ void store_reward_power(const StepArgs& a, int64_t i, const float* pen, double sum_pen,
                                                   float max_pen, float sig_term, float o_sig, float o_pow) {
  float r[VEC], c5[VEC], c6[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    r[v] = reward_value(a, pen[v], sum_pen, max_pen, sig_term);
    c5[v] = o_sig;
    c6[v] = o_pow;
  }
  store_out<VEC>(a.reward, i, r);
  if (a.obs == nullptr) return;
  store_out<VEC>(a.obs + 5 * a.plane, i, c5);
  store_out<VEC>(a.obs + 6 * a.plane, i, c6);
}

// Graph mode: a captured launch carries the pointers of table row 0; the device-resident cursor says where the episode is.
__device__ __forceinline__ void rebase(StepArgs& a) {
  if (a.cursor == nullptr) return;
  const int64_t off = (int64_t)min(a.cursor[0], a.cursor_max) * a.E;
  a.od_old += off;
  a.solar_new += off;
  a.sig_old += off;
  a.sig_new += off;
}


// Graph mode: the last workgroup of a step's last kernel moves the cursor on.  Every thread read the cursor first thing (rebase)
// and then either left the kernel or waits at the barrier below, so when the last workgroup's thread 0 has counted all arrivals
// nobody in this launch reads it any more; whatever reads it next is a later launch on the stream.  Saves each captured step a
// one-thread launch of its own (~2 us of node-to-node latency on a 6-10 us step).  Thread 0 of every workgroup must get here.
__device__ __forceinline__ void cursor_done(const StepArgs& a) {
  if (a.cursor_adv == nullptr) return;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const unsigned total = gridDim.x * gridDim.y * gridDim.z;
  if (atomicAdd(reinterpret_cast<unsigned*>(a.cursor_adv + 3), 1u) == total - 1u) {
    a.cursor_adv[3] = 0;
    a.cursor_adv[0] += 1;
    a.cursor_adv[1] += 1;
  }
}

}  // namespace mdr
