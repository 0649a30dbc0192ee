// Fused policy forward + action sampling (include/mdr_policy.h).
//
// Reference: PPO.select_action (agents/ppo.py:68-75) = Actor.forward (agents/network.py:14-33: Linear-ReLU-Linear-ReLU-
// Linear-softmax) + Categorical.sample, called once per agent and step on a batch of 1.  Here: one wavefront per tile of
// 32 agents, agents on the MFMA column (= lane) index, hidden units on the row index:
//
//   layer 1   H1[128 x 32] = W1e[128 x 2 S1] . Xe^T           v_mfma_f32_32x32x2_f32, 4 row blocks x S1 k-steps
//   layer 2   H2[128 x 32] = W2e[128 x 2 S2] . relu(H1)       the accumulator of layer 1 IS the B operand: a lane holds
//                                                             rows 32 kb + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) of its own
//                                                             column, so k-step q = (kb, reg) takes register [kb][reg] as it
//                                                             is and the weights are stored in that k order - no LDS, no
//                                                             lane movement between the layers
//   head      d = (W3e[0] - W3e[1]) . relu(H2)                VALU over the lane's 64 rows + one cross-half add
//             p0 = 1 / (1 + exp(-d)),  action = u < p0 ? 0 : 1
//
// Biases ride along: input feature F and hidden unit H are constant 1 (W1e / W2e carry the bias column and a row that
// reproduces the 1).  fp32 in, fp32 accumulate: the MFMA is a k-ordered fp32 fma chain, so the result differs from a
// torch fp32 forward only by summation order.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "../../include/mdr.h"
#include "../../include/mdr_policy.h"
#include "mdr_device.h"
#include "mdr_kernels.h"

namespace {

using mdr::philox4x32_10;
using mdr::u32x4;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVES = 8;   // per workgroup: two per SIMD, one hides the other's loads and VALU epilogue
constexpr uint32_t TAG_ACTION = 0x41435431u;

struct ActorArgs {
  const float* frag1;
  const float* frag2;
  const float* wdiff;
  const float* obs;
  uint8_t* action;
  float* a_prob;
  float* probs;
  int64_t A;
  int64_t ntiles;
  int64_t plane;   // 0: obs is [A][F] rows; > 0: feature-major [F][plane] (plane >= A): lanes of a group read consecutive floats
  int F, S1, S2;
  uint32_t k0, k1, step_lo, step_hi;
  const int32_t* step_dev;   // optional: added to the step counter on the device (graph replays: the env's time index)
  int greedy;                // action = argmax instead of a draw (DQNAgent.act)
  int64_t agent0;            // index of this launch's first agent in the whole batch (Philox counters stay functions of the batch index when a launch covers a slice)
  float* rows_out;           // observe -> act only, optional: the observation rows [A][51] in normStateDict order (the transition buffer's `state`)
};

// The Philox key of a draw, hidden from loop-invariant code motion: hipcc otherwise keeps the ten round keys (k + n W) of both
// words in 20 scalar registers for the whole kernel - beyond the scalar file, so they are parked in the lanes of a vector
// register and fetched back one v_readlane at a time.  Re-deriving them where a draw is made is 20 scalar adds.
__device__ __forceinline__ uint32_t loop_local(uint32_t x) {
  asm volatile("" : "+s"(x));
  return x;
}

// A lane-dependent value hidden from loop-invariant code motion: what is derived from it (LDS addresses of the row copy, 64-bit
// products for the Philox counter) is then re-derived where it is used - one or two vector instructions - instead of being
// hoisted out of the tile loop into registers the loop does not have (r02: those were the kernels' scratch spills).
__device__ __forceinline__ int tile_local(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// max(x, 0) in ONE instruction: on the bit pattern, as a signed integer (v_max_i32) - every negative float, -0 included, is a
// negative integer, every non-negative float keeps its bits.  fmaxf costs a canonicalising v_max_f32 x, x before the v_max_f32
// x, 0, and hipcc folds v_med3_f32(x, 0, inf) into that very pair (r02: 2442 relus of this file compiled to 4884 v_max_f32).
// Finite inputs: the same bits as fmaxf(x, 0); a NaN keeps its sign's side (no NaN reaches here: weights and features are finite).
__device__ __forceinline__ float relu(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// X1: compile-time bound on S1 (the lane's S1 input features are prefetched into registers, the next tile's while layer 2
// runs); 0 = any S1, features loaded as layer 1 consumes them.  S2C: compile-time S2, 0 = run-time.
template <int X1, int S2C>
__global__ __launch_bounds__(64 * WAVES) void k_actor_sample(ActorArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* f1 = lds;                       // [S1][64][4]
  float* f2 = f1 + a.S1 * 256;           // [S2][64][4]
  float* wd = f2 + a.S2 * 256;           // [4][16][2]
  const int tid = threadIdx.x;
  for (int i = tid * 4; i < a.S1 * 256; i += 64 * WAVES * 4) *reinterpret_cast<float4*>(f1 + i) = *reinterpret_cast<const float4*>(a.frag1 + i);
  for (int i = tid * 4; i < a.S2 * 256; i += 64 * WAVES * 4) *reinterpret_cast<float4*>(f2 + i) = *reinterpret_cast<const float4*>(a.frag2 + i);
  if (tid < 128) wd[tid] = a.wdiff[tid];
  __syncthreads();

  const int lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int64_t wave = (int64_t)blockIdx.x * WAVES + (tid >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  const int kbase = h * a.S1;            // this lane half's input features: [kbase, kbase + S1)
  const int S2 = S2C ? S2C : a.S2;
  constexpr int XR = X1 ? X1 : 1;
  float xr[XR];
  const int64_t fstride = a.plane ? a.plane : 1;   // distance between two features of one agent
  auto row_of = [&](int64_t t) {
    const int64_t agent = t * 32 + r;
    return a.obs + (agent < a.A ? agent : a.A - 1) * (a.plane ? 1 : (int64_t)a.F);
  };
  auto feature = [&](const float* x, int s) {
    const int k = kbase + s;
    return k < a.F ? x[k * fstride] : (k == a.F ? 1.0f : 0.0f);
  };
  auto prefetch = [&](int64_t t) {
    if (X1 == 0 || t >= a.ntiles) return;
    const float* x = row_of(t);
#pragma unroll
    for (int s = 0; s < XR; ++s)
      if (s < a.S1) xr[s] = feature(x, s);
  };
  prefetch(wave);
  for (int64_t t = wave; t < a.ntiles; t += nwaves) {
    const int64_t agent = t * 32 + r;
    const bool valid = agent < a.A;
    f32x16 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][i] = 0.0f;
    // ---- layer 1
    if (X1) {
#pragma unroll
      for (int s = 0; s < XR; ++s) {
        if (s < a.S1) {
          const float4 w = *reinterpret_cast<const float4*>(f1 + s * 256 + lane * 4);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, xr[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, xr[s], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, xr[s], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, xr[s], acc[3], 0, 0, 0);
        }
      }
    } else {
      const float* x = row_of(t);
      for (int s = 0; s < a.S1; ++s) {
        const float b = feature(x, s);
        const float4 w = *reinterpret_cast<const float4*>(f1 + s * 256 + lane * 4);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, b, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, b, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, b, acc[3], 0, 0, 0);
      }
    }
    // ---- layer 2: relu(H1) straight out of the accumulators
    f32x16 out[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) out[mb][i] = 0.0f;
    // The next tile's input features are fetched between the k-steps, one load per step: every such load touches 64
    // different cache lines (~64 cycles in the texture addresser), and a wave that issues them all back to back cannot
    // issue MFMAs meanwhile (in-order issue) - measured 5 % of the kernel against this interleaving.
    const bool more = X1 != 0 && t + nwaves < a.ntiles;
    const float* xn = row_of(more ? t + nwaves : t);
#pragma unroll
    for (int q = 0; q < 64; ++q) {
      if (X1 != 0 && q < XR && q < a.S1 && more) xr[q < XR ? q : 0] = feature(xn, q);
      if (q < S2) {
        const float b = relu(acc[q >> 4][q & 15]);
        const float4 w = *reinterpret_cast<const float4*>(f2 + q * 256 + lane * 4);
        out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, b, out[0], 0, 0, 0);
        out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, b, out[1], 0, 0, 0);
        out[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, b, out[2], 0, 0, 0);
        out[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, b, out[3], 0, 0, 0);
      }
    }
    // ---- head: d = logit0 - logit1 over this lane's 64 rows, then the other half's
    float d = 0.0f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) d = fmaf(wd[(mb * 16 + i) * 2 + h], relu(out[mb][i]), d);
    d += __shfl_xor(d, 32);
    const float p0 = 1.0f / (1.0f + expf(-d));
    const float p1 = 1.0f / (1.0f + expf(d));
    if (h == 0 && valid) {
      const u32x4 rnd = philox4x32_10((uint32_t)agent, (uint32_t)((uint64_t)agent >> 32), a.step_lo + (a.step_dev ? (uint32_t)*a.step_dev : 0u), TAG_ACTION ^ a.step_hi, loop_local(a.k0), loop_local(a.k1));
      const float u = ((float)(rnd.x >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const int act = a.greedy ? (d >= 0.0f ? 0 : 1) : (u < p0 ? 0 : 1);   // argmax keeps the first maximum, as torch.argmax
      a.action[agent] = (uint8_t)act;
      if (a.a_prob) a.a_prob[agent] = act ? p1 : p0;
      if (a.probs) {
        a.probs[agent * 2] = p0;
        a.probs[agent * 2 + 1] = p1;
      }
    }
  }
}

// ---- the same network on v_mfma_f32_16x16x4_f32: 16 agents per wavefront, hidden units in MB blocks of 16 rows.
// 101 rows pad to 112 (7 blocks) instead of 128, a lane carries 4 accumulator registers per block instead of 16 (56
// instead of 128 for both layers), so twice as many waves fit a SIMD.  C/D: col = lane & 15, row = 4 (lane >> 4) + reg;
// A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15].  Layer 2's k-step q = (kb, reg) takes register
// [kb][reg] of layer 1: lane group g = lane >> 4 supplies row 16 kb + 4 g + reg - again no lane movement.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef MDR_WAVES16
#define MDR_WAVES16 16
#endif
constexpr int WAVES16 = MDR_WAVES16;   // (-DMDR_WAVES16=12: experiment build)
#ifndef MDR_WAVES16_EXT
#define MDR_WAVES16_EXT 16
#endif
constexpr int WAVES16_EXT = MDR_WAVES16_EXT;   // the extended observe -> act form: four waves per SIMD as the default form (its rows are staged before layer 2, while that layer's accumulators do not exist yet); fewer where the windows do not fit the LDS

// MDR_ACTOR_FRAG16T: the last of the MB blocks holds at most 4 hidden units (the reference's 100 = 6 x 16 + 4) and runs on
// v_mfma_f32_4x4x1_16B_f32 instead - 16 independent 4x4 outer products, block = lane >> 2: D[v](lane) += A(lane (lane & ~3) + v) *
// B(lane).  With B the same operand the 16-row blocks take (lane = 16 g + agent: the value of k-index g) and A(lane) = W[unit
// 96 + (lane & 3)][that k], register v of a lane collects unit 96 + v for the lane's agent over its group's share of k; the four
// groups are summed once per layer.  A third of the time of the 16x16x4 instruction it replaces (4.9 vs 15.0 ns per SIMD,
// tools/probe/mfma4x4_probe.hip), for a block that would run three quarters empty.
template <bool TAIL_BLOCK>
__device__ __forceinline__ f32x4 mma16(float w, float b, f32x4 c) {
  if (TAIL_BLOCK) return __builtin_amdgcn_mfma_f32_4x4x1f32(w, b, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 sum_lane_groups(f32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] += __shfl_xor(v[i], 16);
    v[i] += __shfl_xor(v[i], 32);
  }
  return v;
}

__device__ __forceinline__ float pick_register(f32x4 v, int g) { return g == 0 ? v[0] : (g == 1 ? v[1] : (g == 2 ? v[2] : v[3])); }

// XS: feature registers per lane = the k-steps of layer 1 a form is compiled for - 16 (num_state <= 64) or 32 (<= 128: observations
// with the optional message columns, utils.py:858-866 - 81 / 91 / 121 features with ten senders).
template <int MB, bool TAIL, int XS = 16>
__global__ __launch_bounds__(64 * WAVES16) void k_actor_sample16(ActorArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* f1 = lds;                       // [S1][64][8]
  float* f2 = f1 + a.S1 * 512;           // [S2][64][8]
  float* wd = f2 + a.S2 * 512;           // [8][4][4]
  const int tid = threadIdx.x;
  for (int i = tid * 4; i < a.S1 * 512; i += 64 * WAVES16 * 4) *reinterpret_cast<float4*>(f1 + i) = *reinterpret_cast<const float4*>(a.frag1 + i);
  for (int i = tid * 4; i < a.S2 * 512; i += 64 * WAVES16 * 4) *reinterpret_cast<float4*>(f2 + i) = *reinterpret_cast<const float4*>(a.frag2 + i);
  if (tid < 388) wd[tid] = a.wdiff[tid];   // wdiff[128] | b1[8][4 groups][4] | b2[8][4][4] | b3[0] - b3[1], pad
  __syncthreads();

  const int lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * WAVES16 + (tid >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * WAVES16;
  const int kbase = g * a.S1;            // this lane group's input features: [kbase, kbase + S1)
  // Biases start the accumulators (the C operand) instead of riding as a constant-1 feature: no per-feature selects -
  // features past F are read at the clamped index F - 1 and meet zero weights.
  const f32x4* bias1 = reinterpret_cast<const f32x4*>(wd + 128) + g;   // [mb] at stride 4 vectors
  const f32x4* bias2 = reinterpret_cast<const f32x4*>(wd + 256) + g;
  const float bias3 = wd[384];
  float xr[XS];
  // Feature k of agent i sits `(i F + k) * 4` (rows) or `(k plane + i) * 4` (planes) bytes behind a.obs: 32-bit byte offsets from the
  // wave-uniform base (the launcher keeps the buffer below 4 GiB per launch), so a feature load is `global_load v, voffset, s[base]`
  // and what the tile loop keeps per feature is nothing - 64-bit per-feature addresses were 32 registers this kernel does not have.
  const char* const obs_base = reinterpret_cast<const char*>(a.obs);
  const uint32_t fstride4 = (uint32_t)(a.plane ? a.plane : 1) * 4u, astride4 = (uint32_t)(a.plane ? 1 : a.F) * 4u;
  auto row_of = [&](int64_t t) {
    const int64_t agent = t * 16 + r;
    return (uint32_t)(agent < a.A ? agent : a.A - 1) * astride4;
  };
  auto feature = [&](uint32_t x, int s) { return *reinterpret_cast<const float*>(obs_base + (x + (uint32_t)min(tile_local(kbase) + s, a.F - 1) * fstride4)); };
  if (wave < a.ntiles) {
    const uint32_t x = row_of(wave);
#pragma unroll
    for (int s = 0; s < XS; ++s) xr[s] = feature(x, s < a.S1 ? s : 0);
  }
  // One Philox call serves four tiles: lane group g draws for the tile this wave reaches g iterations from now (the
  // counter is that tile's agent index, so the draw stays a function of (seed, step, agent) alone).
  uint32_t rnd = 0;
  int it = 0;
  for (int64_t t = wave; t < a.ntiles; t += nwaves, ++it) {
    const int64_t agent = t * 16 + r;
    const bool valid = agent < a.A;
    if ((it & 3) == 0) {
      const int64_t ag = a.agent0 + (t + tile_local(g) * nwaves) * 16 + r;
      rnd = philox4x32_10((uint32_t)ag, (uint32_t)((uint64_t)ag >> 32), a.step_lo + (a.step_dev ? (uint32_t)*a.step_dev : 0u), TAG_ACTION ^ a.step_hi, loop_local(a.k0), loop_local(a.k1)).x;
    }
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = bias1[mb * 4];
    // ---- layer 1
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      if (s < a.S1) {
        const float4 w0 = *reinterpret_cast<const float4*>(f1 + s * 512 + lane * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(f1 + s * 512 + lane * 8 + 4);
        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = mb == MB - 1 ? mma16<TAIL>(w[mb], xr[s], acc[mb]) : mma16<false>(w[mb], xr[s], acc[mb]);
      }
    }
    if (TAIL) acc[MB - 1] = sum_lane_groups(acc[MB - 1]);   // units 96 + v of the lane's agent, in every lane group (bias: group 0 brought it)
    // ---- layer 2 (the next tile's features are loaded between its k-steps, one per step)
    const bool more = t + nwaves < a.ntiles;
    const uint32_t xn = row_of(more ? t + nwaves : t);
    f32x4 out[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) out[mb] = bias2[mb * 4];
#pragma unroll
    for (int q = 0; q < 4 * MB; ++q) {
      if (q < XS && q < a.S1 && more) xr[q < XS ? q : 0] = feature(xn, q);
      if (q < a.S2) {
        const float b = relu(TAIL && q >= 4 * (MB - 1) ? pick_register(acc[MB - 1], g) : acc[q >> 2][q & 3]);   // tail: k-index g is unit 96 + g
        const float4 w0 = *reinterpret_cast<const float4*>(f2 + q * 512 + lane * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(f2 + q * 512 + lane * 8 + 4);
        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) out[mb] = mb == MB - 1 ? mma16<TAIL>(w[mb], b, out[mb]) : mma16<false>(w[mb], b, out[mb]);
      }
    }
    if (TAIL) out[MB - 1] = sum_lane_groups(out[MB - 1]);
#pragma unroll
    for (int q = 4 * MB; q < XS; ++q)      // feature registers beyond layer 2's k-steps (XS = 32 with seven blocks)
      if (q < a.S1 && more) xr[q] = feature(xn, q);
    // ---- head: this lane's 4 MB rows, then the other three lane groups'
    float d = 0.0f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int i = 0; i < 4; ++i) d = fmaf(wd[(mb * 4 + i) * 4 + g], relu(out[mb][i]), d);
    d += __shfl_xor(d, 16);
    d += __shfl_xor(d, 32);
    d += bias3;
    const float e = expf(-d);                          // exp(l1 - l0); inf for d < -88: p0 = 0, p1 = 1
    const float p0 = 1.0f / (1.0f + e);
    const float p1 = e > 1e30f ? 1.0f : e * p0;
    const uint32_t draw = (uint32_t)__shfl((int)rnd, r + 16 * (it & 3));   // the group that drew for this tile
    if (g == 0 && valid) {
      const float u = ((float)(draw >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const int act = a.greedy ? (d >= 0.0f ? 0 : 1) : (u < p0 ? 0 : 1);   // argmax keeps the first maximum, as torch.argmax
      a.action[agent] = (uint8_t)act;
      if (a.a_prob) a.a_prob[agent] = act ? p1 : p0;
      if (a.probs) {
        a.probs[agent * 2] = p0;
        a.probs[agent * 2 + 1] = p1;
      }
    }
  }
}

// ---- the same network on v_mfma_f32_16x16x32_bf16 with every fp32 operand split into two bf16 halves, x = xh + xl
// (xh = bf16(x), xl = bf16(x - xh): 16 significand bits): w x ~ wh xh + wl xh + wh xl, three MFMAs at 16x the fp32
// matrix rate, fp32 accumulation; the dropped wl xl term is 2^-16 of the product.  16 agents per wavefront, K = 32 per
// k-step.  A[row = lane & 15][k = 8 (lane >> 4) + j], B[k = 8 (lane >> 4) + j][col = lane & 15], j < 8; C/D as 16x16x4.
// Layer 1: k-step s, lane group g, element j <-> input feature (4 s + g) 8 + j (a contiguous run per lane).
// Layer 2: k-step s covers the row blocks 2 s and 2 s + 1 of layer 1: element j <-> row 16 (2 s + (j >> 2)) + 4 g + (j & 3),
// i.e. register [2 s + (j >> 2)][j & 3] of the lane itself - again no LDS and no lane movement for the activations.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// (bf16(hi) << 16) | bf16(lo), round to nearest even.  Inline assembly is opaque to hipcc's hazard recogniser: a VGPR written here and
// read as an MFMA operand by the very next instruction needs two wait states that nobody else inserts (round 1 shipped this
// statement without them; the stale-operand reads surfaced when a new kernel variant changed the instruction schedule:
// half the agents of the second column block came out with garbage logits).  Hence the `s_nop 1` INSIDE the string.  The
// plain vector conversion (__builtin_convertvector to bf16x2) is hazard-safe too and selects the same instruction, but lets the
// scheduler hoist the conversions until k_actor_sample_bf16 spills (1.1 KB of scratch per lane, 6x slower).
// four packed conversions in ONE statement: the last write is two wait states away from whatever follows the statement, the
// earlier ones further - one `s_nop 1` instead of four
__device__ __forceinline__ void cvt_pk_bf16_x4(const float* v, uint32_t* out) {
  asm("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\tv_cvt_pk_bf16_f32 %2, %8, %9\n\tv_cvt_pk_bf16_f32 %3, %10, %11\n\ts_nop 1"
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
}

// eight fp32 values -> their bf16 head and tail fragments
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
  uint32_t h[4], l[4];
  cvt_pk_bf16_x4(v, h);
  float res[8];
#pragma unroll
  for (int p = 0; p < 4; ++p) {   // x - float(bf16(x)); scalar subtractions: packed f32 VALU issues slowly beside MFMAs (MI355X_MICROARCH.md)
    res[2 * p] = v[2 * p] - __uint_as_float(h[p] << 16);
    res[2 * p + 1] = v[2 * p + 1] - __uint_as_float(h[p] & 0xFFFF0000u);
  }
  cvt_pk_bf16_x4(res, l);
  hi = uint4{h[0], h[1], h[2], h[3]};
  lo = uint4{l[0], l[1], l[2], l[3]};
}

constexpr int WAVESB = 8;    // 2 per SIMD: two column blocks of accumulators (~200 registers per lane)
constexpr int NCB = 2;       // 16-agent column blocks per wavefront: every weight fragment read from LDS feeds NCB MFMAs (with one block
                             // the fragment reads, 84 KB per 16 agents, kept the LDS busier than the matrix pipe)

// XF: features per lane and column block = 8 per k-step of layer 1 - 16 (num_state <= 64) or 32 (<= 128).  Both forms hold 16 feature
// registers per column block: the 32-feature form converts its first two k-steps to bf16 fragments, re-uses the registers for the
// loads of the tile's other 16 features and runs the first two k-steps' matrix instructions while those are in flight.
template <int MB, int XF = 16>
__global__ __launch_bounds__(64 * WAVESB) void k_actor_sample_bf16(ActorArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int S2B = (MB + 1) / 2;
  uint4* f1 = reinterpret_cast<uint4*>(lds);                    // [S1][8][2][64] fragments of 8 bf16 (8 row blocks stored, MB used)
  uint4* f2 = f1 + a.S1 * 1024;                                 // [S2B][8][2][64]
  float* wd = reinterpret_cast<float*>(f2 + S2B * 1024);        // [8][4][4] head weights, then the biases (below)
  const int tid = threadIdx.x;
  const uint4* g1 = reinterpret_cast<const uint4*>(a.frag1);
  const uint4* g2 = reinterpret_cast<const uint4*>(a.frag2);
  for (int i = tid; i < a.S1 * 1024; i += 64 * WAVESB) f1[i] = g1[i];
  for (int i = tid; i < S2B * 1024; i += 64 * WAVESB) f2[i] = g2[i];
  if (tid < 388) wd[tid] = a.wdiff[tid];   // wdiff[128] | b1[8][4 groups][4] | b2[8][4][4] | b3[0] - b3[1], pad
  __syncthreads();

  const int lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * WAVESB + (tid >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * WAVESB;
  const int64_t fstride = a.plane ? a.plane : 1;
  // Biases start the accumulators (the C operand) instead of riding as a constant-1 feature: no per-feature selects -
  // features past F are read at the clamped index F - 1 and meet zero weights.
  const f32x4* bias1 = reinterpret_cast<const f32x4*>(wd + 128) + g;   // [mb] at stride 4 vectors
  const f32x4* bias2 = reinterpret_cast<const f32x4*>(wd + 256) + g;
  const float bias3 = wd[384];
  float xr[NCB][16];                      // two k-steps at a time: 8 features per step and column block
  auto row_of = [&](int64_t t, int c) {   // a tile = NCB * 16 consecutive agents
    const int64_t agent = (t * NCB + c) * 16 + r;
    return a.obs + (agent < a.A ? agent : a.A - 1) * (a.plane ? 1 : (int64_t)a.F);
  };
  auto feature = [&](const float* x, int i) {   // i = 8 s + j
    const int k = ((i >> 3) * 4 + g) * 8 + (i & 7);
    return x[min(k, a.F - 1) * fstride];
  };
  if (wave < a.ntiles) {
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const float* x = row_of(wave, c);
#pragma unroll
      for (int i = 0; i < 16; ++i) xr[c][i] = feature(x, i < 8 * a.S1 ? i : 0);
    }
  }
  // One Philox call per agent serves four tiles: lane group g draws for the tile this wave reaches g iterations from now
  // (the counter is that agent's index, so the draw stays a function of (seed, step, agent) alone).
  uint32_t rnd[NCB] = {};
  int it = 0;
  for (int64_t t = wave; t < a.ntiles; t += nwaves, ++it) {
    if ((it & 3) == 0) {
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        const int64_t ag = ((t + tile_local(g) * nwaves) * NCB + c) * 16 + r;
        rnd[c] = philox4x32_10((uint32_t)ag, (uint32_t)((uint64_t)ag >> 32), a.step_lo + (a.step_dev ? (uint32_t)*a.step_dev : 0u),
                               TAG_ACTION ^ a.step_hi, loop_local(a.k0), loop_local(a.k1)).x;
      }
    }
    f32x4 acc[NCB][MB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[c][mb] = bias1[mb * 4];
    // ---- layer 1
    auto mma1 = [&](int s, const bf16x8* Bh, const bf16x8* Bl) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, f1[((s * 8 + mb) * 2 + 0) * 64 + lane]);
        const bf16x8 Al = __builtin_bit_cast(bf16x8, f1[((s * 8 + mb) * 2 + 1) * 64 + lane]);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          acc[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh[c], acc[c][mb], 0, 0, 0);
          acc[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh[c], acc[c][mb], 0, 0, 0);
          acc[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl[c], acc[c][mb], 0, 0, 0);
        }
      }
    };
    auto split_step = [&](int s, bf16x8* Bh, bf16x8* Bl) {   // k-step s of the registers' pair of k-steps
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        uint4 bh, bl;
        split8(xr[c] + 8 * s, bh, bl);
        Bh[c] = __builtin_bit_cast(bf16x8, bh);
        Bl[c] = __builtin_bit_cast(bf16x8, bl);
      }
    };
    if (XF == 16) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s < a.S1) {
          bf16x8 Bh[NCB], Bl[NCB];
          split_step(s, Bh, Bl);
          mma1(s, Bh, Bl);
        }
      }
    } else {   // a.S1 = 3 | 4
      bf16x8 Bh0[NCB], Bl0[NCB], Bh1[NCB], Bl1[NCB];
      split_step(0, Bh0, Bl0);
      split_step(1, Bh1, Bl1);
#pragma unroll
      for (int c = 0; c < NCB; ++c) {      // the registers are free: this tile's features 64 .. 127
        const float* x = row_of(t, c);
#pragma unroll
        for (int i = 0; i < 16; ++i) xr[c][i] = feature(x, 16 + i < 8 * a.S1 ? 16 + i : 16);
      }
      mma1(0, Bh0, Bl0);
      mma1(1, Bh1, Bl1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (2 + s < a.S1) {
          bf16x8 Bh[NCB], Bl[NCB];
          split_step(s, Bh, Bl);
          mma1(2 + s, Bh, Bl);
        }
      }
    }
    // ---- layer 2 (the next tile's features are loaded between its k-steps)
    const bool more = t + nwaves < a.ntiles;
    const float* xn[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) xn[c] = row_of(more ? t + nwaves : t, c);
    f32x4 out[NCB][MB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) out[c][mb] = bias2[mb * 4];
#pragma unroll
    for (int s = 0; s < S2B; ++s) {
      if (more) {
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
          for (int i = 4 * s; i < 4 * s + 4 && i < 16; ++i)
            if (i < 8 * a.S1) xr[c][i] = feature(xn[c], i);
      }
      bf16x8 Bh[NCB], Bl[NCB];
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (2 * s + (j >> 2) < MB) ? relu(acc[c][2 * s + (j >> 2) < MB ? 2 * s + (j >> 2) : 0][j & 3]) : 0.0f;
        uint4 bh, bl;
        split8(v, bh, bl);
        Bh[c] = __builtin_bit_cast(bf16x8, bh);
        Bl[c] = __builtin_bit_cast(bf16x8, bl);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, f2[((s * 8 + mb) * 2 + 0) * 64 + lane]);
        const bf16x8 Al = __builtin_bit_cast(bf16x8, f2[((s * 8 + mb) * 2 + 1) * 64 + lane]);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          out[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh[c], out[c][mb], 0, 0, 0);
          out[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh[c], out[c][mb], 0, 0, 0);
          out[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl[c], out[c][mb], 0, 0, 0);
        }
      }
    }
    if (more && S2B * 4 < 16) {
#pragma unroll
      for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int i = S2B * 4; i < 16; ++i)
          if (i < 8 * a.S1) xr[c][i] = feature(xn[c], i);
    }
    // ---- head: this lane's 4 MB rows, then the other three lane groups'
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const int64_t agent = (t * NCB + c) * 16 + r;
      const bool valid = agent < a.A;
      float d = 0.0f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 4; ++i) d = fmaf(wd[(mb * 4 + i) * 4 + g], relu(out[c][mb][i]), d);
      d += __shfl_xor(d, 16);
      d += __shfl_xor(d, 32);
      d += bias3;
      const float e = expf(-d);                          // exp(l1 - l0); inf for d < -88: p0 = 0, p1 = 1
      const float p0 = 1.0f / (1.0f + e);
      const float p1 = e > 1e30f ? 1.0f : e * p0;
      const uint32_t draw = (uint32_t)__shfl((int)rnd[c], r + 16 * (it & 3));   // the group that drew for this tile
      if (g == 0 && valid) {
        const float u = ((float)(draw >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const int act = a.greedy ? (d >= 0.0f ? 0 : 1) : (u < p0 ? 0 : 1);   // argmax keeps the first maximum, as torch.argmax
        a.action[agent] = (uint8_t)act;
        if (a.a_prob) a.a_prob[agent] = act ? p1 : p0;
        if (a.probs) {
          a.probs[agent * 2] = p0;
          a.probs[agent * 2 + 1] = p1;
        }
      }
    }
  }
}


// =================================================================================================================
// Observe -> act: the same forwards with the B operand built from the compact state instead of observation rows.
//
// Per wavefront and tile of TILE consecutive agents (= houses h0 .. h0 + TILE - 1 of ONE env: N % TILE == 0) the lanes
// p < TILE + 10 load the nine per-house values of house h0 - 5 + p (circular in the env), turn them into
//   the house's SingleHouse.message record   (Ta - target) / 5 | seconds_since_off | curr / norm | max / norm      (env 624-662)
//   its own normStateDict features           (Ta-20)/5 (Tm-20)/5 (target-20)/5 deadband cap/def on lock sso/L L/L S/norm P/norm
// with the very expressions of obs_features() / sender_from_global() in mdr_kernels.hip, and write them into the wave's LDS
// window as ready rows  row[r] = [ message of slot 0..9 (40 floats) | own (11) | L | 1/L | pad ]  (OBS_ROW floats): the record
// of the house at window position p is message slot m of the agents r = p - m - (m >= 5) - up to ten 16-byte stores.
// The matrix-core forward then reads each lane's feature run with 16-byte LDS loads; the sender's seconds_since_off is
// divided by the RECEIVER's lockout (utils.py:849-851) on the way, with div_by_lockout() as k_obs_rows_default does.
// Feature k of a row is normStateDict index (k < 40 ? 11 + k : k - 40): the weights come packed in that order
// (mdr_actor_t.feature_order = 1).  The loads for the next tile are issued before layer 1 and land during it; the rows
// are staged between the k-steps of layer 2, read back before the head, so one window per wave suffices.
// =================================================================================================================
constexpr int OBS_HALO = 5, OBS_C = 10, OBS_ROW = 56, OBS_PAD = 16;   // floats; 56 = 40 + 11 + L + 1/L + 3 (16-byte rows)
// The extended form (ObserveArgs.ext; template parameter EXT): optional state columns, c != 10 circular neighbours, link defects.
// A row is [4 c message floats | own features in normStateDict order | zeros up to 64 | L | 1 / L | pad]: feature k of a row is
// normStateDict index (k < 4 c ? own + k : k - 4 c), the packed weights follow (mdr_actor_t.feature_order = 1), F = 4 c + own <= 64.
// The row stride of the extended form is a run-time value (ObserveArgs.row; L and 1 / L sit in its floats row - 2, row - 1): the
// features rounded up to 16 bytes, so that as many windows fit the LDS as for the default shape where the shape allows it.
constexpr int OBS_MAX_C = 13;                         // (run-time strides: 4 c + own + 2 rounded up to 4 * odd, at most 68)
typedef float v4f_nt __attribute__((ext_vector_type(4)));

// between a wave's window stores and its loads of what OTHER lanes stored
__device__ __forceinline__ void observe_window_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

struct HouseRegs {
  float Ta, Tm, tg, db, cap, pm;
  int sso, lk;
  unsigned fl;
  float sig, pw;
  float Ua, Cm, Ca, Hm, COP, latent;   // EXT: the thermal / hvac columns, already divided by their defaults
  float x_od, x_sd, x_cd, x_sh, x_ch, x_sol;   // EXT: the per-env columns (k_observe_env_extras)
};

__device__ __forceinline__ const double* observe_sig_row(const mdr::ObserveArgs& o) {
  if (o.cursor == nullptr) return o.sig_now;
  return o.sig_now + (int64_t)min(o.cursor[0], o.cursor_max + 1) * o.E;   // as rebase(ObsArgs&) in mdr_kernels.hip
}

// A load at a 32-bit byte offset from a wave-uniform base (`global_load v, voffset, s[base:base+1]`): the extended forms read up to 15
// per-house arrays at ONE house index - one offset register instead of a 64-bit address per array (the launcher keeps 4 A below 2^32)
template <class T>
__device__ __forceinline__ T ld32(const T* base, uint32_t byte_offset) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_offset);
}

// the per-house / per-env columns of the extended form (thermal, hvac: per house, divided by their defaults; the rest: k_observe_env_extras)
template <bool ENV = true>   // ENV = false: the per-env columns come later (observe_env_extras_now: whole-tile staging, one env per tile)
__device__ __forceinline__ void observe_load_extras(const mdr::ObserveArgs& o, HouseRegs& r, uint32_t i4, int e) {
  if (o.f_thermal) {
    if (ENV) r.x_od = o.env_extra_a[e];
    r.Ua = ld32(o.Ua, i4) * o.inv_Ua;
    r.Cm = ld32(o.Cm, i4) * o.inv_Cm;
    r.Ca = ld32(o.Ca, i4) * o.inv_Ca;
    r.Hm = ld32(o.Hm, i4) * o.inv_Hm;
  }
  if (o.f_hvac) {
    r.COP = ld32(o.COP, i4) * o.inv_COP;
    r.latent = ld32(o.latent, i4) * o.inv_latent;
  }
  if (!ENV) return;
  if (o.f_day) {
    r.x_sd = o.env_extra_a[(int64_t)o.E + e];
    r.x_cd = o.env_extra_a[2 * (int64_t)o.E + e];
  }
  if (o.f_hour) {
    r.x_sh = o.env_extra_a[3 * (int64_t)o.E + e];
    r.x_ch = o.env_extra_b[e];
  }
  if (o.f_solar) r.x_sol = o.env_extra_b[(int64_t)o.E + e];
}

// The per-env columns of ONE env, fetched when its tile's rows are staged (whole-tile staging: up to six registers less to carry
// from the loads before layer 1 to the staging behind it; the index is wave-uniform)
__device__ __forceinline__ void observe_env_extras_now(const mdr::ObserveArgs& o, HouseRegs& r, int e_uniform) {
  const int e = __builtin_amdgcn_readfirstlane(e_uniform);
  if (o.f_thermal) r.x_od = o.env_extra_a[e];
  if (o.f_day) {
    r.x_sd = o.env_extra_a[(int64_t)o.E + e];
    r.x_cd = o.env_extra_a[2 * (int64_t)o.E + e];
  }
  if (o.f_hour) {
    r.x_sh = o.env_extra_a[3 * (int64_t)o.E + e];
    r.x_ch = o.env_extra_b[e];
  }
  if (o.f_solar) r.x_sol = o.env_extra_b[(int64_t)o.E + e];
}

// the own features of the extended form in normStateDict order (obs_features() in mdr_kernels.hip), optional ones where they belong;
// L and 1 / L behind them in the row's last two floats
__device__ __forceinline__ void observe_write_own_ext(const mdr::ObserveArgs& o, const HouseRegs& r, float* row, int c, int ROW) {
  const float L = (float)r.lk;
  float* own = row + 4 * c;
  int j = 0;
  own[j++] = (r.Ta + o.obs_tshift) * 0.2f;
  own[j++] = (r.Tm + o.obs_tshift) * 0.2f;
  own[j++] = (r.tg + o.obs_tshift) * 0.2f;
  if (o.f_thermal) own[j++] = r.x_od;
  own[j++] = r.db;
  if (o.f_day) {
    own[j++] = r.x_sd;
    own[j++] = r.x_cd;
  }
  if (o.f_hour) {
    own[j++] = r.x_sh;
    own[j++] = r.x_ch;
  }
  if (o.f_solar) own[j++] = r.x_sol;
  own[j++] = r.cap * o.inv_cap;
  if (o.f_thermal) {
    own[j++] = r.Ua;
    own[j++] = r.Cm;
    own[j++] = r.Ca;
    own[j++] = r.Hm;
  }
  if (o.f_hvac) {
    own[j++] = r.COP;
    own[j++] = r.latent;
  }
  own[j++] = (r.fl & 1u) ? 1.0f : 0.0f;
  own[j++] = (r.fl & 2u) ? 1.0f : 0.0f;
  own[j++] = (float)r.sso / L;
  own[j++] = L / L;
  own[j++] = r.sig;
  own[j++] = r.pw;
  row[ROW - 2] = L;
  row[ROW - 1] = 1.0f / L;
}

// A tile of TILE consecutive agents inside ONE env (N a multiple of the tile): lane l stages the house at window position l, the window
// being the tile's houses and the c around them (`before` of them in front; the default: 5 + 5)
template <int TILE, bool EXT = false>
__device__ __forceinline__ HouseRegs observe_load(const mdr::ObserveArgs& o, const double* sig_row, int e, int h0, int lane) {
  const int before = EXT ? o.before : OBS_HALO, c = EXT ? o.c : 2 * OBS_HALO;
  HouseRegs r{};
  if (lane < TILE + c) {
    int hh = h0 - before + lane;
    hh += hh < 0 ? o.N : 0;
    hh -= hh >= o.N ? o.N : 0;
    const int64_t i = (int64_t)e * o.N + hh;
    if (EXT) {
      const uint32_t i1 = (uint32_t)i, i4 = i1 << 2;
      r.Ta = ld32(o.Ta, i4);
      r.Tm = ld32(o.Tm, i4);
      r.tg = ld32(o.target, i4);
      r.db = ld32(o.deadband, i4);
      r.cap = ld32(o.capacity, i4);
      r.pm = ld32(o.P_max, i4);
      r.sso = ld32(o.sso, i4);
      r.lk = ld32(o.lockout, i4);
      r.fl = ld32(o.flags, i1);
      observe_load_extras<false>(o, r, i4, e);
    } else {
      r.Ta = o.Ta[i];
      r.Tm = o.Tm[i];
      r.tg = o.target[i];
      r.db = o.deadband[i];
      r.cap = o.capacity[i];
      r.pm = o.P_max[i];
      r.sso = o.sso[i];
      r.lk = o.lockout[i];
      r.fl = o.flags[i];
    }
  }
  r.sig = (float)(sig_row[e] * o.inv_obs_norm);   // utils.py:832-841, per env
  r.pw = (float)(o.P[e] * o.inv_obs_norm);
  return r;
}

template <int TILE, bool EXT = false, int ROWC = 0>
__device__ __forceinline__ void observe_stage(const mdr::ObserveArgs& o, const HouseRegs& r, float* rows, int lane, int e = 0) {
  const float4 rec = make_float4((r.Ta - r.tg) * 0.2f, (float)r.sso, ((r.fl & 1u) ? r.pm : 0.0f) * o.inv_norm_reg, r.pm * o.inv_norm_reg);
  if (EXT) {
    const int before = o.before, c = o.c;
    const int ROW = ROWC ? ROWC : o.row;   // (the fp32 forms know their stride at compile time)
    if (lane >= TILE + c) return;
#pragma unroll
    for (int m = 0; m < OBS_MAX_C; ++m) {
      if (m >= c) break;
      const int off = m < before ? m - before : m - before + 1;   // slot m listens to house h + off (env 816-828)
      const int rr = lane - before - off;                         // ... so this house is slot m of the agent at tile row rr
      if (rr >= 0 && rr < TILE) *reinterpret_cast<float4*>(rows + rr * ROW + 4 * m) = rec;
    }
    const int rr = lane - before;
    HouseRegs own = r;
    observe_env_extras_now(o, own, e);
    if (rr >= 0 && rr < TILE) observe_write_own_ext(o, own, rows + rr * ROW, c, ROW);
    return;
  }
  if (lane >= TILE + 2 * OBS_HALO) return;
#pragma unroll
  for (int m = 0; m < OBS_C; ++m) {
    const int rr = lane - m - (m >= OBS_HALO ? 1 : 0);   // the agent this house is sender slot m of (env 816-828)
    if (rr >= 0 && rr < TILE) *reinterpret_cast<float4*>(rows + rr * OBS_ROW + 4 * m) = rec;
  }
  const int rr = lane - OBS_HALO;
  if (rr >= 0 && rr < TILE) {
    const float L = (float)r.lk;
    float* own = rows + rr * OBS_ROW + 4 * OBS_C;
    *reinterpret_cast<float4*>(own) = make_float4((r.Ta + o.obs_tshift) * 0.2f, (r.Tm + o.obs_tshift) * 0.2f, (r.tg + o.obs_tshift) * 0.2f, r.db);
    *reinterpret_cast<float4*>(own + 4) = make_float4(r.cap * o.inv_cap, (r.fl & 1u) ? 1.0f : 0.0f, (r.fl & 2u) ? 1.0f : 0.0f, (float)r.sso / L);
    *reinterpret_cast<float4*>(own + 8) = make_float4(L / L, r.sig, r.pw, L);
    own[12] = 1.0f / L;
  }
}

// ---- any cluster size (N >= 11): a tile of TILE consecutive agents may start anywhere in an env and span several (the reference
// trains with 20 houses and deploys with 50).  The tile is cut into per-env segments; a segment of `len` houses [hs, hs + len)
// stages the window of houses hs - 5 .. hs + len + 4 (circular) - or the whole env when that wraps onto itself (len + 10 >= N).
// Windows are laid out one after the other over the lanes (at most 52 lanes for TILE = 32, 36 for TILE = 16: checked for every
// N and tile start); a staging lane keeps (house, segment bounds, first tile row of the segment) with its loaded values.
struct SegSlot {
  int j, hs, len, rb;   // this lane's house in its segment's env; the segment's receivers [hs, hs + len) = tile rows [rb, rb + len)
  bool live;
  int e;                // ... and that env (the extended form fetches its per-env columns when the rows are staged)
};

template <int TILE, bool EXT = false, bool TABLE = false>
__device__ __forceinline__ HouseRegs observe_load_gen(const mdr::ObserveArgs& o, const double* sig_row, int e0, int h0, int64_t a0, int64_t A,
                                                      int lane, SegSlot& slot) {
  // senders before the house / in all (env 816-828).  TABLE (link tables, random_sample): the window is the tile's houses alone -
  // one lane per agent - and the senders' records are gathered when the rows are staged
  const int before = TABLE ? 0 : (EXT ? o.before : OBS_HALO), c = TABLE ? 0 : (EXT ? o.c : 2 * OBS_HALO);
  HouseRegs r{};
  slot = SegSlot{0, 0, 0, 0, false, 0};
  // the segment walk is wave-uniform: keep it on the scalar unit (the tile index comes out of threadIdx, which the compiler
  // cannot see is uniform across the wave)
  int rem = __builtin_amdgcn_readfirstlane((int)((A - a0) < (int64_t)TILE ? (A - a0) : (int64_t)TILE));
  int e = __builtin_amdgcn_readfirstlane(e0), hs = __builtin_amdgcn_readfirstlane(h0), wb = 0, rb = 0, my_e = 0;
#pragma unroll 1
  while (rem > 0) {                          // at most 4 segments
    const int len = min(o.N - hs, rem);
    const bool whole = len + c >= o.N;
    const int start = whole ? 0 : (hs - before + o.N) % o.N;
    const int wlen = whole ? o.N : len + c;
    if (lane >= wb && lane < wb + wlen) {
      int j = start + (lane - wb);
      j -= j >= o.N ? o.N : 0;
      slot = SegSlot{j, hs, len, rb, true, e};
      my_e = e;
    }
    wb += wlen;
    rb += len;
    rem -= len;
    hs = 0;
    e += 1;
  }
  if (slot.live) {
    const int64_t i = (int64_t)my_e * o.N + slot.j;
    if (EXT) {
      const uint32_t i1 = (uint32_t)i, i4 = i1 << 2;
      r.Ta = ld32(o.Ta, i4);
      r.Tm = ld32(o.Tm, i4);
      r.tg = ld32(o.target, i4);
      r.db = ld32(o.deadband, i4);
      r.cap = ld32(o.capacity, i4);
      r.pm = ld32(o.P_max, i4);
      r.sso = ld32(o.sso, i4);
      r.lk = ld32(o.lockout, i4);
      r.fl = ld32(o.flags, i1);
      observe_load_extras<false>(o, r, i4, my_e);
    } else {
      r.Ta = o.Ta[i];
      r.Tm = o.Tm[i];
      r.tg = o.target[i];
      r.db = o.deadband[i];
      r.cap = o.capacity[i];
      r.pm = o.P_max[i];
      r.sso = o.sso[i];
      r.lk = o.lockout[i];
      r.fl = o.flags[i];
    }
    r.sig = (float)(sig_row[my_e] * o.inv_obs_norm);
    r.pw = (float)(o.P[my_e] * o.inv_obs_norm);
  }
  return r;
}

template <bool EXT = false, int ROWC = 0, bool TABLE = false>
__device__ __forceinline__ void observe_stage_gen(const mdr::ObserveArgs& o, const HouseRegs& r, const SegSlot& slot, float* rows) {
  if (!slot.live) return;
  const int ROW = ROWC ? ROWC : (EXT ? o.row : OBS_ROW);   // (the fp32 extended forms know their stride at compile time)
  const int before = EXT ? o.before : OBS_HALO, c = EXT ? o.c : OBS_C;
  const float4 rec = make_float4((r.Ta - r.tg) * 0.2f, (float)r.sso, ((r.fl & 1u) ? r.pm : 0.0f) * o.inv_norm_reg, r.pm * o.inv_norm_reg);
#pragma unroll
  for (int m = 0; m < (EXT ? OBS_MAX_C : OBS_C); ++m) {
    if (TABLE || (EXT && m >= c)) break;
    const int off = m < before ? m - before : m - before + 1;   // slot m listens to house h + off (env 816-828)
    int h = slot.j - off;                                      // ... so this house is slot m of house j - off
    h += h < 0 ? o.N : 0;
    h -= h >= o.N ? o.N : 0;
    const int k = h - slot.hs;
    if (k >= 0 && k < slot.len) *reinterpret_cast<float4*>(rows + (slot.rb + k) * ROW + 4 * m) = rec;
  }
  const int k = slot.j - slot.hs;
  if (EXT) {
    if (TABLE) {   // this lane's house is a receiver (the window holds nothing else): its c senders by the table, four at a time
      float* row = rows + (slot.rb + k) * ROW;
      const int32_t* ids = o.links + (int64_t)slot.e * o.links_env_stride + (int64_t)slot.j * c;
      const float4* recs = reinterpret_cast<const float4*>(o.msg_rec) + (int64_t)slot.e * o.N;
#pragma unroll 1
      for (int m0 = 0; m0 < c; m0 += 4) {
        const int last = c - 1;
        const int s0 = ids[m0], s1 = ids[min(m0 + 1, last)], s2 = ids[min(m0 + 2, last)], s3 = ids[min(m0 + 3, last)];
        const float4 r0 = recs[s0], r1 = recs[s1], r2 = recs[s2], r3 = recs[s3];
        float* dst = row + 4 * m0;
        *reinterpret_cast<float4*>(dst) = r0;
        if (m0 + 1 < c) *reinterpret_cast<float4*>(dst + 4) = r1;
        if (m0 + 2 < c) *reinterpret_cast<float4*>(dst + 8) = r2;
        if (m0 + 3 < c) *reinterpret_cast<float4*>(dst + 12) = r3;
      }
    }
    if (k >= 0 && k < slot.len) {
      HouseRegs own = r;
      const int e = slot.e;      // per lane here: a tile may span envs
      if (o.f_thermal) own.x_od = o.env_extra_a[e];
      if (o.f_day) {
        own.x_sd = o.env_extra_a[(int64_t)o.E + e];
        own.x_cd = o.env_extra_a[2 * (int64_t)o.E + e];
      }
      if (o.f_hour) {
        own.x_sh = o.env_extra_a[3 * (int64_t)o.E + e];
        own.x_ch = o.env_extra_b[e];
      }
      if (o.f_solar) own.x_sol = o.env_extra_b[(int64_t)o.E + e];
      observe_write_own_ext(o, own, rows + (slot.rb + k) * ROW, c, ROW);
    }
    return;
  }
  if (k >= 0 && k < slot.len) {
    const float L = (float)r.lk;
    float* own = rows + (slot.rb + k) * OBS_ROW + 4 * OBS_C;
    *reinterpret_cast<float4*>(own) = make_float4((r.Ta + o.obs_tshift) * 0.2f, (r.Tm + o.obs_tshift) * 0.2f, (r.tg + o.obs_tshift) * 0.2f, r.db);
    *reinterpret_cast<float4*>(own + 4) = make_float4(r.cap * o.inv_cap, (r.fl & 1u) ? 1.0f : 0.0f, (r.fl & 2u) ? 1.0f : 0.0f, (float)r.sso / L);
    *reinterpret_cast<float4*>(own + 8) = make_float4(L / L, r.sig, r.pw, L);
    own[12] = 1.0f / L;
  }
}

// Link defects (env 988-1002; the draws of k_obs_* in mdr_kernels.hip: Philox stream TAG_COMM, counter (env, house, time index,
// block of four slots), a link delivers iff (float)u > comm_defect_prob): lane group g draws block g of ITS agent and zeroes the
// dead message records in the agent's staged row - the all-zero message(empty=True) of the reference.
__device__ __forceinline__ void observe_apply_defects(const mdr::ObserveArgs& o, float* row, int64_t agent, int g, int64_t A) {
  if (!(o.defect_prob > 0.0f) || 4 * g >= o.c || agent >= A) return;
  const uint32_t ag = (uint32_t)agent, e = ag / (uint32_t)o.N, h = ag - e * (uint32_t)o.N;   // (the launcher keeps A below 2^31 here)
  const uint32_t k = o.cursor ? (uint32_t)o.cursor[1] : (uint32_t)o.k;
  const u32x4 rnd = philox4x32_10(e + (uint32_t)o.env_offset, h + (uint32_t)o.house_offset, k, mdr::TAG_COMM | ((uint32_t)g << 8), o.k0,
                                  o.k1 ^ (o.episode * 0x85EBCA6Bu));
  const uint32_t x[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = 4 * g + j;
    if (m < o.c && !((float)mdr::u01(x[j]) > o.defect_prob)) *reinterpret_cast<float4*>(row + 4 * m) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
}

// The per-env observation columns of the extended form, once per env and step (obs_features() evaluates them per house):
// (OD - 20) / 5 (utils.py:803-805), sin / cos of tm_yday 2 pi / 365 (806-809) and of the integer hour 2 pi / 24 (810-813),
// solar gain / 1000 (0 until the first step, env 573) -> env_extra_a [4][E] | env_extra_b [2][E].
__global__ __launch_bounds__(256) void k_observe_env_extras(mdr::ObserveArgs o) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= o.E) return;
  int64_t off = 0, k = o.k;
  if (o.cursor != nullptr) {
    off = (int64_t)min(o.cursor[0], o.cursor_max + 1) * o.E;
    k = o.cursor[1];
  }
  const int64_t E = o.E;
  if (o.f_thermal) o.env_extra_a[e] = (o.od_now[off + e] + o.obs_tshift) * 0.2f;
  if (o.f_day || o.f_hour) {
    const mdr::Civil c = mdr::civil_from_epoch(o.t0[e] + k * (int64_t)o.dt);
    if (o.f_day) {
      const double ang = (double)c.yday * 6.283185307179586476925286766559 / 365.0;
      o.env_extra_a[E + e] = (float)sin(ang);
      o.env_extra_a[2 * E + e] = (float)cos(ang);
    }
    if (o.f_hour) {
      const double ang = (double)c.hour * 6.283185307179586476925286766559 / 24.0;
      o.env_extra_a[3 * E + e] = (float)sin(ang);
      o.env_extra_b[e] = (float)cos(ang);
    }
  }
  if (o.f_solar) o.env_extra_b[E + e] = k > 0 ? o.solar_now[off + e] * 1e-3f : 0.0f;
}

// Optional side product of observe -> act: the tile's observation rows, in normStateDict order, for the transition buffer
// (train_ppo.py:87-98 stores `state` with every transition).  The window holds them already - in staging order and with the raw
// seconds_since_off, which the gathering lanes replace by the quotient they computed - so the tile's TILE * 51 contiguous
// output floats are copied out with 16-byte non-temporal stores through a source-offset table built once per workgroup
// (output float o = 51 r + n  <-  window float OBS_ROW r + (n < 11 ? 40 + n : n - 11)).
template <int TILE>
__device__ __forceinline__ void observe_build_table(uint16_t* table, int tid, int nthreads) {
  for (int o = tid; o < TILE * 51; o += nthreads) {
    const int r = o / 51, n = o - 51 * r;
    table[o] = (uint16_t)(OBS_ROW * r + (n < 11 ? 4 * OBS_C + n : n - 11));
  }
}
template <int TILE>
__device__ __forceinline__ void observe_build_table_ext(uint16_t* table, int tid, int nthreads, int own, int c, int row) {
  const int F = own + 4 * c;
  for (int o = tid; o < TILE * F; o += nthreads) {
    const int r = o / F, n = o - F * r;
    table[o] = (uint16_t)(row * r + (n < own ? 4 * c + n : n - own));
  }
}

template <int TILE, bool EXT = false>
__device__ __forceinline__ void observe_store_rows(const float* rows, const uint16_t* table, float* out_tile, int lane_in, int nrows = TILE, int F = 51) {
  constexpr int QUADS = TILE * (EXT ? 64 : 51) / 4;   // 408 | 204 (EXT: the bound for F = 64)
  const int lane = tile_local(lane_in);
  if (!EXT) F = 51;
  if (((uintptr_t)out_tile & 15u) != 0) {   // a transition buffer whose per-step slice is not 16-byte aligned (A * F % 4 != 0): 4-byte stores
    for (int i = lane; i < nrows * F; i += 64) __builtin_nontemporal_store(rows[table[i]], out_tile + i);
    return;
  }
  const int quads = nrows * F / 4;      // the last tile of a batch may hold fewer agents (F nrows need not be a multiple of 4)
  if (lane < nrows * F - 4 * quads) out_tile[4 * quads + lane] = rows[table[4 * quads + lane]];
#pragma unroll
  for (int i = 0; i < (QUADS + 63) / 64; ++i) {
    const int q = i * 64 + lane;
    if (q < quads) {
      const uint2 src = *reinterpret_cast<const uint2*>(table + 4 * q);   // four 16-bit window offsets
      v4f_nt v = {rows[src.x & 0xFFFFu], rows[src.x >> 16], rows[src.y & 0xFFFFu], rows[src.y >> 16]};
      __builtin_nontemporal_store(v, reinterpret_cast<v4f_nt*>(out_tile) + q);
    }
  }
}

// (env, first house) of a tile, advanced by a fixed stride without a division per tile
struct TileCursor {
  int e, h0, de, dh, N;
  __device__ __forceinline__ void init(int64_t first_agent, int64_t stride_agents, int n) {
    N = n;
    e = (int)(first_agent / n);
    h0 = (int)(first_agent - (int64_t)e * n);
    de = (int)(stride_agents / n);
    dh = (int)(stride_agents - (int64_t)de * n);
  }
  __device__ __forceinline__ void next() {
    e += de;
    h0 += dh;
    if (h0 >= N) {
      h0 -= N;
      e += 1;
    }
  }
};

// ---- bf16x3 form: 32 agents per wavefront (two 16-agent column blocks), k-step s of layer 1 = row floats [32 s + 8 g, + 8)
// The forward reads 64 floats from the start of EVERY row whatever the row stride (against zero weights past the features), so the
// last row of a window reaches 64 - ROW floats past the rows: the pad behind them covers that.  With the fixed 16 floats a stride
// below 48 (F <= 42: six neighbours or fewer) let the last wave's last row read past the workgroup's LDS - whatever bits a previous
// kernel left there, and a NaN or infinity among them times a zero weight is a NaN logit for that one agent (seen once as a
// with / without rows_out mismatch in tests/test_gpu_observe_act.py).
__host__ __device__ constexpr int observe_bf16_pad(int row) { return 64 - row > OBS_PAD ? 64 - row : OBS_PAD; }
template <int MB, bool STORE, bool GEN, bool EXT = false, bool TABLE = false>   // TABLE (GEN && EXT): senders gathered through a link table
__global__ __launch_bounds__(64 * WAVESB) void k_actor_observe_bf16(ActorArgs a, mdr::ObserveArgs o) {
  const int NW = EXT ? (int)(blockDim.x >> 6) : WAVESB;   // waves per workgroup: the extended form takes as many as its windows leave room for (6 .. 8; two per SIMD either way)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int S2B = (MB + 1) / 2, TILE = 16 * NCB;
  const int ROW = EXT ? o.row : OBS_ROW, WIN = TILE * ROW + observe_bf16_pad(ROW);
  const int F = EXT ? o.own + 4 * o.c : 51;
  // the packed fragments hold 8 row blocks per k-step; the extended form keeps only the MB it uses in LDS (MB = 7: 12 KB less, which
  // is what lets eight 68-float windows fit beside them)
  constexpr int BLK = EXT ? MB : 8, KSTEP = BLK * 128;          // row blocks / uint4 per k-step in LDS
  uint4* f1 = reinterpret_cast<uint4*>(lds);                    // [2][BLK][2][64] fragments of 8 bf16
  const int S1B = EXT ? a.S1 : 2;                               // k-steps of layer 1: 32 features each (the extended form: one for F <= 32)
  uint4* f2 = f1 + 2 * KSTEP;                                   // [S2B][BLK][2][64] (f1: both k-steps, the second zero-filled when S1B = 1)
  float* wd = reinterpret_cast<float*>(f2 + S2B * KSTEP);       // head weights + biases (512 floats reserved)
  const int tid = threadIdx.x;
  float* rows = wd + 512 + (tid >> 6) * WIN;                    // this wave's window
  uint16_t* table = reinterpret_cast<uint16_t*>(wd + 512 + NW * WIN);   // [TILE * 51] (only when rows are stored)
  const uint4* g1 = reinterpret_cast<const uint4*>(a.frag1);
  const uint4* g2 = reinterpret_cast<const uint4*>(a.frag2);
  for (int i = tid; i < 2 * KSTEP; i += 64 * NW) {
    const int st = i / KSTEP, in = i - st * KSTEP;               // k-step, offset inside it (in < BLK * 128: the first BLK blocks)
    f1[i] = st < S1B ? g1[st * 1024 + in] : uint4{0u, 0u, 0u, 0u};
  }
  for (int i = tid; i < S2B * KSTEP; i += 64 * NW) {
    const int st = i / KSTEP, in = i - st * KSTEP;
    f2[i] = g2[st * 1024 + in];
  }
  for (int i = tid; i < 388; i += 64 * NW) wd[i] = a.wdiff[i];   // (a workgroup of the extended bf16 form has 384 threads)
  const int lane = tid & 63;
  for (int i = lane; i < WIN; i += 64) rows[i] = 0.0f;          // pads are read (against zero weights): they must be finite
  constexpr bool store = STORE;   // rows_out != nullptr (a compile-time variant: the plain form keeps its registers)
  if (store) {
    if (EXT) observe_build_table_ext<TILE>(table, tid, 64 * NW, o.own, o.c, ROW);
    else observe_build_table<TILE>(table, tid, 64 * NW);
  }
  __syncthreads();

  const int r = lane & 15, g = lane >> 4;
  const int wave = blockIdx.x * NW + (tid >> 6);   // tile indices fit 32 bits (the launcher checks): one register each, not two
  const int nwaves = gridDim.x * NW;
  const f32x4* bias1 = reinterpret_cast<const f32x4*>(wd + 128) + g;
  const f32x4* bias2 = reinterpret_cast<const f32x4*>(wd + 256) + g;
  const float bias3 = wd[384];
  const double* sig_row = observe_sig_row(o);
  TileCursor tc;
  tc.init((int64_t)wave * TILE, (int64_t)nwaves * TILE, o.N);
  float xr[NCB][16];
  // the lane's 16 features of column block c: two runs of 8 floats of row c * 16 + r.  `first_agent`: the tile's first agent -
  // with `store`, the quotients go back into the window and the wave copies the tile's rows out (observe_store_rows)
  auto gather = [&](int64_t first_agent) {
    if (EXT && o.defect_prob > 0.0f) {   // dead links first: the runs below then read zeros
#pragma unroll
      for (int c = 0; c < NCB; ++c) observe_apply_defects(o, rows + (c * 16 + r) * ROW, first_agent + c * 16 + r, g, a.A);
      observe_window_fence();
    }
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      float* row = rows + (c * 16 + r) * ROW;
      const float L = row[EXT ? ROW - 2 : 4 * OBS_C + 11], y = row[EXT ? ROW - 1 : 4 * OBS_C + 12];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float4 v0 = *reinterpret_cast<const float4*>(row + 32 * s + 8 * g);
        const float4 v1 = *reinterpret_cast<const float4*>(row + 32 * s + 8 * g + 4);
        // message records (their second field is the sender's seconds_since_off): the default shape's 40 message floats are the
        // runs s == 0 and (s == 1, g == 0); the extended form asks per four-float record
        const bool msg = (s == 0) || (g == 0);
        const bool msg_a = EXT ? (32 * s + 8 * g + 4 <= 4 * o.c) : msg, msg_b = EXT ? (32 * s + 8 * g + 8 <= 4 * o.c) : msg;
        xr[c][8 * s + 0] = v0.x;
        xr[c][8 * s + 1] = msg_a ? mdr::div_by_lockout(v0.y, L, y) : v0.y;
        xr[c][8 * s + 2] = v0.z;
        xr[c][8 * s + 3] = v0.w;
        xr[c][8 * s + 4] = v1.x;
        xr[c][8 * s + 5] = msg_b ? mdr::div_by_lockout(v1.y, L, y) : v1.y;
        xr[c][8 * s + 6] = v1.z;
        xr[c][8 * s + 7] = v1.w;
        // (rows past the batch's last agent hold zeros - L = 0, a NaN quotient - and the row BEFORE them reads their first floats
        // against zero weights when the row stride is below 64: they keep their zeros)
        const bool live_row = !GEN || first_agent + c * 16 + r < a.A;
        if (store && msg_a && live_row) row[32 * s + 8 * g + 1] = xr[c][8 * s + 1];
        if (store && msg_b && live_row) row[32 * s + 8 * g + 5] = xr[c][8 * s + 5];
      }
    }
    if (store) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      observe_store_rows<TILE, EXT>(rows, table, a.rows_out + first_agent * F, lane,
                                    GEN ? (int)((a.A - first_agent) < (int64_t)TILE ? (a.A - first_agent) : (int64_t)TILE) : TILE, F);
    }
  };
  SegSlot slot{};
  if (wave < a.ntiles) {
    if (GEN) {
      const HouseRegs first = observe_load_gen<TILE, EXT, TABLE>(o, sig_row, tc.e, tc.h0, (int64_t)wave * TILE, a.A, lane, slot);
      observe_stage_gen<EXT, 0, TABLE>(o, first, slot, rows);
    } else {
      const HouseRegs first = observe_load<TILE, EXT>(o, sig_row, tc.e, tc.h0, lane);
      observe_stage<TILE, EXT>(o, first, rows, lane, tc.e);
    }
    observe_window_fence();
    gather((int64_t)wave * TILE);
  }
  uint32_t rnd[NCB] = {};
  int it = 0;
  const int ntiles = (int)a.ntiles;
  for (int t = wave; t < ntiles; t += nwaves, ++it) {
    if ((it & 3) == 0) {
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        const int64_t ag = (((int64_t)t + (int64_t)tile_local(g) * nwaves) * NCB + c) * 16 + r;
        rnd[c] = philox4x32_10((uint32_t)ag, (uint32_t)((uint64_t)ag >> 32), a.step_lo + (a.step_dev ? (uint32_t)*a.step_dev : 0u),
                               TAG_ACTION ^ a.step_hi, loop_local(a.k0), loop_local(a.k1)).x;
      }
    }
    // the next tile's compact state: issued now, consumed between the k-steps of layer 2
    const bool more = t + nwaves < ntiles;
    tc.next();
    HouseRegs nxt{};
    if (more) nxt = GEN ? observe_load_gen<TILE, EXT, TABLE>(o, sig_row, tc.e, tc.h0, (int64_t)(t + nwaves) * TILE, a.A, lane, slot) : observe_load<TILE, EXT>(o, sig_row, tc.e, tc.h0, lane);
    f32x4 acc[NCB][MB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[c][mb] = bias1[mb * 4];
    // ---- layer 1 (F = 51: both k-steps)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 Bh[NCB], Bl[NCB];
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        uint4 bh, bl;
        split8(xr[c] + 8 * s, bh, bl);
        Bh[c] = __builtin_bit_cast(bf16x8, bh);
        Bl[c] = __builtin_bit_cast(bf16x8, bl);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, f1[((s * BLK + mb) * 2 + 0) * 64 + lane]);
        const bf16x8 Al = __builtin_bit_cast(bf16x8, f1[((s * BLK + mb) * 2 + 1) * 64 + lane]);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          acc[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh[c], acc[c][mb], 0, 0, 0);
          acc[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh[c], acc[c][mb], 0, 0, 0);
          acc[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl[c], acc[c][mb], 0, 0, 0);
        }
      }
    }
    // ---- layer 2; the next tile's rows are staged after its first k-step (this tile's rows were read before its layer 1)
    f32x4 out[NCB][MB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) out[c][mb] = bias2[mb * 4];
#pragma unroll
    for (int s = 0; s < S2B; ++s) {
      if (s == 1 && more) {
        if (GEN) observe_stage_gen<EXT, 0, TABLE>(o, nxt, slot, rows);
        else observe_stage<TILE, EXT>(o, nxt, rows, lane, tc.e);
      }
      bf16x8 Bh[NCB], Bl[NCB];
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (2 * s + (j >> 2) < MB) ? relu(acc[c][2 * s + (j >> 2) < MB ? 2 * s + (j >> 2) : 0][j & 3]) : 0.0f;
        uint4 bh, bl;
        split8(v, bh, bl);
        Bh[c] = __builtin_bit_cast(bf16x8, bh);
        Bl[c] = __builtin_bit_cast(bf16x8, bl);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, f2[((s * BLK + mb) * 2 + 0) * 64 + lane]);
        const bf16x8 Al = __builtin_bit_cast(bf16x8, f2[((s * BLK + mb) * 2 + 1) * 64 + lane]);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          out[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh[c], out[c][mb], 0, 0, 0);
          out[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh[c], out[c][mb], 0, 0, 0);
          out[c][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl[c], out[c][mb], 0, 0, 0);
        }
      }
    }
    if (more) {   // LDS operations of one wave complete in order: the rows staged above are what these loads see
      observe_window_fence();
      gather((int64_t)(t + nwaves) * TILE);
    }
    // ---- head
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const int64_t agent = ((int64_t)t * NCB + c) * 16 + r;
      const bool valid = agent < a.A;
      float d = 0.0f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 4; ++i) d = fmaf(wd[(mb * 4 + i) * 4 + g], relu(out[c][mb][i]), d);
      d += __shfl_xor(d, 16);
      d += __shfl_xor(d, 32);
      d += bias3;
      const float e = expf(-d);
      const float p0 = 1.0f / (1.0f + e);
      const float p1 = e > 1e30f ? 1.0f : e * p0;
      const uint32_t draw = (uint32_t)__shfl((int)rnd[c], r + 16 * (it & 3));
      if (g == 0 && valid) {
        const float u = ((float)(draw >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const int act = a.greedy ? (d >= 0.0f ? 0 : 1) : (u < p0 ? 0 : 1);
        a.action[agent] = (uint8_t)act;
        if (a.a_prob) a.a_prob[agent] = act ? p1 : p0;
        if (a.probs) {
          a.probs[agent * 2] = p0;
          a.probs[agent * 2 + 1] = p1;
        }
      }
    }
  }
}

// ---- exact-fp32 form (v_mfma_f32_16x16x4_f32): 16 agents per wavefront, lane group g holds features [13 g, 13 g + 13) of its agent
// EXTK = 0: the default observation; 13 | 15 | 16: the extended form specialised for that many k-steps of layer 1 (features up to
// 52 | 60 | 64) - the k-steps, the row stride and with it all window addressing are compile-time values as in the default form.
constexpr int observe16_row(int extk) { return extk == 0 ? OBS_ROW : (((4 * extk + 2 + 3) & ~3) | 4); }   // 4 S1 + 2 floats as 4 x odd: 60 | 68 | 68
template <int MB, bool STORE, bool GEN, bool TAIL, int EXTK = 0, bool TABLE = false>   // TABLE (GEN && EXTK): senders gathered through a link table
__global__ __launch_bounds__(64 * (EXTK ? WAVES16_EXT : WAVES16)) void k_actor_observe16(ActorArgs a, mdr::ObserveArgs o) {
  constexpr bool EXT = EXTK != 0;
  const int NW = EXT ? (int)(blockDim.x >> 6) : WAVES16;   // waves per workgroup (the extended form: as many of its 16 as the windows leave room for)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TILE = 16;
  constexpr int ROW = observe16_row(EXTK), WIN = TILE * ROW + OBS_PAD;
  // lane group g holds features [S1L g, S1L g + S1L) of its agent: actors in observe order are packed with exactly the 13 | 15 | 16
  // k-steps of their form (steps1_order below; weights past F are zeros, the row floats they meet are the zeros the window starts with)
  constexpr int S1L = EXT ? EXTK : 13;
#ifndef MDR_EXT_STAGE_Q
#define MDR_EXT_STAGE_Q -1
#endif
  constexpr int STAGE_Q = EXT ? MDR_EXT_STAGE_Q : 8;   // the k-step of layer 2 before which the next tile's rows are staged
  const int F = EXT ? o.own + 4 * o.c : 51;
  float* f1 = lds;                       // [S1L][64][8]
  float* f2 = f1 + S1L * 512;            // [S2][64][8]
  float* wd = f2 + a.S2 * 512;           // head weights + biases (512 floats reserved)
  const int tid = threadIdx.x;
  float* rows = wd + 512 + (tid >> 6) * WIN;
  uint16_t* table = reinterpret_cast<uint16_t*>(wd + 512 + NW * WIN);   // [TILE * 51] (only when rows are stored)
  for (int i = tid * 4; i < S1L * 512; i += 64 * NW * 4) *reinterpret_cast<float4*>(f1 + i) = *reinterpret_cast<const float4*>(a.frag1 + i);
  for (int i = tid * 4; i < a.S2 * 512; i += 64 * NW * 4) *reinterpret_cast<float4*>(f2 + i) = *reinterpret_cast<const float4*>(a.frag2 + i);
  for (int i = tid; i < 388; i += 64 * NW) wd[i] = a.wdiff[i];   // (a workgroup of the extended bf16 form has 384 threads)
  const int lane = tid & 63;
  for (int i = lane; i < WIN; i += 64) rows[i] = 0.0f;
  constexpr bool store = STORE;   // rows_out != nullptr (a compile-time variant: the plain form keeps its registers)
  if (store) {
    if (EXT) observe_build_table_ext<TILE>(table, tid, 64 * NW, o.own, o.c, ROW);
    else observe_build_table<TILE>(table, tid, 64 * NW);
  }
  __syncthreads();

  const int r = lane & 15, g = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * NW + (tid >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * NW;
  const f32x4* bias1 = reinterpret_cast<const f32x4*>(wd + 128) + g;
  const f32x4* bias2 = reinterpret_cast<const f32x4*>(wd + 256) + g;
  const float bias3 = wd[384];
  const double* sig_row = observe_sig_row(o);
  TileCursor tc;
  tc.init(wave * TILE, nwaves * TILE, o.N);
  float xr[16];
  auto gather = [&](int64_t first_agent) {
    float* row = rows + r * ROW;
    const float L = row[EXT ? ROW - 2 : 4 * OBS_C + 11], y = row[EXT ? ROW - 1 : 4 * OBS_C + 12];
    // the senders' seconds_since_off (float 1 of every message record) become quotients by the RECEIVER's lockout, in place: lane
    // group g takes the messages g, g + 4 and g + 8 of its agent's row - three sites instead of a test on each of the 13 features
#pragma unroll
    for (int i = 0; i < (EXT ? 4 : 3); ++i) {
      const int m = g + 4 * i;
      if (m < (EXT ? o.c : OBS_C)) row[4 * m + 1] = mdr::div_by_lockout(row[4 * m + 1], L, y);
    }
    if (EXT && o.defect_prob > 0.0f) {   // after the quotients (LDS operations of a wave complete in issue order): a dead record is all zeros
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      observe_apply_defects(o, row, first_agent + r, g, a.A);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
    for (int s = 0; s < S1L; ++s) xr[s] = row[S1L * g + s];
    if (store) {
      observe_store_rows<TILE, EXT>(rows, table, a.rows_out + first_agent * F, lane,
                                    GEN ? (int)((a.A - first_agent) < (int64_t)TILE ? (a.A - first_agent) : (int64_t)TILE) : TILE, F);
    }
  };
  SegSlot slot{};
  if (wave < a.ntiles) {
    if (GEN) {
      const HouseRegs first = observe_load_gen<TILE, EXT, TABLE>(o, sig_row, tc.e, tc.h0, wave * TILE, a.A, lane, slot);
      observe_stage_gen<EXT, EXT ? ROW : 0, TABLE>(o, first, slot, rows);
    } else {
      const HouseRegs first = observe_load<TILE, EXT>(o, sig_row, tc.e, tc.h0, lane);
      observe_stage<TILE, EXT, EXT ? ROW : 0>(o, first, rows, lane, tc.e);
    }
    observe_window_fence();
    gather(wave * TILE);
  }
  uint32_t rnd = 0;
  int it = 0;
  for (int64_t t = wave; t < a.ntiles; t += nwaves, ++it) {
    const int64_t agent = t * 16 + r;
    const bool valid = agent < a.A;
    if ((it & 3) == 0) {
      const int64_t ag = (t + tile_local(g) * nwaves) * 16 + r;
      rnd = philox4x32_10((uint32_t)ag, (uint32_t)((uint64_t)ag >> 32), a.step_lo + (a.step_dev ? (uint32_t)*a.step_dev : 0u), TAG_ACTION ^ a.step_hi, loop_local(a.k0), loop_local(a.k1)).x;
    }
    const bool more = t + nwaves < a.ntiles;
    tc.next();
    HouseRegs nxt{};
    if (more) nxt = GEN ? observe_load_gen<TILE, EXT, TABLE>(o, sig_row, tc.e, tc.h0, (t + nwaves) * TILE, a.A, lane, slot) : observe_load<TILE, EXT>(o, sig_row, tc.e, tc.h0, lane);
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = bias1[mb * 4];
    // ---- layer 1
#pragma unroll
    for (int s = 0; s < S1L; ++s) {
      const float4 w0 = *reinterpret_cast<const float4*>(f1 + s * 512 + lane * 8);
      const float4 w1 = *reinterpret_cast<const float4*>(f1 + s * 512 + lane * 8 + 4);
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[mb] = mb == MB - 1 ? mma16<TAIL>(w[mb], xr[s], acc[mb]) : mma16<false>(w[mb], xr[s], acc[mb]);
    }
    if (TAIL) acc[MB - 1] = sum_lane_groups(acc[MB - 1]);
    // ---- layer 2; the next tile's rows are staged after a few k-steps, read back at the end (STAGE_Q < 0: before the layer starts,
    // while the accumulators of layer 2 do not exist yet)
    if (STAGE_Q < 0 && more) {
      if (GEN) observe_stage_gen<EXT, EXT ? ROW : 0, TABLE>(o, nxt, slot, rows);
      else observe_stage<TILE, EXT, EXT ? ROW : 0>(o, nxt, rows, lane, tc.e);
    }
    f32x4 out[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) out[mb] = bias2[mb * 4];
#pragma unroll
    for (int q = 0; q < 4 * MB; ++q) {
      if (q == STAGE_Q && more) {
        if (GEN) observe_stage_gen<EXT, EXT ? ROW : 0, TABLE>(o, nxt, slot, rows);
        else observe_stage<TILE, EXT, EXT ? ROW : 0>(o, nxt, rows, lane, tc.e);
      }
      if (q < a.S2) {
        const float b = relu(TAIL && q >= 4 * (MB - 1) ? pick_register(acc[MB - 1], g) : acc[q >> 2][q & 3]);   // tail: k-index g is unit 96 + g
        const float4 w0 = *reinterpret_cast<const float4*>(f2 + q * 512 + lane * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(f2 + q * 512 + lane * 8 + 4);
        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) out[mb] = mb == MB - 1 ? mma16<TAIL>(w[mb], b, out[mb]) : mma16<false>(w[mb], b, out[mb]);
      }
    }
    if (TAIL) out[MB - 1] = sum_lane_groups(out[MB - 1]);
    if (more) {
      observe_window_fence();
      gather((t + nwaves) * TILE);
    }
    // ---- head
    float d = 0.0f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int i = 0; i < 4; ++i) d = fmaf(wd[(mb * 4 + i) * 4 + g], relu(out[mb][i]), d);
    d += __shfl_xor(d, 16);
    d += __shfl_xor(d, 32);
    d += bias3;
    const float e = expf(-d);
    const float p0 = 1.0f / (1.0f + e);
    const float p1 = e > 1e30f ? 1.0f : e * p0;
    const uint32_t draw = (uint32_t)__shfl((int)rnd, r + 16 * (it & 3));
    if (g == 0 && valid) {
      const float u = ((float)(draw >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const int act = a.greedy ? (d >= 0.0f ? 0 : 1) : (u < p0 ? 0 : 1);
      a.action[agent] = (uint8_t)act;
      if (a.a_prob) a.a_prob[agent] = act ? p1 : p0;
      if (a.probs) {
        a.probs[agent * 2] = p0;
        a.probs[agent * 2 + 1] = p1;
      }
    }
  }
}

// PPO.update's Monte-Carlo return scan (agents/ppo.py:123-134), backwards over the T steps of every agent:
// R <- reward[t] + gamma * (done[t] ? bootstrap[t] (or 0) : R).  One thread per agent, steps coalesced across agents.
__global__ __launch_bounds__(256) void k_discounted_returns(const float* reward, const uint8_t* done, const float* bootstrap, float gamma,
                                                            int T, int64_t A, float* out) {
  const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (a >= A) return;
  float run = 0.0f;
  for (int t = T - 1; t >= 0; --t) {
    const int64_t i = (int64_t)t * A + a;
    if (done != nullptr && done[i]) run = bootstrap != nullptr ? bootstrap[i] : 0.0f;
    const float scaled = gamma * run;   // two roundings, as torch's reward[t] + gamma * running (separate statements: no contraction)
    run = reward[i] + scaled;
    out[i] = run;
  }
}

int acc_row_half0(int q) { return 32 * (q >> 4) + (q & 3) + 8 * ((q >> 2) & 3); }

// layout MDR_ACTOR_FRAG32: ceil((F + 1) / 2) k-steps of 2; MDR_ACTOR_FRAG16: ceil((F + 1) / 4) k-steps of 4
int blocks16(int h1, int h2) { return ((h1 > h2 ? h1 : h2) <= 112) ? 7 : 8; }   // 16-row blocks holding the hidden units

int steps1(int layout, int num_state) {
  if (layout == MDR_ACTOR_BF16X3) return (num_state + 31) / 32;   // ceil(F / 32)
  return layout == MDR_ACTOR_FRAG16 || layout == MDR_ACTOR_FRAG16T ? (num_state + 3) / 4 : (num_state + 2) / 2;   // FRAG16: no constant-1 feature
}

// Actors packed in observe order (mdr_actor_t.feature_order = 1) for the exact-fp32 forms carry exactly the k-steps of layer 1 that
// their observe -> act kernel is compiled for: 13 (up to 52 features; the default 51), 15 (up to 60) or 16 (up to 64).
int steps1_order(int layout, int num_state, int order) {
  const int s = steps1(layout, num_state);
  if (order != 1 || (layout != MDR_ACTOR_FRAG16 && layout != MDR_ACTOR_FRAG16T) || s > 16) return s;
  return s <= 13 ? 13 : (s <= 15 ? 15 : 16);
}

int steps2(int layout, int hidden1) {
  if (layout == MDR_ACTOR_BF16X3) return 4;                           // k-steps of two 16-row blocks each: all 8 stored blocks
  if (layout == MDR_ACTOR_FRAG16 || layout == MDR_ACTOR_FRAG16T) return 4 * (hidden1 / 16) + (hidden1 % 16 + 3) / 4;   // a partial last block is stored transposed: its first ceil(rem / 4) registers hold it
  int n = 0;                                                             // FRAG32: (block, register) pairs whose half-0 row is <= hidden1
  for (int q = 0; q < 64; ++q)
    if (acc_row_half0(q) <= hidden1) ++n;
  return n;
}

int floats_per_step(int layout) {   // 4-byte units per k-step: 64 lanes x (4 | 8 floats), or 8 row blocks x (head, tail) x 64 lanes x 8 bf16
  return layout == MDR_ACTOR_BF16X3 ? 4096 : (layout == MDR_ACTOR_FRAG16 || layout == MDR_ACTOR_FRAG16T ? 512 : 256);
}

bool layout_ok(int layout) {
  return layout == MDR_ACTOR_FRAG32 || layout == MDR_ACTOR_FRAG16 || layout == MDR_ACTOR_BF16X3 || layout == MDR_ACTOR_FRAG16T;
}

// MDR_ACTOR_FRAG16T: six full 16-row blocks and a tail of 1..4 units in both hidden layers (the reference's [100, 100])
bool tail_shape_ok(const mdr_actor_t* actor) {
  return actor->hidden1 / 16 == 6 && actor->hidden2 / 16 == 6 && actor->hidden1 % 16 >= 1 && actor->hidden1 % 16 <= 4 &&
         actor->hidden2 % 16 >= 1 && actor->hidden2 % 16 <= 4;
}

}  // namespace

namespace mdr {

// Lanes the general staging needs for a tile of `tile` agents (c senders per house, N houses per env): a tile is cut into per-env
// segments, each staging its houses plus the c around them - or the whole env when that wraps onto itself; worst case over the tile
// starts (the pattern repeats with the env).
static int observe_window_lanes(int N, int c, int tile) {
  int worst = 0;
  const int starts = N < 4096 ? N : 1;   // big envs: at most two segments, tile + 2 c lanes
  for (int h0 = 0; h0 < starts; ++h0) {
    int lanes = 0, hs = h0, rem = tile;
    while (rem > 0) {
      const int len = N - hs < rem ? N - hs : rem;
      lanes += len + c >= N ? N : len + c;
      rem -= len;
      hs = 0;
    }
    if (lanes > worst) worst = lanes;
  }
  if (N >= 4096) worst = tile + 2 * c;
  return worst;
}

int launch_actor_observe(const mdr_actor_t* actor, const ObserveArgs& o, uint64_t seed, uint64_t step, const int32_t* step_dev, uint8_t* action,
                         float* a_prob, float* probs, float* rows_out, hipStream_t stream) {
  if (!actor || actor->struct_size != sizeof(mdr_actor_t) || !action) return MDR_ERR_INVALID;
  if (!actor->frag1 || !actor->frag2 || !actor->wdiff) return MDR_ERR_INVALID;
  const int layout = actor->layout;
  if (layout != MDR_ACTOR_FRAG16 && layout != MDR_ACTOR_BF16X3 && layout != MDR_ACTOR_FRAG16T) return MDR_ERR_UNSUPPORTED;
  if (layout == MDR_ACTOR_FRAG16T && !tail_shape_ok(actor)) return MDR_ERR_UNSUPPORTED;
  const bool ext = o.ext != 0;
  const int c = ext ? o.c : OBS_C, own = ext ? o.own : 11, F = 4 * c + own;
  if (actor->feature_order != 1 || actor->num_state != F || actor->observe_msg_floats != 4 * c) return MDR_ERR_UNSUPPORTED;
  if (F > 64 || c > OBS_MAX_C || c < 0) return MDR_ERR_UNSUPPORTED;
  if (actor->hidden1 <= 0 || actor->hidden2 <= 0 || actor->hidden1 > MDR_ACTOR_MAX_HIDDEN || actor->hidden2 > MDR_ACTOR_MAX_HIDDEN) return MDR_ERR_INVALID;
  const bool lbf = layout == MDR_ACTOR_BF16X3;
  const int tile = lbf ? 16 * NCB : 16, s1 = steps1_order(layout, actor->num_state, actor->feature_order);
  // row stride of the extended form: the floats the forward reads of a row (bf16: the features; fp32: four lane groups of S1) + L, 1 / L
  // ... as a multiple of 4 floats that is ODD in units of 4: the 16 lanes of a lane group read the same offset of 16 consecutive
  // rows, and a stride of 64 floats would put them all on one LDS bank (60 or 68: two lanes per bank)
  int row = OBS_ROW;
  const int extk = !ext || lbf ? 0 : s1;   // the fp32 form's compile-time k-steps of layer 1: 13 | 15 | 16
  if (ext) {
    row = ((lbf ? F : 4 * extk) + 2 + 3) & ~3;
    if ((row & 4) == 0) row += 4;
  }
  int waves = lbf ? WAVESB : (ext ? WAVES16_EXT : WAVES16);
  if (o.N < c + 1) return MDR_ERR_UNSUPPORTED;   // c distinct circular neighbours
  static const bool force_gen = [] { const char* t = getenv("MDR_OBSERVE_GEN"); return t && t[0] == '1'; }();   // experiment knob
  const bool table = ext && o.links != nullptr;    // senders by a link table: gathered from the message records, one staging lane per agent
  if (table && o.msg_rec == nullptr) return MDR_ERR_INVALID;
  const bool gen = o.N % 32 != 0 || force_gen || table;   // tiles that start anywhere in an env / span several: the general staging
  if (gen && observe_window_lanes(o.N, table ? 0 : c, tile) > 64) return MDR_ERR_UNSUPPORTED;
  ActorArgs a{};
  a.frag1 = static_cast<const float*>(actor->frag1); a.frag2 = static_cast<const float*>(actor->frag2); a.wdiff = actor->wdiff;
  a.action = action; a.a_prob = a_prob; a.probs = probs;
  if (rows_out && ((uintptr_t)rows_out & 3u) != 0) return MDR_ERR_INVALID;
  a.rows_out = rows_out;
  a.A = (int64_t)o.E * o.N;
  a.ntiles = (a.A + tile - 1) / tile;
  if (a.ntiles > 0x7FFFFFFF) return MDR_ERR_UNSUPPORTED;   // the kernels count tiles in 32 bits
  if (ext && a.A > 0x3FFFFFFF) return MDR_ERR_UNSUPPORTED;   // the extended forms address a house's arrays by one 32-bit byte offset (and draw link defects from a 32-bit agent index)
  a.F = actor->num_state; a.S1 = s1; a.S2 = steps2(layout, actor->hidden1);
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(seed >> 32);
  a.step_lo = (uint32_t)(step & 0xFFFFFFFFull); a.step_hi = (uint32_t)(step >> 32);
  a.step_dev = step_dev;
  a.greedy = actor->greedy != 0;
  ObserveArgs oo = o;
  oo.row = row;
  const size_t window = (size_t)tile * row + (lbf ? observe_bf16_pad(row) : OBS_PAD);
  auto lds_need = [&](int w) {
    const int s1_lds = ext ? (lbf ? 2 : extk) : a.S1;   // the extended form stages layer 1 zero-padded to its full k-step count
    size_t per_step = (size_t)floats_per_step(layout);
    if (ext && lbf) per_step = per_step * (size_t)blocks16(actor->hidden1, actor->hidden2) / 8;   // ... and only the row blocks it uses
    return ((size_t)(s1_lds + a.S2) * per_step + 512 + (size_t)w * window) * sizeof(float) + (rows_out ? (size_t)tile * F * sizeof(uint16_t) : 0);
  };
  if (ext && lbf)   // the bf16 form keeps two waves per SIMD whatever the count: as many windows as fit beside the weight fragments
    while (waves > 4 && lds_need(waves) > 160 * 1024) --waves;
  if (ext && !lbf)  // the fp32 form: whole waves per SIMD
    while (waves > 8 && lds_need(waves) > 160 * 1024) waves -= 4;
  const size_t lds_bytes = lds_need(waves);
  if (lds_bytes > 160 * 1024) return MDR_ERR_UNSUPPORTED;
  if (ext && (o.f_thermal || o.f_day || o.f_hour || o.f_solar)) {   // the per-env columns, once per env
    hipLaunchKernelGGL(k_observe_env_extras, dim3((unsigned)((o.E + 255) / 256)), dim3(256), 0, stream, oo);
    if (hipGetLastError() != hipSuccess) return MDR_ERR_HIP;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  const int64_t want = (a.ntiles + waves - 1) / waves;
  const unsigned grid = (unsigned)(want < cus ? want : cus);
  auto launch = [&](auto kernel) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return MDR_ERR_HIP;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * waves), lds_bytes, stream, a, oo);
    return hipGetLastError() == hipSuccess ? MDR_OK : MDR_ERR_HIP;
  };
  const int mb = blocks16(actor->hidden1, actor->hidden2);
  if (ext) {   // optional state columns / c != 10 / link defects: the general staging with run-time row shape
    if (lbf) {
#define MDR_OBSERVE_BF16_EXT(MBV)                                                                                                    \
  (table ? (rows_out ? launch(k_actor_observe_bf16<MBV, true, true, true, true>) : launch(k_actor_observe_bf16<MBV, false, true, true, true>)) \
   : gen ? (rows_out ? launch(k_actor_observe_bf16<MBV, true, true, true>) : launch(k_actor_observe_bf16<MBV, false, true, true>))    \
         : (rows_out ? launch(k_actor_observe_bf16<MBV, true, false, true>) : launch(k_actor_observe_bf16<MBV, false, false, true>)))
      return mb == 7 ? MDR_OBSERVE_BF16_EXT(7) : MDR_OBSERVE_BF16_EXT(8);
#undef MDR_OBSERVE_BF16_EXT
    }
    if (extk != 13 && extk != 15 && extk != 16) return MDR_ERR_UNSUPPORTED;
#define MDR_OBSERVE16_EXT_G(MBV, TAILV, GENV)                                                                                                        \
  (extk == 13 ? (rows_out ? launch(k_actor_observe16<MBV, true, GENV, TAILV, 13>) : launch(k_actor_observe16<MBV, false, GENV, TAILV, 13>))   \
   : extk == 15 ? (rows_out ? launch(k_actor_observe16<MBV, true, GENV, TAILV, 15>) : launch(k_actor_observe16<MBV, false, GENV, TAILV, 15>)) \
                : (rows_out ? launch(k_actor_observe16<MBV, true, GENV, TAILV, 16>) : launch(k_actor_observe16<MBV, false, GENV, TAILV, 16>)))
#define MDR_OBSERVE16_EXT_T(MBV, TAILV)                                                                                                                    \
  (extk == 13 ? (rows_out ? launch(k_actor_observe16<MBV, true, true, TAILV, 13, true>) : launch(k_actor_observe16<MBV, false, true, TAILV, 13, true>))   \
   : extk == 15 ? (rows_out ? launch(k_actor_observe16<MBV, true, true, TAILV, 15, true>) : launch(k_actor_observe16<MBV, false, true, TAILV, 15, true>)) \
                : (rows_out ? launch(k_actor_observe16<MBV, true, true, TAILV, 16, true>) : launch(k_actor_observe16<MBV, false, true, TAILV, 16, true>)))
    if (table) {
      if (layout == MDR_ACTOR_FRAG16T) return MDR_OBSERVE16_EXT_T(7, true);
      return mb == 7 ? MDR_OBSERVE16_EXT_T(7, false) : MDR_OBSERVE16_EXT_T(8, false);
    }
    // (the whole-tile staging for the reference's [100, 100] actor; other hidden sizes take the general windows whatever N is)
    if (layout == MDR_ACTOR_FRAG16T) return gen ? MDR_OBSERVE16_EXT_G(7, true, true) : MDR_OBSERVE16_EXT_G(7, true, false);
    return mb == 7 ? MDR_OBSERVE16_EXT_G(7, false, true) : MDR_OBSERVE16_EXT_G(8, false, true);
#undef MDR_OBSERVE16_EXT_G
#undef MDR_OBSERVE16_EXT_T
  }
#define MDR_OBSERVE_VARIANT(KERNEL, ...)                                                                     \
  (rows_out ? (gen ? launch(KERNEL<__VA_ARGS__, true, true>) : launch(KERNEL<__VA_ARGS__, true, false>))   \
            : (gen ? launch(KERNEL<__VA_ARGS__, false, true>) : launch(KERNEL<__VA_ARGS__, false, false>)))
#define MDR_OBSERVE16_VARIANT(MBV, TAILV)                                                                                  \
  (rows_out ? (gen ? launch(k_actor_observe16<MBV, true, true, TAILV>) : launch(k_actor_observe16<MBV, true, false, TAILV>))   \
            : (gen ? launch(k_actor_observe16<MBV, false, true, TAILV>) : launch(k_actor_observe16<MBV, false, false, TAILV>)))
  if (lbf) return mb == 7 ? MDR_OBSERVE_VARIANT(k_actor_observe_bf16, 7) : MDR_OBSERVE_VARIANT(k_actor_observe_bf16, 8);
  if (layout == MDR_ACTOR_FRAG16T) return MDR_OBSERVE16_VARIANT(7, true);
  return mb == 7 ? MDR_OBSERVE16_VARIANT(7, false) : MDR_OBSERVE16_VARIANT(8, false);
#undef MDR_OBSERVE16_VARIANT
#undef MDR_OBSERVE_VARIANT
}

}  // namespace mdr

extern "C" {

int64_t mdr_actor_steps1(int32_t layout, int32_t num_state) { return (layout_ok(layout) && num_state > 0) ? steps1(layout, num_state) : -1; }
int64_t mdr_actor_steps1_order(int32_t layout, int32_t num_state, int32_t feature_order) {
  return (layout_ok(layout) && num_state > 0) ? steps1_order(layout, num_state, feature_order) : -1;
}
int64_t mdr_actor_steps2(int32_t layout, int32_t hidden1) {
  return (layout_ok(layout) && hidden1 > 0 && hidden1 <= MDR_ACTOR_MAX_HIDDEN) ? steps2(layout, hidden1) : -1;
}
int64_t mdr_actor_frag1_floats(int32_t layout, int32_t num_state) {
  return mdr_actor_steps1(layout, num_state) < 0 ? -1 : mdr_actor_steps1(layout, num_state) * floats_per_step(layout);
}
int64_t mdr_actor_frag2_floats(int32_t layout, int32_t hidden1) {
  return mdr_actor_steps2(layout, hidden1) < 0 ? -1 : mdr_actor_steps2(layout, hidden1) * floats_per_step(layout);
}

int mdr_actor_sample(const mdr_actor_t* actor, const float* obs, int64_t obs_plane_stride, int64_t nb_agents, uint64_t seed, uint64_t step,
                     const int32_t* step_dev, uint8_t* action, float* a_prob, float* probs, void* stream) {
  if (!actor || actor->struct_size != sizeof(mdr_actor_t) || !obs || !action || nb_agents < 0) return MDR_ERR_INVALID;
  if (obs_plane_stride != 0 && obs_plane_stride < nb_agents) return MDR_ERR_INVALID;
  if (!actor->frag1 || !actor->frag2 || !actor->wdiff || !layout_ok(actor->layout)) return MDR_ERR_INVALID;
  if (actor->num_state <= 0 || actor->hidden1 <= 0 || actor->hidden2 <= 0) return MDR_ERR_INVALID;
  if (actor->feature_order != 0) return MDR_ERR_INVALID;   // weights packed for mdr_env_actor_sample: observation rows are in normStateDict order
  if (actor->hidden1 > MDR_ACTOR_MAX_HIDDEN || actor->hidden2 > MDR_ACTOR_MAX_HIDDEN) return MDR_ERR_UNSUPPORTED;
  if (nb_agents == 0) return MDR_OK;
  const int layout = actor->layout;
  const bool lbf = layout == MDR_ACTOR_BF16X3;
  const bool l16 = layout == MDR_ACTOR_FRAG16 || layout == MDR_ACTOR_FRAG16T || lbf;             // 16 agents per wavefront
  if (layout == MDR_ACTOR_FRAG16T && !tail_shape_ok(actor)) return MDR_ERR_UNSUPPORTED;
  if (l16 && actor->num_state > 128) return MDR_ERR_UNSUPPORTED;   // 32 feature registers per lane (and column block): pack FRAG32 instead
  ActorArgs a{};
  a.frag1 = static_cast<const float*>(actor->frag1); a.frag2 = static_cast<const float*>(actor->frag2); a.wdiff = actor->wdiff;
  a.obs = obs; a.action = action; a.a_prob = a_prob; a.probs = probs;
  a.A = nb_agents;
  a.plane = obs_plane_stride;
  const int tile = lbf ? 16 * NCB : (l16 ? 16 : 32), waves = lbf ? WAVESB : (l16 ? WAVES16 : WAVES);
  a.ntiles = (nb_agents + tile - 1) / tile;
  a.F = actor->num_state; a.S1 = steps1(layout, actor->num_state); a.S2 = steps2(layout, actor->hidden1);
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(seed >> 32);
  a.step_lo = (uint32_t)(step & 0xFFFFFFFFull); a.step_hi = (uint32_t)(step >> 32);
  a.step_dev = step_dev;
  a.greedy = actor->greedy != 0;
  const size_t lds_bytes = ((size_t)(a.S1 + a.S2) * floats_per_step(layout) + 512) * sizeof(float);   // + head weights (and biases)
  if (lds_bytes > 160 * 1024) return MDR_ERR_UNSUPPORTED;   // num_state beyond ~190 with 100-unit layers
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  const int64_t want = (a.ntiles + waves - 1) / waves;
  const unsigned grid = (unsigned)(want < cus ? want : cus);   // persistent: the weights are staged once per workgroup
  auto launch = [&](auto kernel) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return MDR_ERR_HIP;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * waves), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MDR_OK : MDR_ERR_HIP;
  };
  const int mb = blocks16(actor->hidden1, actor->hidden2);
  if (lbf) {
    if (a.S1 <= 2) return mb == 7 ? launch(k_actor_sample_bf16<7>) : launch(k_actor_sample_bf16<8>);
    return mb == 7 ? launch(k_actor_sample_bf16<7, 32>) : launch(k_actor_sample_bf16<8, 32>);   // 65..128 features
  }
  if (l16) {
    // k_actor_sample16 addresses its features by 32-bit byte offsets from `obs`: a launch covers at most 4 GiB of observations
    // (20 million agents at F = 51); a bigger batch goes out in slices of whole tiles
    const uint64_t limit = 0xFFFFFFFFull;
    if (obs_plane_stride != 0 && ((uint64_t)(a.F - 1) * (uint64_t)obs_plane_stride + (uint64_t)nb_agents) * 4ull > limit)
      return MDR_ERR_UNSUPPORTED;   // feature planes further apart than 32 bits reach: hand over rows
    const int64_t per = obs_plane_stride != 0 ? nb_agents : (int64_t)((limit / ((uint64_t)a.F * 4ull)) & ~15ull);
    for (int64_t first = 0; first < nb_agents; first += per) {
      const int64_t count = nb_agents - first < per ? nb_agents - first : per;
      a.obs = obs + first * (obs_plane_stride != 0 ? 1 : (int64_t)a.F);
      a.action = action + first;
      a.a_prob = a_prob ? a_prob + first : nullptr;
      a.probs = probs ? probs + 2 * first : nullptr;
      a.A = count;
      a.agent0 = first;
      a.ntiles = (count + tile - 1) / tile;
      int rc;
      if (a.S1 <= 16)
        rc = layout == MDR_ACTOR_FRAG16T ? launch(k_actor_sample16<7, true>)
                                         : (mb == 7 ? launch(k_actor_sample16<7, false>) : launch(k_actor_sample16<8, false>));
      else      // 65..128 features: the same kernel with 32 feature registers per lane
        rc = layout == MDR_ACTOR_FRAG16T ? launch(k_actor_sample16<7, true, 32>)
                                         : (mb == 7 ? launch(k_actor_sample16<7, false, 32>) : launch(k_actor_sample16<8, false, 32>));
      if (rc != MDR_OK) return rc;
    }
    return MDR_OK;
  }
  if (a.S1 <= 32 && a.S2 == 52) return launch(k_actor_sample<32, 52>);   // the reference's shape: num_state <= 62, layers [100, 100]
  if (a.S1 <= 32) return launch(k_actor_sample<32, 0>);
  return launch(k_actor_sample<0, 0>);
}

int mdr_discounted_returns(const float* reward, const uint8_t* done, const float* bootstrap, float gamma, int32_t nb_steps, int64_t nb_agents,
                           float* out, void* stream) {
  if (!reward || !out || nb_steps < 0 || nb_agents < 0) return MDR_ERR_INVALID;
  if (nb_steps == 0 || nb_agents == 0) return MDR_OK;
  hipLaunchKernelGGL(k_discounted_returns, dim3((unsigned)((nb_agents + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reward, done, bootstrap,
                     gamma, nb_steps, nb_agents, out);
  return hipGetLastError() == hipSuccess ? MDR_OK : MDR_ERR_HIP;
}

}  // extern "C"
