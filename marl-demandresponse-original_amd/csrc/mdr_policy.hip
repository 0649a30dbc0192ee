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

#include "../../include/mdr.h"
#include "../../include/mdr_policy.h"
#include "mdr_device.h"

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
  int F, S1, S2;
  uint32_t k0, k1, step_lo, step_hi;
};

// max(x, 0) in one instruction (v_med3_f32; fmaxf costs a canonicalising v_max_f32 x, x before the v_max_f32 x, 0)
__device__ __forceinline__ float relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_huge_valf()); }

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
  auto row_of = [&](int64_t t) {
    const int64_t agent = t * 32 + r;
    return a.obs + (agent < a.A ? agent : a.A - 1) * (int64_t)a.F;
  };
  auto feature = [&](const float* x, int s) {
    const int k = kbase + s;
    return k < a.F ? x[k] : (k == a.F ? 1.0f : 0.0f);
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
      const u32x4 rnd = philox4x32_10((uint32_t)agent, (uint32_t)((uint64_t)agent >> 32), a.step_lo, TAG_ACTION ^ a.step_hi, a.k0, a.k1);
      const float u = ((float)(rnd.x >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const int act = u < p0 ? 0 : 1;
      a.action[agent] = (uint8_t)act;
      if (a.a_prob) a.a_prob[agent] = act ? p1 : p0;
      if (a.probs) {
        a.probs[agent * 2] = p0;
        a.probs[agent * 2 + 1] = p1;
      }
    }
  }
}

int acc_row_half0(int q) { return 32 * (q >> 4) + (q & 3) + 8 * ((q >> 2) & 3); }

int steps1(int num_state) { return (num_state + 2) / 2; }   // ceil((F + 1) / 2): F features + the constant 1

int steps2(int hidden1) {   // (block, register) pairs whose half-0 row is <= hidden1 (rows 0..hidden1-1 and the constant unit)
  int n = 0;
  for (int q = 0; q < 64; ++q)
    if (acc_row_half0(q) <= hidden1) ++n;
  return n;
}

}  // namespace

extern "C" {

int64_t mdr_actor_steps1(int32_t num_state) { return num_state > 0 ? steps1(num_state) : -1; }
int64_t mdr_actor_steps2(int32_t hidden1) { return (hidden1 > 0 && hidden1 <= MDR_ACTOR_MAX_HIDDEN) ? steps2(hidden1) : -1; }
int64_t mdr_actor_frag1_floats(int32_t num_state) { return num_state > 0 ? (int64_t)steps1(num_state) * 256 : -1; }
int64_t mdr_actor_frag2_floats(int32_t hidden1) { return mdr_actor_steps2(hidden1) < 0 ? -1 : mdr_actor_steps2(hidden1) * 256; }

int mdr_actor_sample(const mdr_actor_t* actor, const float* obs, int64_t nb_agents, uint64_t seed, uint64_t step, uint8_t* action,
                     float* a_prob, float* probs, void* stream) {
  if (!actor || actor->struct_size != sizeof(mdr_actor_t) || !obs || !action || nb_agents < 0) return MDR_ERR_INVALID;
  if (!actor->frag1 || !actor->frag2 || !actor->wdiff) return MDR_ERR_INVALID;
  if (actor->num_state <= 0 || actor->hidden1 <= 0 || actor->hidden2 <= 0) return MDR_ERR_INVALID;
  if (actor->hidden1 > MDR_ACTOR_MAX_HIDDEN || actor->hidden2 > MDR_ACTOR_MAX_HIDDEN) return MDR_ERR_UNSUPPORTED;
  if (nb_agents == 0) return MDR_OK;
  ActorArgs a{};
  a.frag1 = actor->frag1; a.frag2 = actor->frag2; a.wdiff = actor->wdiff;
  a.obs = obs; a.action = action; a.a_prob = a_prob; a.probs = probs;
  a.A = nb_agents;
  a.ntiles = (nb_agents + 31) / 32;
  a.F = actor->num_state; a.S1 = steps1(actor->num_state); a.S2 = steps2(actor->hidden1);
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(seed >> 32);
  a.step_lo = (uint32_t)(step & 0xFFFFFFFFull); a.step_hi = (uint32_t)(step >> 32);
  const size_t lds_bytes = ((size_t)(a.S1 + a.S2) * 256 + 128) * sizeof(float);
  if (lds_bytes > 160 * 1024) return MDR_ERR_UNSUPPORTED;   // num_state beyond ~190 with 100-unit layers
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  const int64_t want = (a.ntiles + WAVES - 1) / WAVES;
  const unsigned grid = (unsigned)(want < cus ? want : cus);   // persistent: the weights are staged once per workgroup
  auto launch = [&](auto kernel) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return MDR_ERR_HIP;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * WAVES), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MDR_OK : MDR_ERR_HIP;
  };
  if (a.S1 <= 32 && a.S2 == 52) return launch(k_actor_sample<32, 52>);   // the reference's shape: num_state <= 62, layers [100, 100]
  if (a.S1 <= 32) return launch(k_actor_sample<32, 0>);
  return launch(k_actor_sample<0, 0>);
}

}  // extern "C"
