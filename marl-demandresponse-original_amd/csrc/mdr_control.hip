// GreedyMyopic (agents/greedy_myopic_controller.py:6-50), the reference's centralised baseline controller, for every env of the
// batch: the houses of an env sorted by  -(house_temp - target)  ascending - the hottest relative to its target first - then ONE
// greedy pass over them with the power budget  target = reg_signal :
//     take house i  iff  p_i + total < target   or   (|p_i + total - target| < |total - target|  and not hvac_lockout_i)
//     (p_i = cooling_capacity / COP; a taken house adds p_i to total whether or not its lockout lets it start - the reference's rule)
// One workgroup per env: bitonic sort of (key, house) pairs in LDS, the pass by one lane over the sorted powers, actions scattered
// back.  Houses with equal temperature difference are taken in house order (pandas' sort_values leaves their order open).
// The actions land in actions[E][N]; the step then runs with MDR_ACTIONS_EXTERNAL.
#include "mdr_device.h"
#include "mdr_kernels.h"
#include "mdr_step_common.h"

namespace mdr {

constexpr int GREEDY_MAX_HOUSES = 2048;

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_greedy_myopic(StepArgs a, int P) {   // P: power of two >= N, <= GREEDY_MAX_HOUSES
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];       // [P] {~sortable(Ta - target), house}
  float* pw = reinterpret_cast<float*>(keys + P);                                  // [P] power of the house at sorted position s
  uint8_t* flag = reinterpret_cast<uint8_t*>(pw + P);                              // [P] in: hvac_lockout; out: HVAC_status
  rebase(a);   // graph mode: the table row of the device cursor
  const int e = blockIdx.x, tid = threadIdx.x;
  const int64_t base = (int64_t)e * a.N;
  for (int i = tid; i < P; i += THREADS) {
    unsigned long long k = ~0ull;                                                  // padding sorts last
    if (i < a.N) {
      const float d = a.Ta[base + i] - a.target[base + i];                        // house_temp - target (both relative to the same reference)
      uint32_t u = __float_as_uint(d);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                             // ascending in d
      k = ((unsigned long long)(~u) << 32) | (uint32_t)i;                         // ... descending in d, ties by house index
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long x = keys[i], y = keys[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            keys[i] = y;
            keys[l] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int s = tid; s < a.N; s += THREADS) {
    const int h = (int)(uint32_t)keys[s];
    pw[s] = a.P_max[base + h];
    flag[s] = (uint8_t)((a.flags[base + h] >> 1) & 1u);
  }
  __syncthreads();
  if (tid == 0) {
    const double target = a.sig_old[e];                                            // obs["reg_signal"]: the signal of the current time index
    double total = 0.0;
    for (int s = 0; s < a.N; ++s) {
      const double p = (double)pw[s];
      const bool take = (p + total < target) || (fabs(p + total - target) < fabs(total - target) && flag[s] == 0);
      flag[s] = take ? 1 : 0;
      if (take) total += p;
    }
  }
  __syncthreads();
  for (int s = tid; s < a.N; s += THREADS) a.actions[base + (int)(uint32_t)keys[s]] = flag[s];
}

hipError_t launch_greedy_myopic(const StepArgs& a, hipStream_t s) {
  if (a.N < 1 || a.N > GREEDY_MAX_HOUSES || a.actions == nullptr) return hipErrorInvalidValue;
  int P = 2;
  while (P < a.N) P <<= 1;
  const size_t lds = (size_t)P * (sizeof(unsigned long long) + sizeof(float) + 1);
  if (P <= 64) hipLaunchKernelGGL(k_greedy_myopic<64>, dim3((unsigned)a.E), dim3(64), lds, s, a, P);
  else hipLaunchKernelGGL(k_greedy_myopic<256>, dim3((unsigned)a.E), dim3(256), lds, s, a, P);
  return hipGetLastError();
}

}  // namespace mdr
