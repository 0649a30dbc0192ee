// GreedyMyopic (agents/greedy_myopic_controller.py:6-50), the reference's centralised baseline controller, for every env of the
// batch: the houses of an env sorted by  -(house_temp - target)  ascending - the hottest relative to its target first - then ONE
// greedy pass over them with the power budget  target = reg_signal :
//     take house i  iff  p_i + total < target   or   (|p_i + total - target| < |total - target|  and not hvac_lockout_i)
//     (p_i = cooling_capacity / COP; a taken house adds p_i to total whether or not its lockout lets it start - the reference's rule)
// Houses with equal temperature difference are taken in house order (pandas' sort_values leaves their order open).
// The actions land in actions[E][N]; the step then runs with MDR_ACTIONS_EXTERNAL.
//
// Up to 1024 houses per env: ONE WAVEFRONT per env (k_greedy_wave<R, 64>), R = P / 64 sorted positions per lane in registers (position
// s = lane R + r), the bitonic network's exchanges inside a lane for partner distances below R and by lane shuffles above - no LDS,
// no barrier; envs of up to 32 / 16 / 8 houses share a wavefront (k_greedy_wave<1, G>: G lanes per env, the network ends at G).  The budget pass walks RUNS instead of houses: with S(s) the prefix sum of the sorted powers, a run of takes from
// position c on sees total(s) = S(s) + (total_c - S(c)), so every lane tests its R positions at once and the first position that is
// not taken ends the run; the run of refusals behind it is found the same way with the total held.  The pass alternates only
// inside the last house power below the budget - a handful of runs, not N dependent fp64 additions.  The prefix form equals the
// sequential sum bit for bit as long as the fp64 sums of the fp32 powers are exact (exponent spread + 24 + log2(N) <= 53: physical
// HVAC powers by a wide margin); the kernel checks that per env and walks an env that fails it house by house.
// 1025-2048 houses: one workgroup per env (k_greedy_myopic: bitonic sort of (key, house) pairs in LDS, the pass by one lane).
#include "mdr_device.h"
#include "mdr_kernels.h"
#include "mdr_step_common.h"

namespace mdr {

constexpr int GREEDY_MAX_HOUSES = 2048;


__device__ __forceinline__ unsigned long long greedy_key(const StepArgs& a, int64_t base, int i, int n) {
  if (i >= n) return ~0ull;                                                      // padding sorts last
  const float d = a.Ta[base + i] - a.target[base + i];                            // house_temp - target (both relative to the same reference)
  uint32_t u = __float_as_uint(d);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                                 // ascending in d
  return ((unsigned long long)(~u) << 32) | (uint32_t)i;                          // ... descending in d, ties by house index
}

__device__ __forceinline__ bool greedy_take(double p, double total, double target, bool locked) {
  return (p + total < target) || (fabs(p + total - target) < fabs(total - target) && !locked);
}

// G lanes per env (64, or 32 / 16 / 8 with one position per lane: several small envs share a wavefront), P = G R sorted positions,
// position s = gl R + r in register r of the env's lane gl.  Every shuffle is executed by all lanes; what differs between the envs
// of a wavefront (run boundaries, whether an env is finished) is carried in per-lane flags.
template <int R, int G>
__global__ __launch_bounds__(256) void k_greedy_wave(StepArgs a) {
  static_assert(G == 64 || R == 1, "several envs per wavefront only with one position per lane");
  rebase(a);   // graph mode: the table row of the device cursor
  constexpr int P = G * R;
  constexpr int EPW = 64 / G;
  const int lane = threadIdx.x & 63, gl = lane & (G - 1);
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave * EPW >= a.E) return;                                                  // wave-uniform; the kernel has no barrier
  const int e = wave * EPW + lane / G;
  const bool valid = e < a.E;
  const int64_t base = (int64_t)(valid ? e : 0) * a.N;
  const int n = valid ? a.N : 0;                                                  // an idle group sorts padding and takes nothing
  unsigned long long key[R];
#pragma unroll
  for (int r = 0; r < R; ++r) key[r] = greedy_key(a, base, gl * R + r, n);
#pragma unroll
  for (int k = 2; k <= P; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= R) {                                                               // partner: the same register of lane ^ (j / R)
        const int lj = j / R;
        const bool lower = (gl & lj) == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const unsigned long long mine = key[r];
          const unsigned long long other = __shfl_xor(mine, lj);
          const bool up = ((gl * R + r) & k) == 0;
          const bool want_min = lower == up;
          key[r] = ((other < mine) == want_min) ? other : mine;
        }
      } else {                                                                    // partner: register r ^ j of the lane itself
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & j) == 0) {
            const unsigned long long x = key[r], y = key[r | j];
            const bool up = ((gl * R + r) & k) == 0;
            const bool swap = (x > y) == up;
            key[r] = swap ? y : x;
            key[r | j] = swap ? x : y;
          }
        }
      }
    }
  }
  // powers and lockouts in sorted order, prefix sums of the powers
  float pw[R];
  uint32_t locks = 0;
  double lane_sum = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int s = gl * R + r;
    const int h = (int)(uint32_t)key[r];
    pw[r] = 0.f;
    if (s < n) {
      pw[r] = a.P_max[base + h];
      locks |= (uint32_t)((a.flags[base + h] >> 1) & 1u) << r;
    }
    lane_sum += (double)pw[r];
  }
  double incl = lane_sum;                                                         // inclusive scan over the env's lanes
#pragma unroll
  for (int d = 1; d < G; d <<= 1) {
    const double up = __shfl_up(incl, d, G);
    if (gl >= d) incl += up;
  }
  const double lane_off = incl - lane_sum;                                        // S(gl R)
  // The prefix form equals the sequential total only while every partial sum is exact in fp64: the env's powers as multiples of the
  // smallest one's last place must stay below 2^53 - exponent spread + 24 + log2(N) <= 53.  Otherwise (or with a negative power) the
  // env takes the sequential pass below.
  int ex_hi = 0, ex_lo = 255;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t u = __float_as_uint(pw[r]);
    const int ex = (int)((u >> 23) & 0xFFu);
    if ((u << 1) != 0u) {
      ex_hi = max(ex_hi, (int)(u >> 31) != 0 ? 4096 : ex);
      ex_lo = min(ex_lo, ex);
    }
  }
#pragma unroll
  for (int d = 1; d < G; d <<= 1) {
    ex_hi = max(ex_hi, __shfl_xor(ex_hi, d));
    ex_lo = min(ex_lo, __shfl_xor(ex_lo, d));
  }
  const bool exact = ex_hi - ex_lo <= 29 - (32 - __clz(max(a.N - 1, 1)));
  const double target = valid ? a.sig_old[e] : 0.0;                               // obs["reg_signal"]: the signal of the current time index
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull) << (lane & ~(G - 1));
  const int last = (lane & ~(G - 1)) + G - 1;                                      // the env's last lane
  uint32_t taken = 0;
  double total = 0.0, off = 0.0;                                                  // off = total_c - S(c)
  int c = 0;
  bool active = n > 0 && exact;
  if (__any(n > 0 && !exact)) {                                                   // the reference's pass as written, house by house
    double tot = 0.0;
    uint32_t tk = 0;
    for (int q = 0; q < G && q * R < a.N; ++q) {
      const int src = (lane & ~(G - 1)) + q;
      const uint32_t lk = __shfl(locks, src);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double p = (double)__shfl(pw[r], src);
        const bool take = q * R + r < n && greedy_take(p, tot, target, (lk >> r) & 1u);
        if (take) tot += p;
        tk |= (uint32_t)(take && q == gl) << r;
      }
    }
    if (!exact) taken = tk;
  }
  for (int guard = 0; guard <= a.N; ++guard) {
    if (!__any(active)) break;
    // run of takes from c: the first position >= c that is refused given everything from c up to it was taken
    int pos = P;
    double at = 0.0;
    double S = lane_off;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int s = gl * R + r;
      const double tb = S + off;
      const bool refuse = s >= c && s < n && !greedy_take((double)pw[r], tb, target, (locks >> r) & 1u);
      if (refuse && pos == P) { pos = s; at = tb; }
      S += (double)pw[r];
    }
    const unsigned long long refusers = __ballot(active && pos != P) & gmask;
    const int src_a = refusers != 0ull ? __ffsll((long long)refusers) - 1 : last;
    const int f_pos = __shfl(pos, src_a);
    const double f_at = __shfl(at, src_a);
    const double s_end = __shfl(S, last);                                          // the whole env taken (padding adds zero)
    const int f = refusers != 0ull ? f_pos : n;
    if (active) {
      total = refusers != 0ull ? f_at : s_end + off;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int s = gl * R + r;
        taken |= (uint32_t)(s >= c && s < f) << r;
      }
    }
    const bool more = active && f < n;
    // run of refusals from f (refused) on, the total held: the first position > f that is taken
    pos = P;
    at = 0.0;
    S = lane_off;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int s = gl * R + r;
      const bool take = s > f && s < n && greedy_take((double)pw[r], total, target, (locks >> r) & 1u);
      if (take && pos == P) { pos = s; at = S; }
      S += (double)pw[r];
    }
    const unsigned long long takers = __ballot(more && pos != P) & gmask;
    const int src_b = takers != 0ull ? __ffsll((long long)takers) - 1 : last;
    const int g_pos = __shfl(pos, src_b);
    const double g_at = __shfl(at, src_b);
    active = more && takers != 0ull;
    if (active) {
      c = g_pos;
      off = total - g_at;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int s = gl * R + r;
    if (s < n) a.actions[base + (int)(uint32_t)key[r]] = (uint8_t)((taken >> r) & 1u);
  }
}

template <int R, int G>
static void launch_greedy_wave(const StepArgs& a, hipStream_t s) {
  const int64_t waves = ((int64_t)a.E + 64 / G - 1) / (64 / G);
  hipLaunchKernelGGL((k_greedy_wave<R, G>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
}

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
  if (P <= 1024) {
    if (P <= 8) launch_greedy_wave<1, 8>(a, s);
    else if (P == 16) launch_greedy_wave<1, 16>(a, s);
    else if (P == 32) launch_greedy_wave<1, 32>(a, s);
    else if (P == 64) launch_greedy_wave<1, 64>(a, s);
    else if (P == 128) launch_greedy_wave<2, 64>(a, s);
    else if (P == 256) launch_greedy_wave<4, 64>(a, s);
    else if (P == 512) launch_greedy_wave<8, 64>(a, s);
    else launch_greedy_wave<16, 64>(a, s);
    return hipGetLastError();
  }
  const size_t lds = (size_t)P * (sizeof(unsigned long long) + sizeof(float) + 1);
  if (P <= 64) hipLaunchKernelGGL(k_greedy_myopic<64>, dim3((unsigned)a.E), dim3(64), lds, s, a, P);
  else hipLaunchKernelGGL(k_greedy_myopic<256>, dim3((unsigned)a.E), dim3(256), lds, s, a, P);
  return hipGetLastError();
}

}  // namespace mdr
