// Device-side building blocks shared by the reset / table / step kernels (gfx950 only).
//
// Reference citations are to /root/reference (zhimaerfan/marl-demandresponse-original @ 2025-03-14).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdr {

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al., SC'11).  Stream layout:
//   key     = (seed lo, seed hi)
//   counter = (global env index, item index, episode, stream tag)
// ---------------------------------------------------------------------------------------------
enum : uint32_t {
  TAG_HOUSE_TEMPS = 1,   // x0,x1 -> gauss(init air) ; x2,x3 -> gauss(init mass)
  TAG_HOUSE_HVAC = 2,    // x0,x1 -> gauss(target) ; x2 -> capacity choice ; x3 -> lockout noise
  TAG_HOUSE_THERMO = 3,  // x0..x3 -> triangular factors of Ua, Cm, Ca, Hm
  TAG_ENV_START = 4,     // x0 -> days ; x1 -> seconds ; x2 -> phase ; x3 -> artificial ratio
  TAG_OD_NOISE = 5,      // item = time index ; x0,x1 -> gauss
  TAG_PERLIN = 6,        // item = lattice index ; x0 -> gradient
  TAG_COMM = 7,          // item = house ; counter word 2 = time index ; x0..x3 -> link defects (4 links per draw)
  TAG_LINKS = 9,         // item = house ; counter word 2 = time index ; x0..x3 -> round keys of the sender permutation
  TAG_INTERP = 8,        // item = draw index ; counter word 2 = time index ; tag | episode << 8 ; x0 -> sampled house
  ENV_LEVEL = 0xFFFFFFFFu
};

struct u32x4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0, p1 = (uint64_t)M1 * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += W0;
    k1 += W1;
  }
  return {c0, c1, c2, c3};
}

// uint32 -> double in the open interval (0,1)
__device__ __forceinline__ double u01(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

// Box-Muller standard normal
__device__ __forceinline__ double gauss01(uint32_t a, uint32_t b) {
  return sqrt(-2.0 * log(u01(a))) * cos(6.283185307179586476925286766559 * u01(b));
}

// integer in [0, n): (x * n) >> 32
__device__ __forceinline__ int64_t mulhi_pick(uint32_t x, uint32_t n) { return (int64_t)(((uint64_t)x * (uint64_t)n) >> 32); }

// random.triangular(low, high, mode = 1) as CPython evaluates it; reference use utils.py:640-666
__device__ __forceinline__ double triangular_mode1(double u, double low, double high) {
  if (high == low) return low;
  double c = (1.0 - low) / (high - low);
  if (u > c) {
    u = 1.0 - u;
    c = 1.0 - c;
    const double t = low;
    low = high;
    high = t;
  }
  return low + (high - low) * sqrt(u * c);
}

// ---------------------------------------------------------------------------------------------
// Calendar: naive epoch seconds -> civil fields (era-based days algorithm, H. Hinnant, public domain)
// ---------------------------------------------------------------------------------------------
struct Civil {
  int month, day, hour, minute, sod, yday;  // yday = tm_yday (1..366)
};

__device__ __forceinline__ Civil civil_from_epoch(int64_t t) {
  int64_t days = t / 86400;
  int64_t sod = t - days * 86400;
  if (sod < 0) {
    sod += 86400;
    days -= 1;
  }
  const int64_t z = days + 719468;
  const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  const int64_t doe = z - era * 146097;
  const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const int64_t mp = (5 * doy + 2) / 153;
  Civil c;
  c.day = (int)(doy - (153 * mp + 2) / 5 + 1);
  c.month = (int)(mp < 10 ? mp + 3 : mp - 9);
  c.sod = (int)sod;
  c.hour = c.sod / 3600;
  c.minute = (c.sod % 3600) / 60;
  // day of year counted from 1 January: doy above counts from 1 March
  const int64_t y = yoe + era * 400 + (c.month <= 2 ? 1 : 0);
  const bool leap = (y % 4 == 0 && y % 100 != 0) || (y % 400 == 0);
  c.yday = (int)(c.month > 2 ? doy + 60 + (leap ? 1 : 0) : doy - 305);
  return c;
}

// ---------------------------------------------------------------------------------------------
// Solar cooling load, W per m^2 of glazing (utils.house_solar_gain, utils.py:1302-1347): a polynomial
// regression in x = hour + minute/60 - 7.5 and y = month + day/30 - 1, zero outside 0 <= x <= 10.
// Coefficients arranged as C[i][j] of x^i y^j and evaluated by nested Horner.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double solar_cooling_load(int hour, int minute, int month, int day) {
  const double x = (double)hour + (double)minute / 60.0 - 7.5;
  if (x < 0.0 || x > 10.0) return 0.0;
  const double y = (double)month + (double)day / 30.0 - 1.0;
  const double r0 = 4.36579418e01 + y * (8.76635241e01 + y * (-1.47795612e01 + y * (1.04354810e00 + y * -3.97855577e-02)));
  const double r1 = 1.58055357e02 + y * (-3.73313090e01 + y * (4.68950855e00 + y * -1.18302764e-01));
  const double r2 = -4.55944821e01 + y * (3.24275366e00 + y * (-4.56096472e-01 + y * 1.56398008e-02));
  const double r3 = 5.78827663e00 + y * (2.12969604e-02 + y * (2.58881400e-03 + y * -5.11397219e-04));
  const double r4 = -2.71446436e-01;
  return r0 + x * (r1 + x * (r2 + x * (r3 + x * r4)));
}

// ---------------------------------------------------------------------------------------------
// 1-D gradient ("Perlin") noise with the octave structure of utils.Perlin (utils.py:1231-1253).
// Lattice gradients come from the PERLIN Philox stream of the env (this build's own lattice: the
// third-party perlin_noise package is absent from the image, parity unpinned for the VALUES).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double fade5(double w) { return ((6.0 * w - 15.0) * w + 10.0) * w * w * w; }

__device__ __forceinline__ double lattice_noise_1d(double x, uint32_t env, uint32_t episode, uint32_t k0, uint32_t k1) {
  const double l0 = floor(x);
  const double d0 = x - l0;
  const double d1 = d0 - 1.0;
  const uint32_t li = (uint32_t)(int64_t)l0;
  const double g0 = 2.0 * u01(philox4x32_10(env, li, episode, TAG_PERLIN, k0, k1).x) - 1.0;
  const double g1 = 2.0 * u01(philox4x32_10(env, li + 1u, episode, TAG_PERLIN, k0, k1).x) - 1.0;
  return fade5(1.0 - d0) * g0 * d0 + fade5(1.0 + d1) * g1 * d1;
}

// ---------------------------------------------------------------------------------------------
// Thermal model.  SingleHouse.update_temperature (env/MA_DemandResponse.py:664-738) is the closed-form
// solution of a linear 2-node ETP circuit over one time step with Qm = 0; it is therefore a constant
// 2x2 map M on (Ta - Tinf, Tm - Tinf), Tinf = T_od + Qa/Ua.  The kernels apply it
// in difference form
//     Ta' = Ta + k01 (Tm - Ta) + s0 (Ta - Tinf),   s0 = m00 + m01 - 1
//     Tm' = Tm + k10 (Ta - Tm) + s1 (Tm - Tinf),   s1 = m10 + m11 - 1
// (coefficients computed here in fp64) so that fp32 rounding of the large common-mode term Tinf is
// multiplied by the small row-sum leak s0/s1 only.
// ---------------------------------------------------------------------------------------------
struct ThermalMap {
  double k01, s0, k10, s1;
};

__device__ __forceinline__ ThermalMap thermal_map(double Ua, double Cm, double Ca, double Hm, double dt) {
  const double a = Cm * Ca / Hm;
  const double b = Cm * (Ua + Hm) / Hm + Ca;
  const double root = sqrt(b * b - 4.0 * a * Ua);
  const double r1 = (-b + root) / (2.0 * a);
  const double r2 = (-b - root) / (2.0 * a);
  const double A3 = r1 * Ca / Hm + (Ua + Hm) / Hm;
  const double A4 = r2 * Ca / Hm + (Ua + Hm) / Hm;
  const double ax = (r2 + (Hm + Ua) / Ca) / (r2 - r1);
  const double ay = -(Hm / Ca) / (r2 - r1);
  const double e1 = exp(r1 * dt), e2 = exp(r2 * dt);
  const double m00 = ax * e1 + (1.0 - ax) * e2;
  const double m01 = ay * (e1 - e2);
  const double m10 = ax * A3 * e1 + (1.0 - ax) * A4 * e2;
  const double m11 = ay * (A3 * e1 - A4 * e2);
  ThermalMap m;
  m.k01 = m01;
  m.s0 = (m00 - 1.0) + m01;
  m.k10 = m10;
  m.s1 = m10 + (m11 - 1.0);
  return m;
}

// ---------------------------------------------------------------------------------------------
// One house, one step (fp32).  HVAC.step (env 463-492) branch-free, then the thermal map, the
// temperature penalty utils.deadbandL2 (utils.py:1266-1274) and the electric power (env 511-523).
// ---------------------------------------------------------------------------------------------
struct HouseIn {
  float Ta, Tm;
  int sso;
  unsigned flags;
  float k01, s0, k10, s1, inv_Ua, Q_hvac, P_max, target, deadband;
  int lockout;
};

struct HouseOut {
  float Ta, Tm;
  int sso;
  unsigned flags;
  float pen, power;
};

// The same step with the HVAC's on / lockout bits as booleans: the multi-step kernels carry them from step to step as lane masks
// (scalar registers) instead of packing them into `flags` after every step and unpacking them before the next.
struct HouseNext {
  float Ta, Tm;
  int sso;
  bool on, lock;
  float pen, power;
};

__device__ __forceinline__ HouseNext house_advance(const HouseIn& h, bool on, bool cmd, float od_old, float solar, int dt) {
  const int sso1 = on ? h.sso : h.sso + dt;          // env 475-476
  const bool can = on || (sso1 >= h.lockout);        // env 478-481
  const bool on2 = can && cmd;                       // env 483-486
  const int sso2 = on2 ? 0 : sso1;                   // env 487-488
  const bool lock2 = !can || (!on2 && (sso2 + dt < h.lockout));  // env 489-492
  const float Qa = (on2 ? h.Q_hvac : 0.0f) + solar;  // env 690-699
  const float Tinf = fmaf(Qa, h.inv_Ua, od_old);     // uses the PREVIOUS step's outdoor temperature (env 1034)
  const float dm = h.Tm - h.Ta;
  HouseNext o;
  o.Ta = h.Ta + fmaf(h.k01, dm, h.s0 * (h.Ta - Tinf));
  o.Tm = h.Tm + fmaf(-h.k10, dm, h.s1 * (h.Tm - Tinf));
  o.sso = sso2;
  o.on = on2;
  o.lock = lock2;
  const float hi = fmaf(0.5f, h.deadband, h.target);
  const float lo = fmaf(-0.5f, h.deadband, h.target);
  const float above = o.Ta - hi, below = lo - o.Ta;
  const float excess = above > 0.0f ? above : (below > 0.0f ? below : 0.0f);   // one select chain and ONE product: the same bits as
  o.pen = excess * excess;                                                     // squaring inside each arm, without the arms' branches
  o.power = on2 ? h.P_max : 0.0f;
  return o;
}

// ... and with the bits of all 64 lanes as LANE MASKS - wave-uniform 64-bit values in scalar registers.  The HVAC's little state
// machine is then scalar-unit logic on whole masks between the vector compares that feed it (a compare writes its mask straight
// into a scalar register pair), and a mask steers a per-lane select for free (inverse ballot): 9 vector instructions per house
// instead of ~18 with per-lane booleans, which the compiler materialises as 0 / 1 in vector registers around every step.  Must be
// called in wave-uniform control flow (a mask defined under a divergent branch is not one value per wave).  Same arithmetic, same bits.
struct HouseNextM {
  float Ta, Tm;
  int sso;
  uint64_t on, lock;
  float pen, power;
};

__device__ __forceinline__ bool lane_bit(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

__device__ __forceinline__ HouseNextM house_advance_m(const HouseIn& h, uint64_t on_m, uint64_t cmd_m, float od_old, float solar, int dt) {
  const int sso1 = lane_bit(on_m) ? h.sso : h.sso + dt;                                        // env 475-476
  const uint64_t can_m = on_m | __builtin_amdgcn_ballot_w64(sso1 >= h.lockout);                // env 478-481
  const uint64_t on2_m = can_m & cmd_m;                                                        // env 483-486
  const bool on2 = lane_bit(on2_m);
  const int sso2 = on2 ? 0 : sso1;                                                             // env 487-488
  const uint64_t lock2_m = ~can_m | (~on2_m & __builtin_amdgcn_ballot_w64(sso2 + dt < h.lockout));   // env 489-492
  const float Qa = (on2 ? h.Q_hvac : 0.0f) + solar;  // env 690-699
  const float Tinf = fmaf(Qa, h.inv_Ua, od_old);     // uses the PREVIOUS step's outdoor temperature (env 1034)
  const float dm = h.Tm - h.Ta;
  HouseNextM o;
  o.Ta = h.Ta + fmaf(h.k01, dm, h.s0 * (h.Ta - Tinf));
  o.Tm = h.Tm + fmaf(-h.k10, dm, h.s1 * (h.Tm - Tinf));
  o.sso = sso2;
  o.on = on2_m;
  o.lock = lock2_m;
  const float hi = fmaf(0.5f, h.deadband, h.target);
  const float lo = fmaf(-0.5f, h.deadband, h.target);
  const float above = o.Ta - hi, below = lo - o.Ta;
  const float excess = above > 0.0f ? above : (below > 0.0f ? below : 0.0f);
  o.pen = excess * excess;
  o.power = on2 ? h.P_max : 0.0f;
  return o;
}

__device__ __forceinline__ unsigned house_flags(bool on, bool lock) { return (on ? 1u : 0u) | (lock ? 2u : 0u); }

__device__ __forceinline__ HouseOut house_step(const HouseIn& h, bool cmd, float od_old, float solar, int dt) {
  const HouseNext n = house_advance(h, (h.flags & 1u) != 0u, cmd, od_old, solar, dt);
  return HouseOut{n.Ta, n.Tm, n.sso, house_flags(n.on, n.lock), n.pen, n.power};
}

// a / L for integer-valued a and L (exact in fp32) with y = RN(1 / L): q = RN(a y), r = a - q L (exact in one fma),
// RN(q + r y) is the correctly rounded quotient (Markstein; L's significand is never all ones below 2^24 - 1; checked
// exhaustively for a <= 20000, L <= 3000), i.e. the very bits the `/` of the other kernels gives, in 3 ops instead of ~10
__device__ __forceinline__ float div_by_lockout(float a, float L, float y) {
  if (L == 0.0f) return a / L;   // inf / nan exactly as the division produces them
  const float q = a * y;
  return __fmaf_rn(__fmaf_rn(-q, L, a), y, q);
}

// ---------------------------------------------------------------------------------------------
// Reductions over one env: 64-lane wavefront butterflies, then LDS across the waves of a workgroup.
// ---------------------------------------------------------------------------------------------
struct Red3 {
  double sum_p, sum_pen;
  float max_pen;
};

// Cross-lane data movement by DPP (data-parallel primitives: a VALU modifier, no LDS crossbar, ~VALU latency).
// Control words: quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror (i <-> 7-i), row_mirror (i <-> 15-i) swap a lane
// with its partner at distance 1, 2, 4, 8 (as seen by a symmetric reduction); row_bcast15 / row_bcast31 (gfx9 family)
// feed the last lane of the previous 16-lane row(s) into the next.  Lanes whose row is masked out receive 0.
constexpr int DPP_QUAD_SWAP1 = 0xB1, DPP_QUAD_SWAP2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140,
              DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// With every row enabled each lane is written from a valid lane, so bound_ctrl (zero for invalid sources) changes nothing but
// spares the compiler the v_mov 0 that otherwise seeds the destination before each DPP move.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
  return __hiloint2double(hi, lo);
}

// One exchange step on M independent triples at once (the envs that share a lane group, mdr_multi.hip): their dependent chains
// interleave inside one basic block.  PEN: the penalty sum / max are wanted too.  The multi-step kernels pass pen = false in the
// individual_L2 mode, where only the cluster power is reduced - a third of their per-step vector instructions were these exchanges.
template <int CTRL, int ROW_MASK, bool PEN, int M>
__device__ __forceinline__ void red3_dpp(Red3* v) {
#pragma unroll
  for (int m = 0; m < M; ++m) {
    v[m].sum_p += dpp_f64<CTRL, ROW_MASK>(v[m].sum_p);
    if constexpr (PEN) {
      v[m].sum_pen += dpp_f64<CTRL, ROW_MASK>(v[m].sum_pen);
      v[m].max_pen = fmaxf(v[m].max_pen, dpp_f32<CTRL, ROW_MASK>(v[m].max_pen));   // penalties are >= 0: 0 is the identity
    }
  }
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <int WIDTH, bool SKIP1, bool PEN, int M>
__device__ __forceinline__ void lanes_reduce_steps(Red3* v) {
  if constexpr (WIDTH >= 2 && !SKIP1) red3_dpp<DPP_QUAD_SWAP1, 0xF, PEN, M>(v);
  if constexpr (WIDTH >= 4) red3_dpp<DPP_QUAD_SWAP2, 0xF, PEN, M>(v);
  if constexpr (WIDTH >= 8) red3_dpp<DPP_ROW_HALF_MIRROR, 0xF, PEN, M>(v);
  if constexpr (WIDTH >= 16) red3_dpp<DPP_ROW_MIRROR, 0xF, PEN, M>(v);
  if constexpr (WIDTH == 32) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      v[m].sum_p += __shfl_xor(v[m].sum_p, 16, 64);
      if constexpr (PEN) {
        v[m].sum_pen += __shfl_xor(v[m].sum_pen, 16, 64);
        v[m].max_pen = fmaxf(v[m].max_pen, __shfl_xor(v[m].max_pen, 16, 64));
      }
    }
  }
  if constexpr (WIDTH == 64) {
    red3_dpp<DPP_ROW_BCAST15, 0xA, PEN, M>(v);   // rows 1 and 3 += row 0 / row 2 totals
    red3_dpp<DPP_ROW_BCAST31, 0xC, PEN, M>(v);   // rows 2 and 3 += the total of rows 0-1: lane 63 holds everything
#pragma unroll
    for (int m = 0; m < M; ++m) {
      v[m].sum_p = readlane_f64(v[m].sum_p, 63);
      if constexpr (PEN) {
        v[m].sum_pen = readlane_f64(v[m].sum_pen, 63);
        v[m].max_pen = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[m].max_pen), 63));
      }
    }
  }
}

// Reduction over groups of WIDTH consecutive lanes (WIDTH a power of two <= 64); every lane of a group ends with the
// group's totals.  Distances 1..8 by DPP; 16 by one bpermute butterfly (WIDTH == 32) or, for the full wavefront, the
// row broadcasts + a scalar read of lane 63.  `pen` is wave-uniform: ONE branch picks the form with or without the penalties.
// SKIP1: neighbouring lanes 2 j, 2 j + 1 already hold the same pair total - the tree starts at distance 2 (mdr_multi.hip k_rollout_pairs).
template <int WIDTH, bool SKIP1 = false, int M = 1>
__device__ __forceinline__ void lanes_reduce_n(Red3* v, bool pen) {
  if (pen) lanes_reduce_steps<WIDTH, SKIP1, true, M>(v);
  else lanes_reduce_steps<WIDTH, SKIP1, false, M>(v);
}
template <int WIDTH, bool SKIP1 = false>
__device__ __forceinline__ Red3 lanes_reduce(Red3 v, bool pen = true) {
  lanes_reduce_n<WIDTH, SKIP1, 1>(&v, pen);
  return v;
}

template <int THREADS>
__device__ __forceinline__ Red3 block_reduce(Red3 v, double* lds /* [3][THREADS/64] */, bool pen = true) {
  constexpr int WAVES = THREADS / 64;
  v = lanes_reduce<64>(v, pen);
  if constexpr (WAVES == 1) return v;
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    lds[wave] = v.sum_p;
    if (pen) {
      lds[WAVES + wave] = v.sum_pen;
      lds[2 * WAVES + wave] = (double)v.max_pen;
    }
  }
  __syncthreads();
  Red3 t{0.0, 0.0, 0.0f};
#pragma unroll
  for (int w = 0; w < WAVES; ++w) t.sum_p += lds[w];  // same order in every thread: identical totals, no second barrier
  if (pen) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      t.sum_pen += lds[WAVES + w];
      t.max_pen = fmaxf(t.max_pen, (float)lds[2 * WAVES + w]);
    }
  }
  return t;
}

}  // namespace mdr
