// Persistent sharded rollout for MI355X (gfx950, wave64): the closed loop (bang-bang or another rule-based controller) of one env whose houses span several
// workgroups - and several ranks - in ONE launch per time-table window, the houses resident in registers across steps, the
// per-step exchange (cluster power sum, env 1042-1050; penalty sum / max, env 274-321) through a MAILBOX instead of a
// kernel boundary and a collective.
//
//   house workgroup b of env e   steps its 1024 houses (VEC = 4; 256 when nb_houses % 4 != 0), reduces them to the very
//                                (power sum, penalty sum, penalty max) record k_step_partial writes, and publishes the record
//                                into the mailbox of EVERY rank (its own included) - 8-byte granules {tag, 32 data bits}, each
//                                written by ONE write-through store (sc1; sc0 sc1 towards a peer device), so a granule is its
//                                own flag and needs no fence (cdna_hip_programming.md Guideline 16, R2);
//   reducer workgroups of env e  (blockIdx.x >= nblk; reducer j of R takes the steps s with s % R == j - one suffices while a
//                                thread holds one record, four share the work at 977) poll the world * records granule sets of the
//                                step, re-sums them in the fixed order of k_step_finish (thread t: ranks in order, records
//                                t, t + 256, ...; then the workgroup tree) - the totals are bit-identical on every rank and
//                                to the records path - and publishes the totals as granules; it also keeps the per-env
//                                accumulators (power trace, squared signal error);
//   house workgroups             pick the totals up DEPTH steps later (the state does not depend on them - only the
//                                rewards do - so the houses run ahead and the exchange latency is overlapped) and add the
//                                step's rewards to the running sums in step order.  Inside a house workgroup the exchange is
//                                spread over the waves: wave 1 pushes the record (lane = destination rank x granule, one
//                                store), wave 0 picks the totals up at the END of an iteration for the next one.
//
// Tags count steps over the life of the env handle (never reset, never a per-launch memset: a peer may already be pushing
// into this rank's mailbox when the launch begins); slot = tag mod SLOTS.  Every spin is bounded: on expiry the workgroup
// writes an error word into the mailbox header of every rank and leaves; the others see the word and leave too; nothing
// is written back (the state lives in registers), so a failed launch leaves the buffers as they were.
//
// Reference: env/MA_DemandResponse.py:1005-1055 (ClusterHouses.step), 234-373 (rewards); main-deploy.py:99-148 (the loop).
#include <algorithm>
#include <cstdlib>

#include "mdr_device.h"
#include "mdr_kernels.h"
#include "mdr_step_common.h"

namespace mdr {

typedef __attribute__((address_space(1))) unsigned long long gu64;

template <bool SYS>
__device__ __forceinline__ void granule_store(gu64* p, uint32_t tag, uint32_t value) {
  const unsigned long long x = ((unsigned long long)tag << 32) | value;
  if (SYS) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool SYS>
__device__ __forceinline__ unsigned long long granule_load(const gu64* p) {
  if (SYS) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int64_t rec_offset(const PersistArgs& m, int E, int slot, int e, int r, int b) {
  return PERSIST_HDR + ((((int64_t)slot * E + e) * m.world + r) * m.stride + b) * PERSIST_G;
}
__device__ __forceinline__ int64_t tot_offset(const PersistArgs& m, int E, int slot, int e) {
  return PERSIST_HDR + (int64_t)PERSIST_SLOTS * E * m.world * m.stride * PERSIST_G + ((int64_t)slot * E + e) * PERSIST_TOT;
}

enum : uint32_t { PERSIST_FAIL_TOTALS = 1, PERSIST_FAIL_RECORDS = 2 };

// error word (granule 0 of every rank's header): {tag | kind << 28 | workgroup}; a spinner that finds it set leaves as well.
// Lane r < world keeps rank r's header address (`abort_ptr`): the eight mailbox pointers then need not stay live in scalar
// registers across the step loops for the sake of this cold path (they were spilled to vector lanes and reloaded every step).
template <bool SYS>
__device__ __forceinline__ void raise_abort(gu64* abort_ptr, uint32_t tag, uint32_t kind) {
  if (abort_ptr != nullptr) granule_store<SYS>(abort_ptr, tag, (kind << 28) | (blockIdx.x & 0x0FFFFFFFu));
}
template <bool SYS>
__device__ __forceinline__ bool abort_raised(const gu64* own) {
  return granule_load<SYS>(own) != 0ull;
}

// BB: the bang-bang rule compiled in (the default controller; the general rule costs this latency-bound loop 0.3 us per step)
// Experiment build -DMDR_PERSIST_TRACE=1 (tools/scratch): cycle stamps of house workgroup 0 / the reducer of env 0 into the
// power-trace buffer, 8 per step: where a step of the pipeline spends its time.  Never in the product library.
#if defined(MDR_PERSIST_TRACE) && MDR_PERSIST_TRACE
#define MDR_STAMP(buf, step, slot, on) \
  do { if ((on) && (buf) != nullptr) reinterpret_cast<unsigned long long*>(buf)[(int64_t)(step) * 16 + (slot)] = (unsigned long long)clock64(); } while (0)
#define MDR_NOTE(buf, step, slot, on, value) \
  do { if ((on) && (buf) != nullptr) reinterpret_cast<unsigned long long*>(buf)[(int64_t)(step) * 16 + (slot)] = (unsigned long long)(value); } while (0)
#else
#define MDR_STAMP(buf, step, slot, on) do { } while (0)
#define MDR_NOTE(buf, step, slot, on, value) do { } while (0)
#endif

template <int VEC, bool SYS, bool BB>
__global__ __launch_bounds__(256, 4) void k_rollout_persist(StepArgs a, RolloutArgs ro, PersistArgs m) {
  const int D = m.depth;   // steps the houses run ahead of the totals (<= PERSIST_MAX_DEPTH)
  // dynamic LDS: [D + 1][256][VEC] floats - each house's own penalty of the steps in flight - then the env's table rows of this
  // launch: sig_old[T], sig_new[T] (f64), od[T], solar[T] (f32).  The rows are wave-uniform, but with the granule stores in the loop
  // hipcc would fetch them by VECTOR loads, and a wait for one of those also waits for the totals / records in flight behind it
  // (vmcnt retires in order): from LDS they cost a broadcast read on the other counter.
  extern __shared__ __attribute__((aligned(16))) float lds_hist[];
  double* const row_sig_old = reinterpret_cast<double*>(lds_hist + (size_t)(D + 1) * 256 * VEC);
  double* const row_sig_new = row_sig_old + ro.nsteps;
  float* const row_od = reinterpret_cast<float*>(row_sig_new + ro.nsteps);
  float* const row_solar = row_od + ro.nsteps;
  __shared__ double lds_part[2][3 * 4];
  __shared__ double lds_tot[2][3];
  __shared__ int lds_fail[2];
  __shared__ int s_nrec[MDR_MAX_SHARDS];
  const bool need_pen = a.penalty_mode != MDR_PENALTY_INDIVIDUAL_L2;
  const int ng = need_pen ? PERSIST_G : 2;   // granules that travel: the power sum alone unless a common penalty mode needs the rest
  const int e = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and known to be: the wave's roles below become scalar branches
  const int nblk = m.nrec[m.rank];
  const int T = ro.nsteps;
  const bool want_terr = ro.sq_temp_error_sum != nullptr;
  gu64* const own = (gu64*)m.box[m.rank];
  gu64* abort_ptr = nullptr;
#pragma unroll
  for (int r = 0; r < MDR_MAX_SHARDS; ++r)
    if (r < m.world && lane == r) abort_ptr = (gu64*)m.box[r];
  if (tid < MDR_MAX_SHARDS) s_nrec[tid] = tid < m.world ? m.nrec[tid] : 0;
  if (tid < 2) lds_fail[tid] = 0;
  for (int t = tid; t < ro.nsteps; t += 256) {
    const int64_t row = (int64_t)t * a.E + blockIdx.y;
    row_sig_old[t] = a.sig_old[row];
    row_sig_new[t] = a.sig_new[row];
    row_od[t] = a.od_old[row];
    row_solar[t] = a.solar_new[row];
  }
  __syncthreads();

  if (blk >= nblk) {
    // ---------------------------------------------------------------- reducer j of env e on this rank: the steps s = j, j + R, ...
    const int R = m.reducers, j = blk - nblk;
    double serr = 0.0, P_last = 0.0;
    Red3 tot{0.0, 0.0, 0.0f};
    const int last = want_terr ? T : T - 1;   // the pseudo-step T carries the workgroups' squared temperature errors
    // Thread t re-sums ranks in order, records t, t + 256, ... - the order of finish_block (mdr_kernels.hip).  Where its first
    // four records sit (relative to the slot's base) never changes: worked out once; they are fetched for step s + 1 while
    // step s is reduced and published, so a step of the reducer costs no load latency as long as the houses run ahead.
    int64_t off[4];
    bool have[4];
    int r_more = 0, b_more = tid;   // where the fifth record would be (r_more == world: there is none - every practical shape)
    {
      int r = 0, b = tid;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        while (r < m.world && b >= s_nrec[r]) { ++r; b = tid; }
        have[k] = r < m.world;
        off[k] = have[k] ? ((int64_t)r * m.stride + b) * PERSIST_G : 0;
        if (have[k]) b += 256;
      }
      while (r < m.world && b >= s_nrec[r]) { ++r; b = tid; }
      r_more = r;
      b_more = b;
    }
    unsigned long long x[4][PERSIST_G];
    auto fetch = [&](int step) {
      const uint32_t tag = m.tag_base + (uint32_t)step;
      const gu64* base = own + rec_offset(m, a.E, (int)(tag % PERSIST_SLOTS), e, 0, 0);
      const int g_now = (step == T) ? 2 : ng;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int g = 0; g < PERSIST_G; ++g)
          x[k][g] = (have[k] && g < g_now) ? granule_load<SYS>(base + off[k] + g) : ((unsigned long long)tag << 32);
    };
    if (j <= last) fetch(j);
    // (`par`: the reductions of THIS reducer alternate between the two LDS scratch sets - with an even number of reducers the
    // parity of its steps never changes, and one barrier per reduction only separates a set's readers from the writers after next)
    int par = 0;
    for (int s = j; s <= last; s += R, par ^= 1) {
      const uint32_t tag = m.tag_base + (uint32_t)s;
      const int slot = (int)(tag % PERSIST_SLOTS);
      const int g_now = (s == T) ? 2 : ng;
      Red3 acc{0.0, 0.0, 0.0f};
      bool failed = false;
      uint32_t spins = 0;
      for (;;) {
        acc = Red3{0.0, 0.0, 0.0f};
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int g = 0; g < PERSIST_G; ++g) ok &= (uint32_t)(x[k][g] >> 32) == tag;
          if (have[k]) {
            acc.sum_p += __hiloint2double((int)(uint32_t)x[k][1], (int)(uint32_t)x[k][0]);
            acc.sum_pen += __hiloint2double((int)(uint32_t)x[k][3], (int)(uint32_t)x[k][2]);
            acc.max_pen = fmaxf(acc.max_pen, __uint_as_float((uint32_t)x[k][4]));
          }
        }
        for (int r = r_more, b = b_more; r < m.world;) {   // more than four records per thread: on demand, same order
          const gu64* rec = own + rec_offset(m, a.E, slot, e, r, b);
          unsigned long long y[PERSIST_G];
#pragma unroll
          for (int g = 0; g < PERSIST_G; ++g) y[g] = (g < g_now) ? granule_load<SYS>(rec + g) : ((unsigned long long)tag << 32);
#pragma unroll
          for (int g = 0; g < PERSIST_G; ++g) ok &= (uint32_t)(y[g] >> 32) == tag;
          acc.sum_p += __hiloint2double((int)(uint32_t)y[1], (int)(uint32_t)y[0]);
          acc.sum_pen += __hiloint2double((int)(uint32_t)y[3], (int)(uint32_t)y[2]);
          acc.max_pen = fmaxf(acc.max_pen, __uint_as_float((uint32_t)y[4]));
          b += 256;
          while (r < m.world && b >= s_nrec[r]) { ++r; b = tid; }
        }
        if (ok) break;
        ++spins;
        if (spins > m.spin_limit || ((spins & 63u) == 0u && abort_raised<SYS>(own))) {
          failed = true;
          if (spins > m.spin_limit) raise_abort<SYS>(abort_ptr, tag, PERSIST_FAIL_RECORDS);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        fetch(s);
      }
      MDR_STAMP(ro.power_trace, s, 8, tid == 0 && e == 0);
      MDR_NOTE(ro.power_trace, s, 11, tid == 0 && e == 0, spins);
      if (s + R <= last) fetch(s + R);   // in flight across the reduction and the publication below
      if (failed) lds_fail[par] = 1;
      tot = block_reduce<256>(acc, lds_part[par]);   // one barrier: the failure flag rides on it
      if (lds_fail[par]) return;
      MDR_STAMP(ro.power_trace, s, 9, tid == 0 && e == 0);
      if (s == T) {
        if (tid == 0) ro.sq_temp_error_sum[e] += tot.sum_p;
        break;
      }
      if (tid < ng) {   // the totals, one granule per lane
        const uint32_t v = tid == 0 ? (uint32_t)__double2loint(tot.sum_p) : tid == 1 ? (uint32_t)__double2hiint(tot.sum_p)
                         : tid == 2 ? (uint32_t)__double2loint(tot.sum_pen) : tid == 3 ? (uint32_t)__double2hiint(tot.sum_pen)
                                    : __float_as_uint(tot.max_pen);
        granule_store<SYS>(own + tot_offset(m, a.E, slot, e) + tid, tag, v);
      }
      MDR_STAMP(ro.power_trace, s, 10, tid == 0 && e == 0);
      if (tid == 0) {
        const int64_t row = (int64_t)s * a.E + e;
#if !(defined(MDR_PERSIST_TRACE) && MDR_PERSIST_TRACE)
        if (ro.power_trace) ro.power_trace[row] = tot.sum_p;
#endif
        const double d = row_sig_new[s] - tot.sum_p;
        serr += d * d;
        P_last = tot.sum_p;
      }
    }
    if (tid == 0 && T > 0) {
      if ((T - 1) % R == j) a.P[e] = P_last;     // the reducer of the last step
      if (ro.sq_signal_error_sum) {
        if (R == 1) ro.sq_signal_error_sum[e] += serr;
        else m.serr_part[(int64_t)j * a.E + e] = serr;   // launch_persist_combine adds the partial sums in reducer order
      }
    }
    return;
  }

  // ------------------------------------------------------------------ house workgroup
  const int h = (blk * 256 + tid) * VEC;
  const bool live = h < a.N;
  const int64_t i = (int64_t)e * a.N + h;
  HouseIn hs[VEC];
  float rsum[VEC], pen_now[VEC], pen_old[VEC];   // pen_old: the penalty of step it - D, back from LDS
  uint64_t on_m[VEC], lock_m[VEC], cmd_m[VEC];   // the HVAC bits and the latest command as lane masks (house_advance_m)
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    hs[v] = HouseIn{};
    hs[v].lockout = 1;
    rsum[v] = 0.0f;
    pen_now[v] = pen_old[v] = 0.0f;
  }
  if (live) {
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
    for (int v = 0; v < VEC; ++v)
      hs[v] = HouseIn{Ta[v], Tm[v], sso[v], fl[v], k01[v], s0[v], k10[v], s1[v], iu[v], q[v], pm[v], tg[v], db[v], lk[v]};
    if (ro.reward_sum) load_vec<VEC>(ro.reward_sum, i, rsum);   // continue the caller's running sum in step order
  }
  // The state and parameters must have ARRIVED before the loop: a wait for them inside it (hipcc puts one where a register is first
  // used) is, from the second iteration on, a wait for the record granules and the totals request in flight - the counter cannot tell.
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    asm volatile("" : "+v"(hs[v].Ta), "+v"(hs[v].Tm), "+v"(hs[v].sso), "+v"(hs[v].flags), "+v"(hs[v].k01), "+v"(hs[v].s0), "+v"(hs[v].k10));
    asm volatile("" : "+v"(hs[v].s1), "+v"(hs[v].inv_Ua), "+v"(hs[v].Q_hvac), "+v"(hs[v].P_max), "+v"(hs[v].target), "+v"(hs[v].deadband),
                 "+v"(hs[v].lockout), "+v"(rsum[v]));
    on_m[v] = __builtin_amdgcn_ballot_w64((hs[v].flags & 1u) != 0u);
    lock_m[v] = __builtin_amdgcn_ballot_w64((hs[v].flags & 2u) != 0u);
    cmd_m[v] = 0;
  }
  // A single wavefront issues one instruction every four cycles whatever its kind: this loop is bound by its INSTRUCTION COUNT on the
  // busiest wave (r03 trace: 840 per step, half of them scalar address arithmetic and spilled-pointer reloads).  So: what changes from
  // step to step in an address is the slot alone - each lane keeps its destination at slot 0 and adds slot * stride - and the
  // exchange is spread over the waves: wave 0 picks the totals up, wave 1 pushes the record - lane L = (destination rank, granule).
  constexpr int PUSH_WAVE = 1;
  const int64_t rec_slot_stride = (int64_t)a.E * m.world * m.stride * PERSIST_G;   // granules from a slot to the next
  const int64_t tot_slot_stride = (int64_t)a.E * PERSIST_TOT;
  const int push_r = lane / ng, push_g = lane - (lane / ng) * ng;
  const bool pusher = wave == PUSH_WAVE && push_r < m.world;
  gu64* push_base = nullptr;
#pragma unroll
  for (int r = 0; r < MDR_MAX_SHARDS; ++r)
    if (pusher && push_r == r) push_base = (gu64*)m.box[r] + rec_offset(m, a.E, 0, e, m.rank, blk) + push_g;
  const gu64* const tot_base = own + tot_offset(m, a.E, 0, e) + min(lane, ng - 1);
  double terr = 0.0;
  Red3 tot{0.0, 0.0, 0.0f};
  float sig_term = 0.0f;
  unsigned long long pre = 0ull;   // wave 0: the totals granule of the NEXT step to pick up, in flight
  bool have_pre = false;
  for (int it = 0, ring = 0; it < T + D; ++it, ring = (ring == D ? 0 : ring + 1)) {   // ring = it mod (D + 1)
    const int par = it & 1;
    [[maybe_unused]] const bool tr = tid == 0 && blk == 0 && e == 0 && it < T;   // (trace build only)
    MDR_STAMP(ro.power_trace, it, 0, tr);
    // the step `it` itself: needs nothing from the other workgroups
    Red3 acc{0.0, 0.0, 0.0f};
    if (it < T) {
      const float od_old = row_od[it], solar = row_solar[it];
      {   // every lane, idle ones too (blank houses: exact zeros in every sum): the lane masks need wave-uniform control flow
        float p = 0.0f, ps = 0.0f, te = 0.0f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          cmd_m[v] = BB ? __builtin_amdgcn_ballot_w64(hs[v].Ta > hs[v].target)   // agents/bangbang_controllers.py
                        : controller_cmd_m(a.action_source, hs[v].Ta, hs[v].target, hs[v].deadband, on_m[v]);
          const HouseNextM o = house_advance_m(hs[v], on_m[v], cmd_m[v], od_old, solar, a.dt);
          hs[v].Ta = o.Ta;
          hs[v].Tm = o.Tm;
          hs[v].sso = live ? o.sso : 0;
          on_m[v] = o.on;
          lock_m[v] = o.lock;
          pen_now[v] = o.pen;
          p += o.power;
          ps += o.pen;
          acc.max_pen = fmaxf(acc.max_pen, o.pen);
          if (want_terr) {
            const float d = o.Ta - hs[v].target;
            te = fmaf(d, d, te);
          }
        }
        acc.sum_p = (double)p;
        acc.sum_pen = (double)ps;
        terr += (double)te;
        store_vec<VEC>(lds_hist, ((int64_t)ring * 256 + tid) * VEC, pen_now);   // read back by this very thread D iterations on
      }
      acc = lanes_reduce<64>(acc, need_pen);
      if (lane == 0) {
        lds_part[par][wave] = acc.sum_p;
        lds_part[par][4 + wave] = acc.sum_pen;
        lds_part[par][8 + wave] = (double)acc.max_pen;
      }
    }
    MDR_STAMP(ro.power_trace, it, 2, tr);
    __syncthreads();   // the wave partials of step `it`; and what wave 0 picked up at the end of the iteration before (lds_tot / lds_fail [par])
    MDR_STAMP(ro.power_trace, it, 3, tr);
    if (lds_fail[par]) return;
    if (it < T && wave == PUSH_WAVE) {
      // this workgroup's record: the same arithmetic as block_reduce (the wave partials re-added in order), pushed by ONE store -
      // lane (rank r, granule g) writes granule g of the record into rank r's mailbox
      Red3 rec{0.0, 0.0, 0.0f};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        rec.sum_p += lds_part[par][w];
        rec.sum_pen += lds_part[par][4 + w];
        rec.max_pen = fmaxf(rec.max_pen, (float)lds_part[par][8 + w]);
      }
      if (pusher) {
        const uint32_t tag = m.tag_base + (uint32_t)it;
        const uint32_t v = push_g == 0 ? (uint32_t)__double2loint(rec.sum_p) : push_g == 1 ? (uint32_t)__double2hiint(rec.sum_p)
                         : push_g == 2 ? (uint32_t)__double2loint(rec.sum_pen) : push_g == 3 ? (uint32_t)__double2hiint(rec.sum_pen)
                                       : __float_as_uint(rec.max_pen);
        granule_store<SYS>(push_base + (int64_t)(tag % PERSIST_SLOTS) * rec_slot_stride, tag, v);
      }
    }
    MDR_STAMP(ro.power_trace, it, 4, tr);
    if (it >= D) {   // rewards of step it - D, in step order
      tot.sum_p = lds_tot[par][0];
      tot.sum_pen = lds_tot[par][1];
      tot.max_pen = (float)lds_tot[par][2];
      sig_term = signal_term(a, tot.sum_p, row_sig_old[it - D]);
      if (live) load_vec<VEC>(lds_hist, ((int64_t)(ring == D ? 0 : ring + 1) * 256 + tid) * VEC, pen_old);   // slot of step it - D
      if (ro.reward_sum != nullptr && live) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) rsum[v] = __fadd_rn(rsum[v], reward_value(a, pen_old[v], tot.sum_pen, tot.max_pen, sig_term));
      }
    }
    // Wave 0, at the END of the iteration: it picks up the totals the NEXT iteration's rewards need (step it + 1 - D; published by the
    // reducer D - 1 steps of work ago) and leaves them in LDS behind that iteration's barrier, then requests the totals after them.
    // A request thus has a whole iteration - compute, barrier, rewards - before its pick-up: a device-scope load is ~2000 cycles
    // away, and picked up in the middle of the next iteration (r03, first form) wave 0 stood ~700 of them at the wait.
    if (wave == 0 && it + 1 >= D && it + 1 < T + D) {
      const int nxt_par = par ^ 1;
      const uint32_t tag = m.tag_base + (uint32_t)(it + 1 - D);
      // (lanes past the granules that travel read the last one of them: every lane loads, no lane patches its register - a
      // constant moved into the register of a load in flight would cost a wait for everything in flight)
      const gu64* src = tot_base + (int64_t)(tag % PERSIST_SLOTS) * tot_slot_stride;
      uint32_t val = 0;
      bool failed = false;
      unsigned long long x = have_pre ? pre : granule_load<SYS>(src);
      uint32_t spins = 0;
      for (;;) {
        val = (uint32_t)x;
        if (__all((uint32_t)(x >> 32) == tag)) break;
        ++spins;
        if (spins > m.spin_limit || ((spins & 63u) == 0u && abort_raised<SYS>(own))) {
          failed = true;
          if (spins > m.spin_limit) raise_abort<SYS>(abort_ptr, tag, PERSIST_FAIL_TOTALS);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        x = granule_load<SYS>(src);
      }
      MDR_NOTE(ro.power_trace, it, 6, tr, spins);
      const int v0 = __builtin_amdgcn_readlane((int)val, 0), v1 = __builtin_amdgcn_readlane((int)val, 1);
      const int v2 = need_pen ? __builtin_amdgcn_readlane((int)val, 2) : 0, v3 = need_pen ? __builtin_amdgcn_readlane((int)val, 3) : 0;
      const int v4 = need_pen ? __builtin_amdgcn_readlane((int)val, 4) : 0;   // (granules that did not travel: zeros)
      if (lane == 0) {
        lds_tot[nxt_par][0] = __hiloint2double(v1, v0);
        lds_tot[nxt_par][1] = __hiloint2double(v3, v2);
        lds_tot[nxt_par][2] = (double)__int_as_float(v4);
        if (failed) lds_fail[nxt_par] = 1;
      }
      have_pre = it + 2 < T + D;
      if (have_pre) pre = granule_load<SYS>(tot_base + (int64_t)((tag + 1u) % PERSIST_SLOTS) * tot_slot_stride);
    }
    MDR_STAMP(ro.power_trace, it, 5, tr);
  }
  if (T <= 0) return;
  if (want_terr) {   // pseudo-step T: this workgroup's squared temperature error travels as one more record
    Red3 r{terr, 0.0, 0.0f};
    r = block_reduce<256>(r, lds_part[(T + D) & 1]);
    if (tid < 2) {
      const uint32_t tag = m.tag_base + (uint32_t)T;
      const uint32_t v = tid == 0 ? (uint32_t)__double2loint(r.sum_p) : (uint32_t)__double2hiint(r.sum_p);
      const int64_t off = rec_offset(m, a.E, (int)(tag % PERSIST_SLOTS), e, m.rank, blk) + tid;
#pragma unroll
      for (int rr = 0; rr < MDR_MAX_SHARDS; ++rr)
        if (rr < m.world) granule_store<SYS>((gu64*)m.box[rr] + off, tag, v);
    }
  }
  if (!live) return;
  // final state, and the last step's outputs exactly as the single-step kernels leave them
  const float o_sig = (float)(row_sig_new[T - 1] * a.inv_obs_norm);
  const float o_pow = (float)(tot.sum_p * a.inv_obs_norm);
  float nTa[VEC], nTm[VEC], pen[VEC];
  int nsso[VEC], lk[VEC];
  unsigned nfl[VEC], act[VEC];
  HouseOut o[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    nTa[v] = hs[v].Ta;
    nTm[v] = hs[v].Tm;
    nsso[v] = hs[v].sso;
    nfl[v] = house_flags(lane_bit(on_m[v]), lane_bit(lock_m[v]));
    act[v] = lane_bit(cmd_m[v]) ? 1u : 0u;
    lk[v] = hs[v].lockout;
    pen[v] = pen_old[v];
    o[v] = HouseOut{hs[v].Ta, hs[v].Tm, hs[v].sso, nfl[v], pen_old[v], 0.0f};
  }
  store_vec<VEC>(a.Ta, i, nTa);
  store_vec<VEC>(a.Tm, i, nTm);
  store_vec<VEC>(a.sso, i, nsso);
  store_bytes<VEC>(a.flags, i, nfl);
  if (a.actions != nullptr) store_bytes<VEC>(a.actions, i, act);
  store_obs_local<VEC>(a, i, o, lk);
  store_reward_power<VEC>(a, i, pen, tot.sum_pen, tot.max_pen, sig_term, o_sig, o_pow);
  if (ro.reward_sum) store_vec<VEC>(ro.reward_sum, i, rsum);
}

// One reducer while a thread holds at most one record of a step, then one more per 256 records (977 records - 1,000,000 houses on
// one rank, or 8 ranks x 123 - take four): a reducer's step is a round of device-scope loads past the L2s plus a workgroup
// reduction, ~1.5 us at four records per thread, and the houses wait for it; the reducers take the steps in turn.
int persist_reducers(int64_t records_of_all_ranks) {
  static const int knob = [] { const char* t = getenv("MDR_PERSIST_REDUCERS"); return t ? atoi(t) : 0; }();
  if (knob >= 1) return std::min(knob, PERSIST_MAX_REDUCERS);
  return (int)std::max<int64_t>(1, std::min<int64_t>(PERSIST_MAX_REDUCERS, (records_of_all_ranks + 255) / 256));
}

__global__ __launch_bounds__(256) void k_persist_combine(const double* part, int R, int E, double* out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  double sum = 0.0;
  for (int j = 0; j < R; ++j) sum += part[(int64_t)j * E + e];
  out[e] += sum;
}

hipError_t launch_persist_combine(const PersistArgs& m, int E, double* sq_signal_error_sum, hipStream_t s) {
  if (m.reducers <= 1 || sq_signal_error_sum == nullptr) return hipSuccess;
  hipLaunchKernelGGL(k_persist_combine, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, m.serr_part, m.reducers, E, sq_signal_error_sum);
  return hipGetLastError();
}

int64_t persist_mailbox_granules(int E, int world, int stride) {
  return PERSIST_HDR + (int64_t)PERSIST_SLOTS * E * world * stride * PERSIST_G + (int64_t)PERSIST_SLOTS * E * PERSIST_TOT;
}

static size_t persist_lds_bytes(int vec, int depth, int nsteps = PERSIST_MAX_STEPS) {
  return (size_t)(depth + 1) * 256 * vec * sizeof(float) + (size_t)nsteps * 24;
}

template <typename K>
static hipError_t persist_capacity(K kernel, size_t lds, int64_t* blocks) {
  int dev = 0, per_cu = 0, cus = 0;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) return err;
  err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds);
  if (err != hipSuccess) return err;
  err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (err != hipSuccess) return err;
  // the runtime's answer can be one workgroup per CU above what the hardware admits once a kernel needs more than 80 scalar
  // registers (MI355X_MICROARCH.md, residency): this kernel is built for four workgroups per CU (launch bounds), count no more
  *blocks = (int64_t)std::min(per_cu, 4) * cus;
  return hipSuccess;
}

hipError_t persist_resident_blocks(int vec, bool sys, int depth, int64_t* blocks) {   // (the general-controller forms: they hold the most registers)
  const size_t lds = persist_lds_bytes(vec, depth);
  if (vec == 4) return sys ? persist_capacity(k_rollout_persist<4, true, false>, lds, blocks) : persist_capacity(k_rollout_persist<4, false, false>, lds, blocks);
  return sys ? persist_capacity(k_rollout_persist<1, true, false>, lds, blocks) : persist_capacity(k_rollout_persist<1, false, false>, lds, blocks);
}

hipError_t launch_rollout_persist(const StepArgs& a, const RolloutArgs& r, const PersistArgs& m, bool sys, hipStream_t s) {
  if (m.depth < 1 || m.depth > PERSIST_MAX_DEPTH || r.nsteps < 1 || r.nsteps > PERSIST_MAX_STEPS) return hipErrorInvalidValue;
  if (m.reducers < 1 || m.reducers > PERSIST_MAX_REDUCERS || m.reducers > m.depth) return hipErrorInvalidValue;
  if (m.reducers > 1 && r.sq_signal_error_sum != nullptr && m.serr_part == nullptr) return hipErrorInvalidValue;
  const dim3 g((unsigned)(m.nrec[m.rank] + m.reducers), (unsigned)a.E), b(256);
  const bool bb = a.action_source == MDR_ACTIONS_BANGBANG;
#define MDR_PERSIST_LAUNCH(VECV)                                                                                      \
  do {                                                                                                                \
    const size_t lds = persist_lds_bytes(VECV, m.depth, r.nsteps);                                                    \
    if (sys) {                                                                                                        \
      if (bb) hipLaunchKernelGGL((k_rollout_persist<VECV, true, true>), g, b, lds, s, a, r, m);                       \
      else hipLaunchKernelGGL((k_rollout_persist<VECV, true, false>), g, b, lds, s, a, r, m);                         \
    } else {                                                                                                          \
      if (bb) hipLaunchKernelGGL((k_rollout_persist<VECV, false, true>), g, b, lds, s, a, r, m);                      \
      else hipLaunchKernelGGL((k_rollout_persist<VECV, false, false>), g, b, lds, s, a, r, m);                        \
    }                                                                                                                 \
  } while (0)
  if (a.N % 4 == 0) MDR_PERSIST_LAUNCH(4);
  else MDR_PERSIST_LAUNCH(1);
#undef MDR_PERSIST_LAUNCH
  return hipGetLastError();
}

}  // namespace mdr
