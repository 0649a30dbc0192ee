/*
 * Plain-C scalar restatement of the env step  --  TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A third, independent statement of the reference algorithm (the other two: oracle/mdr_oracle.py vectorised
 * NumPy, oracle/loop_port.py object-per-house Python), written against the same reference lines:
 *   HVAC.step ...................... env/MA_DemandResponse.py:463-492
 *   get_Q / power_consumption ...... env/MA_DemandResponse.py:494-523
 *   update_temperature (literal) ... env/MA_DemandResponse.py:664-738
 *   cluster power, rewards ......... env/MA_DemandResponse.py:1042-1050, 234-373 ; utils.py:1266-1274
 * Per-env time functions (outdoor temperature, solar gain, signal) are INPUTS here (arrays per step), produced
 * by the NumPy oracle: this file pins the per-house arithmetic and is the compiled single-core CPU baseline
 * ("what one host core does with the same algorithm without the Python interpreter").
 *
 * Parity status: pinned - tests/test_oracle_c.py replays the reference's golden vectors through it.
 * Build: make -C oracle   (gcc -O2 -shared; see oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>

typedef struct {
  int64_t nb_envs, nb_houses;
  double dt;
  double alpha_temp, alpha_sig, norm_temp, norm_sig;
  int32_t penalty_mode; /* 0 individual_L2, 1 common_L2, 2 common_max, 3 mixture */
  double mix_ind, mix_common, mix_max;
} mdrc_config;

typedef struct {
  /* state [E][N] */
  double *Ta, *Tm;
  int64_t *sso;
  uint8_t *on, *lock;
  /* parameters [E][N] */
  const double *Ua, *Cm, *Ca, *Hm, *capacity, *COP, *latent, *target, *deadband;
  const int64_t *lockout;
  /* outputs */
  double *reward; /* [E][N] */
  double *P;      /* [E] */
} mdrc_buffers;

static double deadband_l2(double target, double deadband, double value) {
  if (target + deadband / 2 < value) return (value - (target + deadband / 2)) * (value - (target + deadband / 2));
  if (target - deadband / 2 > value) return ((target - deadband / 2) - value) * ((target - deadband / 2) - value);
  return 0.0;
}

/* One env step for every env.  od_old[E]: outdoor temperature of the PREVIOUS step (env 1034); solar[E]: solar gain in W
 * at the new time; sig_old[E]: regulation signal before this step (env 196, 246); actions [E][N] (truthy = on). */
void mdrc_step(const mdrc_config *c, const mdrc_buffers *b, const uint8_t *actions, const double *od_old,
               const double *solar, const double *sig_old) {
  const int64_t E = c->nb_envs, N = c->nb_houses;
  const int64_t idt = (int64_t)c->dt;
  for (int64_t e = 0; e < E; ++e) {
    double power = 0.0, pen_sum = 0.0, pen_max = 0.0;
    for (int64_t h = 0; h < N; ++h) {
      const int64_t i = e * N + h;
      /* HVAC.step */
      if (!b->on[i]) b->sso[i] += idt;
      int lock = !(b->on[i] || b->sso[i] >= b->lockout[i]);
      if (lock) {
        b->on[i] = 0;
      } else {
        b->on[i] = actions[i] ? 1 : 0;
        if (b->on[i]) b->sso[i] = 0;
        else if (b->sso[i] + idt < b->lockout[i]) lock = 1;
      }
      b->lock[i] = (uint8_t)lock;
      /* update_temperature, Kelvin offset 273 as in the reference */
      const double Hm = b->Hm[i], Ca = b->Ca[i], Ua = b->Ua[i], Cm = b->Cm[i];
      const double odK = od_old[e] + 273, TaK = b->Ta[i] + 273, TmK = b->Tm[i] + 273;
      const double Qa = (b->on[i] ? -b->capacity[i] / (1 + b->latent[i]) : 0.0) + solar[e];
      const double a = Cm * Ca / Hm, bb = Cm * (Ua + Hm) / Hm + Ca, cc = Ua, d = Qa + Ua * odK;
      const double root = sqrt(bb * bb - 4 * a * cc);
      const double r1 = (-bb + root) / (2 * a), r2 = (-bb - root) / (2 * a);
      const double dT = Hm * TmK / Ca - (Ua + Hm) * TaK / Ca + Ua * odK / Ca + Qa / Ca;
      const double A1 = (r2 * TaK - dT - r2 * d / cc) / (r2 - r1);
      const double A2 = TaK - d / cc - A1;
      const double A3 = r1 * Ca / Hm + (Ua + Hm) / Hm, A4 = r2 * Ca / Hm + (Ua + Hm) / Hm;
      const double e1 = exp(r1 * c->dt), e2 = exp(r2 * c->dt);
      b->Ta[i] = A1 * e1 + A2 * e2 + d / cc - 273;
      b->Tm[i] = A1 * A3 * e1 + A2 * A4 * e2 + d / cc - 273;
      if (b->on[i]) power += b->capacity[i] / b->COP[i];
      const double pen = deadband_l2(b->target[i], b->deadband[i], b->Ta[i]);
      b->reward[i] = pen; /* finished below */
      pen_sum += pen / (double)N;
      if (pen > pen_max) pen_max = pen;
    }
    b->P[e] = power;
    const double s = (power - sig_old[e]) / (double)N;
    const double sig_pen = s * s;
    for (int64_t h = 0; h < N; ++h) {
      const int64_t i = e * N + h;
      double pen = b->reward[i];
      if (c->penalty_mode == 1) pen = pen_sum;
      else if (c->penalty_mode == 2) pen = pen_max;
      else if (c->penalty_mode == 3)
        pen = (c->mix_ind * pen + c->mix_common * pen_sum + c->mix_max * pen_max) / (c->mix_ind + c->mix_common + c->mix_max);
      b->reward[i] = -(c->alpha_temp * pen / c->norm_temp + c->alpha_sig * sig_pen / c->norm_sig);
    }
  }
}

/* bang-bang rule, agents/bangbang_controllers.py:41-61 */
void mdrc_bangbang(const mdrc_config *c, const mdrc_buffers *b, uint8_t *actions) {
  const int64_t n = c->nb_envs * c->nb_houses;
  for (int64_t i = 0; i < n; ++i) actions[i] = b->Ta[i] > b->target[i];
}
