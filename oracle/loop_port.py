"""Object-per-house, pure-Python scalar restatement of the env step  --  TEST INFRASTRUCTURE ONLY.

Purpose: the CPU baseline leg of bench.py (``cpu_baseline.kind == "port"``).  The reference cannot travel to
the GPU box, so its cost structure is mirrored here: one Python object per house and per HVAC, scalar
``math`` arithmetic per house per step, and a per-agent observation dict with neighbour message dicts rebuilt
every step (the reference spends ~58 % of env.step there: SURVEY.md section 3.3).  Single-threaded like the
reference.  It follows the same reference lines as oracle/mdr_oracle.py (see that file's header) and is
pinned by tests/test_loop_port.py against the same golden vectors.

Calibration measured in the build container (8 vCPU Xeon 2.1 GHz): reference 38-51 k house-steps/s per
core; this port is reported beside it in DESIGN.md.
"""
from __future__ import annotations

import datetime as _dt
import math

from oracle.mdr_oracle import SCL_COEFF, parse_config

_SCL = [[float(v) for v in row] for row in SCL_COEFF]


def _deadband_l2(target, deadband, value):  # utils.py:1266-1274
    if target + deadband / 2 < value:
        return (value - (target + deadband / 2)) ** 2
    if target - deadband / 2 > value:
        return ((target - deadband / 2) - value) ** 2
    return 0.0


class PortHvac:
    def __init__(self, capacity, cop, latent, lockout, dt):
        self.cooling_capacity, self.COP, self.latent = capacity, cop, latent
        self.lockout_duration = lockout
        self.turned_on, self.lockout, self.seconds_since_off = False, False, lockout
        self.dt = dt
        self.max_consumption = capacity / cop

    def step(self, command):  # env/MA_DemandResponse.py:463-492
        if not self.turned_on:
            self.seconds_since_off += self.dt
        self.lockout = not (self.turned_on or self.seconds_since_off >= self.lockout_duration)
        if self.lockout:
            self.turned_on = False
        else:
            self.turned_on = bool(command)
            if self.turned_on:
                self.seconds_since_off = 0
            elif self.seconds_since_off + self.dt < self.lockout_duration:
                self.lockout = True

    def heat(self):
        return -self.cooling_capacity / (1 + self.latent) if self.turned_on else 0.0

    def power(self):
        return self.max_consumption if self.turned_on else 0


class PortHouse:
    def __init__(self, idx, Ta, Tm, target, deadband, Ua, Cm, Ca, Hm, hvac):
        self.id = idx
        self.Ta, self.Tm, self.target, self.deadband = Ta, Tm, target, deadband
        self.Ua, self.Cm, self.Ca, self.Hm = Ua, Cm, Ca, Hm
        self.hvac = hvac
        self.solar = 0.0

    def update(self, od, Qsolar, dt):  # env/MA_DemandResponse.py:664-738
        Hm, Ca, Ua, Cm = self.Hm, self.Ca, self.Ua, self.Cm
        odK, TaK, TmK = od + 273, self.Ta + 273, self.Tm + 273
        self.solar = Qsolar
        Qa = self.hvac.heat() + Qsolar
        a = Cm * Ca / Hm
        b = Cm * (Ua + Hm) / Hm + Ca
        c = Ua
        d = Qa + Ua * odK
        root = math.sqrt(b * b - 4 * a * c)
        r1 = (-b + root) / (2 * a)
        r2 = (-b - root) / (2 * a)
        dT = Hm * TmK / Ca - (Ua + Hm) * TaK / Ca + Ua * odK / Ca + Qa / Ca
        A1 = (r2 * TaK - dT - r2 * d / c) / (r2 - r1)
        A2 = TaK - d / c - A1
        A3 = r1 * Ca / Hm + (Ua + Hm) / Hm
        A4 = r2 * Ca / Hm + (Ua + Hm) / Hm
        e1, e2 = math.exp(r1 * dt), math.exp(r2 * dt)
        self.Ta = A1 * e1 + A2 * e2 + d / c - 273
        self.Tm = A1 * A3 * e1 + A2 * A4 * e2 + d / c - 273

    def message(self):  # env/MA_DemandResponse.py:624-662 (default message_properties)
        return {"current_temp_diff_to_target": self.Ta - self.target,
                "hvac_seconds_since_off": self.hvac.seconds_since_off,
                "hvac_curr_consumption": self.hvac.power(),
                "hvac_max_consumption": self.hvac.max_consumption,
                "hvac_lockout_duration": self.hvac.lockout_duration}


def _solar_cooling_load(t):  # utils.py:1302-1347
    x = t.hour + t.minute / 60 - 7.5
    if x < 0 or x > 10:
        return 0.0
    y = t.month + t.day / 30 - 1
    return sum(_SCL[i][j] * x ** i * y ** j for i in range(5) for j in range(5) if _SCL[i][j] != 0.0)


class LoopPortEnv:
    """One environment; houses given explicitly (arrays of length N) or uniform defaults."""

    def __init__(self, config, params=None, od_table=None):
        s = self.spec = parse_config(config, 1)
        n = self.n = s.nb_houses
        p = params or {}
        g = lambda k, default: [float(v) for v in p[k]] if k in p else [float(default)] * n
        Ta, Tm, tg, db = g("Ta", s.init_air), g("Tm", s.init_mass), g("target", s.target), g("deadband", s.deadband)
        Ua, Cm, Ca, Hm = g("Ua", s.Ua), g("Cm", s.Cm), g("Ca", s.Ca), g("Hm", s.Hm)
        cap, cop, lat = g("capacity", s.capacity), g("COP", s.COP), g("latent", s.latent)
        lock = [int(v) for v in p["lockout"]] if "lockout" in p else [s.lockout] * n
        self.houses = [PortHouse(i, Ta[i], Tm[i], tg[i], db[i], Ua[i], Cm[i], Ca[i], Hm[i],
                                 PortHvac(cap[i], cop[i], lat[i], lock[i], s.dt)) for i in range(n)]
        self.t = _dt.datetime(1970, 1, 1) + _dt.timedelta(seconds=int(p.get("t0", s.start_epoch)))
        self.ratio = float(p.get("ratio", s.artificial_ratio))
        self.phase = float(p.get("phase", 0.0))
        self.od_table = od_table
        self.k = 0
        self.max_power = sum(h.hvac.max_consumption for h in self.houses)
        nb_comm = min(s.nb_agents_comm, n - 1)
        before, after = nb_comm // 2, nb_comm - nb_comm // 2    # env 816-828
        self.links = [[(i - before + j) % n for j in range(before)] + [(i + 1 + j) % n for j in range(after)]
                      for i in range(n)]
        self.od = self._od()
        self.signal = self._signal()
        self.power = 0
        self.norm_t = _deadband_l2(s.target, 0, s.target + 1)
        self.norm_s = _deadband_l2(s.norm_reg_sig, 0, 0.75 * s.norm_reg_sig)

    def _od(self):  # env/MA_DemandResponse.py:1057-1081 (noise comes from the table when given)
        if self.od_table is not None:
            return float(self.od_table[self.k])
        s = self.spec
        tday = self.t.hour + self.t.minute / 60.0
        return (s.day_temp - s.night_temp) / 2 * math.sin(2 * math.pi * (tday - 6 + self.phase) / 24) + (s.day_temp + s.night_temp) / 2

    def _signal(self):  # env/MA_DemandResponse.py:1236-1316, deterministic families
        s = self.spec
        base = s.avg_power_per_hvac * self.n
        sod = self.t.hour * 3600 + self.t.minute * 60 + self.t.second
        if s.signal_mode == "flat":
            sig = base
        elif s.signal_mode == "sinusoidals":
            sig = base
            for per, r in zip(s.signal_params["periods"], s.signal_params["amplitude_ratios"]):
                sig += base * r * math.sin(2 * math.pi * sod / per)
        elif s.signal_mode == "regular_steps":
            amp = s.signal_params["amplitude_per_hvac"] * self.n
            per = s.signal_params["period"]
            sig = amp if (sod % per) - (1 - base / amp) * per >= 0 else 0.0
        else:
            raise ValueError("loop port: signal mode %s not restated (use the vectorised oracle)" % s.signal_mode)
        return min(sig * self.ratio, self.max_power)

    def observations(self):  # env/MA_DemandResponse.py:904-1003, 212-232
        obs = {}
        for h in self.houses:
            v = h.hvac
            obs[h.id] = {
                "OD_temp": self.od, "datetime": self.t, "house_temp": h.Ta, "house_mass_temp": h.Tm,
                "hvac_turned_on": v.turned_on, "hvac_seconds_since_off": v.seconds_since_off, "hvac_lockout": v.lockout,
                "house_target_temp": h.target, "house_deadband": h.deadband, "house_Ua": h.Ua, "house_Cm": h.Cm,
                "house_Ca": h.Ca, "house_Hm": h.Hm, "house_solar_gain": h.solar, "hvac_COP": v.COP,
                "hvac_cooling_capacity": v.cooling_capacity, "hvac_latent_cooling_fraction": v.latent,
                "hvac_lockout_duration": v.lockout_duration,
                "message": [self.houses[j].message() for j in self.links[h.id]],
                "reg_signal": self.signal, "cluster_hvac_power": self.power,
            }
        return obs

    def step(self, actions):  # env/MA_DemandResponse.py:174-210
        s = self.spec
        self.t += _dt.timedelta(seconds=s.dt)
        self.k += 1
        Qsolar = s.window_area * s.shading * _solar_cooling_load(self.t) if s.solar_on else 0.0
        for h in self.houses:
            h.hvac.step(actions[h.id])
            h.update(self.od, Qsolar, s.dt)
        self.od = self._od()
        obs = self.observations()
        self.power = 0
        for h in self.houses:
            self.power += h.hvac.power()
        sig_pen = ((self.power - self.signal) / self.n) ** 2
        rewards = {}
        for h in self.houses:
            pen = _deadband_l2(h.target, h.deadband, h.Ta)
            rewards[h.id] = -(s.alpha_temp * pen / self.norm_t + s.alpha_sig * sig_pen / self.norm_s)
        self.signal = self._signal()
        for h in self.houses:
            obs[h.id]["reg_signal"] = self.signal
            obs[h.id]["cluster_hvac_power"] = self.power
        return obs, rewards, {h.id: False for h in self.houses}, {"cluster_hvac_power": self.power}


def bangbang(obs):  # agents/bangbang_controllers.py:41-61, one controller object per house in the reference
    return {i: o["house_temp"] > o["house_target_temp"] for i, o in obs.items()}


def time_baseline(config, seconds=12.0, min_steps=5):
    """house-steps/s of this port on one core: bang-bang closed loop, uniform default houses."""
    import time
    env = LoopPortEnv(config)
    obs = env.observations()
    steps, t0 = 0, time.perf_counter()
    while True:
        obs, *_ = env.step(bangbang(obs))
        steps += 1
        el = time.perf_counter() - t0
        if steps >= min_steps and el >= seconds:
            break
    return env.n * steps / el, steps, el


def _worker(args):
    config, seconds = args
    return time_baseline(config, seconds=seconds)[0]


def time_baseline_parallel(config, processes, seconds=6.0):
    """Aggregate house-steps/s of `processes` independent copies of the port (the reference is single-threaded; its only
    parallelism is running several processes side by side, monteCarlo.py:28-40)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")        # never fork a process that may have initialised the GPU
    with ctx.Pool(processes) as pool:
        rates = pool.map(_worker, [(config, seconds)] * processes)
    return float(sum(rates)), processes
