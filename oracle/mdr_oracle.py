"""CPU oracle for the MADemandResponseEnv step path  --  TEST INFRASTRUCTURE ONLY.

This file is the parity checker, not the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product (``mdr_amd``) never imports anything under ``oracle/``.

It is an independent fp64 NumPy restatement, vectorised over ``[E envs, N houses]``,
of the reference algorithm (paths relative to /root/reference):

* HVAC lockout state machine ............ env/MA_DemandResponse.py:463-492 (init 432-434)
* HVAC heat / electric power ............ env/MA_DemandResponse.py:494-523
* 2-node ETP closed-form update ......... env/MA_DemandResponse.py:664-738
* solar cooling load polynomial ......... utils.py:1277-1350
* outdoor temperature sinusoid + noise .. env/MA_DemandResponse.py:1057-1081
* cluster power sum ..................... env/MA_DemandResponse.py:1042-1050
* rewards (4 temperature-penalty modes) . env/MA_DemandResponse.py:234-373, utils.py:1266-1274
* regulation signal (4 families) ........ env/MA_DemandResponse.py:1236-1316, utils.py:1231-1253
* per-episode parameter sampling ........ utils.py:573-709, env/MA_DemandResponse.py:430,789-793,1116
* step ordering (old OD temp in the thermal update, old signal in the reward,
  new datetime for the solar gain) ...... env/MA_DemandResponse.py:174-210, 1005-1055

Parity status: PINNED for everything except the Perlin lattice values.  The
restatement is checked against golden vectors produced by importing the reference
itself in the build container (tests/golden/make_golden.py, fixtures committed
under tests/golden/) and against the reference's own known-answer HVAC test
(env/unit_tests_MA_DemandResponse.py:36-77).  The third-party ``perlin_noise``
package the reference calls (utils.py:8,1243-1252) is absent from the image and
unpinned by any reference test, so the *lattice gradient values* of the Perlin
signal are "parity unpinned"; its wiring (octave frequencies, weights, clipping,
ratio, max-power clamp) is pinned by driving the reference with this file's own
lattice (`lattice_noise_1d`) as the stand-in.

Random numbers: the reference draws from Python's MT19937 stream, which a
counter-based device generator cannot reproduce.  The oracle therefore defines
the *same* Philox4x32-10 streams as the HIP kernels (see DESIGN.md "RNG streams")
so that device-sampled episodes can be compared draw by draw; golden-vector
replays bypass sampling and load the reference's own parameters and noise.
"""
from __future__ import annotations

import copy
import datetime as _dt
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------- #
# Philox4x32-10 (Salmon, Moraes, Dror, Shaw - "Parallel random numbers: as easy
# as 1, 2, 3", SC'11).  Restated from the paper; checked against the published
# Random123 known-answer vectors in tests/test_oracle_rng.py.
# --------------------------------------------------------------------------- #
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)

# stream (domain) tags, counter word 3
TAG_HOUSE_TEMPS = 1   # x0,x1 -> gauss(init air) ; x2,x3 -> gauss(init mass)
TAG_HOUSE_HVAC = 2    # x0,x1 -> gauss(target)   ; x2 -> capacity choice ; x3 -> lockout noise
TAG_HOUSE_THERMO = 3  # x0..x3 -> triangular factors for Ua, Cm, Ca, Hm
TAG_ENV_START = 4     # x0 -> days ; x1 -> seconds ; x2 -> phase ; x3 -> artificial ratio
TAG_OD_NOISE = 5      # counter word 1 = time index ; x0,x1 -> gauss
TAG_PERLIN = 6        # counter word 1 = lattice index ; x0 -> gradient
TAG_COMM = 7          # counter word 1 = receiving house, word 2 = time index, tag | (slot >> 2) << 8 ; word (slot & 3) -> link defect
TAG_LINKS = 9         # counter word 1 = receiving house, word 2 = time index, tag | (draw >> 2) << 8 ; word (draw & 3) -> sender draw
COMM_KEY_MIX = 0x85EBCA6B   # streams 7 and 9 carry the time index where the others carry the episode: the episode is folded into key word 1
TAG_INTERP = 8        # counter word 1 = draw index q, word 2 = time index ; x0 -> house sampled for the base power
ENV_LEVEL = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32 with 10 rounds.  All inputs broadcastable, uint32 range."""
    c0 = np.asarray(c0, dtype=np.uint64) & _MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & _MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & _MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & _MASK32
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def seed_key(seed: int) -> Tuple[int, int]:
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32


def u01(x):
    """uint32 -> float64 in the open interval (0, 1): (x + 0.5) * 2^-32."""
    return (np.asarray(x, dtype=np.float64) + 0.5) * (1.0 / 4294967296.0)


def gauss01(xa, xb):
    """Box-Muller standard normal from two uint32 words."""
    return np.sqrt(-2.0 * np.log(u01(xa))) * np.cos((2.0 * np.pi) * u01(xb))


def mulhi_pick(x, n):
    """Unbiased-enough integer in [0, n): (x * n) >> 32, exact integer arithmetic."""
    return ((np.asarray(x, dtype=np.uint64) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def triangular_mode1(u, low, high):
    """random.triangular(low, high, 1) as CPython computes it (reference use: utils.py:640-666)."""
    u = np.asarray(u, dtype=np.float64)
    if high == low:
        return np.full(u.shape, float(low))
    c = (1.0 - low) / (high - low)
    flip = u > c
    uu = np.where(flip, 1.0 - u, u)
    cc = np.where(flip, 1.0 - c, c)
    lo = np.where(flip, high, low)
    hi = np.where(flip, low, high)
    return lo + (hi - lo) * np.sqrt(uu * cc)


# --------------------------------------------------------------------------- #
# Calendar (proleptic Gregorian, naive / UTC) from integer epoch seconds
# --------------------------------------------------------------------------- #
_EPOCH = _dt.datetime(1970, 1, 1)


def to_epoch_seconds(d: _dt.datetime) -> int:
    return int((d - _EPOCH).total_seconds())


def civil_from_epoch(t):
    """epoch seconds (int64 array) -> dict(year, month, day, hour, minute, second, sod, yday).

    Days -> (y, m, d) uses the era-based algorithm (public-domain, H. Hinnant's
    "chrono-compatible low-level date algorithms")."""
    t = np.asarray(t, dtype=np.int64)
    days = np.floor_divide(t, 86400)
    sod = t - days * 86400
    z = days + 719468
    era = np.floor_divide(z, 146097)
    doe = z - era * 146097
    yoe = (doe - doe // 1460 + doe // 36524 - doe // 146096) // 365
    y = yoe + era * 400
    doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
    mp = (5 * doy + 2) // 153
    d = doy - (153 * mp + 2) // 5 + 1
    m = np.where(mp < 10, mp + 3, mp - 9)
    y = np.where(m <= 2, y + 1, y)
    leap = ((y % 4 == 0) & (y % 100 != 0)) | (y % 400 == 0)
    cum = np.array([0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334], dtype=np.int64)
    yday = cum[m - 1] + d + np.where(leap & (m > 2), 1, 0)
    return dict(year=y, month=m, day=d, hour=sod // 3600, minute=(sod % 3600) // 60,
                second=sod % 60, sod=sod, yday=yday)


# --------------------------------------------------------------------------- #
# Physics helpers
# --------------------------------------------------------------------------- #
# Solar cooling load regression (utils.py:1302-1347): SCL = sum_ij C[i][j] x^i y^j,
# rows = power of x (0..4), columns = power of y (0..4).
SCL_COEFF = np.array([
    # y^0            y^1              y^2              y^3              y^4
    [4.36579418e01, 8.76635241e01, -1.47795612e01, 1.04354810e00, -3.97855577e-02],   # x^0
    [1.58055357e02, -3.73313090e01, 4.68950855e00, -1.18302764e-01, 0.0],            # x^1
    [-4.55944821e01, 3.24275366e00, -4.56096472e-01, 1.56398008e-02, 0.0],           # x^2
    [5.78827663e00, 2.12969604e-02, 2.58881400e-03, -5.11397219e-04, 0.0],           # x^3
    [-2.71446436e-01, 0.0, 0.0, 0.0, 0.0],                                           # x^4
])


def solar_cooling_load(hour, minute, month, day):
    """W/m^2 of glazing; zero outside 07:30-17:30 (utils.py:1302-1304)."""
    x = np.asarray(hour, dtype=np.float64) + np.asarray(minute, dtype=np.float64) / 60.0 - 7.5
    y = np.asarray(month, dtype=np.float64) + np.asarray(day, dtype=np.float64) / 30.0 - 1.0
    acc = np.zeros(np.broadcast(x, y).shape)
    for i in range(5):
        for j in range(5):
            if SCL_COEFF[i, j] != 0.0:
                acc = acc + SCL_COEFF[i, j] * x ** i * y ** j
    return np.where((x < 0) | (x > 10), 0.0, acc)


def deadband_l2(target, deadband, value):
    """utils.py:1266-1274."""
    hi = target + deadband / 2.0
    lo = target - deadband / 2.0
    return np.where(value > hi, (value - hi) ** 2, np.where(value < lo, (lo - value) ** 2, 0.0))


def etp_closed_form(Ta, Tm, od, Qa, Ua, Cm, Ca, Hm, dt):
    """Literal closed-form two-exponential solution (env/MA_DemandResponse.py:681-738).

    Kelvin offset is 273 as in the reference (685-687); Qm = 0 (701)."""
    odK = od + 273.0
    TaK = Ta + 273.0
    TmK = Tm + 273.0
    a = Cm * Ca / Hm
    b = Cm * (Ua + Hm) / Hm + Ca
    c = Ua
    d = Qa + Ua * odK
    root = np.sqrt(b * b - 4.0 * a * c)
    r1 = (-b + root) / (2.0 * a)
    r2 = (-b - root) / (2.0 * a)
    dTa0 = Hm * TmK / Ca - (Ua + Hm) * TaK / Ca + Ua * odK / Ca + Qa / Ca
    A1 = (r2 * TaK - dTa0 - r2 * d / c) / (r2 - r1)
    A2 = TaK - d / c - A1
    A3 = r1 * Ca / Hm + (Ua + Hm) / Hm
    A4 = r2 * Ca / Hm + (Ua + Hm) / Hm
    e1 = np.exp(r1 * dt)
    e2 = np.exp(r2 * dt)
    newTa = A1 * e1 + A2 * e2 + d / c
    newTm = A1 * A3 * e1 + A2 * A4 * e2 + d / c
    return newTa - 273.0, newTm - 273.0


def etp_affine_coefficients(Ua, Cm, Ca, Hm, dt):
    """The same update written as a constant 2x2 map on (Ta-Tinf, Tm-Tinf).

    Returns m00, m01, m10, m11.  Used by tests to prove the affine form the kernels
    use is the closed form above, not by the oracle's own stepping."""
    a = Cm * Ca / Hm
    b = Cm * (Ua + Hm) / Hm + Ca
    root = np.sqrt(b * b - 4.0 * a * Ua)
    r1 = (-b + root) / (2.0 * a)
    r2 = (-b - root) / (2.0 * a)
    A3 = r1 * Ca / Hm + (Ua + Hm) / Hm
    A4 = r2 * Ca / Hm + (Ua + Hm) / Hm
    ax = (r2 + (Hm + Ua) / Ca) / (r2 - r1)
    ay = -(Hm / Ca) / (r2 - r1)
    e1 = np.exp(r1 * dt)
    e2 = np.exp(r2 * dt)
    return (ax * e1 + (1 - ax) * e2, ay * (e1 - e2),
            ax * A3 * e1 + (1 - ax) * A4 * e2, ay * (A3 * e1 - A4 * e2))


def hvac_transition(on, sso, lockout_duration, cmd, dt):
    """One HVAC.step (env/MA_DemandResponse.py:463-492) for arrays.  Returns on', lock', sso'."""
    sso1 = np.where(on, sso, sso + dt)
    can = on | (sso1 >= lockout_duration)
    on2 = can & cmd
    sso2 = np.where(on2, 0, sso1)
    lock2 = (~can) | ((~on2) & (sso2 + dt < lockout_duration))
    return on2, lock2, sso2


def fade5(w):
    return ((6.0 * w - 15.0) * w + 10.0) * w * w * w


def lattice_noise_1d(x, gradient_fn):
    """1-D gradient noise: sum over the two bracketing lattice points l of
    fade(1-|x-l|) * g(l) * (x-l), g(l) in (-1, 1).  This is the published structure of
    the `perlin_noise` package the reference calls; gradient VALUES are this build's own."""
    x = np.asarray(x, dtype=np.float64)
    l0 = np.floor(x)
    d0 = x - l0
    d1 = d0 - 1.0
    g0 = gradient_fn(l0.astype(np.int64))
    g1 = gradient_fn(l0.astype(np.int64) + 1)
    return fade5(1.0 - d0) * g0 * d0 + fade5(1.0 + d1) * g1 * d1


def perlin_octaves(x_over_period, gradient_fn, nb_octaves, octaves_step):
    """utils.Perlin.calculate_noise (utils.py:1247-1253) incl. the last-octave weight 1/(2**n - 1)."""
    total = 0.0
    for j in range(nb_octaves):
        n = lattice_noise_1d(x_over_period * (2 ** j * octaves_step), gradient_fn)
        w = 1.0 / (2 ** j) if j < nb_octaves - 1 else 1.0 / (2 ** nb_octaves - 1)
        total = total + n * w
    return total


# --------------------------------------------------------------------------- #
# Interpolated base power (monteCarlo/interpolation.py:113-142, env/MA_DemandResponse.py:1195-1234)
# --------------------------------------------------------------------------- #
INTERP_KEYS = ("Ua_ratio", "Cm_ratio", "Ca_ratio", "Hm_ratio", "air_temp", "mass_temp", "OD_temp", "HVAC_power", "hour", "date")
INTERP_NEAREST = (0, 1, 2, 3, 7)      # thermal ratios and HVAC power: nearest grid value (interpolation.py:121-134)
INTERP_LINEAR = (4, 5, 6, 8, 9)       # air, mass, OD temperature differences, hour, date: multilinear (interpn)


class InterpGrid:
    """PowerInterpolator.interpolateGridFast restated for arrays of points.

    values: flat array in the reference's C order over the 10 axes of `axes` (dict key -> grid coordinates)."""

    def __init__(self, values, axes: dict, keys=INTERP_KEYS):
        self.keys = list(keys)
        if tuple(self.keys) != INTERP_KEYS:
            raise ValueError("interpolation grid axes must be " + ", ".join(INTERP_KEYS))
        self.axes = [np.asarray(axes[k], dtype=np.float64) for k in self.keys]
        self.dims = [len(a) for a in self.axes]
        self.values = np.asarray(values, dtype=np.float64).reshape(self.dims)

    def clip(self, point):
        """utils.clipInterpolationPoint (utils.py:1214-1221) on [..., 10] points."""
        out = np.array(point, dtype=np.float64, copy=True)
        for d, ax in enumerate(self.axes):
            out[..., d] = np.clip(out[..., d], ax.min(), ax.max())
        return out

    def lookup(self, point):
        p = self.clip(point)
        idx = [None] * 10
        for d in INTERP_NEAREST:
            idx[d] = np.argmin(np.abs(self.axes[d][None, :] - p[..., d].reshape(-1, 1)), axis=1).reshape(p.shape[:-1])
        lo, w = {}, {}
        for d in INTERP_LINEAR:
            ax = self.axes[d]
            i = np.clip(np.searchsorted(ax, p[..., d], side="right") - 1, 0, len(ax) - 2)
            lo[d] = i
            w[d] = (p[..., d] - ax[i]) / (ax[i + 1] - ax[i])
        total = np.zeros(p.shape[:-1])
        for corner in range(32):
            weight = np.ones(p.shape[:-1])
            for bit, d in enumerate(INTERP_LINEAR):
                up = (corner >> bit) & 1
                idx[d] = lo[d] + up
                weight = weight * (w[d] if up else 1.0 - w[d])
            total = total + weight * self.values[tuple(idx)]
        return total


# --------------------------------------------------------------------------- #
# Config parsing (schema: config.py of the reference)
# --------------------------------------------------------------------------- #
PENALTY_MODES = ("individual_L2", "common_L2", "common_max", "mixture")


@dataclass
class OracleSpec:
    nb_envs: int
    nb_houses: int
    dt: int
    # defaults
    init_air: float
    init_mass: float
    target: float
    deadband: float
    Ua: float
    Cm: float
    Ca: float
    Hm: float
    window_area: float
    shading: float
    solar_on: bool
    COP: float
    capacity: float
    latent: float
    lockout: int
    lockout_noise: int
    # noise
    std_start: float
    std_target: float
    f_low: float
    f_high: float
    cap_list: List[float]
    # start time
    start_mode: str
    start_epoch: int
    # weather
    day_temp: float
    night_temp: float
    temp_std: float
    random_phase: bool
    # grid
    base_power_mode: str
    avg_power_per_hvac: float
    signal_mode: str
    signal_params: dict
    artificial_ratio: float
    ratio_range: float
    # reward
    alpha_temp: float
    alpha_sig: float
    norm_reg_sig: float
    penalty_mode: str
    mix: Tuple[float, float, float]
    # obs normalisation
    cfg_nb_agents: int
    nb_agents_comm: int
    comm_mode: str


def parse_config(config: dict, nb_envs: int = 1) -> OracleSpec:
    env = config["default_env_prop"]
    house = config["default_house_prop"]
    hvac = config["default_hvac_prop"]
    nh = config["noise_house_prop"]
    nv = config["noise_hvac_prop"]
    cl = env["cluster_prop"]
    pg = env["power_grid_prop"]
    rw = env["reward_prop"]
    hp = nh["noise_parameters"][nh["noise_mode"]]
    vp = nv["noise_parameters"][nv["noise_mode"]]
    tp = cl["temp_parameters"][cl["temp_mode"]]
    if env["start_datetime_mode"] not in ("random", "fixed"):
        raise ValueError("start_datetime_mode must be random or fixed")
    if pg["base_power_mode"] not in ("constant", "interpolation"):
        raise ValueError("The base_power_mode parameter in the config file can only be 'constant' or 'interpolation'")
    mode = pg["signal_mode"]
    if not (mode in ("flat", "sinusoidals", "regular_steps") or "perlin" in mode):
        raise ValueError("Invalid power grid signal mode: {}".format(mode))
    if rw["temp_penalty_mode"] not in PENALTY_MODES:
        raise ValueError("Unknown temperature penalty mode: {}".format(rw["temp_penalty_mode"]))
    if rw["sig_penalty_mode"] != "common_L2":
        raise ValueError("Unknown signal penalty mode: {}".format(rw["sig_penalty_mode"]))
    mixp = rw["temp_penalty_parameters"].get("mixture", {})
    start = _dt.datetime.strptime(env["start_datetime"], "%Y-%m-%d %H:%M:%S")
    return OracleSpec(
        nb_envs=nb_envs, nb_houses=cl["nb_agents"], dt=int(env["time_step"]),
        init_air=house["init_air_temp"], init_mass=house["init_mass_temp"],
        target=house["target_temp"], deadband=house["deadband"],
        Ua=house["Ua"], Cm=house["Cm"], Ca=house["Ca"], Hm=house["Hm"],
        window_area=house["window_area"], shading=house["shading_coeff"],
        solar_on=bool(house["solar_gain_bool"]),
        COP=hvac["COP"], capacity=hvac["cooling_capacity"], latent=hvac["latent_cooling_fraction"],
        lockout=int(hvac["lockout_duration"]), lockout_noise=int(hvac["lockout_noise"]),
        std_start=hp["std_start_temp"], std_target=hp["std_target_temp"],
        f_low=hp["factor_thermo_low"], f_high=hp["factor_thermo_high"],
        cap_list=[float(c) for c in vp["cooling_capacity_list"][hvac["cooling_capacity"]]],
        start_mode=env["start_datetime_mode"], start_epoch=to_epoch_seconds(start),
        day_temp=tp["day_temp"], night_temp=tp["night_temp"], temp_std=tp["temp_std"],
        random_phase=bool(tp["random_phase_offset"]),
        base_power_mode=pg["base_power_mode"],
        avg_power_per_hvac=pg["base_power_parameters"]["constant"]["avg_power_per_hvac"],
        signal_mode=mode, signal_params=copy.deepcopy(pg["signal_parameters"][mode]),
        artificial_ratio=pg["artificial_ratio"], ratio_range=pg["artificial_signal_ratio_range"],
        alpha_temp=rw["alpha_temp"], alpha_sig=rw["alpha_sig"], norm_reg_sig=rw["norm_reg_sig"],
        penalty_mode=rw["temp_penalty_mode"],
        mix=(mixp.get("alpha_ind_L2", 1), mixp.get("alpha_common_L2", 1), mixp.get("alpha_common_max", 0)),
        cfg_nb_agents=cl["nb_agents"], nb_agents_comm=cl["nb_agents_comm"], comm_mode=cl["agents_comm_mode"],
    )


# --------------------------------------------------------------------------- #
# The batched oracle environment
# --------------------------------------------------------------------------- #
class OracleEnv:
    """fp64 batched restatement of MADemandResponseEnv (one instance == E independent envs).

    Two ways to start an episode:
      * ``reset(seed, episode)``   - sample with the Philox streams the HIP reset kernel uses;
      * ``load_episode(params)``   - load arrays captured from the reference (golden replays).
    ``od_table`` (shape [T+1, E]) replaces the modelled outdoor temperature when given, which is
    how a replay feeds the reference's own Gaussian draws.  ``perlin_gradient_fn(env_idx, l)``
    can be overridden the same way.
    """

    def __init__(self, config: dict, nb_envs: int = 1, env_offset: int = 0, house_offset: int = 0,
                 nb_houses_local: Optional[int] = None):
        self.spec = parse_config(config, nb_envs)
        s = self.spec
        self.E = nb_envs
        self.N_total = s.nb_houses
        self.N = nb_houses_local if nb_houses_local is not None else s.nb_houses
        self.env_offset = env_offset
        self.house_offset = house_offset
        self.k = 0
        self.episode = 0
        self.seed = 0
        self.od_table = None
        self.interp_grid = None       # InterpGrid, required when base_power_mode == "interpolation"
        ip = config["default_env_prop"]["power_grid_prop"]["base_power_parameters"].get("interpolation", {})
        self.interp_period = ip.get("interp_update_period", 300)
        self.interp_nb_agents = ip.get("interp_nb_agents", 100)
        self.default_ratios = (self.spec.Ua, self.spec.Cm, self.spec.Ca, self.spec.Hm)
        self.group_reduce = None  # hook for sharded runs: fn(sumP, sumPen, maxPen) -> global values
        norm_t = float(deadband_l2(np.float64(s.target), 0.0, np.float64(s.target + 1)))
        norm_s = float(deadband_l2(np.float64(s.norm_reg_sig), 0.0, np.float64(0.75 * s.norm_reg_sig)))
        self.norm_temp_pen = norm_t
        self.norm_sig_pen = norm_s

    # ---- episode start ---------------------------------------------------- #
    def _env_ids(self):
        return np.arange(self.E, dtype=np.int64) + self.env_offset

    def _house_ids(self):
        return np.arange(self.N, dtype=np.int64) + self.house_offset

    def reset(self, seed: int = 0, episode: int = 0):
        s = self.spec
        self.seed, self.episode = int(seed), int(episode)
        k0, k1 = seed_key(seed)
        e = self._env_ids()[:, None]
        h = self._house_ids()[None, :]
        a = philox4x32_10(e, h, episode, TAG_HOUSE_TEMPS, k0, k1)
        b = philox4x32_10(e, h, episode, TAG_HOUSE_HVAC, k0, k1)
        c = philox4x32_10(e, h, episode, TAG_HOUSE_THERMO, k0, k1)
        Ta = s.init_air + np.abs(s.std_start * gauss01(a[0], a[1]))
        Tm = s.init_mass + np.abs(s.std_start * gauss01(a[2], a[3]))
        target = s.target + np.abs(s.std_target * gauss01(b[0], b[1]))
        caps = np.asarray(s.cap_list, dtype=np.float64)
        cap = caps[mulhi_pick(b[2], len(caps))]
        lock = s.lockout + (-s.lockout_noise + mulhi_pick(b[3], 2 * s.lockout_noise + 1))
        Ua = s.Ua * triangular_mode1(u01(c[0]), s.f_low, s.f_high)
        Cm = s.Cm * triangular_mode1(u01(c[1]), s.f_low, s.f_high)
        Ca = s.Ca * triangular_mode1(u01(c[2]), s.f_low, s.f_high)
        Hm = s.Hm * triangular_mode1(u01(c[3]), s.f_low, s.f_high)
        d = philox4x32_10(self._env_ids(), ENV_LEVEL, episode, TAG_ENV_START, k0, k1)
        if s.start_mode == "random":
            t0 = s.start_epoch + mulhi_pick(d[0], 364) * 86400 + mulhi_pick(d[1], 86400)
        else:
            t0 = np.full(self.E, s.start_epoch, dtype=np.int64)
        phase = u01(d[2]) * 24.0 if s.random_phase else np.zeros(self.E)
        ratio = s.artificial_ratio * np.power(float(s.ratio_range), 2.0 * u01(d[3]) - 1.0)
        self._install(dict(Ta=Ta, Tm=Tm, target=target, deadband=np.full((self.E, self.N), float(s.deadband)),
                           Ua=Ua, Cm=Cm, Ca=Ca, Hm=Hm, capacity=cap,
                           COP=np.full((self.E, self.N), float(s.COP)),
                           latent=np.full((self.E, self.N), float(s.latent)),
                           lockout=lock, t0=t0, phase=phase, ratio=ratio))
        return self

    def load_episode(self, params: Dict[str, np.ndarray], od_table: Optional[np.ndarray] = None):
        """params: arrays named as in `_install`; [E,N] for house items, [E] for env items."""
        p = {k: np.array(v) for k, v in params.items()}
        self.od_table = None if od_table is None else np.asarray(od_table, dtype=np.float64)
        self._install(p)
        return self

    def _install(self, p):
        E, N = self.E, self.N
        f = lambda name: np.broadcast_to(np.asarray(p[name], dtype=np.float64), (E, N)).copy()
        self.Ta, self.Tm = f("Ta"), f("Tm")
        self.target, self.deadband = f("target"), f("deadband")
        self.Ua, self.Cm, self.Ca, self.Hm = f("Ua"), f("Cm"), f("Ca"), f("Hm")
        self.capacity, self.COP, self.latent = f("capacity"), f("COP"), f("latent")
        self.lockout = np.broadcast_to(np.asarray(p["lockout"], dtype=np.int64), (E, N)).copy()
        if np.any(self.lockout < 0):
            raise ValueError("Lockout duration must be positive")
        self.t0 = np.broadcast_to(np.asarray(p["t0"], dtype=np.int64), (E,)).copy()
        self.phase = np.broadcast_to(np.asarray(p.get("phase", 0.0), dtype=np.float64), (E,)).copy()
        self.ratio = np.broadcast_to(np.asarray(p.get("ratio", self.spec.artificial_ratio), dtype=np.float64), (E,)).copy()
        self.on = np.zeros((E, N), dtype=bool)            # env 432
        self.lock = np.zeros((E, N), dtype=bool)          # env 433
        self.sso = self.lockout.copy()                    # env 434
        self.Pmax = self.capacity / self.COP              # env 436
        self.Qhvac = -self.capacity / (1.0 + self.latent)  # env 505
        local_max = self.Pmax.sum(axis=1)
        if self.group_reduce is not None:
            local_max = self.group_reduce(local_max, None, None)[0]
        self.max_power = local_max                        # env 798-802, 125
        self.k = 0
        self.P = np.zeros(E)                              # env 796-801 (all off)
        self.solar = np.zeros(E)
        self.OD = self._od_temp(0)                        # env 793
        self.base_power = np.zeros(E)                     # env 1190
        self.tsli = self.interp_period + 1                # env 1155
        self.cumulated_abs_noise = np.zeros(E)            # env 1117
        self.grid_steps = 0                               # env 1118
        self.S = self._signal(0)                          # env 133

    # ---- per-env time functions ------------------------------------------- #
    def _time(self, j: int):
        return civil_from_epoch(self.t0 + j * self.spec.dt)

    def _od_temp(self, j: int):
        if self.od_table is not None:
            return self.od_table[j].astype(np.float64).copy()
        s = self.spec
        c = self._time(j)
        amp = (s.day_temp - s.night_temp) / 2.0
        bias = (s.day_temp + s.night_temp) / 2.0
        tday = c["hour"] + c["minute"] / 60.0
        temp = amp * np.sin(2.0 * np.pi * (tday + (-6.0 + self.phase)) / 24.0) + bias
        k0, k1 = seed_key(self.seed)
        g = philox4x32_10(self._env_ids(), j, self.episode, TAG_OD_NOISE, k0, k1)
        return temp + s.temp_std * gauss01(g[0], g[1])

    def perlin_gradient(self, env_ids, lattice):
        k0, k1 = seed_key(self.seed)
        x = philox4x32_10(env_ids, np.asarray(lattice, dtype=np.int64) & 0xFFFFFFFF, self.episode, TAG_PERLIN, k0, k1)
        return 2.0 * u01(x[0]) - 1.0

    def _signal(self, j: int):
        s = self.spec
        c = self._time(j)
        n_hvac = self.N_total
        if s.base_power_mode == "constant":
            base = np.full(self.E, s.avg_power_per_hvac * n_hvac, dtype=np.float64)   # env 1249
        else:                                                                         # env 1250-1255
            self.tsli += s.dt
            if self.tsli >= self.interp_period:
                self.base_power = self._interpolate_power(j, c)
                self.tsli = 0
            base = self.base_power
        sod = c["sod"].astype(np.float64)
        mode = s.signal_mode
        if mode == "flat":
            sig = base
        elif mode == "sinusoidals":
            periods, ratios = s.signal_params["periods"], s.signal_params["amplitude_ratios"]
            if len(periods) != len(ratios):
                raise ValueError("periods and amplitude_ratios lists should have the same length")
            sig = base.copy()
            for per, r in zip(periods, ratios):
                sig = sig + base * r * np.sin(2.0 * np.pi * sod / per)
        elif mode == "regular_steps":
            amp = s.signal_params["amplitude_per_hvac"] * n_hvac
            ratio = base / amp
            per = s.signal_params["period"]
            sig = amp * np.heaviside(np.mod(sod, per) - (1.0 - ratio) * per, 1.0)
        else:  # perlin family
            amp = s.signal_params["amplitude_ratios"]
            x = sod / s.signal_params["period"]          # mktime % 86400 == seconds of day in UTC
            ids = self._env_ids()
            noise = perlin_octaves(x, lambda l: self.perlin_gradient(ids, l),
                                   s.signal_params["nb_octaves"], s.signal_params["octaves_step"])
            sig = np.maximum(0.0, base + base * amp * noise)
            self.cumulated_abs_noise = self.cumulated_abs_noise + np.abs(base * amp * noise)   # env 1301
            self.grid_steps += 1                                                               # env 1302
        sig = sig * self.ratio                               # env 1312
        return np.minimum(sig, self.max_power)               # env 1314

    def _interpolate_power(self, j, cal):
        """PowerGrid.interpolatePower (env 1195-1234): sum of the grid's bang-bang average power over (a sample of)
        the houses, scaled back to the whole cluster."""
        if self.interp_grid is None:
            raise ValueError("base_power_mode='interpolation' needs an interpolation grid (OracleEnv.interp_grid)")
        s = self.spec
        E, N = self.E, self.N
        if N <= self.interp_nb_agents:
            ids = np.broadcast_to(np.arange(N)[None, :], (E, N))
            factor = 1.0
        else:   # random.choices(all_ids, k=interp_nb_agents) -> Philox stream TAG_INTERP
            k0, k1 = seed_key(self.seed)
            q = np.arange(self.interp_nb_agents)[None, :]
            x = philox4x32_10(self._env_ids()[:, None], q, j, TAG_INTERP | (self.episode << 8), k0, k1)
            ids = mulhi_pick(x[0], N)
            factor = float(N) / self.interp_nb_agents
        rows = np.arange(E)[:, None]
        g = lambda a: a[rows, ids]
        if s.solar_on:   # env 1198-1202: tm_yday and seconds since midnight
            date = np.broadcast_to(cal["yday"].astype(np.float64)[:, None], ids.shape)
            hour = np.broadcast_to(cal["sod"].astype(np.float64)[:, None], ids.shape)
        else:
            date = np.zeros(ids.shape)
            hour = np.zeros(ids.shape)
        target = g(self.target)
        point = np.stack([g(self.Ua) / s.Ua, g(self.Cm) / s.Cm, g(self.Ca) / s.Ca, g(self.Hm) / s.Hm,
                          g(self.Ta) - target, g(self.Tm) - target, self.OD[:, None] - target, g(self.capacity),
                          hour, date], axis=-1)
        return self.interp_grid.lookup(point).sum(axis=1) * factor

    # ---- the step ----------------------------------------------------------- #
    def step(self, actions):
        s = self.spec
        cmd = np.asarray(actions).astype(bool).reshape(self.E, self.N)
        self.k += 1
        cal = self._time(self.k)
        self.on, self.lock, self.sso = hvac_transition(self.on, self.sso, self.lockout, cmd, s.dt)
        if s.solar_on:
            self.solar = s.window_area * s.shading * solar_cooling_load(cal["hour"], cal["minute"], cal["month"], cal["day"])
        else:
            self.solar = np.zeros(self.E)
        Qa = np.where(self.on, self.Qhvac, 0.0) + self.solar[:, None]
        self.Ta, self.Tm = etp_closed_form(self.Ta, self.Tm, self.OD[:, None], Qa,
                                           self.Ua, self.Cm, self.Ca, self.Hm, float(s.dt))
        self.OD = self._od_temp(self.k)
        pen = deadband_l2(self.target, self.deadband, self.Ta)
        sumP = np.where(self.on, self.Pmax, 0.0).sum(axis=1)
        sumPen = pen.sum(axis=1)
        maxPen = pen.max(axis=1)
        if self.group_reduce is not None:
            sumP, sumPen, maxPen = self.group_reduce(sumP, sumPen, maxPen)
        self.P = sumP
        n = float(self.N_total)
        sig_pen = ((self.P - self.S) / n) ** 2                 # uses the OLD signal (env 196, 246)
        if s.penalty_mode == "individual_L2":
            tpen = pen
        elif s.penalty_mode == "common_L2":
            tpen = np.broadcast_to((sumPen / n)[:, None], pen.shape)
        elif s.penalty_mode == "common_max":
            tpen = np.broadcast_to(maxPen[:, None], pen.shape)
        else:
            a_i, a_c, a_m = s.mix
            tpen = (a_i * pen + a_c * (sumPen / n)[:, None] + a_m * maxPen[:, None]) / (a_i + a_c + a_m)
        self.reward = -(s.alpha_temp * tpen / self.norm_temp_pen
                        + s.alpha_sig * sig_pen[:, None] / self.norm_sig_pen)
        self.S = self._signal(self.k)
        return self.reward

    # ---- observation columns the kernels emit --------------------------------- #
    def dynamic_obs(self):
        """The 7 per-step-varying entries of the default normStateDict vector (utils.py:800-841)."""
        den = self.spec.norm_reg_sig * self.spec.cfg_nb_agents
        E, N = self.E, self.N
        return np.stack([
            (self.Ta - 20.0) / 5.0,
            (self.Tm - 20.0) / 5.0,
            self.on.astype(np.float64),
            self.lock.astype(np.float64),
            self.sso / self.lockout,
            np.broadcast_to((self.S / den)[:, None], (E, N)),
            np.broadcast_to((self.P / den)[:, None], (E, N)),
        ])

    def circular_links(self, nb_comm):
        """agents_comm_mode 'neighbours' (env 816-828): floor(c/2) before, ceil(c/2) after, circular."""
        n = self.N
        before, after = nb_comm // 2, nb_comm - nb_comm // 2
        i = np.arange(n)[:, None]
        return np.concatenate([(i - before + np.arange(before)[None, :]) % n,
                               (i + 1 + np.arange(after)[None, :]) % n], axis=1).astype(np.int64)

    # ---- the random part of the message gather (env 976-1002), as the HIP kernels draw it ------------------------------------- #
    def _comm_key(self):
        k0, k1 = seed_key(self.seed)
        return k0, k1 ^ ((self.episode * COMM_KEY_MIX) & 0xFFFFFFFF)

    def link_keep(self, c: int, defect_prob: float) -> np.ndarray:
        """`np.random.rand() > comm_defect_prob` per link, in link order (env 992), restated on Philox stream 7: one block of four
        words serves message slots 4b..4b+3 of a (env, receiving house, time index).  The comparison is done in fp32, as the
        kernels do it.  -> bool [E, N, c], True = the message is delivered."""
        E, N = self.E, self.N
        keep = np.ones((E, N, c), dtype=bool)
        if not (np.float32(defect_prob) > np.float32(0.0)):
            return keep
        k0, k1 = self._comm_key()
        e = self._env_ids()[:, None]
        h = self._house_ids()[None, :]
        for b in range((c + 3) // 4):
            words = philox4x32_10(e, h, self.k, TAG_COMM | (b << 8), k0, k1)
            for w in range(4):
                m = 4 * b + w
                if m < c:
                    keep[:, :, m] = u01(words[w]).astype(np.float32) > np.float32(defect_prob)
        return keep

    def sampled_senders(self, c: int) -> np.ndarray:
        """agents_comm_mode 'random_sample' (env 976-983): `random.sample(others, k=nb_comm)` per house and step - an ORDERED draw
        without replacement - restated on Philox stream 9 by rejection: slot m takes uniform picks `(x * (N - 1)) >> 32` among the
        N - 1 other houses (draw d of a house = word d & 3 of block d >> 2) until one is not among the m senders already chosen;
        pick p maps to house id p (p < own id) or p + 1.  -> int64 [E, N, c] global house ids."""
        E, N = self.E, self.N
        D = self.N_total - 1
        k0, k1 = self._comm_key()
        e = np.broadcast_to(self._env_ids()[:, None], (E, N))
        hg = np.broadcast_to(self._house_ids()[None, :], (E, N))
        picks = np.full((E, N, c), -1, dtype=np.int64)
        draws = np.zeros((E, N), dtype=np.int64)
        for m in range(c):
            pending = np.ones((E, N), dtype=bool)
            while pending.any():
                words = philox4x32_10(e, hg, self.k, TAG_LINKS | ((draws >> 2) << 8), k0, k1)
                x = np.choose(draws & 3, words)
                pick = mulhi_pick(x, D)
                taken = (picks[:, :, :m] == pick[:, :, None]).any(axis=2)
                accept = pending & ~taken
                picks[:, :, m] = np.where(accept, pick, picks[:, :, m])
                draws = draws + pending           # every pending house consumed one draw
                pending = pending & taken
        return np.where(picks < hg[:, :, None], picks, picks + 1)

    def norm_state(self, config, links=None, keep=None):
        """utils.normStateDict (utils.py:740-880) for every house -> float64 [E, N, F], messages from
        SingleHouse.message (env 624-662) gathered through `links` ([N, c] sender ids shared by the envs, or [E, N, c];
        None = the mode's own: circular 'neighbours', or the step's `sampled_senders` for 'random_sample').
        `keep` (bool [E, N, c], default `link_keep` at the config's comm_defect_prob): links that deliver; the others
        carry the all-zero message of SingleHouse.message(empty=True) (env 996-1001)."""
        env = config["default_env_prop"]
        sp, mp = env["state_properties"], env["message_properties"]
        house, hvac = config["default_house_prop"], config["default_hvac_prop"]
        s = self.spec
        E, N = self.E, self.N
        c = int(min(s.nb_agents_comm, s.cfg_nb_agents - 1))
        if s.comm_mode == "no_message":
            links = np.zeros((N, 0), dtype=np.int64)
        elif links is None:
            links = self.sampled_senders(c) if s.comm_mode == "random_sample" else self.circular_links(c)
        links = np.asarray(links, dtype=np.int64)
        if links.ndim == 2:
            links = np.broadcast_to(links[None, :, :], (E,) + links.shape)
        if keep is None:
            keep = self.link_keep(links.shape[2], float(env["cluster_prop"].get("comm_defect_prob", 0.0)))
        rows = np.arange(E)[:, None]
        den = s.norm_reg_sig * s.cfg_nb_agents
        cal = self._time(self.k)
        ones = np.ones((E, N))
        cols = [(self.Ta - 20) / 5, (self.Tm - 20) / 5, (self.target - 20) / 5]
        if sp["thermal"]:
            cols.append(((self.OD - 20) / 5)[:, None] * ones)
        cols.append(self.deadband)
        if sp["day"]:
            cols += [np.sin(cal["yday"] * 2 * np.pi / 365)[:, None] * ones, np.cos(cal["yday"] * 2 * np.pi / 365)[:, None] * ones]
        if sp["hour"]:
            cols += [np.sin(cal["hour"] * 2 * np.pi / 24)[:, None] * ones, np.cos(cal["hour"] * 2 * np.pi / 24)[:, None] * ones]
        if sp["solar_gain"]:
            cols.append((self.solar / 1000)[:, None] * ones)
        cols.append(self.capacity / hvac["cooling_capacity"])
        if sp["thermal"]:
            cols += [self.Ua / house["Ua"], self.Cm / house["Cm"], self.Ca / house["Ca"], self.Hm / house["Hm"]]
        if sp["hvac"]:
            cols += [self.COP / hvac["COP"], self.latent / hvac["latent_cooling_fraction"]]
        cols += [self.on.astype(float), self.lock.astype(float), self.sso / self.lockout, self.lockout / self.lockout,
                 (self.S / den)[:, None] * ones, (self.P / den)[:, None] * ones]
        for m in range(links.shape[2]):
            j = links[:, :, m]
            z = keep[:, :, m].astype(np.float64)
            g = lambda a: a[rows, j]              # the sender's value for every (env, receiving house)
            msg = [g(self.Ta - self.target) / 5, g(self.sso) / self.lockout,
                   g(np.where(self.on, self.Pmax, 0.0)) / s.norm_reg_sig, g(self.Pmax) / s.norm_reg_sig]
            if mp["thermal"]:
                msg += [g(self.Ua) / house["Ua"], g(self.Cm) / house["Cm"], g(self.Ca) / house["Ca"], g(self.Hm) / house["Hm"]]
            if mp["hvac"]:
                msg += [g(self.COP) / hvac["COP"], g(self.latent) / hvac["latent_cooling_fraction"],
                        g(self.capacity) / hvac["cooling_capacity"]]
            cols += [z * v for v in msg]
        return np.stack([np.broadcast_to(cc, (E, N)) for cc in cols], axis=-1)

    def bangbang_actions(self):
        """agents/bangbang_controllers.py:41-61: on iff house_temp > target."""
        return self.Ta > self.target

    def deadband_actions(self):
        """agents/bangbang_controllers.py:13-38 (DeadbandBangBangController) == 64-88 (BasicController): off below
        target - deadband / 2, on above target + deadband / 2, else hvac_turned_on."""
        half = self.deadband / 2
        return np.where(self.Ta < self.target - half, False, np.where(self.Ta > self.target + half, True, self.on.astype(bool)))

    def greedy_myopic_actions(self):
        """agents/greedy_myopic_controller.py:29-50: houses sorted by -(house_temp - target) ascending (equal ones in house order:
        pandas leaves that open), then one greedy pass against the budget reg_signal; a house is taken iff
        p + total < target  or  (|p + total - target| < |total - target| and not hvac_lockout),  p = cooling_capacity / COP."""
        E, N = self.Ta.shape
        out = np.zeros((E, N), dtype=bool)
        power = self.capacity / self.COP
        for e in range(E):
            order = np.argsort(-(self.Ta[e] - self.target[e]), kind="stable")
            target, total = float(self.S[e]), 0.0
            for h in order:
                p = float(power[e, h])
                if p + total < target or (abs(p + total - target) < abs(total - target) and not self.lock[e, h]):
                    total += p
                    out[e, h] = True
        return out

    def always_on_actions(self):
        """agents/bangbang_controllers.py:1-10 (AlwaysOnController)."""
        return np.ones(self.Ta.shape, dtype=bool)
