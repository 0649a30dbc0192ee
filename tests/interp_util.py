"""Interpolated base power (PowerGrid.interpolatePower, env 1195-1234): how the device-vs-oracle comparison is bounded.

The device evaluates the 5-D multilinear lookup in fp64, but its QUERY is built from fp32 house temperatures (and fp32 copies of the
per-house parameters).  Two comparisons separate the two effects:

  * `DeviceFedOracle` - the oracle with ONE change: interpolatePower reads the device's own state as its query.  Same inputs, both
    lookups in fp64: base power and signal must agree to rounding (1e-9) - an interval-selection or nearest-index slip would show here.
  * the pure fp64 oracle - the base power may differ by what the temperature differences can move it: the lookup is piecewise
    multilinear, so along each interpolated axis its slope is bounded by the steepest grid edge (`edge_slopes`), and
        |d base_power| <= nb_agents * (G_air |dTa| + G_mass |dTm| + G_od |dOD| + (G_air + G_mass + G_od) |dtarget|).
"""
import numpy as np

from oracle import mdr_oracle as mo


def edge_slopes(grid: "mo.InterpGrid"):
    """max |V[i+1] - V[i]| / (ax[i+1] - ax[i]) along the air-, mass- and outdoor-temperature axes (W per deg C)."""
    out = []
    for d in (4, 5, 6):
        ax = grid.axes[d]
        dv = np.abs(np.diff(grid.values, axis=d))
        shape = [1] * grid.values.ndim
        shape[d] = len(ax) - 1
        out.append(float((dv / np.diff(ax).reshape(shape)).max()))
    return out


class DeviceFedOracle(mo.OracleEnv):
    """OracleEnv whose interpolatePower takes its query (house temperatures, targets, thermal parameters, capacities, outdoor
    temperature) from `self.device_env` - the device's fp32 state widened to fp64 - instead of from its own fp64 state."""

    device_env = None

    def _interpolate_power(self, j, cal):
        env = self.device_env
        names = ("Ta", "Tm", "target", "Ua", "Cm", "Ca", "Hm", "capacity", "OD")
        saved = {k: getattr(self, k) for k in names}
        try:
            self.Ta = env.house_temp().cpu().numpy()
            self.Tm = env.house_mass_temp().cpu().numpy()
            self.target = env.target_temp().cpu().numpy()
            for k in ("Ua", "Cm", "Ca", "Hm", "capacity"):
                setattr(self, k, env.t[k].double().cpu().numpy())
            self.OD = env.od_temp().cpu().numpy()
            return super()._interpolate_power(j, cal)
        finally:
            for k, v in saved.items():
                setattr(self, k, v)


def nearest_axis_flips(grid, env, ora) -> np.ndarray:
    """bool [E]: envs with a house whose NEAREST grid index (the four thermal ratios and the HVAC power are looked up by
    `np.argmin(|axis - value|)`, monteCarlo/interpolation.py:113-142) differs between the device's fp32 copy of the parameter and
    the fp64 value - a parameter within fp32 rounding of the midpoint of two axis values (seen once in 31,000 fuzz cases: Ua ratio
    1.0029999995 against the axis [1.0, 1.006]).  The lookup is discontinuous there, so no slope bound relates the two base
    powers; the device-fed comparison still holds the device to its own inputs."""
    s = ora.spec
    flips = np.zeros(ora.E, dtype=bool)
    for d, (name, default) in enumerate((("Ua", s.Ua), ("Cm", s.Cm), ("Ca", s.Ca), ("Hm", s.Hm), ("capacity", 1.0))):
        ax = grid.axes[d if d < 4 else 7]
        v32 = np.clip(env.t[name].double().cpu().numpy() / default, ax.min(), ax.max())
        v64 = np.clip(getattr(ora, name) / default, ax.min(), ax.max())
        i32 = np.argmin(np.abs(ax[None, None, :] - v32[..., None]), axis=-1)
        i64 = np.argmin(np.abs(ax[None, None, :] - v64[..., None]), axis=-1)
        flips |= (i32 != i64).any(axis=1)
    return flips


def base_power_bound(grid, env, ora, nb_agents):
    """Upper bound of |device base power - fp64 oracle base power| from the CURRENT state differences (see module docstring)."""
    g_air, g_mass, g_od = edge_slopes(grid)
    d_ta = float(np.max(np.abs(env.house_temp().cpu().numpy() - ora.Ta)))
    d_tm = float(np.max(np.abs(env.house_mass_temp().cpu().numpy() - ora.Tm)))
    d_tg = float(np.max(np.abs(env.target_temp().cpu().numpy() - ora.target)))
    d_od = float(np.max(np.abs(env.od_temp().cpu().numpy() - ora.OD)))
    return nb_agents * (g_air * d_ta + g_mass * d_tm + g_od * d_od + (g_air + g_mass + g_od) * d_tg)
