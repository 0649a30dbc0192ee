"""The reference's Monte-Carlo base-power grid (monteCarlo/monteCarlo.py:133-278) regenerated on the GPU.

For each grid point the reference builds a 1-house env (no noise, fixed start date/hour, constant outdoor
temperature = target + OD_temp, lockout 1 s, thermal parameters scaled by the four ratios, start temperatures =
target + air/mass offsets), runs 75 bang-bang steps and records the "stabilised" running-average power (mean of the
last 10 running averages).  Here the grid points are batches of independent 1-house envs pushed through the fused
multi-step rollout kernel (mdr_env_rollout_fused) with the per-step cluster power traced; merge.py's flattening is
the C order of the axes.  The full 4,199,040-point grid takes ~2 s on one MI355X (the reference: >= 2 CPU-hours).
tests/test_gpu_montecarlo.py compares against grid points computed by the reference itself.
"""
from __future__ import annotations

import datetime as dt
from typing import Dict, Optional, Sequence

import numpy as np

from .config import DEFAULT_INTERP_AXES, INTERP_KEYS, default_config, to_epoch_seconds

NB_TIME_STEPS_BY_SIM, NB_TIME_STEPS_AVG = 75, 10      # monteCarlo.py:23-24


def grid_points(axes, keys, start, stop):
    """Rows [start, stop) of itertools.product(*axes) as an index array [n, 10] (C order, like merge.py's flattening)."""
    dims = [len(axes[k]) for k in keys]
    flat = np.arange(start, stop, dtype=np.int64)
    return np.stack(np.unravel_index(flat, dims), axis=1)


def evaluate(idx, axes, keys, device="cuda:0"):
    """Bang-bang stabilised average power (W) of the grid points idx [n, 10]."""
    import torch
    from .batched_env import BatchedDemandResponseEnv

    n = idx.shape[0]
    val = {k: np.asarray(axes[k], dtype=np.float64)[idx[:, d]] for d, k in enumerate(keys)}
    cfg = default_config()
    house, hvac = cfg["default_house_prop"], cfg["default_hvac_prop"]
    cfg["noise_house_prop"]["noise_mode"] = "no_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "no_noise"
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 1
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    # the recorded quantity is the HVAC power, which the regulation signal does not influence (it only enters the reward):
    # "flat" spares the per-env Perlin evaluation the reference's default signal_mode would cost for 4.2 M envs
    cfg["default_env_prop"]["power_grid_prop"]["signal_mode"] = "flat"
    cfg["default_hvac_prop"]["lockout_duration"] = 1
    tgt = float(house["target_temp"])
    d0 = to_epoch_seconds(dt.datetime(2021, 1, 1))
    hour = val["hour"]
    # monteCarlo.py:137-140: the start time is truncated to whole seconds
    sec = (hour // 3600).astype(np.int64) * 3600 + (hour % 3600 // 60).astype(np.int64) * 60 + (hour % 60).astype(np.int64)
    col = lambda a: np.asarray(a, dtype=np.float64).reshape(n, 1)
    params = dict(
        Ta=col(tgt + val["air_temp"]), Tm=col(tgt + val["mass_temp"]), target=col(np.full(n, tgt)),
        deadband=col(np.full(n, float(house["deadband"]))),
        Ua=col(house["Ua"] * val["Ua_ratio"]), Cm=col(house["Cm"] * val["Cm_ratio"]),
        Ca=col(house["Ca"] * val["Ca_ratio"]), Hm=col(house["Hm"] * val["Hm_ratio"]),
        capacity=col(val["HVAC_power"]), COP=col(np.full(n, float(hvac["COP"]))),
        latent=col(np.full(n, float(hvac["latent_cooling_fraction"]))),
        lockout=np.ones((n, 1), dtype=np.int64),
        t0=d0 + val["date"].astype(np.int64) * 86400 + sec,
    )
    od = np.broadcast_to((tgt + val["OD_temp"])[None, :], (NB_TIME_STEPS_BY_SIM + 1, n))
    env = BatchedDemandResponseEnv(cfg, nb_envs=n, device=device, table_steps=NB_TIME_STEPS_BY_SIM)
    env.load_episode(params, od_table=od)
    res = env.rollout_fused(NB_TIME_STEPS_BY_SIM, power_trace=True, accumulate=False)
    trace = res["power_trace"]                                              # [75, n] cluster_hvac_power
    running = torch.cumsum(trace, dim=0)
    steps = torch.arange(1, NB_TIME_STEPS_BY_SIM + 1, device=trace.device, dtype=torch.float64)[:, None]
    tail = (running / (steps * NB_TIME_STEPS_AVG))[NB_TIME_STEPS_BY_SIM - NB_TIME_STEPS_AVG:]   # monteCarlo.py:197-198
    return tail.sum(dim=0).cpu().numpy()



_cache: Dict[tuple, np.ndarray] = {}


def generate_grid(axes: Optional[dict] = None, device="cuda:0", chunk: int = 1 << 19, progress=None) -> np.ndarray:
    """Flat float64 grid (C order over INTERP_KEYS) for `axes` (default: the reference's 4,199,040-point grid)."""
    axes = DEFAULT_INTERP_AXES if axes is None else axes
    keys = list(INTERP_KEYS)
    key = (str(device),) + tuple(tuple(float(v) for v in axes[k]) for k in keys)
    if key in _cache:
        return _cache[key]
    dims = [len(axes[k]) for k in keys]
    total = int(np.prod(dims))
    out = np.empty(total, dtype=np.float64)
    for start in range(0, total, chunk):
        stop = min(total, start + chunk)
        out[start:stop] = evaluate(grid_points(axes, keys, start, stop), axes, keys, device=device)
        if progress:
            progress(stop, total)
    _cache[key] = out
    return out
