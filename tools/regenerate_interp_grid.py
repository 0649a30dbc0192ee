#!/usr/bin/env python3
"""Regenerate the base-power interpolation grid the reference does not ship, on the GPU.

    python tools/regenerate_interp_grid.py [--out-dir ./monteCarlo] [--chunk 1048576] [--check]

What monteCarlo/monteCarlo.py:133-278 of the reference does in >= 2 CPU-hours - for each of the 4,199,040 grid
points build a 1-house env (no noise, fixed start date/hour, constant outdoor temperature = target + OD_temp,
lockout 1 s, thermal parameters scaled by the four ratios, start temperatures = target + air/mass offsets), run 75
bang-bang steps and record the "stabilised" running-average power (mean of the last 10 running averages) - is done
here as batches of independent 1-house envs through the fused multi-step rollout kernel (mdr_env_rollout_fused) with
the per-step cluster power traced; merge.py's flattening is the C order of the axes.  Writes, in the reference's own
file formats and default locations (config.py:343-345):

    <out-dir>/mergedGridSearchResultFinal.npy   flat float64, C order over the axes
    <out-dir>/interp_parameters_dict.json       axis values
    <out-dir>/interp_dict_keys.csv              axis order

so that the reference's DEFAULT config (base_power_mode="interpolation") runs.  `--check` compares against the
48 grid points captured from the reference itself (tests/golden/montecarlo_points.npz).
"""
from __future__ import annotations

import argparse
import csv
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", default=os.path.join(".", "monteCarlo"))
    ap.add_argument("--axes", default=None, help="JSON file with the axis values (default: the reference's grid)")
    ap.add_argument("--chunk", type=int, default=1 << 19)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    from mdr_amd.config import DEFAULT_INTERP_AXES, INTERP_KEYS
    from mdr_amd.montecarlo import NB_TIME_STEPS_BY_SIM, generate_grid
    axes = DEFAULT_INTERP_AXES
    if args.axes:
        with open(args.axes) as f:
            axes = json.load(f)
    keys = list(INTERP_KEYS)
    dims = [len(axes[k]) for k in keys]
    total = int(np.prod(dims))
    t0 = time.perf_counter()
    out = generate_grid(axes, chunk=args.chunk,
                        progress=lambda done, tot: print("grid points %d of %d  (%.1f s)" % (done, tot, time.perf_counter() - t0), flush=True))
    el = time.perf_counter() - t0
    print("%d grid points x %d steps in %.1f s = %.3g house-steps/s" % (total, NB_TIME_STEPS_BY_SIM, el, total * NB_TIME_STEPS_BY_SIM / el))
    os.makedirs(args.out_dir, exist_ok=True)
    np.save(os.path.join(args.out_dir, "mergedGridSearchResultFinal.npy"), out)
    with open(os.path.join(args.out_dir, "interp_parameters_dict.json"), "w") as f:
        json.dump({k: list(axes[k]) for k in keys}, f)
    with open(os.path.join(args.out_dir, "interp_dict_keys.csv"), "w") as f:
        csv.writer(f).writerow(keys)
    if args.check:
        z = np.load(os.path.join(ROOT, "tests", "golden", "montecarlo_points.npz"))
        flat = np.ravel_multi_index(z["index"].T, dims)
        got, ref = out[flat], z["hvac_average_power"]
        exact = np.isclose(got, ref, rtol=1e-6, atol=1e-6)
        print("check vs reference: %d/%d grid points identical, max |rel diff| %.3g" % (
            exact.sum(), len(ref), float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))))
        if exact.mean() < 0.9:
            raise SystemExit("regenerated grid disagrees with the reference's grid points")


if __name__ == "__main__":
    main()
