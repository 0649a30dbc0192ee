#!/usr/bin/env python3
"""Per-step time of the dict adapter (the reference's own calling convention, E = 1) - a compatibility surface."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdr_amd  # noqa: E402

for N in (10, 50, 200):
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    env = mdr_amd.MADemandResponseEnv(cfg)
    obs = env.reset()
    act = {i: True for i in obs}
    for _ in range(20):
        obs, r, d, info = env.step(act)
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        obs, r, d, info = env.step(act)
    dt = (time.perf_counter() - t0) / n
    print("N=%d  %.1f us per step  %.3g house-steps/s" % (N, dt * 1e6, N / dt), flush=True)
