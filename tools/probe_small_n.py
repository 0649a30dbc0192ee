#!/usr/bin/env python3
"""Where a step's time goes for small envs: the per-house kernel vs the per-env time tables (which are O(E) per step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import mdr_amd  # noqa: E402
from bench_kernels import cfg_for  # noqa: E402

for (E, N) in ((4194304, 1), (419430, 10), (83886, 50)):
    for light in (False, True):
        kw = {"default_env_prop.power_grid_prop.signal_mode": "flat",
              "default_env_prop.cluster_prop.temp_mode": "constant"} if light else {}
        env = mdr_amd.BatchedDemandResponseEnv(cfg_for(N, **kw), nb_envs=E, seed=1)
        env.reset()
        env.rollout(2)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        env.rollout(40)          # inside the first table chunk (K = 64): no refill
        e1.record()
        env.rollout(64)          # exactly one refill
        e2.record()
        torch.cuda.synchronize()
        in_chunk = e0.elapsed_time(e1) / 40 * 1e3
        with_fill = e1.elapsed_time(e2) / 64 * 1e3
        print("%8d x %-4d %-13s kernel %.1f us/step, with table refills %.1f us/step" % (
            E, N, "flat/constant" if light else "perlin/noisy", in_chunk, with_fill), flush=True)
        del env
        torch.cuda.empty_cache()
