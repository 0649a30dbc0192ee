#!/usr/bin/env python3
"""Step rate at the reference's own cluster sizes (cli.py:53 trains with 20 houses, cli.py:629 deploys with 50) against the
headline shape, every batch ~4.19 M houses of BASELINE's C3 configuration (heterogeneous houses, noisy weather, solar, Perlin
signal): microseconds per step inside a time-table window and INCLUDING the table refills (640 steps = 10 windows), and the
rate relative to N = 1024.  One JSON line per shape.

    python tools/bench_small_n.py [N ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import mdr_amd  # noqa: E402

TOTAL = 4096 * 1024


def measure(N):
    E = TOTAL // N
    cfg = bench.c3_config(mdr_amd)
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, seed=2024, table_steps=64)
    env.reset(episode=0)
    env.rollout(64 + 8)          # warm: past the first refill, 8 steps into a window
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    env.rollout(48)              # inside one window: no refill
    ev[1].record()
    env.rollout(640)             # ten refills
    ev[2].record()
    torch.cuda.synchronize()
    row = {"N": N, "E": E, "houses": E * N, "us_in_window": ev[0].elapsed_time(ev[1]) / 48 * 1e3,
           "us_with_refills": ev[1].elapsed_time(ev[2]) / 640 * 1e3}
    row["house_steps_per_s"] = E * N / row["us_with_refills"] * 1e6
    del env
    torch.cuda.empty_cache()
    return row


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1024, 10, 20, 50, 64, 100]
    base = None
    for N in sizes:
        row = measure(N)
        if N == 1024:
            base = row["house_steps_per_s"]
        if base:
            row["rate_vs_N1024"] = row["house_steps_per_s"] / base
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)


if __name__ == "__main__":
    main()
