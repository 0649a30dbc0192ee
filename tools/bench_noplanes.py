#!/usr/bin/env python3
"""The C3 step WITHOUT its seven per-step observation planes (mdr_buffers_t.obs = NULL: what rollout collection with observe -> act
runs, train_ppo.py:69-72): 71 algorithmic bytes per house-step (53 read, 18 written) instead of 99.  HIP events; one JSON line.

    python tools/bench_noplanes.py [--steps 1000] [--warmup 50] [--planes]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import mdr_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--warmup", type=int, default=50)
ap.add_argument("--planes", action="store_true", help="keep the planes (the headline's 99 B form), for an A/B in one process")
args = ap.parse_args()
env = mdr_amd.BatchedDemandResponseEnv(bench.c3_config(mdr_amd), nb_envs=bench.E_PER_GPU, device="cuda:0", seed=2024, table_steps=64,
                                       obs_planes=args.planes)
env.reset(episode=0)
env.rollout(args.warmup)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
env.rollout(args.steps)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / args.steps * 1e3
houses = bench.E_PER_GPU * bench.N_HOUSES
b_alg = 99 if args.planes else 71
print(json.dumps({"kernel": "k_step_fused<4,1,256>", "obs_planes": bool(args.planes), "algorithmic_bytes_per_house_step": b_alg,
                  "us_per_step": round(us, 3), "house_steps_per_s": houses / us * 1e6, "GBps": round(houses * b_alg / us * 1e-3, 1),
                  "frac_of_8TBps": round(houses * b_alg / us * 1e-3 / 8000.0, 4), "steps": args.steps}))
