#!/usr/bin/env python3
"""Infinity-Cache control for the headline kernel: k_step_fused<4,1,256> at E x 1024 houses for growing E.

The C3 batch re-reads 53 B/house (state 13 + parameters 40) = 222 MB every step - less than the 256 MiB Infinity Cache
once the 134 MB of output-only streams are written non-temporally - so part of C3's "HBM" rate may be cache-assisted
(FETCH_SIZE / WRITE_SIZE count cache hits too, MI355X_MICROARCH.md).  This sweep grows the re-read set to 8x the cache
(E = 32768: 1.8 GB) with the product build (non-temporal output stores) and with -DMDR_NT_STORES=0, HIP events on the
launch stream, one JSON line per point:

    python tools/bench_size_sweep.py [--envs 1024,2048,...] [--steps 200] > profiles/r02_size_sweep.jsonl

Every (variant, E) point runs in its own child process started before this one touches the GPU.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B_ALG, N = 99, 1024


def child(E, steps, variant):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import mdr_amd
    cfg = bench.c3_config(mdr_amd)
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, seed=2024, table_steps=64)
    env.reset(episode=0)
    env.rollout(30)
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(steps)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / steps * 1e3
        best = us if best is None else min(best, us)
    houses = E * N
    print(json.dumps({"kernel": "k_step_fused<4,1,256>", "variant": variant, "envs": E, "houses": houses, "us_per_step": round(best, 2),
                      "GBps": round(houses * B_ALG / best * 1e-3, 1), "house_steps_per_s": houses / best * 1e6,
                      "reread_set_MB": round(houses * 53 / 1e6, 1), "resident_MB": round(houses * 114 / 1e6, 1),
                      "output_streams_MB": round(houses * 32 / 1e6, 1), "steps": steps, "reps": 3}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", default="1024,2048,4096,6144,8192,16384,32768")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    if args.child:
        E, variant = args.child.split(":")
        return child(int(E), args.steps, variant)
    from mdr_amd.build import OUTPUT, build_variant
    libs = {"nt_stores=1 (product)": OUTPUT, "nt_stores=0": build_variant("nt0", ["MDR_NT_STORES=0"])}
    for variant, lib in libs.items():
        for E in [int(x) for x in args.envs.split(",")]:
            env = dict(os.environ, MDR_HIP_LIB=lib)
            rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "%d:%s" % (E, variant), "--steps", str(args.steps)],
                                env=env).returncode
            if rc != 0:
                raise SystemExit("point %s E=%d failed (rc %d)" % (variant, E, rc))


if __name__ == "__main__":
    main()
