#!/usr/bin/env python3
"""Secondary timings on one MI355X (not the headline bench): per-kernel micro-benchmarks with HIP events.

    python tools/bench_kernels.py [what ...]     what in: step obs c2 c1 c5 reset
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mdr_amd  # noqa: E402


def cfg_for(n, **kw):
    cfg = mdr_amd.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = n
    env["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["noise_house_prop"]["noise_mode"] = "house_big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    for k, v in kw.items():
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def timeit(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def report(name, us, houses, bytes_per_house):
    print(json.dumps({"what": name, "us": round(us, 2), "house_per_s": houses / us * 1e6,
                      "GBps": houses * bytes_per_house / us * 1e-3, "bytes_per_house": bytes_per_house}), flush=True)


def main():
    what = sys.argv[1:] or ["step", "obs", "c2", "c1", "c5", "fused", "reset"]
    if "step" in what:
        env = mdr_amd.BatchedDemandResponseEnv(cfg_for(1024), nb_envs=4096, seed=1)
        env.reset()
        report("C3 step, in-kernel bang-bang", timeit(lambda: env.rollout(50), 10) / 50, 4096 * 1024, 99)
        act = (torch.rand(4096, 1024, device="cuda") < 0.5).to(torch.uint8)
        report("C3 step, external actions", timeit(lambda: env.rollout(50, act), 10) / 50, 4096 * 1024, 99)
        del env
    if "obs" in what:
        env = mdr_amd.BatchedDemandResponseEnv(cfg_for(1024), nb_envs=4096, seed=1)
        env.reset()
        env.rollout(10)
        F = env.obs_vector_length()
        report("C3 obs_vector planes F=%d" % F, timeit(lambda: env.obs_vector("planes"), 200), 4096 * 1024, 4 * F + 25)
        report("C3 obs_vector rows F=%d" % F, timeit(lambda: env.obs_vector("rows"), 200), 4096 * 1024, 4 * F + 25)
        del env
    if "obs_small" in what:
        for (E, N) in ((209715, 20), (83886, 50), (32768, 128)):
            env = mdr_amd.BatchedDemandResponseEnv(cfg_for(N), nb_envs=E, seed=1)
            env.reset()
            env.rollout(3)
            F = env.obs_vector_length()
            report("obs_vector planes %dx%d F=%d" % (E, N, F), timeit(lambda: env.obs_vector("planes"), 10), E * N, 4 * F + 25)
            report("obs_vector rows %dx%d F=%d" % (E, N, F), timeit(lambda: env.obs_vector("rows"), 10), E * N, 4 * F + 25)
            del env
            torch.cuda.empty_cache()
    if "c2" in what:
        env = mdr_amd.BatchedDemandResponseEnv(cfg_for(50), nb_envs=1024, seed=1)
        env.reset()
        report("C2 1024x50 step (launch-bound)", timeit(lambda: env.rollout(200), 10) / 200, 1024 * 50, 99)
        del env
    if "c1" in what:
        env = mdr_amd.BatchedDemandResponseEnv(cfg_for(10), nb_envs=1, seed=1)
        env.reset()
        report("C1 1x10 step (launch-bound)", timeit(lambda: env.rollout(200), 10) / 200, 10, 99)
        del env
    if "c5" in what:
        n = 1_000_000
        env = mdr_amd.BatchedDemandResponseEnv(cfg_for(n), nb_envs=1, seed=1)
        env.reset()
        report("1 env x 1e6 houses, split path on one GPU", timeit(lambda: env.rollout(50), 10) / 50, n, 111)
        n8 = 125_000
        env = mdr_amd.BatchedDemandResponseEnv(cfg_for(n8), nb_envs=1, seed=1)
        env.reset()
        report("1 env x 125k houses (C5 per-GPU share), split path", timeit(lambda: env.rollout(50), 10) / 50, n8, 111)
        del env
    if "fused" in what:
        for (E, N, K) in ((4096, 1024, 64), (4096, 1024, 1024), (1024, 50, 1024), (1, 10, 1024), (32768, 50, 1024)):
            env = mdr_amd.BatchedDemandResponseEnv(cfg_for(N), nb_envs=E, seed=1, table_steps=K)
            env.reset()
            steps = 1024
            us = timeit(lambda: env.rollout_fused(steps, accumulate=True), 3, warm=1) / steps
            report("fused multi-step rollout %dx%d, table_steps=%d (per step)" % (E, N, K), us, E * N, 0)
            del env
    if "shapes" in what:
        for (E, N) in ((4194304, 1), (419430, 10), (83886, 50), (65536, 64), (41943, 100), (16384, 256), (8192, 512), (4194, 1000),
                       (4190, 1001), (2048, 2048), (1024, 4096), (512, 8192), (64, 65536), (4, 1048576)):
            env = mdr_amd.BatchedDemandResponseEnv(cfg_for(N), nb_envs=E, seed=1)
            env.reset()
            report("step %d x %d" % (E, N), timeit(lambda: env.rollout(20), 5, warm=2) / 20, E * N, 99)
            del env
            torch.cuda.empty_cache()
    if "reset" in what:
        env = mdr_amd.BatchedDemandResponseEnv(cfg_for(1024), nb_envs=4096, seed=1)
        report("C3 reset (sample + derive + tables)", timeit(lambda: env.reset(), 5, warm=1), 4096 * 1024, 114)


if __name__ == "__main__":
    main()
