#!/usr/bin/env python3
"""Headline benchmark: house-steps/s of the fused env step at 4096 envs x 1024 houses per MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one MADemandResponseEnv.step for every env of the batch (one fused HIP launch) on BASELINE.json
configs[2] (C3): heterogeneous houses (house_big_noise) and HVAC capacities (big_noise), noisy sinusoidal
heat-wave outdoor temperature (one Gaussian per env-step, Philox), solar gain, Perlin regulation signal,
random start date per env, bang-bang actions evaluated in-kernel on the previous observation (closed loop,
no host round trip).  All state is resident in HBM before the timed region.  With N GPUs every rank owns
its own 4096 envs (independent replicas, env_offset = rank * 4096, no data-path collective): weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# algorithmic bytes per house-step of this layout
#   state 13 R + 13 W (Ta, Tm f32; sso i32; flags u8) + parameters 40 R (9 f32 + lockout i32)
#   + action 1 (read, or written when the bang-bang rule runs in-kernel) + reward 4 W + 7 obs planes 28 W
B_ALG = 99
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling

E_PER_GPU, N_HOUSES = 4096, 1024


def c3_config(mdr):
    cfg = mdr.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = N_HOUSES
    env["cluster_prop"]["temp_mode"] = "noisy_sinusoidal_heatwave"
    env["power_grid_prop"]["base_power_mode"] = "constant"
    env["power_grid_prop"]["signal_mode"] = "perlin"
    env["start_datetime_mode"] = "random"
    cfg["noise_house_prop"]["noise_mode"] = "house_big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    cfg["default_hvac_prop"]["lockout_noise"] = 0
    cfg["default_house_prop"]["solar_gain_bool"] = True
    return cfg


def cpu_baseline(cfg_full, seconds):
    """The oracle's loop-form port (reference cost structure) timed on one host core, bounded sample."""
    import copy
    from oracle import loop_port
    cfg = copy.deepcopy(cfg_full)
    cfg["default_env_prop"]["power_grid_prop"]["signal_mode"] = "sinusoidals"   # deterministic family the port restates
    rate, steps, el = loop_port.time_baseline(cfg, seconds=seconds)
    out = {"value": rate, "unit": "house-steps/s", "cores": 1, "kind": "port",
           "sample": "oracle/loop_port.py (object-per-house pure-Python restatement, obs dicts + 10-neighbour "
                     "messages as the reference builds them), 1 env x %d houses x %d bang-bang steps, %.1f s"
                     % (N_HOUSES, steps, el)}
    try:   # the reference's only parallelism: independent processes side by side
        procs = max(1, min(16, (os.cpu_count() or 1)))
        mp_rate, procs = loop_port.time_baseline_parallel(cfg, procs, seconds=min(6.0, seconds))
        out["multi_process_value"] = mp_rate
        out["multi_process_cores"] = procs
    except Exception as exc:
        out["multi_process_value"] = None
        out["multi_process_cores"] = "unavailable: %s" % exc
    try:   # the same per-house arithmetic compiled (oracle/mdr_oracle_c.c): what one core does without the interpreter
        from oracle import c_port
        crate, csteps, cel = c_port.time_baseline(cfg, nb_envs=4, seconds=min(4.0, seconds))
        out["c_port_value"] = crate
        out["c_port_sample"] = "oracle/mdr_oracle_c.c (plain C, literal closed-form update, gcc -O2), 4 envs x %d houses x %d steps, %.1f s, 1 core" % (N_HOUSES, csteps, cel)
    except Exception as exc:   # no compiler on the box: the Python port above still stands
        out["c_port_value"] = None
        out["c_port_sample"] = "unavailable: %s" % exc
    try:   # the vectorised NumPy oracle (fp64, [E, N] arrays): SURVEY 8d's third CPU form
        import time
        import numpy as np
        from oracle import mdr_oracle as mo
        ora = mo.OracleEnv(cfg, nb_envs=4).reset(seed=2024, episode=0)
        nsteps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min(3.0, seconds):
            ora.step(ora.bangbang_actions().astype(np.uint8))
            nsteps += 1
        el = time.perf_counter() - t0
        out["numpy_value"] = 4 * N_HOUSES * nsteps / el
        out["numpy_sample"] = "oracle/mdr_oracle.py (vectorised NumPy fp64), 4 envs x %d houses x %d steps, %.1f s, 1 process" % (N_HOUSES, nsteps, el)
    except Exception as exc:
        out["numpy_value"] = None
        out["numpy_sample"] = "unavailable: %s" % exc
    return out


def traffic_from_profiles():
    """HBM bytes per launch of the step kernel from the committed PMC passes (profiles/*traffic*.json), or None."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            with open(path) as f:
                best = json.load(f)
        except Exception:
            pass
    return None if best is None else best.get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stagger", type=int, default=int(os.environ.get("MDR_STAGGER", "2304")))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mdr_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # one process per GPU; MDR_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the multi-rank code path
    # on a one-GPU box (RCCL refuses two ranks on one device) - never used for reported numbers
    backend = os.environ.get("MDR_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    cfg = c3_config(mdr_amd)
    e_per_gpu = int(os.environ.get("MDR_BENCH_ENVS", E_PER_GPU))   # rehearsal knob only; the reported config is 4096
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=e_per_gpu, device=device, seed=2024,
                                           env_offset=rank * e_per_gpu, table_steps=64, stagger_bytes=args.stagger)
    env.reset(episode=0)
    env.rollout(args.warmup)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record()               # same stream the kernels are launched on (torch's current stream)
    env.rollout(args.steps)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps   # average launch-to-launch duration of the step kernel

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    houses = e_per_gpu * N_HOUSES * world
    value = houses * args.steps / elapsed

    # sanity: the rollout really advanced and the state is finite (cheap, outside the timed region)
    assert env.steps_taken == args.warmup + args.steps
    assert bool(torch.isfinite(env.t["Ta"]).all()) and bool(torch.isfinite(env.t["reward"]).all())

    if rank == 0:
        achieved = B_ALG * e_per_gpu * N_HOUSES / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "house-steps/sec at 4096 envs x 1024 houses; achieved HBM GB/s vs roofline",
            "value": value, "unit": "house-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: %d envs x %d houses per GPU, house_big_noise + big_noise HVAC, "
                                   "noisy_sinusoidal_heatwave OD temp, solar gain, perlin signal, random start, "
                                   "in-kernel bang-bang closed loop" % (e_per_gpu, N_HOUSES),
                       "envs_per_gpu": e_per_gpu, "houses_per_env": N_HOUSES, "sharding": "independent env replicas, no collective",
                       "seed": 2024, "table_steps": 64},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profiles(),
                         "kernel": "k_step_fused<4,1,256>", "algorithmic_bytes_per_house_step": B_ALG,
                         "kernel_ms": kernel_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
