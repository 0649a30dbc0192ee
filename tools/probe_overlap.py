#!/usr/bin/env python3
"""Can the per-step RCCL all-gather of the sharded step hide behind the step kernels (separate streams)?  Upper bound at a world of one:
K split-path steps of a 1,000,000-house env on the main stream while K all-gathers of the record block run on a side stream."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import mdr_amd
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = mdr_amd.default_config()
cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=1, device=dev, seed=1)
env.reset(episode=0)
part = env.t["partials"]
out = torch.empty((1,) + tuple(part.shape), dtype=part.dtype, device=dev)
side = torch.cuda.Stream()
K = 200


def steps():
    env.rollout(K)


def gathers():
    with torch.cuda.stream(side):
        for _ in range(K):
            dist.all_gather_into_tensor(out.view(part.shape[0], part.shape[1], 3), part)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6


for f in (steps, gathers):
    f(); torch.cuda.synchronize()
res = {"houses": env.nb_agents, "steps_alone_us": round(timed(steps), 2), "gathers_alone_us": round(timed(gathers), 2)}


def both():
    # interleave the host issue as the pipelined loop would: one gather, one step
    for _ in range(K):
        with torch.cuda.stream(side):
            dist.all_gather_into_tensor(out.view(part.shape[0], part.shape[1], 3), part)
        env.rollout(1)


res["interleaved_two_streams_us"] = round(timed(both), 2)
print(json.dumps(res))
dist.destroy_process_group()
