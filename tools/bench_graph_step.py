#!/usr/bin/env python3
"""Graph mode, launch-bound shapes: microseconds per captured-and-replayed step (16 steps per graph), bang-bang step alone and
with the fused observe -> act kernel in front of it.  MDR_CURSOR_ATOMIC_BLOCKS picks where the cursor advance runs (0 = always
the one-thread launch).  One JSON line per shape."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdr_amd
from mdr_amd.policy import FEATURES_OBSERVE, FusedActor
from mdr_amd.rollout import ActorMLP

U, REPS = 16, 40
for E, N in ((16, 50), (64, 50), (128, 50), (256, 20), (512, 50), (2048, 50), (64, 1024), (512, 1024)):
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=1, table_steps=U * REPS + 8, graph_mode=True)
    torch.manual_seed(0)
    fused = FusedActor.from_module(ActorMLP(env.obs_vector_length()).cuda(), feature_order=FEATURES_OBSERVE)
    act = torch.empty(E * N, dtype=torch.uint8, device="cuda:0")
    prob = torch.empty(E * N, device="cuda:0")

    def bang():
        env.step_bangbang()

    def policy():
        fused.sample_env(env, 7, 0, action=act, a_prob=prob, step_dev=env.device_time_index)
        env.step(act.view(E, N))

    row = {"envs": E, "houses": N, "atomic_blocks": os.environ.get("MDR_CURSOR_ATOMIC_BLOCKS", "default")}
    for name, one in (("bangbang_us", bang), ("observe_act_step_us", policy)):
        env.reset(episode=0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            one()
        torch.cuda.current_stream().wait_stream(side)
        env.graph_replayed(0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(U):
                one()
        g.replay()
        env.graph_replayed(U)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPS - 2):
            g.replay()
        torch.cuda.synchronize()
        row[name] = round((time.perf_counter() - t0) / ((REPS - 2) * U) * 1e6, 2)
        env.graph_replayed(U * (REPS - 2))
    print(json.dumps(row), flush=True)
