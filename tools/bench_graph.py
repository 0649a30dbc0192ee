#!/usr/bin/env python3
"""Launch-bound regime: one policy-in-the-loop step (obs rows -> fused actor -> env step) eager vs replayed from a graph.

    python tools/bench_graph.py [--shapes 1x10,64x50,1024x50,8192x50] [--steps 512]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="1x10,64x50,1024x50,8192x50")
    ap.add_argument("--steps", type=int, default=512)
    args = ap.parse_args()
    import mdr_amd
    from mdr_amd.policy import FusedActor
    from mdr_amd.rollout import ActorMLP
    for shape in args.shapes.split(","):
        E, N = (int(x) for x in shape.split("x"))
        cfg = mdr_amd.default_config()
        cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
        cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=1, table_steps=1024, graph_mode=True)
        env.reset(episode=0)
        F = env.obs_vector_length()
        torch.manual_seed(0)
        fused = FusedActor.from_module(ActorMLP(F).cuda())
        obs = torch.empty((E, N, F), device="cuda:0")
        act = torch.empty(E * N, dtype=torch.uint8, device="cuda:0")
        prob = torch.empty(E * N, device="cuda:0")

        def step():
            env.obs_vector("rows", out=obs)
            fused.sample(obs.view(-1, F), 7, 0, action=act, a_prob=prob, step_dev=env.device_time_index)
            env.step(act.view(E, N))

        def timed(fn, n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fn(n)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3

        for _ in range(20):
            step()
        eager_us = timed(lambda n: [step() for _ in range(n)], args.steps)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()

        def replay(n):
            done = 0
            while done < n:
                m = min(env.graph_room(), n - done)
                for _ in range(m):
                    g.replay()
                env.graph_replayed(m)
                done += m

        replay(50)
        graph_us = timed(replay, args.steps)
        print(json.dumps({"shape": shape, "agents": E * N, "eager_step_us": round(eager_us, 1), "graph_step_us": round(graph_us, 1),
                          "speedup": round(eager_us / graph_us, 2), "agent_steps_per_s_graph": round(E * N / (graph_us * 1e-6))}), flush=True)


if __name__ == "__main__":
    main()
