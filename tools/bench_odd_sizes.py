import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, mdr_amd
for E, N in ((4096, 1024), (4096, 1023), (4096, 1022), (4096, 1021), (83886, 50), (83886, 51), (20971, 200), (20971, 201)):
    cfg = bench.c3_config(mdr_amd)
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=1, table_steps=64)
    env.reset(episode=0); env.rollout(70)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.rollout(300); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 300 * 1e3
    print(json.dumps({"envs": E, "houses": N, "us": round(us, 1), "GBps": round(E * N * 99 / us / 1e3)}), flush=True)
    del env; torch.cuda.empty_cache()
