#!/usr/bin/env python3
"""obs_vector('rows') for observation shapes other than the default one (optional columns, link defects, fewer neighbours) at C3's batch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdr_amd


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, state, msg, defect, comm in (("default", False, False, 0.0, 10), ("default + 10 % link defects", False, False, 0.1, 10),
                                       ("state thermal+hvac", "th", False, 0.0, 10), ("every optional column", True, True, 0.0, 10),
                                       ("6 neighbours", False, False, 0.0, 6)):
    cfg = mdr_amd.default_config()
    env_p = cfg["default_env_prop"]
    env_p["cluster_prop"].update(nb_agents=1024, nb_agents_comm=comm, comm_defect_prob=defect)
    env_p["power_grid_prop"]["base_power_mode"] = "constant"
    if state is True:
        env_p["state_properties"].update(hour=True, day=True, solar_gain=True, thermal=True, hvac=True)
    elif state == "th":
        env_p["state_properties"].update(thermal=True, hvac=True)
    if msg:
        env_p["message_properties"].update(thermal=True, hvac=True)
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=4096, seed=1)
    env.reset(episode=0)
    env.rollout(5)
    F = env.obs_vector_length()
    us = timeit(lambda: env.obs_vector("rows"))
    mb = 4096 * 1024 * (4 * F + 25) / 1e6
    print(json.dumps({"shape": name, "F": F, "rows_us": round(us, 1), "MB": round(mb), "GBps": round(mb / us * 1e3)}), flush=True)
    del env
    torch.cuda.empty_cache()
