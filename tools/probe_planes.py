import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch, mdr_amd
from mdr_amd.rollout import ActorMLP, _fused_policy
from bench_observe_ext import timeit
for flags in ((), ("thermal",), ("thermal", "hvac")):
    cfg = mdr_amd.default_config(); ep = cfg["default_env_prop"]; ep["cluster_prop"]["nb_agents"] = 1024
    ep["power_grid_prop"]["base_power_mode"] = "constant"
    for f in flags: ep["message_properties"][f] = True
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=4096, device="cuda:0", seed=1); env.reset(episode=0); env.rollout(5)
    F = env.obs_vector_length(); actor = ActorMLP(F).to("cuda:0")
    row = {"F": F}
    for prec in ("fp32", "bf16x3"):
        pol = _fused_policy(actor, env.device, prec)
        rows = env.obs_vector("rows").view(-1, F); planes = env.obs_vector("planes")
        pol.sample(rows, 0, 0); pol.sample(planes, 0, 0)
        row[prec + "_rows_us"] = round(timeit(lambda: pol.sample(rows, 0, 0)), 1)
        row[prec + "_planes_us"] = round(timeit(lambda: pol.sample(planes, 0, 0)), 1)
    row["obs_rows_us"] = round(timeit(lambda: env.obs_vector("rows")), 1); row["obs_planes_us"] = round(timeit(lambda: env.obs_vector("planes")), 1)
    print(json.dumps(row), flush=True)
    del env; torch.cuda.empty_cache()
