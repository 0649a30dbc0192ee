#!/usr/bin/env python3
"""Soak run: many simulated days of the C3 batch in closed loop; checks that the state stays finite and physical, and that the fused
rollout (houses resident in registers) ends the same episode in the same bits as the step-by-step run.

    python tools/soak.py [steps]      (default 432,000 steps = 20 simulated days at 4 s)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mdr_amd  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 432000
    cfg = mdr_amd.default_config()
    env_p = cfg["default_env_prop"]
    env_p["cluster_prop"]["nb_agents"] = 1024
    env_p["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["noise_house_prop"]["noise_mode"] = "house_big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=4096, device="cuda:0", seed=2024)
    env.reset(episode=0)
    t0 = time.perf_counter()
    done = 0
    report = []
    while done < steps:
        n = min(43200, steps - done)          # two simulated days per chunk
        env.rollout(n)
        done += n
        Ta, Tm = env.house_temp(), env.house_mass_temp()
        ok = bool(torch.isfinite(env.t["Ta"]).all() and torch.isfinite(env.t["Tm"]).all() and torch.isfinite(env.t["reward"]).all())
        on = env.hvac_turned_on()
        row = {"steps": done, "finite": ok, "Ta_min": Ta.min().item(), "Ta_max": Ta.max().item(), "Tm_min": Tm.min().item(),
               "Tm_max": Tm.max().item(), "mean_abs_temp_error": (Ta - env.target_temp()).abs().mean().item(),
               "fraction_on": on.float().mean().item(), "sso_max": int(env.t["sso"].max().item()),
               "mean_reward": env.t["reward"].mean().item(), "wall_s": round(time.perf_counter() - t0, 1)}
        report.append(row)
        print(json.dumps(row), flush=True)
        assert ok and 10.0 < row["Ta_min"] and row["Ta_max"] < 45.0, row
        assert bool((env.t["sso"][on] == 0).all())
    el = time.perf_counter() - t0
    print(json.dumps({"total_steps": done, "house_steps": done * 4096 * 1024, "wall_s": round(el, 1),
                      "house_steps_per_s": done * 4096 * 1024 / el}))
    # the same episode once more with the houses resident in registers (mdr_env_rollout_fused): must end in the same bits
    twin = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=4096, device="cuda:0", seed=2024)
    twin.reset(episode=0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    done = 0
    while done < steps:
        n = min(43200, steps - done)
        twin.rollout_fused(n, accumulate=False)
        done += n
    torch.cuda.synchronize()
    el = time.perf_counter() - t1
    same = {name: bool(torch.equal(twin.t[name], env.t[name])) for name in ("Ta", "Tm", "sso", "flags", "reward", "P")}
    print(json.dumps({"fused_rollout_total_steps": done, "wall_s": round(el, 1), "house_steps_per_s": done * 4096 * 1024 / el,
                      "bit_identical_to_the_stepwise_run": same}))
    assert all(same.values()), same


if __name__ == "__main__":
    main()
