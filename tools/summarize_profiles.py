#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the tracked summaries under profiles/.

    python tools/summarize_profiles.py r01

Inputs (written on the GPU box by the commands recorded in profiles/<round>_README.md):
  gpurun_out/prof_<round>/step_kernel_stats.csv       rocprofv3 --kernel-trace --stats
  gpurun_out/pmc_fetch/fetch_counter_collection.csv   rocprofv3 --pmc FETCH_SIZE   (own pass)
  gpurun_out/pmc_write/write_counter_collection.csv   rocprofv3 --pmc WRITE_SIZE   (own pass)
Outputs: profiles/<round>_kernel_stats.csv, profiles/<round>_traffic.json, profiles/<round>_bench.json
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled;
WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def counter_mean(path, kernel_substr):
    vals = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel_substr in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in vals.items()}


def traffic(stats_csv, fetch_csv, write_csv, houses):
    """One kernel_stats row + the two PMC passes -> the traffic summary of k_step_fused at `houses` houses per launch."""
    with open(stats_csv) as f:
        rows = list(csv.DictReader(f))
    step = [r for r in rows if "k_step_fused" in r["Name"]][0]
    fetch, nf = counter_mean(fetch_csv, "k_step_fused")["FETCH_SIZE"]
    write, nw = counter_mean(write_csv, "k_step_fused")["WRITE_SIZE"]
    fetch_bytes = fetch * 1024 * 2      # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads
    write_bytes = write * 1024
    return {
        "kernel": step["Name"], "houses_per_launch": houses, "calls": int(step["Calls"]), "avg_ns": float(step["AverageNs"]),
        "min_ns": float(step["MinNs"]), "max_ns": float(step["MaxNs"]),
        "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write, "pmc_dispatches": [nf, nw],
        "fetch_bytes_corrected_x2": fetch_bytes, "write_bytes": write_bytes,
        "hbm_bytes_per_launch": fetch_bytes + write_bytes,
        "algorithmic_bytes_per_launch": 99 * houses,
        "algorithmic_read_bytes": 53 * houses, "algorithmic_write_bytes": 46 * houses,
        "traffic_over_algorithmic": (fetch_bytes + write_bytes) / (99.0 * houses),
        "achieved_GBps_from_rocprof_avg": 99 * houses / float(step["AverageNs"]),
        "note": "reads: state 13 + parameters 40 B/house; writes: state 13 + action 1 (in-kernel bang-bang) + reward 4 + obs 28 B/house; "
                "FETCH_SIZE / WRITE_SIZE count requests at the L2's memory side, Infinity-Cache hits included (MI355X_MICROARCH.md)",
    }


def main_r02(rnd):
    """Round 2 layout: the C3 headline (gpurun_out/prof_r02, pmc_*_r02), the same kernel with the re-read set at 8x the
    Infinity Cache (prof_r02_big, pmc_*_big: 32768 envs x 1024 houses) and the split-path kernels of C5 (prof_r02_c5)."""
    os.makedirs(P, exist_ok=True)
    shutil.copy(os.path.join(G, "prof_" + rnd, "step_kernel_stats.csv"), os.path.join(P, rnd + "_kernel_stats.csv"))
    shutil.copy(os.path.join(G, "prof_" + rnd + "_big", "big_kernel_stats.csv"), os.path.join(P, rnd + "_kernel_stats_32768envs.csv"))
    shutil.copy(os.path.join(G, "prof_" + rnd + "_c5", "c5_kernel_stats.csv"), os.path.join(P, rnd + "_c5_kernel_stats.csv"))
    c3 = traffic(os.path.join(G, "prof_" + rnd, "step_kernel_stats.csv"), os.path.join(G, "pmc_fetch_" + rnd, "fetch_counter_collection.csv"),
                 os.path.join(G, "pmc_write_" + rnd, "write_counter_collection.csv"), 4096 * 1024)
    big = traffic(os.path.join(G, "prof_" + rnd + "_big", "big_kernel_stats.csv"), os.path.join(G, "pmc_fetch_big", "fetch_counter_collection.csv"),
                  os.path.join(G, "pmc_write_big", "write_counter_collection.csv"), 32768 * 1024)
    with open(os.path.join(P, rnd + "_traffic.json"), "w") as f:
        json.dump(c3, f, indent=1)
    with open(os.path.join(P, rnd + "_traffic_32768envs.json"), "w") as f:
        json.dump(big, f, indent=1)
    print(json.dumps({"c3": c3, "x8": big}, indent=1))


def main_r03(rnd):
    """Round 3 layout: the C3 headline again (gpurun_out/prof_r03, pmc_fetch_r03, pmc_write_r03) and the same kernel without the
    seven observation planes (prof_r03_noplanes, pmc_*_r03_noplanes: 71 algorithmic bytes per house-step)."""
    os.makedirs(P, exist_ok=True)
    shutil.copy(os.path.join(G, "prof_" + rnd, "step_kernel_stats.csv"), os.path.join(P, rnd + "_kernel_stats.csv"))
    c3 = traffic(os.path.join(G, "prof_" + rnd, "step_kernel_stats.csv"), os.path.join(G, "pmc_fetch_" + rnd, "fetch_counter_collection.csv"),
                 os.path.join(G, "pmc_write_" + rnd, "write_counter_collection.csv"), 4096 * 1024)
    with open(os.path.join(P, rnd + "_traffic.json"), "w") as f:
        json.dump(c3, f, indent=1)
    out = {"c3": c3}
    stats = os.path.join(G, "prof_" + rnd + "_noplanes", "np_kernel_stats.csv")
    if os.path.isfile(stats):
        shutil.copy(stats, os.path.join(P, rnd + "_kernel_stats_noplanes.csv"))
        bare = traffic(stats, os.path.join(G, "pmc_fetch_" + rnd + "_noplanes", "fetch_counter_collection.csv"),
                       os.path.join(G, "pmc_write_" + rnd + "_noplanes", "write_counter_collection.csv"), 4096 * 1024)
        houses = 4096 * 1024
        bare.update({"algorithmic_bytes_per_launch": 71 * houses, "algorithmic_write_bytes": 18 * houses,
                     "traffic_over_algorithmic": bare["hbm_bytes_per_launch"] / (71.0 * houses),
                     "achieved_GBps_from_rocprof_avg": 71 * houses / bare["avg_ns"],
                     "note": "mdr_buffers_t.obs == NULL: reads as the headline (53 B/house), writes state 13 + action 1 + reward 4 = 18 B/house"})
        with open(os.path.join(P, rnd + "_traffic_noplanes.json"), "w") as f:
            json.dump(bare, f, indent=1)
        out["noplanes"] = bare
    print(json.dumps(out, indent=1))


def main(rnd):
    if rnd == "r03":
        return main_r03(rnd)
    if rnd != "r01":
        return main_r02(rnd)
    os.makedirs(P, exist_ok=True)
    stats = os.path.join(G, "prof_" + rnd, "step_kernel_stats.csv")
    shutil.copy(stats, os.path.join(P, rnd + "_kernel_stats.csv"))
    with open(stats) as f:
        rows = list(csv.DictReader(f))
    step = [r for r in rows if "k_step_fused" in r["Name"]][0]
    kernel = "k_step_fused"
    fetch, nf = counter_mean(os.path.join(G, "pmc_fetch", "fetch_counter_collection.csv"), kernel)["FETCH_SIZE"]
    write, nw = counter_mean(os.path.join(G, "pmc_write", "write_counter_collection.csv"), kernel)["WRITE_SIZE"]
    houses = 4096 * 1024
    fetch_bytes = fetch * 1024 * 2      # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads
    write_bytes = write * 1024
    out = {
        "kernel": step["Name"], "calls": int(step["Calls"]), "avg_ns": float(step["AverageNs"]),
        "min_ns": float(step["MinNs"]), "max_ns": float(step["MaxNs"]),
        "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write, "pmc_dispatches": [nf, nw],
        "fetch_bytes_corrected_x2": fetch_bytes, "write_bytes": write_bytes,
        "hbm_bytes_per_launch": fetch_bytes + write_bytes,
        "algorithmic_bytes_per_launch": 99 * houses,
        "algorithmic_read_bytes": 53 * houses, "algorithmic_write_bytes": 46 * houses,
        "achieved_GBps_from_rocprof_avg": 99 * houses / float(step["AverageNs"]),
        "note": "reads: state 13 + parameters 40 B/house; writes: state 13 + action 1 (in-kernel bang-bang) + reward 4 + obs 28 B/house",
    }
    with open(os.path.join(P, rnd + "_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    bench = os.path.join(G, "bench_" + rnd + ".json")
    if os.path.isfile(bench):
        shutil.copy(bench, os.path.join(P, rnd + "_bench.json"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
