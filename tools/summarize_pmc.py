#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc pass (…_counter_collection.csv) as JSON.

    python tools/summarize_pmc.py gpurun_out/pmc_r02_ppo/mfma_counter_collection.csv [kernel-substring ...]

Adds `mfma_util` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) when both counters are present
(GRBM_GUI_ACTIVE is reported summed over the 8 XCDs)."""
import collections
import csv
import json
import sys


def main():
    path, wanted = sys.argv[1], sys.argv[2:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if wanted and not any(w in name for w in wanted):
                continue
            vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for name, counters in vals.items():
        d = {k: sum(v) / len(v) for k, v in counters.items()}
        d["dispatches"] = max(len(v) for v in counters.values())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
            d["mfma_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "SQ_INSTS_VALU" in d and "SQ_INSTS_MFMA" in d and d["SQ_INSTS_MFMA"] > 0:
            d["valu_per_mfma"] = (d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / d["SQ_INSTS_MFMA"]
        out[name] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
