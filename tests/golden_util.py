"""Helpers to read the golden vectors in tests/golden/ (data captured from the reference)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "s*.npz")))


def _intkeys(cfg):
    """JSON turned the integer keys of cooling_capacity_list into strings; undo that."""
    for top in ("noise_hvac_prop", "noise_hvac_prop_test"):
        for mode in cfg[top]["noise_parameters"].values():
            if "cooling_capacity_list" in mode:
                mode["cooling_capacity_list"] = {int(k): v for k, v in mode["cooling_capacity_list"].items()}
    return cfg


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.name = name
        self.a = {k: z[k] for k in z.files}
        self.meta = json.loads(str(self.a["meta"]))
        self.config = _intkeys(self.meta["config"])
        self.T = self.meta["T"]
        self.N = self.meta["N"]
        self.seed = self.meta["seed"]

    def params(self):
        """Episode parameters in the naming OracleEnv.load_episode / the product's load_episode use ([1,N] / [1])."""
        a = self.a
        p = {k: a["p_" + k][None, :] for k in ("Ta", "Tm", "target", "deadband", "Ua", "Cm", "Ca", "Hm",
                                                "capacity", "COP", "latent", "lockout")}
        p["t0"] = np.array([a["p_t0"]], dtype=np.int64)
        p["phase"] = np.array([a["p_phase"]], dtype=np.float64)
        p["ratio"] = np.array([a["p_ratio"]], dtype=np.float64)
        return p

    def interp_grid(self):
        """(values, axes dict) of the synthetic base-power grid of an interpolation-mode fixture, or None."""
        if "interp_grid_file" not in self.a:
            return None
        z = np.load(os.path.join(GOLDEN_DIR, str(self.a["interp_grid_file"])), allow_pickle=False)
        return z["values"].astype(np.float64), json.loads(str(z["axes"]))

    def od_table(self):
        return self.a["od"][:, None]       # [T+1, E=1]


def reference_env_config():
    with open(os.path.join(GOLDEN_DIR, "reference_env_config.json")) as f:
        return _intkeys(json.load(f))
