"""The CPU-side C of the test infrastructure (oracle/mdr_oracle_c.c) under AddressSanitizer + UndefinedBehaviorSanitizer:
golden replays (every penalty mode, lockout edges, N = 1) and a bang-bang run through the instrumented build, in a child process
with libasan preloaded; any report aborts the child (-fno-sanitize-recover, ASan's default abort).  Sanitizers exist on the CPU
build only - the GPU pool offers neither GPU ASan nor XNACK."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import numpy as np
from oracle import c_port
from oracle import mdr_oracle as mo
from tests import golden_util as gu
assert c_port.SANITIZED and c_port.build().endswith("_asan.so")
for name in ("s4_penalty_mixture", "s4_penalty_common_max", "s6_lockout_onoff", "s9_single_house", "s3_c3_heterogeneous"):
    g = gu.Golden(name)
    ora = mo.OracleEnv(g.config, nb_envs=1)
    ora.seed, ora.episode = g.seed, 0
    ora.load_episode(g.params(), od_table=g.od_table())
    port = c_port.CPort(ora)
    for t in range(min(g.T, 120)):
        od_old, sig_old = ora.OD.copy(), ora.S.copy()
        ora.step(g.a["actions"][t][None, :])
        port.step_arrays(g.a["actions"][t][None, :], od_old, ora.solar, sig_old)
        np.testing.assert_allclose(port.a["Ta"][0], g.a["Ta"][t], rtol=1e-11)
        np.testing.assert_allclose(port.a["reward"][0], g.a["reward"][t], rtol=1e-9, atol=1e-12)
cfg = gu.reference_env_config()
cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 37
cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
rate, steps, el = c_port.time_baseline(cfg, nb_envs=3, seconds=0.3)
assert steps > 0
print("sanitized ok", steps)
"""


def test_oracle_c_under_asan_and_ubsan():
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    env = dict(os.environ, MDR_ORACLE_C_SANITIZED="1", LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "sanitized ok" in res.stdout
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr
