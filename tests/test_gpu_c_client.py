"""The C ABI driven from a torch-free C program (tools/c_abi_client.c: plain C99 built with gcc, hipMalloc'ed buffers,
plain structs) gives the same numbers as the Python host for the same configuration and seed: the boundary carries
no Python / torch state."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    import mdr_amd
    mdr_amd.build_native()
    exe = tmp_path_factory.mktemp("cabi") / "c_abi_client"
    csrc = os.path.join(ROOT, "marl-demandresponse-original_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tools", "c_abi_client.c"),
                    "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-L" + csrc, "-lmdr_hip",
                    "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("E,N,steps,mode", [(4, 64, 100, ""), (3, 1024, 70, ""), (2, 5000, 40, ""), (50, 20, 130, ""),
                                            (2, 5000, 150, "records"), (1, 125000, 70, "records")])
def test_c_client_matches_python_host(client, E, N, steps, mode):
    import mdr_amd
    out = subprocess.run([client, str(E), str(N), "7", str(steps)] + ([mode] if mode else []), check=True, capture_output=True, text=True).stdout
    got = json.loads(out.strip().splitlines()[-1])
    if mode == "records":      # one launch per step between the exchanges, except where the 64-row time tables are refilled
        assert got["fused"] == steps - 1 - (steps - 1) // 64

    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["default_env_prop"]["power_grid_prop"]["signal_mode"] = "sinusoidals"
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=7)
    env.reset(episode=0)
    env.rollout(steps)
    assert got["steps"] == steps == env.steps_taken
    assert got["sum_sso"] == int(env.t["sso"].sum().item())
    assert got["sum_P"] == pytest.approx(env.t["P"].sum().item(), rel=1e-12)
    assert got["Ta0"] == pytest.approx(env.t["Ta"].flatten()[0].item(), rel=1e-9)      # printed with 10 digits: same fp32 value
    assert got["TaLast"] == pytest.approx(env.t["Ta"].flatten()[-1].item(), rel=1e-9)
    assert got["sum_Ta"] == pytest.approx(env.t["Ta"].double().sum().item(), rel=1e-9)
    assert got["sum_reward"] == pytest.approx(env.t["reward"].double().sum().item(), rel=1e-9)
