"""bench.py end to end on the GPU box: the default single-GPU line (with its secondary legs) and the self-launched two-rank form
(gloo, both ranks sharing the one GPU - RCCL refuses two ranks on one device; reduced batch, never a reported number)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    return lines[0]


@pytest.mark.gpu
def test_single_gpu_line_with_legs_over_rccl():
    line = _run({"MDR_BENCH_ENVS": "512"}, "--steps", "50", "--warmup", "5", "--no-cpu-baseline", "--ppo-steps", "3", "--c5-steps", "20")
    assert line["n_gpus"] == 1 and line["value"] > 1e9 and line["roofline"]["bound"] == "hbm"
    # (MDR_BENCH_ENVS=512: the whole 52 MB batch lives in the caches - its algorithmic rate can pass the 8 TB/s HBM figure by a hair)
    assert 0 < line["roofline"]["frac"] < 1.25 and line["roofline"]["frac_of_measured_copy"] > line["roofline"]["frac"]
    assert "error" not in line["ppo_rollout"], line["ppo_rollout"]
    for prec in ("fp32", "bf16x3"):
        for key in ("transitions", "no_states"):
            assert line["ppo_rollout"][prec][key]["agent_steps_per_s"] > 1e7
    assert "error" not in line["c5"], line["c5"]        # the exchange of a world of one ran over RCCL
    assert line["c5"]["backend"] == "rccl" and line["c5"]["houses_per_rank"] == 1_000_000 and line["c5"]["value"] > 1e9
    assert line["c5"]["collective_us_per_step"] > 0 and line["c5"]["kernel_us_per_step"] > 0
    g = line["c5_graph"]                                # the same step, captured with its all-gather in a hipGraph and replayed
    assert "error" not in g, g
    assert g["captured"] and g["backend"] == "rccl" and g["checksum_Ta"] == line["c5"]["checksum_Ta"]
    assert g["us_per_step"] < line["c5"]["us_per_step"]
    p = line["c5_persistent"]                           # the same env without a kernel boundary or a collective per step
    assert "error" not in p, p
    assert p["checksum_Ta"] == line["c5"]["checksum_Ta"] and p["houses_per_rank"] == 1_000_000
    assert p["us_per_step"] < g["us_per_step"] and p["us_per_step_no_accumulators"] < g["us_per_step"]
    assert line["degraded"] is False and "degraded_reasons" not in line
    assert 0.4 < line["roofline"]["frac_out_of_cache"] < 1.0


@pytest.mark.gpu
def test_two_ranks_self_launched_gloo_on_one_gpu():
    line = _run({"MDR_BENCH_BACKEND": "gloo", "MDR_BENCH_ENVS": "256"}, "--gpus", "2", "--steps", "20", "--warmup", "5", "--ppo-steps", "2",
                "--c5-steps", "10")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 1e8
    assert line["config"]["envs_per_gpu"] == 256
    assert line["ppo_rollout"]["n_gpus"] == 2 and line["ppo_rollout"]["value"] > 1e6
    assert line["c5"]["n_gpus"] == 2 and line["c5"]["houses_per_rank"] == 500_000 and line["c5"]["backend"] == "gloo"
    assert line["c5_graph"]["captured"] is False and line["c5_graph"]["checksum_Ta"] == line["c5"]["checksum_Ta"]   # gloo: not capturable, stepped eagerly
    assert "cpu_baseline" not in line
    p = line["c5_persistent"]      # one child process per rank, peer mailboxes over hipIpc between the two (both on the one GPU here)
    assert "error" not in p, p
    assert p["houses_per_rank"] == 500_000 and len(p["checksum_Ta_ranks"]) == 2 and "hipIpc" in p["exchange"]
    assert line["degraded"] is True and any("gloo" in r for r in line["degraded_reasons"])


@pytest.mark.gpu
def test_a_leg_that_does_not_return_never_costs_the_headline():
    """--leg-timeout: the headline is measured before the secondary legs; if one hangs (here: a timeout shorter than any leg) rank 0
    still prints the ONE line - with the leg marked - and the process exits 0."""
    line = _run({"MDR_BENCH_ENVS": "256"}, "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--leg-timeout", "0.2")
    assert line["value"] > 1e9 and line["roofline"]["frac"] > 0
    assert "did not finish" in line["c5_graph"]["error"]                      # the last leg cannot have made it in 0.2 s
    assert "value" in line["ppo_rollout"] or "did not finish" in line["ppo_rollout"]["error"]


@pytest.mark.gpu
def test_a_leg_that_dies_on_one_rank_costs_neither_the_line_nor_the_exit_code():
    """Rank 1 raises inside the c5 leg while rank 0 enters its collectives: rank 1 skips the remaining legs and leaves through the
    watchdog (its close() barrier never completes), rank 0's watchdog prints the line with the leg marked; exit code 0."""
    line = _run({"MDR_BENCH_BACKEND": "gloo", "MDR_BENCH_ENVS": "256", "MDR_BENCH_FAIL_LEG": "c5:1"}, "--gpus", "2", "--steps", "20",
                "--warmup", "5", "--ppo-steps", "2", "--c5-steps", "10", "--leg-timeout", "25", timeout=300)
    assert line["n_gpus"] == 2 and line["value"] > 1e8
    assert "value" in line["ppo_rollout"]                          # finished before the failure
    assert "error" in line["c5"] and "error" in line["c5_graph"]


@pytest.mark.gpu
def test_a_leg_that_dies_on_rank_0_is_reported_and_the_rest_skipped():
    line = _run({"MDR_BENCH_ENVS": "256", "MDR_BENCH_FAIL_LEG": "ppo_rollout:0"}, "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                "--c5-steps", "10")
    assert "injected failure" in line["ppo_rollout"]["error"]
    assert "value" in line["c5"] and "value" in line["c5_graph"]   # a world of one: the other legs still run


@pytest.mark.gpu
def test_the_fence_falls_back_to_gloo_when_rccl_does_not_come_up():
    """The env replicas of the headline need a fence, not RCCL: a communicator that fails to initialise must not cost the line."""
    line = _run({"MDR_BENCH_ENVS": "256", "MDR_BENCH_FORCE_DIST": "1", "MDR_BENCH_BREAK_NCCL": "1"}, "--steps", "20", "--warmup", "5",
                "--no-cpu-baseline", "--ppo-steps", "2", "--c5-steps", "10")
    assert line["value"] > 1e9 and "fence over gloo" in line["backend_note"]
    assert line["c5"]["backend"] == "gloo" and line["c5_graph"]["captured"] is False
