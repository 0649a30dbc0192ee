"""bench.py's launch contract (CPU, gloo): `python bench.py --gpus N` starts its own ranks, rank 0 prints ONE JSON line, a failing
rank fails the launch.  MDR_BENCH_DRY=1 rehearses launcher, rendezvous, fence and max-over-ranks without a GPU (no kernel runs and
the line says so); the same launch with real work is tests/test_gpu_bench.py on the GPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config")


def _run(extra_env, *argv, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MDR_BENCH_BACKEND="gloo", **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)


def _json_lines(text):
    out = []
    for line in text.splitlines():
        if line.startswith("{"):
            out.append(json.loads(line))
    return out


def test_gpus_2_launches_its_own_ranks_and_prints_one_json_line():
    res = _run({"MDR_BENCH_DRY": "1"}, "--gpus", "2", "--steps", "7", "--warmup", "2")
    assert res.returncode == 0, res.stderr[-3000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout
    line = lines[0]
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line["n_gpus"] == 2 and line["steps"] == 7 and line["warmup"] == 2
    assert line["dry_run"] is True and "dry-run" in line["data"]
    assert line["ms_per_step"] >= 0.02 * 1e3 / 7 * 0.9      # max over ranks: rank 1 sleeps 20 ms


def test_gpus_8_dry_launch_is_the_shape_the_driver_uses():
    """Eight ranks on one node (the driver's scaling run): rendezvous on 127.0.0.1, fence, max over ranks, ONE line from rank 0."""
    res = _run({"MDR_BENCH_DRY": "1", "OMP_NUM_THREADS": "1"}, "--gpus", "8", "--steps", "3", "--warmup", "1", timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["dry_run"] is True
    assert lines[0]["ms_per_step"] >= 0.08 * 1e3 / 3 * 0.9      # max over ranks: rank 7 sleeps 80 ms


def test_a_failing_rank_fails_the_launch():
    res = _run({"MDR_BENCH_DRY": "fail1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert res.returncode != 0
    assert not _json_lines(res.stdout)


def test_world_size_mismatch_is_refused():
    res = _run({"MDR_BENCH_DRY": "1", "WORLD_SIZE": "1", "RANK": "0"}, "--gpus", "2")
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stderr + res.stdout)


def test_bench_never_touches_the_gpu_before_the_ranks_exist():
    """The parent of `--gpus N` must not initialise HIP (a launcher that has may not start other GPU programs on this pool): the
    launch decision comes before any torch / mdr_amd import."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main_body = src[src.index("def main():"):]
    assert main_body.index("launch_ranks(args, argv)") < main_body.index("run_rank(args)")
    head = src[:src.index("def c3_config")]
    assert "import torch" not in head and "import mdr_amd" not in head
