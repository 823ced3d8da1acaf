"""bench.py's own launcher and its multi-rank bookkeeping, end to end on CPU.

`python bench.py --gpus 2` (no torch.distributed.run environment) must start its two ranks itself, shard the roots, run the
pipelined exchange and print ONE JSON line from rank 0 carrying both scaling readings.  The device search
is replaced by bench.py's stub (RGL_BENCH_STUB_SEARCH=1: gloo, CPU tensors, no kernels) -- this covers the launcher, not the
path; the line it prints is marked as not being a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env=None, launcher=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RGL_BENCH_STUB_SEARCH="1", OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + argv
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-3000:] + out.stderr[-3000:]
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks_and_reports_both_readings():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--roots", "11"])
    assert r["stub_search"] is True and r["metric"].startswith("STUB SEARCH")
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["steps"] == 4 and r["warmup"] == 1
    # `value`: the fixed-total reading -- 11 roots split 6 + 5
    assert r["scaling"] == "strong" and r["config"]["total_roots"] == 11 and r["config"]["roots_per_gpu"] == 6
    assert r["multi_gpu"]["roots_per_rank"] == [6, 5]
    assert len(r["multi_gpu"]["search_ms_per_step_by_rank"]) == 2 and len(r["multi_gpu"]["exchange_ms_per_step_by_rank"]) == 2
    assert abs(r["value"] - 249 * 11 * 4 / (r["ms_per_step"] * 4e-3)) < 1e-6 * r["value"]
    # ... and the per-GPU-fixed reading from the same invocation
    assert r["weak_total_roots"] == 22 and r["weak_roots_per_gpu"] == 11 and r["weak_value"] > 0
    assert r["weak_multi_gpu"]["roots_per_rank"] == [11, 11] and r["weak_multi_gpu"]["ranks_seen"] == 2
    assert r["cpu_baseline"] is None and r["roofline"] is None
    # the untimed set-up phase lasts RGL_BENCH_INIT_MS at least (round 6: the device's clock ramp is a matter of time): more than its
    # 40 steps here, the same count on every rank -- a step is an exchange, ranks that disagreed would hang -- and the line says so
    assert r["config"]["init_ms"] == 80.0 and r["config"]["init_steps"] > 40


def test_eight_ranks_uneven_shards_and_the_single_process_decisions():
    """The shape the driver's 8-GPU node will see, on gloo: 8 ranks through bench.py's own launcher, the
    fixed-total reading over 2048 roots (configs[2]: 256 per rank), 4096 roots at depth 3 (configs[3]: 512 per rank) and a root
    count 8 does not divide (2051: three ranks carry one root more).  Every rank is seen, the shards add up, the `multi_gpu`
    block is there, and the gathered decisions of the 8-rank run are the single-process run's, digest for digest."""
    for argv, total, per_rank in (
            (["--roots", "2048"], 2048, [256] * 8),
            (["--scaling", "strong", "--total-roots", "4096", "--depth", "3"], 4096, [512] * 8),
            (["--scaling", "strong", "--total-roots", "2051"], 2051, [257] * 3 + [256] * 5)):
        r8 = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"] + argv)
        assert r8["n_gpus"] == 8 and r8["ranks_seen"] == 8 and r8["scaling"] == "strong"
        assert r8["config"]["total_roots"] == total and r8["multi_gpu"]["roots_per_rank"] == per_rank
        assert sum(r8["multi_gpu"]["roots_per_rank"]) == total
        assert len(r8["multi_gpu"]["search_ms_per_step_by_rank"]) == 8 and len(r8["multi_gpu"]["exchange_ms_per_step_by_rank"]) == 8
        assert r8["decisions"]["roots"] == total
        one = ["--scaling", "strong", "--total-roots", str(total)] + (["--depth", "3"] if "--depth" in argv else [])
        r1 = _run(["--gpus", "1", "--steps", "2", "--warmup", "1"] + one)
        assert r1["n_gpus"] == 1 and r1["config"]["total_roots"] == total and "multi_gpu" not in r1
        assert r1["decisions"] == r8["decisions"], (argv, r1["decisions"], r8["decisions"])
        if argv[0] == "--roots":                      # `both`: the per-GPU-fixed reading rides along (2048 roots on each of 8 ranks)
            assert r8["weak_total_roots"] == 8 * 2048 and r8["weak_multi_gpu"]["roots_per_rank"] == [2048] * 8


def test_single_rank_line_keeps_its_shape():
    r = _run(["--steps", "3", "--warmup", "1", "--roots", "7"])
    assert r["n_gpus"] == 1 and r["scaling"] == "weak" and r["config"]["total_roots"] == 7 and "weak_value" not in r
    assert "multi_gpu" not in r and "ranks_seen" not in r
    assert r["config"]["init_steps"] >= 40 and r["config"]["init_ms"] == 80.0
    short = _run(["--steps", "3", "--warmup", "1", "--roots", "7"], extra_env={"RGL_BENCH_INIT_MS": "0", "RGL_BENCH_INIT_STEPS": "3"})
    assert short["config"]["init_steps"] == 3 and short["decisions"] == r["decisions"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "step_ms_device", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "decisions"):
        assert key in r, key


def test_the_drivers_torchrun_command_gives_the_same_multi_rank_line():
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29541"]
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--roots", "8", "--scaling", "strong", "--total-roots", "9"],
             launcher=launcher)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["multi_gpu"]["roots_per_rank"] == [5, 4] and "weak_value" not in r
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--roots", "8", "--scaling", "weak"], launcher=launcher)
    assert r["scaling"] == "weak" and r["config"]["total_roots"] == 16 and r["multi_gpu"]["roots_per_rank"] == [8, 8]


def test_gpu_count_mismatch_is_an_error_not_a_silent_single_rank_run():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(RGL_BENCH_STUB_SEARCH="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29542")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 4" in (out.stderr + out.stdout)


def test_clearance_scenes():
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    robot, humans = bench.synth_scenes(1000, 64, 19)
    p = humans[:, :, :2].numpy().astype(np.float64)
    d = np.linalg.norm(p[:, :, None] - p[:, None], axis=3) + np.eye(19) * 99
    assert d.min() >= bench.CLEARANCE - 1e-6
    assert np.linalg.norm(p - robot[:, None, :2].numpy(), axis=2).min() >= bench.CLEARANCE - 1e-6
    r2, h2 = bench.synth_scenes(1000, 64, 19)
    assert (r2 == robot).all() and (h2 == humans).all()                       # seeded
    r3, h3 = bench.synth_scenes(1000, 64, 19, "uniform")                      # the round-1/2 generator: same robots
    assert (r3 == robot).all() and not (h3 == humans).all()
