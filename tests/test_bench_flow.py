"""bench.py's control flow on CPU: world_size-2 gloo dry run with an injected stub solver.

What is under test is everything of `bench.run` that is not the HIP workload: strong-scaling shard selection
(global batch 4096 split N ways, contiguous), W warm-up + exactly K timed steps between barriers, the single
gather of y* to rank 0, max-over-ranks timing, the C4 (nIter = 30) extra, and the JSON contract."""
import json
import os
import sys
import subprocess

import pytest

import spawn_util

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from bench_stub import StubWorkload  # noqa: E402


def _worker(rank, world, port, argv, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    captured = {}

    class Factory(StubWorkload):
        def __init__(self, *a):
            super().__init__(*a)
            captured["wl"] = self

    out = bench.run(bench.parse_args(argv), workload_factory=Factory, backend="gloo")
    out["_calls"] = captured["wl"].calls
    out["_local_batch"] = captured["wl"].local_batch
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as fh:
        json.dump(out, fh)


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_two_rank_dry_run_of_the_bench_control_flow(tmp_path, scaling):
    world, steps, warmup = 2, 4, 2
    argv = ["--gpus", "2", "--steps", str(steps), "--warmup", str(warmup), "--batch", "10", "--scaling", scaling,
            "--c4-steps", "2", "--cpu-sample", "0", "--weak-steps", "0"]
    spawn_util.spawn(_worker, lambda port: (world, port, argv, str(tmp_path)), world)
    outs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    o = outs[0]
    assert o["n_gpus"] == 2 and o["world_size"] == 2 and o["steps"] == steps and o["warmup"] == warmup
    assert o["scaling"] == scaling and o["higher_is_better"] is True and o["vs_baseline"] is None
    gb = 10 if scaling == "strong" else 20
    assert o["config"]["global_batch"] == gb
    assert [x["_local_batch"] for x in outs] == ([5, 5] if scaling == "strong" else [10, 10])
    # probe + warm-up + K timed headline steps, then 1 + 2 steps of the nIter = 30 extra
    assert outs[0]["_calls"] == [10] * (1 + warmup + steps) + [30] * 3
    # whole-job value from the slowest rank's clock
    assert len(o["per_rank_ms_per_step"]) == 2
    assert o["ms_per_step"] == pytest.approx(max(o["per_rank_ms_per_step"]))
    assert o["value"] == pytest.approx(gb * 10 * 1e3 / o["ms_per_step"])
    assert o["per_rank_ms_per_step"][1] >= 4.0                      # rank 1 sleeps 4 ms per step
    assert o["extra"]["c4"]["value"] == pytest.approx(gb * 30 * 1e3 / o["extra"]["c4"]["ms_per_step"])
    assert "gather to rank 0" in o["config"]["parallelism"]
    # where a step goes, per rank: the solve of its shard and the one collective, from each rank's own events
    for blk in (o, o["extra"]["c4"]):
        assert len(blk["per_rank_solve_ms"]) == 2 and len(blk["per_rank_gather_ms"]) == 2
        assert blk["per_rank_solve_ms"][1] >= 4.0 > blk["per_rank_solve_ms"][0] >= 2.0     # the stub sleeps 2 ms / 4 ms
        assert all(g >= 0.0 for g in blk["per_rank_gather_ms"])


def test_single_process_contract_fields():
    sys.path.insert(0, REPO)
    import bench
    args = bench.parse_args(["--steps", "3", "--warmup", "1", "--batch", "8", "--c4-steps", "0", "--cpu-sample", "0"])
    assert args.gpus == 1 and args.scaling == "strong"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    out = bench.run(args, workload_factory=StubWorkload, backend="gloo")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in out
    assert out["n_gpus"] == 1 and out["config"]["global_batch"] == 8 and "extra" not in out


def _run_bench(argv, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_plain_shell_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no torchrun and no WORLD_SIZE in the environment (how the driver spells the N = 1
    command): bench.py starts the two ranks itself, rank 0 prints the one JSON line, exit status 0."""
    p = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "10", "--c4-steps", "1", "--cpu-sample", "0",
                    "--backend", "gloo", "--workload", "tests/bench_stub.py:StubWorkload"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["world_size"] == 2 and o["steps"] == 3 and o["warmup"] == 1
    assert len(o["per_rank_ms_per_step"]) == 2 and len(o["per_rank_solve_ms"]) == 2 and len(o["per_rank_gather_ms"]) == 2
    assert o["config"]["global_batch"] == 10 and o["config"]["per_gpu_batch"] == 5
    assert o["value"] == pytest.approx(10 * 10 * 1e3 / o["ms_per_step"])
    assert o["per_rank_solve_ms"][1] >= 4.0 > o["per_rank_solve_ms"][0]
    # a strong-scaling run also reports the weak-scaling point: --batch samples PER RANK
    w = o["extra"]["weak"]
    assert w["scaling"] == "weak" and w["steps"] == 10 and len(w["per_rank_ms_per_step"]) == 2
    assert w["value"] == pytest.approx(20 * 10 * 1e3 / w["ms_per_step"])


def test_self_launch_reports_a_failed_rank():
    p = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "10", "--c4-steps", "0", "--cpu-sample", "0",
                    "--backend", "gloo", "--workload", "tests/bench_stub.py:FailsOnRankOne", "--launch-timeout", "120"])
    assert p.returncode != 0
    assert "rank 1 exited" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_plain_single_gpu_command_and_world_size_mismatch():
    argv = ["--steps", "1", "--warmup", "0", "--batch", "8", "--c4-steps", "0", "--cpu-sample", "0",
            "--workload", "tests/bench_stub.py:StubWorkload"]
    p = _run_bench(argv)
    assert p.returncode == 0, p.stderr[-2000:]
    o = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert o["n_gpus"] == 1 and "per_rank_ms_per_step" in o
    # a launcher that started 2 ranks for --gpus 1: a message and a non-zero status before any rendezvous, not a hang
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    q = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=120)
    assert q.returncode != 0 and "WORLD_SIZE=2" in q.stderr


def test_self_launch_refuses_more_ranks_than_gpus():
    """The product workload on a box with fewer GPUs than --gpus (here: none): a message and exit status 2 before any rank
    is started, not N processes that all die in cudaSetDevice."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    p = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--c4-steps", "0", "--cpu-sample", "0"], timeout=120)
    assert p.returncode == 2 and "GPU(s) visible" in p.stderr and not p.stdout.strip()
