#!/usr/bin/env python3
"""bench.py's N > 1 path with the REAL workload on a one-GPU box, through the plain command line the driver uses:

    python bench.py --gpus 2 --one-device --backend gloo ...

bench.py starts its two ranks itself (bench.self_launch); both use cuda:0 and gloo is the backend (RCCL refuses two ranks on
one device).  Exercises what the CPU dry run cannot: HipWorkload's sharded context producer (all-reduce of the BatchNorm
sums on device tensors), the per-shard fused solves, the gather and the per-rank timings.  GPU box only."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--one-device", "--backend", "gloo",
                        "--steps", "5", "--warmup", "2", "--c4-steps", "2", "--cpu-sample", "0"] + sys.argv[1:],
                       env=env, capture_output=True, text=True, timeout=1200)
    sys.stderr.write(p.stderr[-4000:])
    if p.returncode != 0:
        sys.exit(p.returncode)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    out_dir = os.path.join(REPO, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    json.dump(d, open(os.path.join(out_dir, "two_ranks.json"), "w"))
    print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "world_size", "per_rank_solve_ms", "per_rank_gather_ms")})
    print(d["config"])
    print({k: v for k, v in d["extra"]["c4"].items() if k in ("ms_per_step", "per_rank_solve_ms", "per_rank_gather_ms", "kernel")})
