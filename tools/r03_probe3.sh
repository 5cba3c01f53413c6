#!/bin/bash
# round 3, third GPU call: grouped tile kernel (fixed LDS budget) + four-samples-per-wave dual step (GPU box only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p3
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s -k "small_rows or persistent or stragglers or config4_full or config5_full" > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/new_tests.log
grep -E "passed|failed|C4 full|C5 full|FAILED|differs" $O/new_tests.log | tail -30
timeout 600 python tools/bench_configs.py C4 > $O/c4.log 2>&1; cut -c1-330 $O/c4.log
timeout 600 python tools/bench_configs.py C5 > $O/c5.log 2>&1; cut -c1-330 $O/c5.log
timeout 300 python tools/dual_phase_profile.py 30 4096 > $O/dual_phase_30_4096.txt 2>&1; cat $O/dual_phase_30_4096.txt
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "all gpu tests rc=$?" | tee -a $O/gpu_tests.log
tail -15 $O/gpu_tests.log
