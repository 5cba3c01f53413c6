#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p8
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "all gpu tests rc=$?" | tee -a $O/gpu_tests.log
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -12 | cut -c1-300
timeout 600 python bench.py --cpu-sample 0 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_p8/bench.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['solve_stats'])
print('c4', d['extra']['c4']['ms_per_step'], d['extra']['c4']['roofline']['frac'], d['extra']['c4']['solve_stats'])
PY
timeout 900 python tools/bench_configs.py C4 > $O/c4.log 2>&1; cut -c1-250 $O/c4.log
timeout 900 python tools/bench_configs.py C3 > $O/c3.log 2>&1; cut -c1-250 $O/c3.log
timeout 900 python tools/bench_configs.py C2 > $O/c2.log 2>&1; cut -c1-250 $O/c2.log
timeout 300 python tools/conv_dual_phase_profile.py 5 > $O/conv_phase_5.txt 2>&1; head -3 $O/conv_phase_5.txt; tail -1 $O/conv_phase_5.txt
timeout 300 python tools/tile_budget_sweep.py 8 12 0 > $O/budget.txt 2>&1; cat $O/budget.txt
