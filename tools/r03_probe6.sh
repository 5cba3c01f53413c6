#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p6
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s -k "callback or more_iterations or graph or sharded_context or small_rows_golden or small_rows" > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/new_tests.log
grep -E "passed|failed|callback replay|nIter=|sample |FAILED|differs|Error|sharded context" $O/new_tests.log | tail -40 | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "all gpu tests rc=$?" | tee -a $O/gpu_tests.log
tail -12 $O/gpu_tests.log | cut -c1-300
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-3000 $O/bench.json; tail -3 $O/bench.err
timeout 600 python tools/bench_configs.py C3 > $O/c3.log 2>&1; cut -c1-330 $O/c3.log
