#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p5
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s -k "small_rows or config4_full or config5_full or config3_reference" > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/new_tests.log
grep -E "passed|failed|C4 full|C5 full|C3 at|sample |FAILED|differs" $O/new_tests.log | tail -30 | cut -c1-300
timeout 600 python tools/bench_configs.py C5 > $O/c5.log 2>&1; cut -c1-230 $O/c5.log | tail -1
timeout 900 python tools/tile_budget_sweep.py > $O/budget.txt 2>&1; cat $O/budget.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_c5 -- python $GRAFT_REPO_ROOT/tools/bench_configs.py C5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/stats_c5 -name "*kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-200; cp "$f" $O/c5_kernel_stats.csv; rm -rf $O/stats_c5
