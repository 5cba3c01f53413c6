#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p4
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/c4_outliers.py > $O/outliers.txt 2>&1; cat $O/outliers.txt | cut -c1-600
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_c4 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 --c4-steps 5 > $GRAFT_REPO_ROOT/$O/bench_c4.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/stats_c4 -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200; cp "$f" $O/c4_kernel_stats.csv; rm -rf $O/stats_c4
cut -c1-1800 $O/bench_c4.log | tail -3
timeout 600 python -m pytest tests -m gpu -q -k "context or adam or repack" 2>&1 | tail -4
