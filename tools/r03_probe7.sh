#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p7
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "all gpu tests rc=$?" | tee -a $O/gpu_tests.log
tail -8 $O/gpu_tests.log | cut -c1-300
timeout 600 python tools/bench_configs.py F4 > $O/f4.log 2>&1; cut -c1-400 $O/f4.log | head -3
timeout 300 python tools/small_batch_time.py > $O/small_batch.txt 2>&1; cat $O/small_batch.txt | tail -12
bash tools/prof_round.sh r03_a > $O/prof_round.log 2>&1; tail -25 $O/prof_round.log | cut -c1-250
bash tools/prof_shapes.sh r03_a_shapes "c4 c4shard c3 c3n30 c5 adam" > $O/prof_shapes.log 2>&1; grep -E "^\{|^\| (fused|dual|fc_fg|conv|adam)" $O/prof_shapes.log | cut -c1-250
