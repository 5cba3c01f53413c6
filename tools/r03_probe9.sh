#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p9
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "all gpu tests rc=$?" | tee -a $O/gpu_tests.log
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -12 | cut -c1-300
timeout 300 python tools/dual_phase_profile.py 30 512 > $O/dual_phase_30_512.txt 2>&1; cat $O/dual_phase_30_512.txt
timeout 300 python tools/dual_phase_profile.py 30 4096 > $O/dual_phase_30_4096.txt 2>&1; cat $O/dual_phase_30_4096.txt
timeout 300 python tools/dual_phase_profile.py 10 4096 > $O/dual_phase_10_4096.txt 2>&1; cat $O/dual_phase_10_4096.txt
bash tools/prof_round.sh r03_b > $O/prof_round.log 2>&1; tail -4 $O/prof_round.log | cut -c1-250
bash tools/prof_shapes.sh r03_b_shapes "c4 c4shard c3 c3n30 c2" > $O/prof_shapes.log 2>&1; grep -E "^\{" $O/prof_shapes.log | cut -c1-250
