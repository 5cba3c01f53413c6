#!/bin/bash
# round 3, second GPU call: grouped persistent tile kernel for nIter > 15, 32-slot kernels on a register diet (GPU box only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p2
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s -k "persistent or stragglers or config4_full or time_sliced or fused_matches_chain or fused_bibtex or period3 or cycle_shortcut or rl_variant" > $O/tile_tests.log 2>&1; echo "tile tests rc=$?" | tee -a $O/tile_tests.log
grep -E "passed|failed|C4 full|rounds:" $O/tile_tests.log | tail -12
timeout 600 python tools/bench_configs.py C4 > $O/c4.log 2>&1; cut -c1-330 $O/c4.log
timeout 300 python tools/dual_phase_profile.py 30 4096 > $O/dual_phase_30_4096.txt 2>&1; cat $O/dual_phase_30_4096.txt
timeout 300 python tools/dual_phase_profile.py 30 512 > $O/dual_phase_30_512.txt 2>&1; cat $O/dual_phase_30_512.txt
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 > $O/bench.json 2> $O/bench.err; cut -c1-1200 $O/bench.json
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "all gpu tests rc=$?" | tee -a $O/gpu_tests.log
tail -15 $O/gpu_tests.log
