#!/bin/bash
# round 3, first GPU call: new full-size parity tests + where the time of C4 / C5 goes (GPU box only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p1
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -s -k "config4_full or config5_full or config3_reference or repack or rl_variant" > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/new_tests.log
tail -5 $O/new_tests.log
tools/probes/_bin/mfma_f64_order_probe > $O/mfma_order.txt 2>&1; cat $O/mfma_order.txt
timeout 300 python tools/dual_phase_profile.py 30 512 > $O/dual_phase_30_512.txt 2>&1
timeout 300 python tools/dual_phase_profile.py 30 4096 two > $O/dual_phase_30_4096_two.txt 2>&1
timeout 300 python tools/dual_phase_profile.py 10 4096 > $O/dual_phase_10_4096.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_c5 -- python $GRAFT_REPO_ROOT/tools/bench_configs.py C5 > $GRAFT_REPO_ROOT/$O/c5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_c4 -- python $GRAFT_REPO_ROOT/tools/bench_configs.py C4 > $GRAFT_REPO_ROOT/$O/c4.log 2>&1
cd $GRAFT_REPO_ROOT
for d in stats_c5 stats_c4; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -12 "$f"; cp "$f" $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "all gpu tests rc=$?" | tee -a $O/gpu_tests.log
tail -3 $O/gpu_tests.log
