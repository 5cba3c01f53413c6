#!/bin/bash
# Evidence of one build, on the GPU box:   gpurun -- 'bash tools/prof_round.sh r02_a'
# Writes gpurun_out/<tag>/: bench.json (the bench line), kernel_stats.csv (rocprofv3 --kernel-trace --stats of the
# bench's timed loop), pmc.md + traffic.json (separate --pmc passes: FETCH_SIZE alone, WRITE_SIZE alone, one SQ pass;
# never combined with other trace domains).  Copy what is to be judged into profiles/.
TAG=${1:-r02_x}
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
BENCH="python $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 --c4-steps 0 --c3-steps 0"
cd $R && python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --c4-steps 0 --c3-steps 0 > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --c4-steps 0 --c3-steps 0 > $O/pmc_sq.log 2>&1
cd $R && python tools/prof_collect.py $O $TAG
ls -la $O
