#!/bin/bash
# Per-shape evidence of one build on the GPU box:   gpurun -- 'bash tools/prof_shapes.sh r03_a "c4 c4shard c3 c3n30 c5 adam"'
# For every shape: rocprofv3 --kernel-trace --stats, then separate --pmc passes (FETCH_SIZE alone, WRITE_SIZE alone, one SQ
# pass; never combined with other trace domains) of `tools/prof_target.py <shape>`.  tools/prof_collect_shapes.py condenses
# gpurun_out/<tag>/<shape>/ into <tag>_<shape>_kernel_stats.csv / _pmc.md and merges profiles-ready traffic rows.
TAG=${1:-r03_x}
SHAPES=${2:-"c4 c4shard c3 c3n30 c5 adam"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in $SHAPES; do
  O=$R/gpurun_out/$TAG/$s
  mkdir -p $O
  T="python $R/tools/prof_target.py $s"
  $T > $O/run.json 2> $O/run.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $T > $O/stats.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- $T 2 > $O/pmc_$c.log 2>&1
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- $T 2 > $O/pmc_sq.log 2>&1
  cat $O/run.json
done
cd $R && python tools/prof_collect_shapes.py gpurun_out/$TAG $TAG
