set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R && python bench.py > $O/bench_d.json 2> $O/bench_d.err; tail -c 600 $O/bench_d.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d -- python $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $O/prof_d.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_d_$c -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/pmc_d_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_d_sq -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/pmc_d_sq.log 2>&1
find $O/prof_d $O/pmc_d_FETCH_SIZE $O/pmc_d_WRITE_SIZE $O/pmc_d_sq -name "*.csv" | head -20
