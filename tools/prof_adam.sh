# rocprofv3 kernel statistics of the RL agent's Adam inner optimiser (tools/bench_configs.py F4) -> gpurun_out/prof_adam
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_adam -- python $R/tools/bench_configs.py F4 > $O/prof_adam.log 2>&1
find $O/prof_adam -name "*kernel_stats.csv" | head -3
