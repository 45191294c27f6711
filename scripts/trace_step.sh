# kernel trace of a few replayed steps -> gpurun_out/<tag>/ (timeline analysis: scripts/step_timeline.py)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-trace}
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/profiled_bench.log 2>&1
find $O/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace.csv
rm -rf $O/prof
