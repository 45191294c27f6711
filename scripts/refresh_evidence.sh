set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r03d}
mkdir -p $O
timeout -k 10 420 python -m pytest tests -q -m gpu -x > $O/gputests.log 2>&1; tail -3 $O/gputests.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 scripts/kbench.py --only fwd,bwd,hash --iters 3 > $O/pmc_$c.log 2>&1
done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_SQ -- python3 scripts/kbench.py --only fwd,bwd,hash --iters 3 > $O/pmc_SQ.log 2>&1
python scripts/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ --tag ${1:-r03d} > $O/pmc_summary.log 2>&1; tail -5 $O/pmc_summary.log
cp profiles/${1:-r03d}_pmc* $O/ 
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.log; tail -c 600 $O/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --warmup 3 > $O/profiled_bench.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
timeout -k 10 200 python scripts/kbench.py --graph --iters 200 > $O/kbench.log 2>&1; cat $O/kbench.log
