#!/bin/bash
# Round profile collection on the GPU box (run from the repo root through gpurun): kernel-trace stats, per-step table, PMC passes.
# Every rocprofv3 command puts the program itself after `--` and collects counters in passes of their own (no trace domains beside --pmc
# other than --kernel-trace).
set -o pipefail
R=${1:-r03}
O=$PWD/gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $BENCH --steps 10 --warmup 5 --no-cpu-baseline --no-roofline > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --kernel-trace -d $O/trace -o run -- python3 $BENCH --steps 12 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_bench.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 $BENCH --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 $BENCH --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o run -- python3 $BENCH --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_mfma.err
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" | head -20
DB=$(find $O/trace -name "*.db" | head -1)
python tools/prof_summary.py $DB 10 > $O/per_step_kernel_table.txt 2>&1
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) > $O/pmc_traffic.json 2> $O/pmc_traffic.err
python tools/pmc_mfma.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) > $O/pmc_mfma.json 2> $O/pmc_mfma.err
head -40 $O/per_step_kernel_table.txt; cat $O/pmc_traffic.json $O/pmc_mfma.json
