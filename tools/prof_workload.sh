#!/bin/bash
# Per-step kernel table of one bench workload (run from the repo root through gpurun): bash tools/prof_workload.sh <out name> <bench args...>
set -o pipefail
O=$PWD/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o run -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 10 --warmup 4 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/trace.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find $O/trace -name "*.db" | head -1) 10 > $O/per_step_kernel_table.txt 2>&1
cat $O/bench.json; head -60 $O/per_step_kernel_table.txt
