#!/bin/bash
# SQ counters of the trunk's kernels whose name matches <pattern> (tools only; run from the repo root through gpurun): bash tools/trunk_pmc.sh <out> <pattern>
set -o pipefail
O=$PWD/gpurun_out/$1; PAT=$2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/p1 -o run -- python3 $GRAFT_REPO_ROOT/tools/trunk_bench.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -o run -- python3 $GRAFT_REPO_ROOT/tools/trunk_bench.py > $O/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$O" "$PAT" <<'PY'
import csv, glob, sys, collections
O, PAT = sys.argv[1], sys.argv[2]
for p in ("p1", "p2"):
    fs = glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        n = row["Kernel_Name"]
        if PAT not in n: continue
        acc[n[:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key, cs in acc.items():
        print(p, key)
        for c, v in cs.items():
            print(f"    {c:30s} {sum(v) / len(v):16.1f}  (x{len(v)})")
PY
