#!/bin/bash
# SQ counters of single convolution launches (tools only): where the waves' cycles go.  Run on the GPU box from the repo root.
# usage: bash tools/conv_pmc.sh <out dir under gpurun_out> <conv spec> [<conv spec> ...]   (specs as tools/conv_stamps.py takes them)
set -o pipefail
O=$PWD/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -o run -- python3 $GRAFT_REPO_ROOT/tools/conv_stamps.py "$@" > $O/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -o run -- python3 $GRAFT_REPO_ROOT/tools/conv_stamps.py "$@" > $O/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for p in ("p1", "p2"):
    fs = glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        n = row["Kernel_Name"]
        if "conv" not in n and "tile8" not in n: continue
        key = (n[:90], row.get("Grid_Size"), row.get("LDS_Block_Size") or row.get("LDS_Block_Size_v"))
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key, cs in acc.items():
        print(p, key)
        for c, v in cs.items():
            print(f"    {c:28s} {sum(v) / len(v):16.1f}  (x{len(v)})")
PY
