#!/bin/bash
# Per-kernel times of the trunk forward ALONE (run from the repo root through gpurun): bash tools/prof_trunk.sh <out name>   (env passes through)
set -o pipefail
O=$PWD/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $GRAFT_REPO_ROOT/tools/trunk_bench.py > $O/bench.txt 2> $O/err.txt
cd $GRAFT_REPO_ROOT
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(f"{O}/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(open(f"{O}/bench.txt").read().strip())
print(f"kernel time total {tot / 24 / 1e3:.1f} us per pass (24 passes)")
for r in rows[:26]:
    print(f"{float(r['TotalDurationNs']) / 24 / 1e3:8.1f} us/pass  x{int(r['Calls']) / 24:5.1f}  avg {float(r['AverageNs']) / 1e3:7.2f} us  {r['Name'][:110]}")
PY
