cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof_cfg5 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/r3_prof_cfg5.json 2> $O/r3_prof_cfg5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof_cfg4 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/r3_prof_cfg4.json 2> $O/r3_prof_cfg4.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for wl, steps in (("cfg4", 9), ("cfg5", 9)):
    f = glob.glob(f"gpurun_out/r3_prof_{wl}/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(wl, "total kernel ms over the run:", round(tot / 1e6, 2))
    for r in rows[:22]:
        print(f'{float(r["TotalDurationNs"]) / 1e3 / steps:9.1f} us/step  x{int(r["Calls"]) / steps:6.1f}  avg {float(r["AverageNs"]) / 1e3:8.2f} us  {r["Name"][:110]}')
PY
