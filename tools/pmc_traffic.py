"""HBM bytes per launch of the trunk kernels from two rocprofv3 PMC passes (tools only).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
  python tools/pmc_traffic.py A/run_counter_collection.csv B/run_counter_collection.csv > profiles/rNN_pmc_traffic.json

bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: counters are in KB; gfx950 FETCH_SIZE reports half of a wide
coalesced read)."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_common import TRUNK_CONV_LAUNCHES, is_trunk_conv
def per_kernel(path, counter):
    acc = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            key = "conv" if is_trunk_conv(name) else "bn_act" if "bn_act_kernel" in name else None
            if key is None:
                continue
            a = acc.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc
f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python3 bench.py --steps 3 --warmup 2 "
                 "--no-cpu-baseline; MI355X, cfg2 ResNet-50 bf16; aggregated by tools/pmc_traffic.py",
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read; counters are in KB)"}
for key, label in (("conv", "conv_bnstats"), ("bn_act", "bn_act")):
    if key in f and key in w:
        fk, wk = f[key][1] / f[key][0], w[key][1] / w[key][0]
        if key == "conv" and (f[key][0] % TRUNK_CONV_LAUNCHES or w[key][0] % TRUNK_CONV_LAUNCHES):
            raise SystemExit(f"conv launches profiled ({f[key][0]} / {w[key][0]}) are not a whole number of {TRUNK_CONV_LAUNCHES}-convolution trunk passes: "
                             "the kernel-name filter has drifted from the kernels")
        out[label] = {"launches_profiled": f[key][0], "fetch_kb_per_launch": round(fk, 1), "write_kb_per_launch": round(wk, 1),
                      "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
print(json.dumps(out, indent=1))
