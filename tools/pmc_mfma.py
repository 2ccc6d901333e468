"""MFMA utilisation of the trunk's convolution kernels from one rocprofv3 PMC pass (tools only).

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d D -o run -- \
      python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline
  python tools/pmc_mfma.py D/run_counter_collection.csv > profiles/rNN_pmc_mfma.json

util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): SQ_VALU_MFMA_BUSY_CYCLES sums, over every SIMD, the
cycles its matrix pipe was busy (MI355X_MICROARCH.md: "counts cycles", 16 per v_mfma_f32_16x16x32_bf16); GRBM_GUI_ACTIVE is reported
summed over the 8 XCDs, so /8 = the dispatch's duration in shader cycles."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_common import is_trunk_conv
acc = {}
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row["Kernel_Name"]
        # the trunk's convolution kernels: tile8 (implicit GEMM), the patch-resident 3x3, the streaming and the panel-resident 1x1 kernels
        conv = is_trunk_conv(name)
        key = "conv" if conv else ("bn_act" if "bn_act_kernel" in name else None)
        if key is None:
            continue
        d = acc.setdefault(key, {})
        did = row.get("Dispatch_Id") or row.get("Correlation_Id")
        d.setdefault(did, {})[row["Counter_Name"]] = float(row["Counter_Value"])
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 3 --warmup 2 "
                 "--no-cpu-baseline --no-roofline; MI355X, cfg2 ResNet-50 bf16; aggregated by tools/pmc_mfma.py",
       "formula": "mfma_util = sum(SQ_VALU_MFMA_BUSY_CYCLES) / sum(GRBM_GUI_ACTIVE / 8 * 256 * 4)"}
for key, d in acc.items():
    rows = [v for v in d.values() if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v]
    if not rows:
        continue
    busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in rows)
    cyc = sum(v["GRBM_GUI_ACTIVE"] / 8.0 for v in rows)
    out[key] = {"dispatches": len(rows), "mfma_busy_cycles_per_dispatch": round(busy / len(rows), 1),
                "shader_cycles_per_dispatch": round(cyc / len(rows), 1), "mfma_util": round(busy / (cyc * 256 * 4), 4)}
print(json.dumps(out, indent=1))
