"""Per-step kernel summary from a rocprofv3 rocpd database (tools only): python tools/prof_summary.py run_results.db [steps]"""
import collections, re, sqlite3, subprocess, sys
c = sqlite3.connect(sys.argv[1])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "pack_image" in r[0]]
sel = rows[idx[-nsteps - 1]:idx[-1]]
def demangle(n):
    if n.startswith("_Z"):
        try: n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        except Exception: pass
    return n
cache = {}
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, g, w in sel:
    if n not in cache:
        d = demangle(n).replace("(anonymous namespace)::", "").replace("void ", "")
        d = re.sub(r"\(.*", "", d)
        d = d.replace("__bf16", "bf16").replace("(bool)", "")
        cache[n] = d[:120]
    a = agg[cache[n]]; a[0] += 1; a[1] += (e - s) / 1e3
span = (sel[-1][2] - sel[0][1]) / nsteps / 1e3
tot = sum(v[1] for v in agg.values()) / nsteps
print(f"{nsteps} steps: span {span:.1f} us/step, kernel-time sum {tot:.1f} us/step, {len(sel) / nsteps:.0f} launches/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1] / nsteps:9.1f} us/step  x{v[0] / nsteps:6.1f}  avg {v[1] / v[0]:7.2f} us  {k}")
