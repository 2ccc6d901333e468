"""Average duration per kernel name from a rocprofv3 rocpd database (tools only): python tools/kernel_avgs.py run_results.db [min_count]"""
import collections, re, sqlite3, subprocess, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    a = agg[n]; a[0] += 1; a[1] += (e - s) / 1e3
def demangle(n):
    if n.startswith("_Z"):
        try: n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        except Exception: pass
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:110]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1]:10.1f} us total  x{v[0]:6d}  avg {v[1] / v[0]:8.2f} us  {demangle(k)}")
