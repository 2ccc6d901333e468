"""Host-side cost of one train step (tools only): cProfile over the bench loop.  python tools/host_profile.py [steps]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + ["--no-cpu-baseline", "--no-roofline", "--steps", sys.argv[1] if len(sys.argv) > 1 else "200"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000], file=sys.stderr)
