import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--steps", "300", "--no-cpu-baseline", "--no-roofline"]
sys.path.insert(0, os.getcwd())
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:3500])
