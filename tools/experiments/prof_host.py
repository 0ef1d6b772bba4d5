import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--steps", "200", "--warmup", "3", "--no-cpu-baseline", "--no-hbm-leg", "--no-graph"]
sys.path.insert(0, os.getcwd())
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
open("gpurun_out/host_profile.txt", "w").write(s.getvalue())
