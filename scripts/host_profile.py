"""cProfile of the host side of the bench's frame loop (where the Python time between launches goes).
usage (GPU box): python scripts/host_profile.py [bench args]"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-parity", "--c4-iters", "0", "--skip-downsampled",
            "--events", "none"] + sys.argv[1:]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue()[:7000])
