import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--events", "none"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pin_slam_amd.dropin.utils import mapper as M
from pin_slam_amd import preprocess as PP
prof = cProfile.Profile()
orig_pf = M.Mapper.process_frame
def pf(self, *a, **k):
    prof.enable(); r = orig_pf(self, *a, **k); prof.disable(); return r
M.Mapper.process_frame = pf
orig_call = PP.ScanPreprocessor.__call__
def pc(self, *a, **k):
    prof.enable(); r = orig_call(self, *a, **k); prof.disable(); return r
PP.ScanPreprocessor.__call__ = pc
bench.main()
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
