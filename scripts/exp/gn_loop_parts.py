"""What each launch of the registration iteration costs IN the frame: bench.py's frame with one of the three launches of the
iteration taken out (the results are then wrong -- this is a timing experiment, nothing else): PIN_EXP_GN_PARTS = nosolve (the
pose never moves, the sums are never cleared), noknn (the search of iteration 0 only), noacc (search + solve).  The odometry
stage of the printed line against an unpatched run is that launch's share, launch boundary included.
  usage: PIN_EXP_GN_PARTS=nosolve python scripts/exp/gn_loop_parts.py --steps 20 --warmup 5 --events none --no-parity ..."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pin_slam_amd import _lib  # noqa: E402

L = _lib.lib()
mode = os.environ.get("PIN_EXP_GN_PARTS", "")
acc_dev, knn_listed, solve = L.pin_gn_accumulate_dev, L.pin_gn_knn_listed, L.pin_gn_solve
if "nosolve" in mode:
    L.pin_gn_accumulate_solve = lambda f, gp, ct, lp, cur, nbr, nn, lab, n, sums, st, stream: acc_dev(f, gp, ct, cur, nbr, nn, lab, n, sums, st, stream)
if "noacc" in mode:
    L.pin_gn_accumulate_solve = lambda f, gp, ct, lp, cur, nbr, nn, lab, n, sums, st, stream: solve(sums, st, lp, stream)
if "noknn" in mode:
    L.pin_gn_knn_listed = lambda sp, bc, src, n, k, st, cur, nbr, nn, cell, lst, first, stream: (knn_listed(sp, bc, src, n, k, st, cur, nbr, nn, cell, lst, first, stream) if first else 0)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"), run_name="__main__")
