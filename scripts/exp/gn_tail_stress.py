"""Is the fused end of the registration iteration (the tile kernel's last block solves) ever short of a block's sums?  The same
one-iteration registration from the same pose, many times, in both forms: the pose after the step is the same up to the order of
the atomics (1e-7); a block whose sums were not visible to the last block would show as an outlier.  Also with other work on a
second stream (the e2e pipeline has the brick build there)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pin_slam_amd import engine, ops
from tests import golden_util as G, gpu_util as U
from tests.test_gpu_parity import _gn_params

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for case in G.CASES:
    d = G.load(case)
    d["table"] = G.dense_table(d)
    st, fs = U.search_state(d, d["table"].astype(np.int32)), U.field_state(d, local=True)
    src_all = U.dev(d["reg_src"])
    for n in sorted({min(3000, src_all.shape[0]), src_all.shape[0]}):
        src = src_all[:n].contiguous()
        for busy in (False, True):
            side = torch.cuda.Stream()
            a = torch.randn(2048, 2048, device="cuda")
            out = {}
            for form in ("own", "fused"):
                gn = engine.GNTracker(st, fs, _gn_params(d), d["cfg_reg_lm_lambda"], n)
                gn.bricks = ops.BrickCache(d["neighbor_dx"], int(d["num_nei_cells"])).build(st)
                gn.fuse_solve = form == "fused"
                P, C_ = [], []
                for r in range(reps):
                    if busy and r % 4 == 0:
                        with torch.cuda.stream(side):
                            b = a @ a
                    T, cnt, res, it, valid, x = gn.track(src, d["reg_Tinit"], 3, early_exit=False)
                    P.append(T[:3, 3].copy()); C_.append(cnt)
                P = np.array(P); C_ = np.array(C_)
                out[form] = (np.median(P, 0), np.abs(P - np.median(P, 0)).max(), np.bincount(C_ - C_.min())[:6], C_.min())
            dm = np.abs(out["own"][0] - out["fused"][0]).max()
            print(case, "n", n, "busy", busy, "| own max dev %.2e counts %s | fused max dev %.2e counts %s | medians differ %.2e" %
                  (out["own"][1], out["own"][2], out["fused"][1], out["fused"][2], dm), flush=True)
