import numpy as np, torch, sys
sys.path.insert(0, '.')
from tests import golden_util as G, gpu_util as U
from pin_slam_amd import ops
d = G.load("c2_wf"); 
st = U.search_state(d)
for tf in (0,1):
    d2, idx = ops.radius_search(st, U.dev(d["query"]), time_filtering=bool(tf))
    d2 = d2.cpu().numpy(); idx = idx.cpu().numpy()
    ref = d[f"rs_d2_tf{tf}"]; ri = d[f"rs_idx_tf{tf}"]
    bad = np.argwhere(d2.view(np.uint32) != ref.view(np.uint32))
    print(tf, "mismatch", len(bad), "of", d2.size, "idx mismatch", (idx!=ri).sum())
    for b in bad[:8]:
        print(b, d2[tuple(b)], ref[tuple(b)], idx[tuple(b)], ri[tuple(b)])
