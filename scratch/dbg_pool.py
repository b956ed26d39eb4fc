import numpy as np, torch
from tests import golden_util as G
from tests.test_gpu_process import _cfg
from pin_slam_amd import pool as P
d = G.load("process")
f = "f0_"
scan = torch.from_numpy(d[f + "scan"]).cuda()
rnd = tuple(torch.from_numpy(d[f + k]).cuda() for k in ("rnd_surface", "rnd_front", "rnd_behind"))
pool = P.SamplePool(capacity=1024)
pool.append_samples(scan, P.sample_params(_cfg(d), d[f + "pose"], 0), rnd=rnd)
for name, key in (("coord", "s_coord"), ("sdf_label", "s_label"), ("weight", "s_weight")):
    got = pool.view(name).cpu().numpy(); ref = d[f + key]
    bad = got.view(np.uint32) != ref.view(np.uint32)
    if bad.ndim == 2: bad = bad.any(1)
    idx = np.nonzero(bad)[0]
    print(name, len(idx), "of", len(ref), "by j:", np.bincount(idx % 7, minlength=7), "maxabs", np.abs(got - ref).max())
    for i in idx[:3]:
        print("  ", i, got[i], ref[i])
