"""PIN_DW_DELTA=hi (the deltas of the recomputing weight gradient streamed as their high fp16 pieces only) against the full
split: the decoder gradient of ONE 2^20-sample iteration on the c3 bench map, element by element, relative to the largest
element of the full gradient; and the time of a 6-iteration Mapper.mapping call either way."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _map_scale_worker as W

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
cfg, npts, dec, cdec, mp = W.build("c3", bs, 6)
mp.reuse_pool_records = True
t = mp._get_trainer(peek_bricks=True)
grads = {}
def hook(g):
    grads.setdefault(os.environ.get("PIN_DW_DELTA", "full"), []).append(g[:dec.flat_params().numel()].detach().cpu().numpy().astype(np.float64).copy())
state0 = (npts.local_geo_features.data.clone(), dec.flat_params().clone(), npts.local_point_certainties.clone(), npts.local_point_ts_update.clone())
def restore():
    npts.local_geo_features.data.copy_(state0[0]); dec.flat_params().copy_(state0[1])
    npts.local_point_certainties.copy_(state0[2]); npts.local_point_ts_update.copy_(state0[3])
for mode in ("full", "hi"):
    os.environ["PIN_DW_DELTA"] = mode
    restore()
    torch.manual_seed(7)
    t.on_grads = None
    # gradient of the first iteration: a hook on the trainer (engine.MapTrainer.on_grads sees [decoder | ...] gradients)
    seen = []
    t.on_grads = lambda g: seen.append(g[:dec.flat_params().numel()].detach().float().cpu().numpy().astype(np.float64).copy())
    mp.mapping(1)
    torch.cuda.synchronize()
    grads[mode] = seen[0] if seen else None
    t.on_grads = None
    ms = []
    for rep in range(3):
        restore(); torch.manual_seed(7 + rep)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mp.mapping(6)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3 / 6)
    print(mode, "ms per iteration (3 calls of 6):", [round(x, 3) for x in ms], flush=True)
gf, gh = grads["full"], grads["hi"]
if gf is not None and gh is not None:
    d = np.abs(gf - gh)
    print("decoder gradient: max |g| %.3e; hi-only vs full: max |diff| %.3e = %.2e of max |g|; median rel (|g| > 1e-3 max) %.2e; share of elements beyond 1e-4 max|g|: %.4f"
          % (np.abs(gf).max(), d.max(), d.max() / np.abs(gf).max(), np.median((d / np.abs(gf))[np.abs(gf) > 1e-3 * np.abs(gf).max()]), float((d > 1e-4 * np.abs(gf).max()).mean())))
