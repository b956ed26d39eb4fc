"""The per-frame sorts timed alone: pin_voxel_downsample_fast (keys kernel + sort + emit) and pin_spatial_sort for a few sizes,
HIP events over `reps` calls, and the outputs' checksums (two builds must print the same ones).  PIN_LIBPINHIP picks the build."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_slam_amd import _lib, ops, synth  # noqa: E402
from pin_slam_amd._lib import check  # noqa: E402

L = _lib.lib()
m = synth.build_map(layers=16)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
print("lib", _lib.LIB_PATH)
for n, vox in ((300_000, 0.05), (100_000, 0.08), (60_000, 0.4), (30_000, 0.4)):
    scan = torch.from_numpy(synth.make_scan(m, n=n, seed=3)).cuda().contiguous()
    n = scan.shape[0]
    ws = torch.empty((int(L.pin_maint_workspace_bytes(n)),), dtype=torch.uint8, device="cuda")
    sel = torch.empty((n,), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((1,), dtype=torch.int32, device="cuda")
    out = torch.empty_like(scan)
    perm = torch.empty((n,), dtype=torch.int32, device="cuda")
    st = ops._stream()

    def vds():
        check(L.pin_voxel_downsample_fast(scan.data_ptr(), n, float(np.float32(vox)), sel.data_ptr(), cnt.data_ptr(), ws.data_ptr(),
                                          ws.numel(), st), "vds")

    def ssort():
        check(L.pin_spatial_sort(scan.data_ptr(), n, 0.1, out.data_ptr(), perm.data_ptr(), ws.data_ptr(), ws.numel(), st), "sort")

    res = {}
    for name, fn in (("vds", vds), ("spatial_sort", ssort)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) / reps * 1e3
    c = int(cnt.item())
    print(f"n={n:7d} vox={vox}: vds {res['vds']:6.1f} us (kept {c}, checksum {int(sel[:c].to(torch.int64).sum().item()) ^ int(sel[:c][::7].to(torch.int64).sum().item())})"
          f"   spatial_sort {res['spatial_sort']:6.1f} us (checksum {int((perm.to(torch.int64) * torch.arange(n, device='cuda')).sum().item())})")
