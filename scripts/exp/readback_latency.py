"""Round trip of a count read-back (a small launch, a copy into pinned memory, a stream synchronisation) as the host sees it,
for the runtime's wait settings: run as  [ROC_ACTIVE_WAIT_TIMEOUT=us] [HSA_ENABLE_INTERRUPT=0] python scripts/exp/readback_latency.py"""
import os, time
import torch
dev = torch.zeros(4, dtype=torch.int32, device="cuda")
host = torch.zeros(4, dtype=torch.int32).pin_memory()
big = torch.zeros(1 << 22, device="cuda")
s = torch.cuda.current_stream()
def rt(work_us):
    ts = []
    for _ in range(300):
        if work_us:
            for _ in range(work_us):
                big.add_(1.0)  # ~ 6 us each
        dev.add_(1)
        t0 = time.perf_counter()
        host.copy_(dev, non_blocking=True)
        s.synchronize()
        v = host[0].item()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts)[10:-10]
    return 1e6 * sum(ts) / len(ts), 1e6 * ts[len(ts) // 2]
print({k: os.environ.get(k) for k in ("ROC_ACTIVE_WAIT_TIMEOUT", "HSA_ENABLE_INTERRUPT")})
for w in (0, 10, 40):
    m, med = rt(w)
    print(f"  {w:3d} launches queued in front: copy + synchronize mean {m:7.1f} us  median {med:7.1f} us")
