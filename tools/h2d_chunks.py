import time, torch
for chunk_mb in (4.5, 18, 72):
    n = int(chunk_mb * (1 << 20)); k = max(1, int(288 / chunk_mb))
    hs = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for _ in range(min(k, 16))]
    ds = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(min(k, 16))]
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    for nstreams in (1, 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(k):
            with torch.cuda.stream(s[i % nstreams]): ds[i % len(ds)].copy_(hs[i % len(hs)], non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"pinned H2D, chunks of {chunk_mb} MB on {nstreams} stream(s): {k * n / dt / 1e9:6.1f} GB/s")
import numpy as np
a = np.empty(288 << 20, np.uint8); d = torch.empty(288 << 20, dtype=torch.uint8, device="cuda")
t = torch.from_numpy(a)
d.copy_(t); torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(t); torch.cuda.synchronize(); print(f"pageable H2D 288 MB: {a.size / (time.perf_counter() - t0) / 1e9:6.1f} GB/s")
