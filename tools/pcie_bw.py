#!/usr/bin/env python3
"""Host link check for the PCIe-inclusive numbers of DESIGN.md 7: pinned H2D alone, D2H alone, and both at once on two
streams (what mlvfs_amd_process_frames_host's three-stream pipeline can at best reach)."""
import time, torch
n = 512 << 20
h_in = torch.empty(n, dtype=torch.uint8, pin_memory=True); h_out = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_a = torch.empty(n, dtype=torch.uint8, device="cuda"); d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(up, down, reps=6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1): d_a.copy_(h_in, non_blocking=True)
        if down:
            with torch.cuda.stream(s2): h_out.copy_(d_b, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return reps * n / dt / 1e9
run(True, True, 1)
print(f"H2D alone {run(True, False):6.1f} GB/s   D2H alone {run(False, True):6.1f} GB/s   both at once {run(True, True):6.1f} GB/s each")
