"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by FRAMES: once a clip's two artefacts exist (pixel
map, 8 stripe coefficients) every frame is independent, so ranks process disjoint
frame ranges with no data-path collective.  The only exchange is once per clip:

  row-sharded first-frame stripes histogram
    1. every rank counts the accepted add_pixel calls of its rows
    2. all_gather(counts)  -> each rank's offset into the glibc rand() stream
       (the reference consumes two rand() per accepted call in raster order)
    3. every rank bins its rows with its slice of the stream
    4. all_reduce(SUM) of int32[8][65536] (2 MiB) + int32[8]
    5. every rank solves the same coefficients locally
  Integer adds commute, so the result does not depend on the number of ranks.

The compute callbacks are injected so the same code runs with the HIP kernels
(bench.py, GPU) and with CPU callbacks in the world_size-2 gloo tests.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np
import torch
import torch.distributed as dist


def describe_ranks(device_ids, rehearsal: bool = False) -> dict:
    """What a multi-process run really ran on: `device_ids[r]` is the identity of the card rank r computes
    on -- its PCI bus id, gathered from every rank.  Returns {"ranks_seen", "devices_seen", "device_ids", "shared"} and raises
    when two ranks share a card outside a rehearsal (N ranks on fewer cards measure contention, not scaling)."""
    ids = [str(x) for x in device_ids]
    seen = {}
    for r, d in enumerate(ids):
        seen.setdefault(d, []).append(r)
    shared = {d: rs for d, rs in seen.items() if len(rs) > 1}
    info = {"ranks_seen": len(ids), "devices_seen": len(seen), "device_ids": ids,
            "shared": {d: rs for d, rs in sorted(shared.items())} or None}
    if shared and not rehearsal:
        raise RuntimeError("ranks share a GPU outside rehearsal mode: " +
                           ", ".join(f"{d}: ranks {rs}" for d, rs in sorted(shared.items())))
    return info


def device_identity(bus_id: str, local_rank: int, host: str | None = None) -> str:
    """The identity of a rank's card for describe_ranks: "<hostname>/<PCI bus id>".  Bus ids repeat across the nodes of a job (rank 0 of
    every node sits on the same 0000:xx:00.0), and so does the fallback "cuda:<local rank>" where a bus id cannot be read: without the
    host in front a multi-node run would end with "ranks share a GPU" (ADVICE r4 #3)."""
    import socket
    return f"{host if host is not None else socket.gethostname()}/{bus_id if bus_id else f'cuda:{local_rank}'}"


def gather_device_ids(my_id: str, group=None):
    """all_gather of each rank's device identity (a group of one included: the collective still runs)."""
    if not dist.is_initialized():
        return [my_id]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, my_id, group=group)
    return out


def frame_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of frames for `rank` (frames are the unit of parallelism)."""
    return n_frames * rank // world, n_frames * (rank + 1) // world


def row_range(height: int, rank: int, world: int) -> Tuple[int, int]:
    return height * rank // world, height * (rank + 1) // world


def sharded_stripes_histogram(count_rows: Callable[[int, int], int],
                              hist_rows: Callable[[int, int, int, int], Tuple[torch.Tensor, torch.Tensor]],
                              height: int, device: torch.device, group=None):
    """Steps 1-4 above.

    count_rows(row0, row1) -> accepted calls in those rows
    hist_rows(row0, row1, first_call, n_calls) -> (hist int32[8*65536], num int32[8]) on `device`,
        binned with dither values 2*first_call .. 2*(first_call+n_calls)-1 of the rand()%1024 stream
    Returns (hist, num, total_calls) with hist/num summed over all ranks.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    row0, row1 = row_range(height, rank, world)
    mine = int(count_rows(row0, row1))
    grouped = dist.is_initialized()          # a group of one still takes the collectives (a one-card box can exercise RCCL that way)
    if grouped:
        parts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(parts, torch.tensor([mine], dtype=torch.int64, device=device), group=group)
        counts = [int(p.item()) for p in parts]
    else:
        counts = [mine]
    first = int(sum(counts[:rank]))
    hist, num = hist_rows(row0, row1, first, mine)
    if grouped:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(num, op=dist.ReduceOp.SUM, group=group)
    return hist, num, int(sum(counts))


def solve_coefficients(hist: torch.Tensor, num: torch.Tensor, frame_size: int, init=None):
    """Step 5 (host, libm pow): mlvfs_amd_stripes_solve, stripes.c:207-246."""
    from . import lib
    L = lib.load()
    h = np.ascontiguousarray(hist.detach().cpu().numpy().astype(np.int32).reshape(-1))
    n = np.ascontiguousarray(num.detach().cpu().numpy().astype(np.int32))
    co = np.zeros(8, np.int32) if init is None else np.array(init, np.int32)
    needed = L.mlvfs_amd_stripes_solve(lib.ptr(h), lib.ptr(n), frame_size, lib.ptr(co))
    return needed, co


_pinned_rand = None


def glibc_rand_slice(first_call: int, n_calls: int, pinned: bool = False) -> np.ndarray:
    """rand()%1024 values 2*first_call .. 2*(first_call+n_calls)-1 of a fresh process (seed 1).

    pinned: generate into a page-locked buffer that is kept between calls (the 56 MB of a 3584x1320 frame then upload at
    link speed instead of through the runtime's pageable staging); the returned array is only valid until the next call."""
    global _pinned_rand
    from . import lib
    L = lib.load()
    n = 2 * n_calls + 2
    if pinned:
        if _pinned_rand is None or _pinned_rand.numel() < n:
            _pinned_rand = torch.empty(n, dtype=torch.int16, pin_memory=True)
        out = _pinned_rand.numpy()[:n].view(np.uint16)
    else:
        out = np.empty(n, np.uint16)                 # filled (and first touched) by the library's generator threads
    out[-2:] = 0
    L.mlvfs_amd_rand_stream(lib.ptr(out), 2 * n_calls, 2 * first_call, 1)
    return out


def gpu_callbacks(stream, frame: torch.Tensor):
    """count_rows / hist_rows backed by the HIP kernels for one device frame."""
    import ctypes as C
    from . import lib
    L, geom = stream.L, stream.geom
    cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def count_rows(r0, r1):
        acc = C.c_int64(0)
        lib.check(L.mlvfs_amd_stripes_count_dev(C.byref(geom), C.c_void_p(frame.data_ptr()), r0, r1, C.byref(acc), cur))
        return acc.value

    def hist_rows(r0, r1, first, n):
        # the dither values of this shard's calls, generated on the device (the host stream would be 5.5 ms + its upload)
        rnd = torch.empty(2 * n + 16, dtype=torch.int16, device=frame.device)
        rnd[2 * n:] = 0
        lib.check(L.mlvfs_amd_rand_stream_dev(C.c_void_p(rnd.data_ptr()), 2 * n, 2 * first, 1, cur), "rand_stream_dev")
        hist = torch.zeros(8 * 65536, dtype=torch.int32, device=frame.device)
        num = torch.zeros(8, dtype=torch.int32, device=frame.device)
        acc = C.c_int64(0)
        lib.check(L.mlvfs_amd_stripes_count_dev(C.byref(geom), C.c_void_p(frame.data_ptr()), r0, r1, C.byref(acc), cur))
        lib.check(L.mlvfs_amd_stripes_hist_dev(C.byref(geom), C.c_void_p(frame.data_ptr()), r0, r1,
                                               C.c_void_p(rnd.data_ptr()), 2 * n, C.c_void_p(hist.data_ptr()),
                                               C.c_void_p(num.data_ptr()), cur))
        torch.cuda.current_stream().synchronize()
        return hist, num

    return count_rows, hist_rows
