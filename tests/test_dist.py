"""Multi-GPU host logic on CPU: world_size-2 gloo processes run mlvfs_amd.dist with CPU
callbacks (the oracle's row-shard histogram) in place of the HIP kernels.

Checks SURVEY.md 8e: the row-sharded first-frame stripes histogram (all_gather of the
accepted-call counts -> rand() offsets, all_reduce of int32[8][65536]) gives every rank
the coefficients of the single-process reference, and frames shard without overlap."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlvfs_amd import dist as mdist, synth
    from oracle.bindings import Oracle           # CPU stand-in for the kernels (test only)
    o = Oracle()
    f = synth.normal_frame(w, h)
    black, white = synth.BLACK, synth.WHITE

    def count_rows(r0, r1):
        return o.stripes_hist_rows(f, r0, r1, black, white)

    def hist_rows(r0, r1, first, n):
        rnd = mdist.glibc_rand_slice(first, n)
        _, hist, num = o.stripes_hist_rows(f, r0, r1, black, white, rnd)
        return torch.from_numpy(hist), torch.from_numpy(num)

    hist, num, calls = mdist.sharded_stripes_histogram(count_rows, hist_rows, h, torch.device("cpu"))
    needed, co = mdist.solve_coefficients(hist, num, w * h * 14 // 8)
    lo, hi = mdist.frame_range(1000, rank, world)
    q.put((rank, needed, [int(c) for c in co], calls, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])                  # 8: the rank count of BASELINE.json configs[4] (one 8-GPU node)
def test_row_sharded_stripes_over_gloo(oracle, world):
    from mlvfs_amd import synth
    w, h = 256, 130
    want_needed, want_co, hist, num = oracle.stripes_compute(synth.normal_frame(w, h), synth.BLACK, synth.WHITE, want_hist=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = []
    for rank, needed, co, calls, lo, hi in res:
        assert needed == want_needed and co == [int(c) for c in want_co]
        assert calls == int(num.sum())
        covered += list(range(lo, hi))
    assert covered == list(range(1000))                      # frames: disjoint, complete


def test_ranges():
    from mlvfs_amd import dist as mdist
    for n, world in ((1000, 8), (7, 3), (1, 4)):
        got = [mdist.frame_range(n, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    assert mdist.row_range(1320, 3, 8) == (495, 660)


def test_device_identity_tells_two_hosts_with_the_same_bus_id_apart():
    """rank 0 of every node reports the same PCI bus id (and the same fallback where none can be read)."""
    from mlvfs_amd import dist as mdist
    a = mdist.device_identity("0000:05:00.0", 0, host="node-a")
    b = mdist.device_identity("0000:05:00.0", 0, host="node-b")
    assert a != b and a.endswith("/0000:05:00.0")
    info = mdist.describe_ranks([a, b])
    assert info["devices_seen"] == 2 and info["shared"] is None
    fa, fb = mdist.device_identity("", 0, host="node-a"), mdist.device_identity("", 0, host="node-b")
    assert fa == "node-a/cuda:0" and mdist.describe_ranks([fa, fb])["devices_seen"] == 2
    with pytest.raises(RuntimeError):                             # the same card of the same host twice is still refused
        mdist.describe_ranks([a, mdist.device_identity("0000:05:00.0", 1, host="node-a")])


def test_describe_ranks_counts_devices_and_refuses_shared_cards():
    """bench.py gathers every rank's PCI bus id; N ranks on fewer than N cards is an error outside the rehearsal mode."""
    from mlvfs_amd import dist as mdist
    ids = [f"0000:{b:02x}:00.0" for b in (0x0c, 0x22, 0x38, 0x5c, 0x9f, 0xaf, 0xbf, 0xdf)]
    info = mdist.describe_ranks(ids)
    assert info["ranks_seen"] == 8 and info["devices_seen"] == 8 and info["shared"] is None and info["device_ids"] == ids
    with pytest.raises(RuntimeError, match="share a GPU"):
        mdist.describe_ranks([ids[0], ids[1], ids[0]])
    reh = mdist.describe_ranks([ids[0], ids[0]], rehearsal=True)
    assert reh["ranks_seen"] == 2 and reh["devices_seen"] == 1 and reh["shared"] == {ids[0]: [0, 1]}
    assert mdist.gather_device_ids("x") == ["x"]                      # no process group: the rank itself


def _gather_ids_worker(rank, world, port, q):
    import torch.distributed as dist
    from mlvfs_amd import dist as mdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        q.put((rank, mdist.gather_device_ids(f"card-{rank // 2}")))
    finally:
        dist.destroy_process_group()


def test_device_ids_are_gathered_from_every_rank_gloo_world3():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    ps = [ctx.Process(target=_gather_ids_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert all(got[r] == ["card-0", "card-0", "card-1"] for r in range(3))
    from mlvfs_amd import dist as mdist
    with pytest.raises(RuntimeError):
        mdist.describe_ranks(got[0])
