"""bench.py's command line: `--gpus N` starts its own ranks (BASELINE.json configs[4]).

CPU part: the argument / environment logic that runs before anything touches the GPU.
GPU part: (a) on any box, a two-rank REHEARSAL of the whole multi-rank control flow (gloo, both ranks on
device 0: the one-GPU box has one card) started by `python bench.py --gpus 2` itself; (b) where two devices
exist, the row-sharded first-frame stripes histogram over RCCL (backend "nccl") with world size 2 against the
oracle's coefficients (stripes.c:143-248)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_gpus_must_agree_with_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in r.stderr and r.stdout == ""


def test_parent_spawns_ranks_and_relays_their_failure():
    """Without a GPU the ranks cannot run: the parent must still be the one that started them (torchrun's own
    failure report names bench.py), print no result line and return a failure code -- never a silent n_gpus: 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--frames-per-step", "2",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    assert "bench.py" in r.stderr


@pytest.mark.gpu
def test_bench_gpus2_starts_two_ranks_rehearsal(gpu):
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["full_size"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MLVFS_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-per-step", "4",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "frames x2"
    # the row-sharded histogram over two ranks gives the single-process coefficients (= the reference's)
    assert res["config"]["stripe_coeffs"] == golden["B_cs5_badpix_stripes_coeffs"]
    assert res["parity"]["ok"] is True and res["parity"]["hashes"] == res["parity"]["reference"]
    assert res["roofline"]["frac"] and res["value"] > 0
    # the run describes what it ran on: two ranks, ONE card (both on device 0 -- allowed in a rehearsal only), each rank's own rate
    c = res["config"]
    assert c["ranks_seen"] == 2 and c["devices_seen"] == 1 and len(c["device_ids"]) == 2 and c["device_ids"][0] == c["device_ids"][1]
    assert c["ranks_sharing_a_device"] == {c["device_ids"][0]: [0, 1]}
    assert len(c["per_rank_kernel_fps"]) == 2 and all(v > 0 for v in c["per_rank_kernel_fps"])
    sp = c["first_frame_split_ms"]
    assert abs(sp["local"] + sp["collectives_and_sharded_histogram"] - c["first_frame_ms"]) < 0.02


@pytest.mark.gpu
def test_bench_two_ranks_on_one_card_is_refused_outside_a_rehearsal(gpu):
    """Two ranks, one visible card, no rehearsal switch: the run must not produce a number."""
    if gpu.mlvfs_amd_device_count() >= 2:
        pytest.skip("two cards visible: the ranks get one each")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MLVFS_BENCH_REHEARSAL")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--frames-per-step", "2",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout


@pytest.mark.gpu
def test_bench_gpus4_rehearsal_weak_and_strong(gpu):
    """Four ranks on the one card (gloo; the box allows six processes on its GPU, this pytest process is one of them, so the
    8-rank run of BASELINE.json configs[4] is rehearsed with 4 here and with 8 CPU-only ranks in tests/test_dist.py): the frame
    ranges, the row-sharded histogram over four shards, the broadcast of the pixel map, the strong-scaling pass."""
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["full_size"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MLVFS_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "2", "--warmup", "1", "--frames-per-step", "6", "--preheat-ms", "20",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert res["n_gpus"] == 4 and res["config"]["parallelism"] == "frames x4" and res["scaling"] == "weak"
    assert res["config"]["stripe_coeffs"] == golden["B_cs5_badpix_stripes_coeffs"] and res["parity"]["ok"] is True
    st = res["config"]["strong_scaling"]
    assert st["frames_total"] == 12 and st["frames_per_rank"] == [3, 3, 3, 3] and st["fps"] > 0


def _nccl_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    from mlvfs_amd import dist as mdist, synth
    from mlvfs_amd.stream import ClipStream
    w, h = 640, 402
    s = ClipStream(w, h, 14, synth.BLACK, synth.WHITE, device=rank)
    f = synth.normal_frame(w, h)
    frame = torch.from_numpy(f.view(np.int16)).to(f"cuda:{rank}")
    count_rows, hist_rows = mdist.gpu_callbacks(s, frame)
    hist, num, calls = mdist.sharded_stripes_histogram(count_rows, hist_rows, h, torch.device(f"cuda:{rank}"))
    needed, co = mdist.solve_coefficients(hist, num, s.frame_size)
    q.put((rank, int(needed), [int(c) for c in co], int(calls), int(num.sum().item())))
    dist.barrier()
    dist.destroy_process_group()
    s.close()


@pytest.mark.gpu
def test_row_sharded_stripes_over_rccl_world2(gpu, oracle):
    if gpu.mlvfs_amd_device_count() < 2:
        pytest.skip("needs two HIP devices (the multi-GPU node); the control flow is covered by the rehearsal test "
                    "above and by tests/test_dist.py over gloo")
    import torch.multiprocessing as mp
    from mlvfs_amd import synth
    w, h = 640, 402
    want_needed, want_co, hist, num = oracle.stripes_compute(synth.normal_frame(w, h), synth.BLACK, synth.WHITE, want_hist=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, needed, co, calls, nsum in res:
        assert needed == want_needed and co == [int(c) for c in want_co]
        assert calls == int(num.sum()) == nsum


@pytest.mark.gpu
def test_row_sharded_stripes_over_rccl_world1(gpu, oracle):
    """What a one-GPU box can show of the RCCL path: the same worker as above as the only rank of an "nccl" process group (the
    communicator is built, the all_gather / all_reduce of the sharded histogram go through RCCL), against the oracle."""
    import torch.multiprocessing as mp
    from mlvfs_amd import synth
    w, h = 640, 402
    want_needed, want_co, hist, num = oracle.stripes_compute(synth.normal_frame(w, h), synth.BLACK, synth.WHITE, want_hist=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    p = ctx.Process(target=_nccl_worker, args=(0, 1, port, q))
    p.start()
    rank, needed, co, calls, nsum = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert needed == want_needed and co == [int(c) for c in want_co]
    assert calls == int(num.sum()) == nsum


@pytest.mark.gpu
def test_bench_single_rank_in_an_rccl_group(gpu):
    """bench.py's N = 1 run as the only rank of an "nccl" process group (MLVFS_BENCH_GROUP1=1): the broadcasts of the pixel map and of
    frame 0, the sharded histogram's collectives, the barriers and the max over ranks all go through RCCL on this one card, and the
    output still equals the reference's."""
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["full_size"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MLVFS_BENCH_GROUP1="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--frames-per-step", "6", "--preheat-ms", "20",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert res["n_gpus"] == 1 and res["parity"]["ok"] is True
    assert res["config"]["stripe_coeffs"] == golden["B_cs5_badpix_stripes_coeffs"]
    assert "RCCL" in res["config"]["collective"] and res["config"]["strong_scaling"]["frames_per_rank"] == [12]
