#!/usr/bin/env python3
"""bench.py -- throughput of the MLVFS per-frame hot path on MI355X.

Metric (BASELINE.json): Mpix/s for 3584x1320 14-bit unpack + cs5x5 + stripes
(+ bad-pixel repair: configs[2]) with the packed stream already resident in HBM.

  step      = one fused pass (k_pixfix + k_frame) over one batch of --frames-per-step
              synthetic frames of the resident stream
  value     = all ranks' pixels / wall time of exactly --steps steps (barrier +
              synchronize on both sides, max over ranks)
  roofline  = dominant kernel k_frame<5,packed,patch,stripes>: algorithmic bytes per launch
              (3.75 B/px x pixels per launch) / mean launch duration measured with HIP
              events on the launching stream inside the timed region; peak 8 TB/s
  parity    = after the timed region frames 0 and 1 of rank 0's output are hashed and compared
              with the hashes of the REFERENCE's output for the same seeded frames
              (tests/golden/golden.json); a mismatch fails the run
  cpu_baseline = the reference's own code (oracle/_ref, kind "reference") or the C
              restatement (kind "port") on the host cores, bounded sample, rank 0 @ N=1:
              one thread, and the best of a sweep of thread counts (one frame per thread)
  extra     = same process, outside the timed headline, rank 0 @ N=1 only: configs[1] (cs2x2),
              other footage kinds through the same kernel, configs[3] (full dual-ISO), and the
              PCIe-inclusive rates (batch API, drop-in symbols).  Never part of `value`.

Multi-GPU: one rank per GPU.  `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the
environment) starts its own ranks: the parent -- before it imports torch or touches HIP --
launches `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process,
relays rank 0's JSON line and exits with the child's return code.  Under an external
torchrun (WORLD_SIZE set) it is a rank.  Frames are sharded over ranks (weak scaling: every
rank owns its own stream of the same size); the only collective is the once-per-clip
row-sharded stripes histogram all-reduce over RCCL, outside the timed region.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, BPP = 3584, 1320, 14
BYTES_PER_PX = 14 / 8 + 2            # packed in + 16-bit out
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s


# ------------------------------------------------------------------------------------------ ranks
def spawn_ranks(args) -> int:
    """Parent of a multi-GPU run.  Nothing here may touch the GPU: the ranks are CHILD processes."""
    # --standalone: torchrun opens its own rendezvous on a free port (a port picked here could be taken before torchrun binds it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--preheat-ms", str(args.preheat_ms),
           "--frames-per-step", str(args.frames_per_step), "--cs", str(args.cs)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_extras:
        cmd.append("--no-extras")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in res.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    elif res.returncode == 0:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        return 1
    return res.returncode


# ------------------------------------------------------------------------------------------ CPU baseline
def host_cores() -> int:
    """Cores this process may really use: scheduler affinity, capped by the cgroup's CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(budget_s: float = 28.0):
    """Reference CPU path, steady state (map + coefficients known), one frame per thread the way
    libfuse parallelises process_frame: 1 thread, then a sweep of thread counts; best one reported."""
    from concurrent.futures import ThreadPoolExecutor
    from mlvfs_amd import synth
    from oracle import bindings               # checker / baseline only
    distinct = [synth.normal_frame(W, H, seed=1, frame=k) for k in range(2)]
    packed = [synth.pack14(f).astype("<u2") for f in distinct]
    packed = [np.concatenate([p, np.zeros(4, "<u2")]) for p in packed]
    npx = W * H
    if bindings.have_ref():
        kind, impl = "reference", bindings.Reference()
        co = np.zeros(8, np.int32)
        needed = C.c_int(0)
        out0 = np.zeros((H, W), np.uint16)
        C.CDLL(None).srand(1)
        impl.L.ref_process_frame.argtypes = None
        def run(p, out, compute):
            impl.L.ref_process_frame(p.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), W, H, BPP,
                                     synth.BLACK, synth.WHITE, 5, 1, 1, C.c_uint64(0x1234), co.ctypes.data_as(C.c_void_p),
                                     C.byref(needed), compute)
        run(packed[0], out0, 1)                # first frame of the clip: detect + stripes compute (not timed)
    else:
        kind, impl = "port", bindings.Oracle()
        img0, corr = impl.process_frame(packed[0], W, H, synth.BLACK, synth.WHITE, cs=5, bad_pix=0, stripes=1)
        pixels = impl.detect_bad_pixels(impl.unpack(packed[0], W, H).reshape(H, W), synth.BLACK, 0)
        def run(p, out, compute):
            img = impl.unpack(p, W, H).reshape(H, W)
            img = impl.apply_bad_pixels(img, synth.BLACK, pixels)
            img = impl.chroma_smooth(img, synth.BLACK, 5)
            out[:] = impl.stripes_apply(img, synth.BLACK, synth.WHITE, *corr)

    def timed(threads, frames_per_thread):
        outs = [np.zeros((H, W), np.uint16) for _ in range(threads)]
        def worker(t):
            for k in range(frames_per_thread):
                run(packed[(t + k) % len(packed)], outs[t], 0)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(worker, range(threads)))
        return threads * frames_per_thread / (time.perf_counter() - t0)

    REPS = 5                                        # the median of >= 5 repetitions per sweep point
    t_start = time.perf_counter()
    timed(1, 1)                                     # page in the library and the output buffers (not part of the sample)
    cores = host_cores()
    sweep, spread, reps_done = {}, {}, {}
    for t in [1] + sorted({max(1, cores // 4), max(1, cores // 2), cores} - {1}):
        # a sweep point is only started if its repetitions fit the rest of the budget at the per-thread rate seen so far
        per_rep = (1.0 / sweep[1]) * 1.15 if sweep else 0.5
        if sweep and (time.perf_counter() - t_start) + REPS * per_rep > budget_s:
            break
        r = sorted(timed(t, 1) for _ in range(REPS))
        sweep[t], spread[t], reps_done[t] = round(r[len(r) // 2], 3), [round(r[0], 3), round(r[-1], 3)], len(r)
    fps1 = sweep[1]
    best_t = max(sweep, key=lambda k: sweep[k])
    n_frames = sum(t * reps_done[t] for t in sweep)
    return {"value": round(sweep[best_t] * npx / 1e6, 2), "unit": "Mpix/s", "cores": best_t, "kind": kind,
            "fps": sweep[best_t],
            "one_thread": {"value": round(fps1 * npx / 1e6, 2), "unit": "Mpix/s", "fps": round(fps1, 3), "cores": 1},
            "threads_sweep_fps": {str(k): v for k, v in sweep.items()},
            "threads_sweep_min_max_fps": {str(k): v for k, v in spread.items()}, "repetitions": REPS, "host_cores_usable": cores,
            "os_cpu_count": os.cpu_count(), "seconds": round(time.perf_counter() - t_start, 1),
            "sample": f"{n_frames} frames of {W}x{H} (2 distinct synthetic frames), steady state unpack+badpix+cs5x5+stripes, "
                      f"one frame per thread and repetition, gcc -O2; per sweep point ({list(sweep)} threads) the MEDIAN of "
                      f"{REPS} repetitions; `value` is the best sweep point"}


# ------------------------------------------------------------------------------------------ extras
def _preheat(fn, ms=60.0):
    """Run fn() back to back for `ms` milliseconds (clock settling, see main)."""
    import torch
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()


def _kernel_timed(L, lib, fn, launches=1):
    """Run fn() (which launches k_frame `launches` times) under the HIP-event timer; mean ms per launch."""
    import torch
    lib.check(L.mlvfs_amd_timer_begin(launches), "timer_begin")
    fn()
    torch.cuda.synchronize()
    ms = np.zeros(launches, np.float32)
    n = L.mlvfs_amd_timer_end(lib.ptr(ms), launches)
    return float(ms[:n].mean()) if n else float("nan")


def extra_cs2x2(golden, fnv1a, F=400):
    """configs[1]: 3584x1320 unpack + cs2x2 (no pixel map, no stripes), resident stream; output checked against the
    reference's hashes for frames 0 and 1.  F frames per launch: the headline's 400 since round 5 (rounds 1-4: 50 -- a launch of
    0.27 ms loses a tenth to its tail and the gaps around it; the same kernel at 50 / 400 frames per launch: 5.4 / 4.8 us per frame)."""
    import torch
    from mlvfs_amd import lib, synth
    from mlvfs_amd.stream import ClipStream, to_numpy_u16
    s = ClipStream(W, H, BPP, synth.BLACK, synth.WHITE, device=0)
    base = s.synth_packed(8, seed=1)
    packed = s.alloc_packed(F)
    for i in range(0, F, 8):
        packed[i:i + 8] = base[:min(8, F - i)]
    out = s.alloc_out(F)
    run = lambda: s.process(packed, out, cs=2, fix_pixels=False, stripes=False)
    _preheat(run)
    ms = [_kernel_timed(s.L, lib, run) for _ in range(5)]
    t0 = time.perf_counter()
    for _ in range(4):
        run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 4
    got = [fnv1a(to_numpy_u16(out[k])) for k in range(2)]
    want = [golden["B_cs2_frame0"], golden["B_cs2_frame1"]]
    s.close()
    kms = float(np.median(ms))
    return {"workload": "configs[1]: 3584x1320 14-bit unpack + cs2x2, stream resident in HBM", "frames_per_launch": F,
            "kernel": "k_frame_s (cs2x2 without a pixel map: one wave per column of the frame, no barriers; MLVFS_AMD_KF_S=0: k_frame<2>)",
            "fps": round(F / wall, 1), "Mpix/s": round(F * W * H / wall / 1e6, 1), "kernel_us_per_frame": round(kms * 1e3 / F, 2),
            "hbm_frac": round(F * W * H * BYTES_PER_PX / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "parity": {"hashes": got, "reference": want, "ok": got == want}}


def extra_footage(F=400):
    """The headline pass (cs5x5 + pixel map + stripes) on footage that is not the benchmark's gradient: an underexposed
    scene (pixels at or below black) and hard colour edges (several EV of colour balance from patch to patch).  F frames per launch:
    the headline's 400 since the end of round 5 (until then 48: the same kernels take 10.9 / 12.1 us per frame at 48 and 8.2-8.5 /
    10.4 at 400 frames per launch -- short launches lose a fifth to their tails on this footage)."""
    import torch
    from mlvfs_amd import lib, synth
    from mlvfs_amd.stream import ClipStream
    res = {}
    for kind, gen, seed0 in (("low_light", synth.low_light_frame, 21), ("colour_cast", synth.colour_cast_frame, 11)):
        s = ClipStream(W, H, BPP, synth.BLACK, synth.WHITE, device=0)
        frames = [gen(W, H, seed=seed0 + i) for i in range(4)]
        base = s.upload_packed([synth.pack_bits(f) for f in frames])
        packed = s.alloc_packed(F)
        for i in range(0, F, 4):
            packed[i:i + 4] = base[:min(4, F - i)]
        out = s.alloc_out(F)
        frame0 = s.unpack(packed[:1])
        s.detect_bad_pixels(frame0[0], 0)
        s.set_stripes(1, [65536, 65536, 65354, 65738, 65241, 65868, 65450, 65640])
        run = lambda: s.process(packed, out, cs=5, fix_pixels=True, stripes=True)
        _preheat(run)
        kms = float(np.median([_kernel_timed(s.L, lib, run) for _ in range(5)]))
        res[kind] = {"frames_per_launch": F, "kernel_us_per_frame": round(kms * 1e3 / F, 2), "fps_kernel": round(F / kms * 1e3, 0),
                     "hbm_frac": round(F * W * H * BYTES_PER_PX / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "share_at_or_below_black": round(float(np.mean([(f <= synth.BLACK).mean() for f in frames])), 4),
                     "t16_layout": s.get_t16_layout()}
        s.close()
    return res


def extra_dualiso(golden, fnv1a, batch=8, reps=4, threads=3):
    """configs[3]: full dual-ISO conversion (AMaZE + edge-directed interpolation, full-res, alias map) of 3584x1320 frames resident
    in HBM: (a) one conversion at a time (latency), (b) mlvfs_amd_cr2hdr20_batch_dev with `batch` frames per submission from one
    host thread, (c) the same from `threads` and `threads + 1` host threads, each with its own stream and batch (the analysis kernels of
    one batch overlap the AMaZE tiles of another).  Every converted frame is hashed against the reference's output."""
    import torch
    from mlvfs_amd import lib, synth
    L = lib.load()
    f = synth.dual_iso_frame(W, H)
    src = torch.from_numpy(f.view(np.int16)).cuda()
    geom = lib.Geom(W, H, 14, synth.BLACK, synth.WHITE, 0, 0)
    L.mlvfs_amd_dualiso_reset()
    res = {"workload": "configs[3]: 3584x1320 cr2hdr20 amaze-edge, fullres, alias map, no chroma smooth; frames resident in HBM"}
    want = golden["dualiso_3584x1320_i0_f1_a1_cs0"]

    # (a) single conversions, one after the other
    one = src.clone()
    L.mlvfs_amd_cr2hdr20_dev(C.byref(geom), C.c_void_p(one.data_ptr()), 0, 1, 1, 0, None)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        one.copy_(src); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = L.mlvfs_amd_cr2hdr20_dev(C.byref(geom), C.c_void_p(one.data_ptr()), 0, 1, 1, 0, None)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        if ok != 1:
            raise RuntimeError("cr2hdr20_dev did not convert the frame")
    res["single"] = {"ms_per_frame": round(float(np.median(ts)) * 1e3, 3), "conversions_per_s": round(1 / float(np.median(ts)), 1)}
    hashes = {fnv1a(one.cpu().numpy().view(np.uint16))}

    # (b), (c) batches
    def batch_worker(i, bufs, streams, n_batches, t_begin, t_end, oks, go):
        lib.check(L.mlvfs_amd_init(0))
        r = np.zeros(batch, np.int32)
        sp = C.c_void_p(streams[i].cuda_stream)
        def run():
            with torch.cuda.stream(streams[i]):                # fresh frames: a device copy on the same stream
                bufs[i].copy_(src.unsqueeze(0).expand(batch, -1, -1))
            rc = L.mlvfs_amd_cr2hdr20_batch_dev(C.byref(geom), C.c_void_p(bufs[i].data_ptr()), W * H * 2, batch, 0, 1, 1, 0, lib.ptr(r), sp)
            streams[i].synchronize()
            return int(rc == 0) * int(r.sum())
        run()                                                  # the thread's work buffers
        go.wait()
        t_begin[i] = time.perf_counter()
        oks[i] = sum(run() for _ in range(n_batches))
        t_end[i] = time.perf_counter()

    for T in (1, threads, threads + 1):
        streams = [torch.cuda.Stream() for _ in range(T)]
        bufs = [torch.empty((batch, H, W), dtype=torch.int16, device="cuda") for _ in range(T)]
        t_begin, t_end, oks, go = [0.0] * T, [0.0] * T, [0] * T, threading.Barrier(T)
        nb = reps if T == 1 else 2 * reps                      # (several threads: a longer run, the start-up is uneven)
        ths = [threading.Thread(target=batch_worker, args=(i, bufs, streams, nb, t_begin, t_end, oks, go)) for i in range(T)]
        for t in ths: t.start()
        for t in ths: t.join()
        torch.cuda.synchronize()
        n = T * batch * nb
        if sum(oks) != n:
            raise RuntimeError("cr2hdr20_batch_dev did not convert every frame")
        dt = max(t_end) - min(t_begin)
        for bb in bufs:
            for k in (0, batch - 1):
                hashes.add(fnv1a(bb[k].cpu().numpy().view(np.uint16)))
        res[f"batch_{batch}_threads_{T}"] = {"ms_per_frame": round(dt / n * 1e3, 3), "conversions_per_s": round(n / dt, 1),
                                             "Mpix/s": round(n * W * H / dt / 1e6, 1)}
        del bufs
        torch.cuda.empty_cache()
    # (c') the reference's default interpolator (mean23, `--dual-iso 2` without `--amaze-edge`), batches of 8 from one thread
    try:
        bb = torch.empty((batch, H, W), dtype=torch.int16, device="cuda")
        r8 = np.zeros(batch, np.int32)
        def run_m23():
            bb.copy_(src.unsqueeze(0).expand(batch, -1, -1))
            rc = L.mlvfs_amd_cr2hdr20_batch_dev(C.byref(geom), C.c_void_p(bb.data_ptr()), W * H * 2, batch, 1, 1, 1, 0, lib.ptr(r8), None)
            torch.cuda.synchronize()
            return rc == 0 and int(r8.sum()) == batch
        ok = run_m23()
        t0 = time.perf_counter()
        for _ in range(reps): ok = run_m23() and ok
        dt = time.perf_counter() - t0
        h23 = fnv1a(bb[0].cpu().numpy().view(np.uint16))
        res["batch_%d_mean23" % batch] = {"conversions_per_s": round(batch * reps / dt, 1), "ok": bool(ok),
                                          "parity": {"hash": h23, "reference": golden.get("dualiso_3584x1320_i1_f1_a1_cs0"), "equal": h23 == golden.get("dualiso_3584x1320_i1_f1_a1_cs0")}}
        del bb
    except Exception as e:
        res["batch_%d_mean23" % batch] = {"failed": str(e)[:200]}
    # (d) one conversion per call through the drop-in symbol, host memory in and out: the reference's own process_frame text
    # (oracle/_ref/ref_host_amd_wrap, main.c's functions linked against the library) with --dual-iso 2 --amaze-edge from 8 threads
    try:
        import shutil, tempfile
        from mlvfs_amd import mlvfile
        exe = os.path.join(ROOT, "oracle", "_ref", "ref_host_amd_wrap")
        if os.path.exists(exe):
            d = tempfile.mkdtemp(prefix="mlvfs_amd_bench_di_")
            try:
                nfr = 8
                pl = [np.ascontiguousarray(synth.pack_bits(synth.dual_iso_frame(W, H, frame=k)), "<u2").tobytes() for k in range(nfr)]
                mlvfile.write_clip(os.path.join(d, "B02-0001.MLV"), pl, W, H, chunks=2)
                vp = ["/B02-0001.MLV/B02-0001_%06d.dng" % k for k in range(nfr)]
                r = subprocess.run([exe, d, "-", "dual_iso=2", "hdr_interp=0", "threads=8", "loops=8", "--", *vp], capture_output=True, text=True, timeout=300)
                line = [ln for ln in r.stderr.splitlines() if ln.startswith("{")]
                res["reference_process_frame_text_host_8_threads"] = {"fps": json.loads(line[-1])["fps"] if line else f"failed rc {r.returncode}",
                                                                      "includes": "file reads, unpack, upload, one conversion per call, download"}
            finally:
                shutil.rmtree(d, ignore_errors=True)
        else:
            res["reference_process_frame_text_host_8_threads"] = {"skipped": "oracle/_ref/ref_host_amd_wrap not built"}
    except Exception as e:
        res["reference_process_frame_text_host_8_threads"] = {"failed": str(e)[:200]}
    best = max(v["conversions_per_s"] for k, v in res.items() if k.startswith("batch_") and "_threads_" in k)
    # SURVEY 8(d)'s convention for this path too: packed in + 16-bit out per frame against the HBM peak
    res["hbm_frac_compulsory_bytes"] = round(best * W * H * BYTES_PER_PX / 1e9 / HBM_PEAK_GBS, 5)
    # ... and what the conversion really moves and executes ("report the actual multi-pass bytes separately"): counters of a
    # batch of 8 from separate rocprofv3 --pmc passes (tools/dualiso_traffic.sh -> profiles/r05/di_fused/dualiso_traffic.json), scaled by
    # THIS run's batch rate.  The path is bound by instruction issue in AMaZE and by table gathers behind it, not by HBM.
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r05", "di_fused", "dualiso_traffic.json")))
        rate = res["batch_8_threads_1"]["conversions_per_s"]
        res["roofline"] = {
            "bound": "instruction issue (AMaZE) / L2 gathers (interpolation, blend); HBM for reference",
            "traffic_bytes_per_frame": tj["traffic_bytes_per_frame"], "traffic_over_compulsory": tj["traffic_over_compulsory"],
            "achieved_GBps": round(tj["traffic_bytes_per_frame"] * rate / 1e9, 1),
            "hbm_frac_actual_bytes": round(tj["traffic_bytes_per_frame"] * rate / 1e9 / HBM_PEAK_GBS, 4),
            "valu_wave_insts_per_frame": tj["valu_wave_insts_per_frame"],
            # share of the chip's vector issue slots (1024 SIMDs, one wave-instruction per 4 clocks at 2.35 GHz) the batch uses
            "valu_issue_share": round(tj["valu_wave_insts_per_frame"] * rate * 4 / (1024 * 2.35e9), 3),
            "dominant_kernel": tj["dominant_kernel"], "dominant_kernel_us_per_frame": tj["dominant_kernel_us_per_frame"],
            "kernel_us_per_frame_sum": tj["kernel_us_per_frame_sum"], "amaze_share_of_kernel_sum": tj["amaze_share_of_kernel_sum"],
            "k_amaze_share_of_kernel_sum": tj["k_amaze_share_of_kernel_sum"],
            "source": "profiles/r05/di_fused/dualiso_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE, SQ_INSTS_VALU, --kernel-trace --stats; separate passes): "
                      "not measured in this run, scaled by its batch_8_threads_1 rate"}
    except Exception as e:
        res["roofline"] = {"unavailable": str(e)[:120]}
    res["parity"] = {"hashes": sorted(hashes), "reference": want, "ok": hashes == {want}}
    return res


def extra_pcie(N=128, threads=16):
    """PCIe-inclusive rates, frames start and end in HOST memory: (a) the batch API (pinned buffers, chunks of 8 frames);
    (b) the five drop-in symbols called per frame in process_frame's order (main.c:942-997) from `threads` host threads."""
    import torch
    from mlvfs_amd import abi, lib, synth
    from mlvfs_amd.stream import ClipStream
    s = ClipStream(W, H, BPP, synth.BLACK, synth.WHITE, device=0)
    L = s.L
    base = s.synth_packed(8, seed=1)
    s.analyse_first_frame(base, cs=5, bad_pix=1, stripes=True, rand_mode=1)
    host_in = torch.empty((N, s.packed_stride), dtype=torch.uint8, pin_memory=True)
    host_in.view(N // 8, 8, s.packed_stride)[:] = base.cpu()
    host_out = torch.empty((N, s.out_stride), dtype=torch.uint8, pin_memory=True)
    s.process_host(host_in, host_out, cs=5, fix_pixels=True, stripes=True, chunk=8)
    t0 = time.perf_counter()
    for _ in range(3):
        s.process_host(host_in, host_out, cs=5, fix_pixels=True, stripes=True, chunk=8)
    dt = (time.perf_counter() - t0) / 3
    res = {"batch_api": {"fps": round(N / dt, 0), "Mpix/s": round(N * W * H / dt / 1e6, 0),
                         "pcie_GBs_both_directions": round(N * (s.packed_stride + s.out_stride) / dt / 1e9, 1),
                         "entry": "mlvfs_amd_process_frames_host, pinned buffers, chunks of 8, unpack+badpix+cs5x5+stripes"}}
    want = host_out[:2].clone()
    s.close()

    want1 = synth.fnv1a(want[1].numpy().view(np.uint16))
    # (b) the drop-in symbols, in a child process per mode (the library reads MLVFS_AMD_RESIDENT once per process); "wrap": the
    # frame bracket where process_frame calls mlvfs_load_chunks / mlvfs_close_chunks (integration/mlvfs_amd_wrap.c)
    for mode in ("0", "1", "wrap"):
        env = dict(os.environ, MLVFS_AMD_RESIDENT=mode) if mode != "wrap" else dict(os.environ, DROPIN_BRACKET="1")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_bench.py"), str(threads), "12"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            res["dropin_symbols_" + ("wrap" if mode == "wrap" else f"resident{mode}")] = {"failed": f"rc {r.returncode}"}
            continue
        d = json.loads(line[-1])
        res["dropin_symbols_" + ("wrap" if mode == "wrap" else f"resident{mode}")] = {"fps_1_thread": d["fps_1_threads"], f"fps_{threads}_threads": d[f"fps_{threads}_threads"],
                                                 "equals_batch_api": d["frame1_hash"] == want1,
                                                 "identical_between_threads": d["identical_between_threads"]}
    # (c) the same from a C host with pthreads (what MLVFS is): tools/dropin_bench_c.c, malloc'ed buffers, a memcpy per frame for the
    # file read.  ONE source, which calls the reference's symbols only, linked twice: plainly, and with integration/mlvfs_amd_wrap.c
    # + --wrap of mlvfs_load_chunks / mlvfs_close_chunks (the link-only route: no line of the host changes)
    try:
        exe = os.path.join(ROOT, "build", "dropin_bench_c")
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        so_dir = os.path.join(ROOT, "mlvfs_amd")
        srcs = [os.path.join(ROOT, "tools", "dropin_bench_c.c"), os.path.join(ROOT, "tests", "c_host_chunks.c")]
        link = ["-L", so_dir, "-lmlvfs_amd", "-Wl,-rpath," + so_dir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lm"]
        cc = ["gcc", "-std=gnu99", "-O2", "-pthread", "-I", os.path.join(ROOT, "include")]
        subprocess.run(cc + srcs + ["-o", exe] + link, check=True, capture_output=True, timeout=120)
        subprocess.run(cc + srcs + [os.path.join(ROOT, "integration", "mlvfs_amd_wrap.c"), "-Wl,--wrap=mlvfs_load_chunks",
                                    "-Wl,--wrap=mlvfs_close_chunks", "-o", exe + "_wrap"] + link, check=True, capture_output=True, timeout=120)
        # ... and with the optional second shim (integration/mlvfs_amd_wrap_alloc.c: process_frame's malloc'ed buffers come from the
        # library's page-locked pool, the fused kernel writes the frame itself over the link)
        subprocess.run(cc + srcs + [os.path.join(ROOT, "integration", "mlvfs_amd_wrap.c"), os.path.join(ROOT, "integration", "mlvfs_amd_wrap_alloc.c"),
                                    "-Wl,--wrap=mlvfs_load_chunks", "-Wl,--wrap=mlvfs_close_chunks", "-Wl,--wrap=malloc", "-Wl,--wrap=calloc",
                                    "-Wl,--wrap=realloc", "-Wl,--wrap=free", "-o", exe + "_wrap_alloc"] + link, check=True, capture_output=True, timeout=120)
        frames = []
        for k in range(2):
            fn = os.path.join(ROOT, "build", f"dropin_frame{k}.bin")
            synth.pack14(synth.normal_frame(W, H, seed=1, frame=k)).astype("<u2").tofile(fn)
            frames.append(fn)
        chost = {}
        for mode in ("0", "1", "wrap", "wrap_alloc"):
            key = mode if mode.startswith("wrap") else f"resident{mode}"
            dump = os.path.join(ROOT, "build", "dropin_frame1_out.bin")
            if os.path.exists(dump):
                os.remove(dump)
            # (48 frames per thread after two of warm-up: the steady state of a long-running mount; 24 left thread start-up in the figure)
            r = subprocess.run([exe + "_" + mode if mode.startswith("wrap") else exe, frames[0], frames[1], str(threads), "48", "0"],
                               env=dict(os.environ, MLVFS_AMD_RESIDENT="0" if mode.startswith("wrap") else mode, DROPIN_DUMP=dump),
                               capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stderr.splitlines() if ln.startswith("{") and '"fps_1_threads"' in ln]
            if line:
                d = json.loads(line[-1])
                chost[key] = {"fps_1_thread": d["fps_1_threads"], f"fps_{threads}_threads": d[f"fps_{threads}_threads"],
                              "identical_between_threads": d["identical_between_threads"], "equals_batch_api": os.path.exists(dump) and bool(np.array_equal(np.fromfile(dump, "<u2"), want[1].numpy().view(np.uint16))),
                              "fused_at_fetch": d.get("fused_at_fetch")}
            else:
                chost[key] = {"failed": f"rc {r.returncode}"}
        res["dropin_symbols_c_host"] = chost
    except Exception as e:                                   # no compiler on the box: the Python harness above stands
        res["dropin_symbols_c_host"] = {"skipped": str(e)[:200]}
    # (d) the REFERENCE'S OWN process_frame text (main.c:908-1005 with mlv_get_frame_headers, get_image_data, the index walk and the
    # fopen of every chunk per frame, sliced out of main.c / resource_manager.c by oracle/Makefile) linked against the library:
    # plainly, and with the wrap shim.  File reads (page cache) included.  oracle/_ref/ref_host_* travel as binaries.
    try:
        import shutil
        import tempfile
        from mlvfs_amd import mlvfile
        hosts = {k: os.path.join(ROOT, "oracle", "_ref", "ref_host_" + k) for k in ("amd", "amd_wrap", "amd_wrap_alloc")}
        if all(os.path.exists(h) for h in hosts.values()):
            d = tempfile.mkdtemp(prefix="mlvfs_amd_bench_")
            try:
                nfr = 32
                pl = [synth.pack14(synth.normal_frame(W, H, seed=1, frame=k % 2)).astype("<u2").tobytes() + b"\0" * 4 for k in range(nfr)]
                mlvfile.write_clip(os.path.join(d, "B01-0001.MLV"), pl, W, H, chunks=2)
                vp = ["/B01-0001.MLV/B01-0001_%06d.dng" % k for k in range(nfr)]
                rh = {}
                for k, exe in hosts.items():
                    r = subprocess.run([exe, d, "-", "cs=5", "badpix=1", "stripes=1", f"threads={threads}", "loops=2", "--", *vp],
                                       capture_output=True, text=True, timeout=300, env=dict(os.environ, MLVFS_AMD_RESIDENT="0"))
                    line = [ln for ln in r.stderr.splitlines() if ln.startswith("{")]
                    rh[k] = json.loads(line[-1])["fps"] if line else f"failed rc {r.returncode}"
                res["reference_process_frame_text_host"] = {f"fps_{threads}_threads_plain_link": rh["amd"], f"fps_{threads}_threads_wrap_link": rh["amd_wrap"],
                                                            f"fps_{threads}_threads_wrap_and_alloc_shim_link": rh["amd_wrap_alloc"],
                                                            "includes": "mlv_get_frame_headers + index walk + fopen/fread of a 32-frame two-chunk clip per frame (page cache)"}
            finally:
                shutil.rmtree(d, ignore_errors=True)
        else:
            res["reference_process_frame_text_host"] = {"skipped": "oracle/_ref/ref_host_* not built"}
    except Exception as e:
        res["reference_process_frame_text_host"] = {"failed": str(e)[:200]}
    res["note"] = "resident1: MLVFS_AMD_RESIDENT=1, a stage takes up the device copy the previous stage left for the same host buffer " \
                  "(upload skipped; every call still downloads what it changed before it returns).  wrap: the host is linked with " \
                  "integration/mlvfs_amd_wrap.c and -Wl,--wrap=mlvfs_load_chunks -Wl,--wrap=mlvfs_close_chunks -- no source line of the " \
                  "host changes --: the stages between the two calls are recorded and run as one fused launch, the frame crosses " \
                  "PCIe once each way (INTEGRATION.md section 1); a C host with pthreads: tools/dropin_bench_c.sh"
    return res


# ------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--preheat-ms", type=float, default=200.0, help="untimed run of the hot path before the warm-up steps (clock settling)")
    ap.add_argument("--frames-per-step", type=int, default=400,
                    help="frames per launch of the fused kernel (one step); 400 = 3.3 GB of packed stream in, 3.8 GB of pixels out")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cs", type=int, default=5)
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus is None:
            args.gpus = 1
        if args.gpus > 1:
            sys.exit(spawn_ranks(args))          # children are the ranks; this process never touches the GPU
    else:
        if args.gpus is None:
            args.gpus = int(env_world)
        if args.gpus != int(env_world):
            sys.exit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={env_world}")

    # stdout carries exactly ONE line, the result: everything else that writes to fd 1 while the bench runs (the bad-pixel
    # list that fix_bad_pixels prints like the reference, cs.c:307-311, through C stdio) goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from mlvfs_amd import dist as mdist
    from mlvfs_amd import lib, synth
    from mlvfs_amd.stream import ClipStream, to_numpy_u16

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal on a one-GPU box (never what the driver runs): MLVFS_BENCH_REHEARSAL=1 puts every rank on device 0 and uses
    # gloo, so that the multi-rank control flow of this file can be exercised where only one card exists.
    rehearsal = world > 1 and os.environ.get("MLVFS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    # MLVFS_BENCH_GROUP1=1 (never what the driver runs): the N = 1 run as the only rank of an "nccl" process group, so that every
    # collective of this file -- broadcasts, the sharded histogram's all_gather / all_reduce, barriers, the max over ranks -- goes
    # through RCCL on a box with one card (tests/test_bench_cli.py)
    grouped = world > 1 or os.environ.get("MLVFS_BENCH_GROUP1") == "1"
    if grouped and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if grouped:
        if not rehearsal and torch.cuda.device_count() < world:
            sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPUs visible")
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    dev = torch.device(f"cuda:{local}")
    F, K, Wm = args.frames_per_step, args.steps, args.warmup
    # ---- what this run really runs on: every rank's card by PCI bus id, gathered; two ranks on one card is an error outside the
    # rehearsal mode (mlvfs_amd.dist.describe_ranks)
    _L0 = lib.load()
    _bus = C.create_string_buffer(64)
    my_dev_id = mdist.device_identity(_bus.value.decode() if _L0.mlvfs_amd_device_pci_bus_id(local, _bus, 64) == 0 and _bus.value else "", local)
    try:
        ranks_info = mdist.describe_ranks(mdist.gather_device_ids(my_dev_id) if grouped else [my_dev_id], rehearsal=rehearsal)
    except RuntimeError as e:
        sys.exit(f"bench.py: {e}")

    s = ClipStream(W, H, BPP, synth.BLACK, synth.WHITE, device=local)
    L = s.L
    # the resident stream: K*F frames per rank (every rank owns its own clip: weak scaling);
    # built in HBM from at most 64 distinct synthetic frames per rank
    # one clip of world*K*F frames, rank r owns frames [r*K*F, (r+1)*K*F)
    # Every timed step reads and writes frames of its own (no reuse between steps) as long as that fits: the resident stream is capped
    # at 150 GB or 60 % of the free HBM, whichever is less; beyond that the steps wrap around (a step moves 7 GB at the default size,
    # the chip's last-level cache holds 256 MB: nothing of a slot survives until it comes round again)
    per_frame = s.packed_stride + W * H * 2
    free_b = torch.cuda.mem_get_info(dev)[0]
    slots = max(1, min(K, int(min(150e9, 0.6 * free_b) // (per_frame * F))))
    distinct = min(slots * F, 64)
    base = s.synth_packed(distinct, seed=1, first_frame=rank * K * F)
    packed = s.alloc_packed(slots * F)
    for i in range(0, slots * F, distinct):
        n = min(distinct, slots * F - i)
        packed[i:i + n] = base[:n]
    out = s.alloc_out(slots * F)

    # ---- first frame of the clip (once per clip, not in the timed region) ------------
    # main.c:969-988: the clip's pixel map and stripe coefficients come from frame 0, which
    # lives on rank 0; the other ranks receive the map and the post-chroma-smooth frame, and
    # the stripes histogram of that frame is row-sharded over all ranks
    def first_frame(st):
        """The clip's pixel map and stripe coefficients from its frame 0, on stream object `st`; returns (split in ms, total ms, ...)."""
        torch.cuda.synchronize()
        ff = {}                                        # where the first frame's time goes (ms, wall, each part synchronised)
        def _lap(name, t_from):
            torch.cuda.synchronize()
            ff[name] = round((time.perf_counter() - t_from) * 1e3, 3)
            return time.perf_counter()
        t0 = time.perf_counter()
        tl = t0
        pix = None
        if rank == 0:
            frame0 = st.unpack(packed[:1])
            tl = _lap("unpack", tl)
            pix = st.detect_bad_pixels(frame0[0], 0)
            tl = _lap("detect_bad_pixels", tl)
            st.fix_pixels(frame0)
            tl = _lap("fix_pixels", tl)
            frame0 = st.chroma_smooth(frame0, args.cs) if args.cs else frame0
            tl = _lap("chroma_smooth", tl)
        else:
            frame0 = st.alloc_out(1)
        t_coll = time.perf_counter()
        if grouped:
            n_pix = torch.tensor([len(pix) if rank == 0 else 0], dtype=torch.int64, device=dev)
            dist.broadcast(n_pix, src=0)
            pix_t = torch.from_numpy(pix.copy()).to(dev) if rank == 0 else \
                torch.zeros((int(n_pix.item()), 2), dtype=torch.int32, device=dev)
            dist.broadcast(pix_t, src=0)
            dist.broadcast(frame0.view(torch.uint8), src=0)
            if rank != 0:
                st.set_pixel_map(pix_t.cpu().numpy())
        tl = _lap("broadcasts", t_coll)
        count_rows, hist_rows = mdist.gpu_callbacks(st, frame0[0])
        hist, num, calls = mdist.sharded_stripes_histogram(count_rows, hist_rows, H, dev)
        tl = _lap("stripes_count_rand_hist_allreduce", tl)
        needed, coeffs = mdist.solve_coefficients(hist, num, st.frame_size)
        st.set_stripes(needed, coeffs)
        tl = _lap("stripes_solve", tl)
        total = (time.perf_counter() - t0) * 1e3
        ff["collectives_and_sharded_histogram"] = round(ff.get("broadcasts", 0) + ff["stripes_count_rand_hist_allreduce"], 3)
        ff["local"] = round(total - ff["collectives_and_sharded_histogram"], 3)
        return ff, total, needed, coeffs

    ff, first_frame_ms, needed, coeffs = first_frame(s)
    # the same on a SECOND clip object: what every further clip of a mounted directory costs, without what the process pays once
    # (code objects of the analysis kernels, the black level's output table, the thread's scratch buffers)
    s2 = ClipStream(W, H, BPP, synth.BLACK, synth.WHITE, device=local)
    ff2, first_frame_next_clip_ms, needed2, coeffs2 = first_frame(s2)
    assert int(needed2) == int(needed) and [int(c) for c in coeffs2] == [int(c) for c in coeffs]
    s2.close()

    def step(b):
        b %= slots
        s.process(packed[b * F:(b + 1) * F], out[b * F:(b + 1) * F], cs=args.cs, fix_pixels=True, stripes=True)

    # Steady state: the GPU needs tens of milliseconds of this load before its clocks settle (the same launch is 8 % slower in
    # the first 10 ms than after 50: profiles/r02/README.md), so the hot path runs for --preheat-ms before the W warm-up steps.
    # Untimed, like the warm-up; the K timed steps below are unchanged.
    t_pre = time.perf_counter()
    i = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        for _ in range(8):
            step(i % K)
            i += 1
        torch.cuda.synchronize()
    for i in range(Wm):
        step(i % K)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    lib.check(L.mlvfs_amd_timer_begin(K), "timer_begin")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    dt = time.perf_counter() - t0
    ms = np.zeros(K, np.float32)
    n_timed = L.mlvfs_amd_timer_end(lib.ptr(ms), K)
    # each rank's own rate from its kernel timer (HIP events around its launches): what the slowest-rank wall time hides
    my_kernel_fps = float(K * F / (ms[:n_timed].sum() * 1e-3)) if n_timed else 0.0
    per_rank_fps = [round(my_kernel_fps, 1)]
    if grouped:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        parts = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(parts, torch.tensor([my_kernel_fps], dtype=torch.float64, device=dev))
        per_rank_fps = [round(float(p.item()), 1) for p in parts]

    # ---- the same hot path at 100 frames per launch, the default of rounds 1-3 (round-to-round deltas must not mix kernel
    # changes with the change of the launch size): a few launches, wall clock, beside the headline -- never `value`
    at100 = None
    if F > 100 and not grouped and not args.no_extras:       # (an extra: --no-extras keeps a profiled run to launches of one size)
        n100 = 16
        for i in range(4):
            s.process(packed[(i % 4) * 100:(i % 4 + 1) * 100], out[(i % 4) * 100:(i % 4 + 1) * 100], cs=args.cs, fix_pixels=True, stripes=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n100):
            s.process(packed[(i % 4) * 100:(i % 4 + 1) * 100], out[(i % 4) * 100:(i % 4 + 1) * 100], cs=args.cs, fix_pixels=True, stripes=True)
        torch.cuda.synchronize()
        d100 = time.perf_counter() - t1
        at100 = {"fps": round(n100 * 100 / d100, 1), "Mpix/s": round(n100 * 100 * W * H / d100 / 1e6, 1), "launches": n100}

    # ---- strong scaling, reported beside the weak headline (never `value`): ONE clip of K*F frames -- the N = 1 workload -- split
    # into contiguous frame ranges (mlvfs_amd.dist.frame_range), no data-path collective; time = slowest rank between two barriers
    strong = None
    if grouped:
        a, b = mdist.frame_range(K * F, rank, world)
        def strong_pass():
            for lo in range(a, b, F):
                n = min(lo + F, b) - lo
                at = min(lo % (slots * F), slots * F - n)          # (the resident stream may hold fewer than K*F frames: wraps)
                s.process(packed[at:at + n], out[at:at + n], cs=args.cs, fix_pixels=True, stripes=True)
        strong_pass()
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        strong_pass()
        torch.cuda.synchronize()
        dist.barrier()
        ts = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        strong = {"frames_total": K * F, "frames_per_rank": [mdist.frame_range(K * F, r, world)[1] - mdist.frame_range(K * F, r, world)[0] for r in range(world)],
                  "ms": round(float(ts.item()) * 1e3, 3), "fps": round(K * F / float(ts.item()), 1),
                  "note": "one clip of the N = 1 size split by frame ranges; launches of <= frames_per_step frames; beside the weak headline"}

    # ---- the timed output is checked: frames 0 and 1 of the clip against the REFERENCE's hashes --------------------
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["full_size"]
    parity = None
    if rank == 0 and args.cs == 5 and K * F >= 2:
        got = [synth.fnv1a(to_numpy_u16(out[k])) for k in range(2)]
        want = [golden["B_cs5_badpix_stripes_frame0"], golden["B_cs5_badpix_stripes_frame1"]]
        co_ok = [int(c) for c in coeffs] == golden["B_cs5_badpix_stripes_coeffs"] and int(needed) == golden["B_cs5_badpix_stripes_needed"]
        map_ok = int(len(s.get_pixel_map())) == golden["B_cs5_badpix_stripes_badpix_count"]
        parity = {"checked": "out[0], out[1] after the timed region vs tests/golden/golden.json B_cs5_badpix_stripes_frame0/1 "
                             "(hashes of the reference's output), stripe coefficients, pixel-map size",
                  "hashes": got, "reference": want, "ok": bool(got == want and co_ok and map_ok)}
    # ... and the rest of the timed output against those: the resident stream repeats its `distinct` synthetic frames, so frame i of
    # every launch must come out like frame i % distinct -- the last frames of a launch, of the last step, a seeded sample in between
    # (a launch walks its tile list across 400 frames: the first two alone would not show a fault further down the list)
    rep_ok, rep_n = True, 0
    if slots * F > distinct:
        gsel = torch.Generator().manual_seed(1 + rank)
        sample = {slots * F - 1, F - 1, min(F, slots * F - 1), min(2 * F + 17, slots * F - 1)}
        sample |= set(torch.randint(distinct, slots * F, (28,), generator=gsel).tolist())
        for i in sorted(sample):
            if i >= distinct:
                rep_ok = rep_ok and bool(torch.equal(out[i], out[i % distinct]))
                rep_n += 1
    if parity is not None:
        parity["later_frames"] = {"checked": rep_n, "equal_to_their_first_occurrence": rep_ok}
        parity["checked"] += "; %d later frames of the resident stream against the first occurrence of their content" % rep_n
        parity["ok"] = bool(parity["ok"] and rep_ok)
    ok_flag = torch.tensor([1 if ((parity is None or parity["ok"]) and rep_ok) else 0], dtype=torch.int32, device=dev)
    if grouped:
        dist.all_reduce(ok_flag, op=dist.ReduceOp.MIN)

    npx = W * H
    total_px = world * K * F * npx
    kern_ms = float(ms[:n_timed].mean()) if n_timed else float("nan")
    achieved = F * npx * BYTES_PER_PX / (kern_ms * 1e-3) / 1e9 if n_timed else None
    traffic, valu, traffic_src = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = int(tj.get("k_frame_bytes_per_frame") * F)     # per launch, like `achieved`
            traffic_src = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this " \
                          "command, FETCH doubled per the gfx950 correction; " + str(tj.get("source", "")) + "): not measured in this run"
            # what actually bounds the kernel: vector-ALU issue.  Wave-instructions per frame from the
            # SQ_INSTS_VALU counter, the chip's issue rate for them from tools/valu_rate2.hip, measured time from this run.
            if tj.get("valu_insts_per_frame") and n_timed:
                per_frame_us = kern_ms * 1e3 / F
                flat_us = tj["valu_insts_per_frame"] / (tj["simds"] * tj["valu_issue_per_clk_per_simd"] * tj["clock_ghz"] * 1e3)
                valu = {"wave_insts_per_frame": int(tj["valu_insts_per_frame"]), "measured_us_per_frame": round(per_frame_us, 2),
                        "flat_floor_us_per_frame": round(flat_us, 2), "flat_issue_per_clk_per_simd": tj["valu_issue_per_clk_per_simd"],
                        "source": "profiles/traffic.json (SQ_INSTS_VALU pass)"}
                # class-weighted floor: every instruction of the kernel's common path priced at the measured
                # issue rate of ITS class -- tools/isa_classes.py on the compiler's assembly, rates from tools/valu_rate*.hip
                if tj.get("class_weighted_clk_per_wave_inst"):
                    cw_us = tj["valu_insts_per_frame"] * tj["class_weighted_clk_per_wave_inst"] / (tj["simds"] * tj["clock_ghz"] * 1e3)
                    valu.update({"floor_us_per_frame": round(cw_us, 2), "frac_of_valu_floor": round(cw_us / per_frame_us, 3),
                                 "floor": "class-weighted: " + str(tj.get("class_source", "")), "class_shares": tj.get("class_shares")})
                else:
                    valu.update({"floor_us_per_frame": round(flat_us, 2), "frac_of_valu_floor": round(flat_us / per_frame_us, 3), "floor": "flat"})
        except Exception:
            traffic = None

    two_kernels = args.cs == 5 and os.environ.get("MLVFS_AMD_KF_P", "1") != "0"
    # which kernel does the tiles' first pass: the library's rule (csrc/k_frame_p.hip: frame_p5_takes) -- the streaming form
    # k_frame_p5 takes launches of at least 3.5 tasks (a 62-item column x 60 rows) per wave, k_frame_p the shorter ones
    streaming = False
    if two_kernels and os.environ.get("MLVFS_AMD_KF_P5", "1") != "0":
        cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        tasks = F * ((W // 8 + 61) // 62) * ((H // 2 + 29) // 30)           # (tasks of 60 rows, or of 30 for launches half as long)
        streaming = os.environ.get("MLVFS_AMD_KF_P5") == "2" or tasks * 2 >= cus * 16 * 7
    first_pass = "void mlv::k_frame_p5<false, 1>(mlv::FrameArgs, int, int, int, int, int)" if streaming else "void mlv::k_frame_p<5, true, 1, false>(mlv::FrameArgs)"
    result = {
        "metric": "Mpix/s, 3584x1320 14-bit unpack+badpix+cs5x5+stripes (fused, stream resident in HBM)",
        "value": round(total_px / dt / 1e6, 1),
        "unit": "Mpix/s",
        "fps": round(world * K * F / dt, 1),
        "n_gpus": world, "steps": K, "warmup": Wm, "preheat_ms": args.preheat_ms,
        "ms_per_step": round(dt / K * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16/int32", "data": "synthetic",
        "config": {"workload": "configs[2]: 3584x1320 unpack + cs5x5 + stripes + bad-pix, frame stream resident in HBM",
                   "frames_per_step": F, "value_at_100_frames_per_step": at100, "frames_per_rank": K * F, "resident_frames_per_rank": slots * F, "chroma_smooth": args.cs,
                   "bad_pixels_in_map": int(len(s.get_pixel_map())), "stripe_coeffs": [int(c) for c in coeffs],
                   "parallelism": f"frames x{world}", "first_frame_ms": round(first_frame_ms, 2), "first_frame_split_ms": ff,
                   "first_frame_next_clip_ms": round(first_frame_next_clip_ms, 2), "first_frame_next_clip_split_ms": ff2,
                   "ranks_seen": ranks_info["ranks_seen"], "devices_seen": ranks_info["devices_seen"], "device_ids": ranks_info["device_ids"],
                   "ranks_sharing_a_device": ranks_info["shared"], "per_rank_kernel_fps": per_rank_fps, "strong_scaling": strong,
                   "collective": None if not grouped else ("gloo (rehearsal)" if rehearsal else "RCCL all_gather + all_reduce int32[8][65536], once per clip")},
        "roofline": {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": (first_pass if two_kernels else "void mlv::k_frame<5, true, 1, false>(mlv::FrameArgs)") if args.cs == 5 else f"void mlv::k_frame<{args.cs}, true, 1, false>(mlv::FrameArgs)",
                     "kernel_template_arguments": "k_frame / k_frame_p: METHOD (chroma smoothing 2 / 3 / 5), PACKED (14-bit stream in), VEC (1: width % 16 == 0; 2: width % 16 == 8; 0: any width), SPREAD (dark-clip table layout); k_frame_p5: SPREAD, VEC",
                     "kernel_ms_per_launch": round(kern_ms, 4),
                     "kernel_ms_per_launch_covers": ("one pass = " + ("k_frame_p5 (the packed-once pass as a streaming kernel: a wave per 62-item column of the frame; long launches)"
                                                                     if streaming else "k_frame_p (every tile whose packed medians are certain)") +
                                                     " + the list-mode k_frame that follows it on the stream (the tiles the first kernel listed; none on these frames: its "
                                                     "workgroups end at once): the HIP events bracket both, the rocprofv3 averages of the two add up to this") if two_kernels else "k_frame",
                     "algorithmic_bytes_per_launch": int(F * npx * BYTES_PER_PX)},
        "parity": parity,
    }
    if valu:
        result["valu"] = valu
    all_ok = bool(int(ok_flag.item()))
    # the headline's buffers are not needed any more: the extras allocate their own
    del packed, out, base
    torch.cuda.empty_cache()
    if rank == 0 and world == 1:
        if not args.no_extras:
            extra = {}
            for name, fn in (("configs1_cs2x2", lambda: extra_cs2x2(golden, synth.fnv1a)), ("footage", extra_footage),
                             ("configs3_dualiso", lambda: extra_dualiso(golden, synth.fnv1a)), ("pcie", extra_pcie)):
                t0 = time.perf_counter()
                try:
                    extra[name] = fn()
                except Exception as e:          # an extra is a report, never a reason to lose the headline
                    extra[name] = {"failed": f"{type(e).__name__}: {e}"}
                extra[name]["took_s"] = round(time.perf_counter() - t0, 1)
                torch.cuda.empty_cache()
            result["extra"] = extra
        if not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                result["cpu_baseline"] = {"value": None, "unit": "Mpix/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    sys.stdout.flush()
    C.CDLL(None).fflush(None)
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)
        os.dup2(2, 1)                              # whatever is still buffered in C stdio at exit stays off stdout
    if grouped:
        dist.destroy_process_group()
    s.close()
    if not all_ok:
        sys.stderr.write("bench.py: the timed output differs from the reference's (parity check failed)\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
