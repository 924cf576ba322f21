"""libfuse calls process_frame from a pool of worker threads (SURVEY.md 8b "Threading"): different frames of different
clips go through the drop-in symbols concurrently.  Every thread must get exactly the single-threaded result."""
import ctypes as C
import threading

import numpy as np
import pytest

from mlvfs_amd import abi, lib, pipeline, synth

pytestmark = pytest.mark.gpu
BLACK, WHITE = synth.BLACK, synth.WHITE


def test_process_frame_from_many_threads(gpu, oracle):
    w, h, nthreads, frames_per_thread = 256, 130, 8, 3
    opts = [pipeline.MlvfsOptions(chroma_smooth=cs, fix_bad_pixels=bad, fix_stripes=st)
            for cs, bad, st in ((5, 1, 1), (2, 0, 0), (3, 2, 1), (0, 1, 1), (5, 0, 0), (5, 1, 0), (3, 0, 1), (2, 2, 1))]
    frames = [[synth.normal_frame(w, h, seed=10 + t, frame=k) for k in range(frames_per_thread)] for t in range(nthreads)]
    # Expected results, clip by clip.  fileGuid = 0 makes every frame detect its own bad pixels (cs.c:233: a zero guid
    # never matches the map cache), like the oracle's per-frame detection.  The stripe coefficients of each clip are
    # computed beforehand and registered under the clip's name: stripes_compute_correction draws from the process-global
    # rand() stream, which concurrent clips would interleave (in the reference just the same).
    want = []
    for t in range(nthreads):
        o = opts[t]
        corr = None
        if o.fix_stripes:
            img0, corr = oracle.process_frame(synth.pack_bits(frames[t][0]), w, h, BLACK, WHITE, o.chroma_smooth, o.fix_bad_pixels, 1)
            node = gpu.stripes_new_correction(f"thread{t}.MLV".encode())
            node.contents.correction_needed = int(corr[0])
            for i in range(8):
                node.contents.coeffficients[i] = int(corr[1][i])
        want.append([oracle.process_frame(synth.pack_bits(f), w, h, BLACK, WHITE, o.chroma_smooth, o.fix_bad_pixels,
                                          o.fix_stripes, correction=corr)[0] for f in frames[t]])
    got = [[None] * frames_per_thread for _ in range(nthreads)]
    errors = []
    start = threading.Barrier(nthreads)

    def worker(t):
        try:
            start.wait()
            for k, f in enumerate(frames[t]):
                fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE, guid=0)
                got[t][k] = pipeline.process_frame(synth.pack_bits(f), fh, opts[t], mlv_filename=f"thread{t}.MLV")
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(nthreads):
        for k in range(frames_per_thread):
            assert np.array_equal(got[t][k], want[t][k]), f"thread {t} frame {k}: {(got[t][k] != want[t][k]).sum()} px differ"
    gpu.stripes_free_corrections()


def test_concurrent_stripes_analyses_share_the_applications_rand_stream(gpu, oracle):
    """Several clips' first frames at once: each takes its dither from the application's libc generator (in bulk: clip.cpp,
    runtime.cpp take_app_state / put_app_state).  Which clip gets which stretch of the stream depends on who comes first -- in the
    reference too -- but no stretch may be handed out twice: afterwards the stream stands where the same analyses one after the
    other leave it, and each clip's coefficients are those of SOME order's stretch."""
    import ctypes as C
    w, h, n = 416, 264, 6
    libc = C.CDLL(None)
    frames = [synth.normal_frame(w, h, seed=40 + t) for t in range(n)]
    libc.srand(9)
    for f in frames:
        oracle.stripes_compute(f, BLACK, WHITE, reseed=False)
    want_next = [libc.rand() for _ in range(8)]
    libc.srand(9)
    corrs = [gpu.stripes_new_correction(f"conc{t}.MLV".encode()) for t in range(n)]
    start, errors = threading.Barrier(n), []

    def worker(t):
        try:
            fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
            start.wait()
            gpu.stripes_compute_correction(C.byref(fh), corrs[t], lib.ptr(frames[t]), 0, frames[t].size)
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert [libc.rand() for _ in range(8)] == want_next
    gpu.stripes_free_corrections()


def test_dual_iso_from_many_threads(gpu, oracle):
    """The 20-bit tables are process-global caches (like the reference's function statics, but locked)."""
    w, h, nthreads = 136, 72, 6
    f = synth.dual_iso_frame(w, h)
    variants = [(0, 1, 1, 0), (1, 1, 1, 0), (0, 0, 1, 0), (1, 1, 0, 5), (0, 1, 1, 5), (1, 0, 0, 0)]
    gpu.mlvfs_amd_dualiso_reset()
    want = [oracle.cr2hdr20(f, BLACK, WHITE, *v, reset=(i == 0))[1] for i, v in enumerate(variants)]
    got, errors = [None] * nthreads, []
    start = threading.Barrier(nthreads)

    def worker(t):
        try:
            start.wait()
            for _ in range(3):
                fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
                fh_opt = pipeline.MlvfsOptions(dual_iso=2, hdr_interpolation_method=variants[t][0], hdr_no_fullres=1 - variants[t][1],
                                               hdr_no_alias_map=1 - variants[t][2], chroma_smooth=variants[t][3])
                got[t] = pipeline.process_frame(synth.pack_bits(f), fh, fh_opt)
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(nthreads):
        assert np.array_equal(got[t], want[t]), f"variant {variants[t]}: {(got[t] != want[t]).sum()} px differ"


def test_one_clip_shared_by_many_threads(gpu, oracle, tmp_path, monkeypatch):
    """libfuse's workers serve different frames of ONE clip at the same time: same fileGuid, hence one cached bad-pixel map
    (cs.c:233-239) and one focus-pixel map shared by every thread.  Each thread must get its own frame's single-threaded
    result: the repair of one frame must not see another frame's values (every thread repairs into its own patch buffer),
    and frames of the same clip with another crop (panPos) or in dual-ISO mode use their own derivation of the same map."""
    import ctypes as C
    from mlvfs_amd import lib
    w, h, nthreads, rounds = 416, 264, 12, 6
    guid, camera = 0x5EED0001, 0x80000331
    rng = np.random.default_rng(11)
    pts = [(int(rng.integers(0, w)), int(rng.integers(0, h))) for _ in range(900)]
    pts += [(60, 30), (62, 30), (61, 30), (60, 32), (60, 30)]                       # dependent + duplicate entries
    monkeypatch.chdir(tmp_path)
    (tmp_path / f"{camera:x}_{w}x{h}.fpm").write_text("".join(f"{x} \t {y}\n" for x, y in pts))
    gpu.free_focus_pixel_maps()
    frames = [synth.normal_frame(w, h, seed=40, frame=t, hot=300, cold=300) for t in range(nthreads)]
    pans = [(0, 0), (16, 2)]

    def headers(pan):
        fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE, guid=guid, pan=pan, camera=camera)
        fh.rawi_hdr.raw_info.width, fh.rawi_hdr.raw_info.height = w, h
        return fh

    # the map comes from the first frame served (frame 0, pan 0) after the focus repair, as in process_frame's order
    crop0 = ((pans[0][0] + 7) & ~7, pans[0][1] & ~1)
    f0 = oracle.apply_focus_pixels(frames[0], BLACK, np.array(pts, np.int32), crop0, 0)
    pixels = oracle.detect_bad_pixels(f0, BLACK, 0, crop0)
    assert len(pixels) > 300
    first = frames[0].copy()
    fh = headers(pans[0])
    gpu.fix_focus_pixels(C.byref(fh), lib.ptr(first), 0)
    gpu.fix_bad_pixels(C.byref(fh), lib.ptr(first), 0, 0)
    assert np.array_equal(first, oracle.apply_bad_pixels(f0, BLACK, pixels, crop0))

    def expected(t):
        pan = pans[t % 2]
        crop = ((pan[0] + 7) & ~7, pan[1] & ~1)
        di = int(t % 3 == 2)
        img = oracle.apply_focus_pixels(frames[t], BLACK, np.array(pts, np.int32), crop, di)
        return oracle.apply_bad_pixels(img, BLACK, pixels, crop, di)

    want = [expected(t) for t in range(nthreads)]
    got, errors = [None] * nthreads, []
    start = threading.Barrier(nthreads)

    def worker(t):
        try:
            fh = headers(pans[t % 2])
            di = int(t % 3 == 2)
            start.wait()
            for _ in range(rounds):
                img = frames[t].copy()
                gpu.fix_focus_pixels(C.byref(fh), lib.ptr(img), di)
                gpu.fix_bad_pixels(C.byref(fh), lib.ptr(img), 0, di)
                if got[t] is not None and not np.array_equal(got[t], img):
                    errors.append((t, "result changed between rounds"))
                got[t] = img
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(nthreads):
        assert np.array_equal(got[t], want[t]), f"thread {t}: {(got[t] != want[t]).sum()} px differ"
    gpu.free_focus_pixel_maps()


def test_retired_worker_threads_give_their_buffers_back(gpu):
    """libfuse creates and retires worker threads; what a retired thread held on the GPU (stream, staging buffers, the fused
    kernel's ticket counters) must be released with it: free device memory does not shrink from generation to generation."""
    import ctypes as C
    import torch
    from mlvfs_amd import lib
    w, h = 1024, 512
    f = synth.normal_frame(w, h)
    fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)

    def worker():
        img = f.copy()
        gpu.chroma_smooth(C.byref(fh), lib.ptr(img), 5)

    def generation(n=8):
        ths = [threading.Thread(target=worker) for _ in range(n)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    generation()
    free1 = generation()
    for _ in range(6):
        free2 = generation()
    # 48 retired threads held 2 x 1 MiB of staging each (plus stream and counters); what the HIP runtime itself keeps per
    # destroyed stream (about 0.2 MiB here) is not ours to free
    assert free1 - free2 < (24 << 20), f"{(free1 - free2) >> 20} MiB of device memory lost over 48 retired threads"


def test_host_buffer_pool(gpu):
    """mlvfs_amd_host_alloc / _free: a freed buffer comes back for the next request of its size (a host may allocate one per frame),
    other sizes get other buffers, foreign pointers are ignored, and the buffers work with the host entry point."""
    gpu.mlvfs_amd_init(0)
    a = gpu.mlvfs_amd_host_alloc(9461760)
    b = gpu.mlvfs_amd_host_alloc(9461760)
    assert a and b and a != b
    gpu.mlvfs_amd_host_free(a)
    c = gpu.mlvfs_amd_host_alloc(9461760 - 100)            # same 64 KiB class
    assert c == a
    d = gpu.mlvfs_amd_host_alloc(1 << 20)
    assert d and d not in (a, b)
    gpu.mlvfs_amd_host_free(C.c_void_p(12345))             # not ours: ignored
    gpu.mlvfs_amd_host_free(None)
    view = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_uint16)), shape=(1000,))
    view[:] = 7
    assert int(view.sum()) == 7000
    for p in (b, c, d):
        gpu.mlvfs_amd_host_free(p)
    again = [gpu.mlvfs_amd_host_alloc(9461760) for _ in range(2)]
    assert set(again) == {a, b}
    for p in again:
        gpu.mlvfs_amd_host_free(p)


def test_drop_in_threads_spread_over_all_visible_gpus(gpu, oracle):
    """host threads that never called mlvfs_amd_init are bound round-robin to the visible GPUs (the
    frame-parallel multi-GPU mode of the drop-in path: runtime.cpp thread_ctx).  Sixteen workers, each through process_frame's
    sequence with the frame bracket, must all deliver the oracle's frame whichever card served them (on the one-GPU box: all on
    device 0; on the 8-GPU node of BASELINE.json configs[4]: two workers per card)."""
    import threading
    w, h = 416, 264
    f = synth.normal_frame(w, h, seed=77, hot=40, cold=40)
    want = oracle.chroma_smooth(oracle.fix_bad_pixels(f, BLACK, 0, 0), BLACK, 5)
    packed = synth.pack_bits(f)
    got, errs = {}, []

    def worker(i):
        try:
            fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
            fh.file_hdr.fileGuid = 0                                   # every frame detects its own bad pixels: no shared state
            for rep in range(3):
                gpu.mlvfs_amd_frame_begin()
                img = pipeline.get_image_data(fh, packed).reshape(h, w)
                gpu.fix_bad_pixels(C.byref(fh), lib.ptr(img), 0, 0)
                gpu.chroma_smooth(C.byref(fh), lib.ptr(img), 5)
                assert gpu.mlvfs_amd_frame_end() == 0
                got[i] = img
        except Exception as e:                                         # noqa: BLE001 - reported below
            errs.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(16)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for i in range(16):
        assert np.array_equal(got[i], want), i
