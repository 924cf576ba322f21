"""libfuse calls process_frame from a pool of worker threads (SURVEY.md 8b "Threading"): different frames of different
clips go through the drop-in symbols concurrently.  Every thread must get exactly the single-threaded result."""
import threading

import numpy as np
import pytest

from mlvfs_amd import abi, pipeline, synth

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
