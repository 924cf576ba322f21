"""Frame bracket (mlvfs_amd_frame_begin / mlvfs_amd_frame_end, what integration/mlvfs_amd_wrap.c makes of process_frame's
mlvfs_load_chunks / mlvfs_close_chunks): second half of this file.
MLVFS_AMD_RESIDENT=1: the drop-in symbols keep the frame they just handed back on the device, and the next symbol called on the
same host buffer (process_frame's order, main.c:942-997) works on that copy instead of uploading the frame again.  Results must be
what they are without the variable -- the library reads it once per process, so the resident runs happen in a child process:
this file re-runs the drop-in / thread tests there, and the tests below (child only) check the mode itself."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from mlvfs_amd import abi, lib, pipeline, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLACK, WHITE = synth.BLACK, synth.WHITE
RESIDENT = os.environ.get("MLVFS_AMD_RESIDENT") == "1"
WRAPPED_CHILD = os.environ.get("MLVFS_AMD_PIPELINE_WRAPPED") == "1"


@pytest.mark.skipif(RESIDENT or WRAPPED_CHILD, reason="this is the child")
def test_dropin_suite_in_resident_mode(gpu):
    env = dict(os.environ, MLVFS_AMD_RESIDENT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_resident.py", "tests/test_gpu_threads.py",
                        "tests/test_gpu_parity.py", "tests/test_gpu_c_host.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.skipif(not RESIDENT, reason="needs MLVFS_AMD_RESIDENT=1 (run by test_dropin_suite_in_resident_mode)")
def test_resident_frames_equal_the_oracle_and_notice_host_changes(gpu, oracle):
    w, h = 416, 264
    f = synth.normal_frame(w, h, hot=100, cold=100)
    fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
    # process_frame's order on ONE buffer: unpack -> bad pixels -> chroma smooth -> stripes apply
    img = pipeline.get_image_data(fh, synth.pack_bits(f)).reshape(h, w)
    assert np.array_equal(img, f)
    gpu.fix_bad_pixels(C.byref(fh), lib.ptr(img), 0, 0)
    want = oracle.fix_bad_pixels(f, BLACK, 0, 0)
    assert np.array_equal(img, want)
    gpu.chroma_smooth(C.byref(fh), lib.ptr(img), 5)
    want = oracle.chroma_smooth(want, BLACK, 5)
    assert np.array_equal(img, want)
    # the caller rewrites the buffer (every pixel): the next call must work on what is in host memory now
    img[:] = synth.normal_frame(w, h, seed=5)
    want = oracle.chroma_smooth(img, BLACK, 3)
    gpu.chroma_smooth(C.byref(fh), lib.ptr(img), 3)
    assert np.array_equal(img, want)
    # another buffer with the same content is not the resident one either
    other = img.copy()
    want = oracle.chroma_smooth(other, BLACK, 2)
    gpu.chroma_smooth(C.byref(fh), lib.ptr(other), 2)
    assert np.array_equal(other, want)
    # stripes on the resident copy, window arguments as main.c passes them
    corr = gpu.stripes_new_correction(b"resident.MLV")
    corr.contents.correction_needed = 1
    co = [65536, 65536, 65354, 65738, 65241, 65868, 65450, 65640]
    for k2 in range(8):
        corr.contents.coeffficients[k2] = co[k2]
    want = oracle.stripes_apply(other, BLACK, WHITE, 1, np.array(co, np.int32))
    gpu.stripes_apply_correction(C.byref(fh), corr, lib.ptr(other), 0, other.size)
    assert np.array_equal(other, want)
    gpu.stripes_free_corrections()


# ---------------------------------------------------------------------------------------------- frame bracket
CO = [65536, 65536, 65354, 65738, 65241, 65868, 65450, 65640]


@pytest.mark.skipif(RESIDENT or WRAPPED_CHILD, reason="this is the child")
def test_process_frame_mirror_suites_with_the_bracket(gpu):
    """The parity and thread tests that go through pipeline.process_frame, re-run with the mirror's mlvfs_load_chunks /
    mlvfs_close_chunks doing what the wrap shim makes of them (a child process: the switch is read at import)."""
    env = dict(os.environ, MLVFS_AMD_PIPELINE_WRAPPED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_parity.py", "tests/test_gpu_threads.py",
                        "-k", "process_frame or hdr_preview or pattern_noise or thread"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def _unpack_into(gpu, fh, f, img):
    packed = np.ascontiguousarray(synth.pack_bits(f), np.uint16)
    assert gpu.dng_get_image_data(C.byref(fh), lib.ptr(packed), lib.ptr(img), 0, img.nbytes) == img.nbytes


def _stripes(gpu, name):
    corr = gpu.stripes_new_correction(name)
    corr.contents.correction_needed = 1
    for k2 in range(8):
        corr.contents.coeffficients[k2] = CO[k2]
    return corr


def test_bracketed_stages_leave_the_host_buffer_alone_until_the_bracket_ends(gpu, oracle):
    w, h = 416, 264
    f = synth.normal_frame(w, h, hot=100, cold=100)
    fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
    img = np.full((h, w), 0xABCD, np.uint16)
    assert gpu.mlvfs_amd_frame_begin() == 0
    _unpack_into(gpu, fh, f, img)
    gpu.fix_focus_pixels(C.byref(fh), lib.ptr(img), 0)
    gpu.fix_bad_pixels(C.byref(fh), lib.ptr(img), 0, 0)
    gpu.chroma_smooth(C.byref(fh), lib.ptr(img), 5)
    corr = _stripes(gpu, b"deferred.MLV")
    gpu.stripes_apply_correction(C.byref(fh), corr, lib.ptr(img), 0, img.size)
    assert (img == 0xABCD).all(), "a stage wrote the host buffer"
    assert gpu.mlvfs_amd_frame_end() == 0
    want = oracle.stripes_apply(oracle.chroma_smooth(oracle.fix_bad_pixels(f, BLACK, 0, 0), BLACK, 5), BLACK, WHITE, 1, np.array(CO, np.int32))
    assert np.array_equal(img, want)
    assert gpu.mlvfs_amd_frame_end() == 0 and gpu.mlvfs_amd_frame_sync(lib.ptr(img)) == 0 and np.array_equal(img, want)   # nothing pending: no-ops
    # mlvfs_amd_frame_sync inside a bracket: the frame now, the bracket stays open
    img2 = np.full((h, w), 0xABCD, np.uint16)
    gpu.mlvfs_amd_frame_begin()
    _unpack_into(gpu, fh, f, img2)
    gpu.fix_bad_pixels(C.byref(fh), lib.ptr(img2), 0, 0)
    gpu.chroma_smooth(C.byref(fh), lib.ptr(img2), 5)
    assert (img2 == 0xABCD).all()
    assert gpu.mlvfs_amd_frame_sync(lib.ptr(img2)) == 0
    assert np.array_equal(img2, oracle.chroma_smooth(oracle.fix_bad_pixels(f, BLACK, 0, 0), BLACK, 5))
    gpu.stripes_apply_correction(C.byref(fh), corr, lib.ptr(img2), 0, img2.size)      # takes the fetched copy up, result deferred again
    assert gpu.mlvfs_amd_frame_end() == 0 and np.array_equal(img2, want)
    gpu.stripes_free_corrections()


def test_outside_a_bracket_every_call_completes_at_once_gif_sequence(gpu, oracle):
    """gif_get_data (gif.c:82-221) never passes through mlvfs_load_chunks / mlvfs_close_chunks around its pixel work: per preview
    frame it calls mlv_get_frame_headers -- which does bracket its header walk with that pair (main.c:434,555) -- and then
    get_image_data -> dng_get_image_data, and reads the buffer in the next statement (gif.c:164-190).  The same sequence here, on a
    thread that has served bracketed frames before and after."""
    w, h = 416, 264
    fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
    frames = [synth.normal_frame(w, h, frame=k) for k in range(3)]
    warm = np.empty((h, w), np.uint16)
    gpu.mlvfs_amd_frame_begin()                                       # a process_frame of this worker, earlier
    _unpack_into(gpu, fh, frames[0], warm)
    gpu.chroma_smooth(C.byref(fh), lib.ptr(warm), 2)
    assert gpu.mlvfs_amd_frame_end() == 0
    assert np.array_equal(warm, oracle.chroma_smooth(frames[0], BLACK, 2))
    image_data = np.full((h, w), 0xABCD, np.uint16)                   # gif.c:141: ONE buffer for all ten frames
    for f in frames:
        gpu.mlvfs_amd_frame_begin(); gpu.mlvfs_amd_frame_end()        # mlv_get_frame_headers (gif.c:159): an empty bracket
        _unpack_into(gpu, fh, f, image_data)                          # gif.c:164
        assert np.array_equal(image_data, f), "gif_get_data would have rendered stale pixels"
    # ... and with the one bracket call that can be left open: the failure branch of process_frame (main.c:924-928) returns
    # without mlvfs_close_chunks, but then the shim has not armed anything either (integration/mlvfs_amd_wrap.c); an explicit
    # begin without end followed by the header walk's pair is closed by that pair
    gpu.mlvfs_amd_frame_begin()
    gpu.mlvfs_amd_frame_begin(); gpu.mlvfs_amd_frame_end()
    _unpack_into(gpu, fh, frames[1], image_data)
    assert np.array_equal(image_data, frames[1])


def test_bracket_is_per_thread_and_disabled_by_environment(gpu, oracle):
    import threading
    w, h = 416, 264
    f = synth.normal_frame(w, h)
    fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
    gpu.mlvfs_amd_frame_begin()                                       # this thread is inside a bracket ...
    seen = {}

    def other():                                                      # ... another one is not: its unpack completes at once
        img = np.full((h, w), 0xABCD, np.uint16)
        fh2 = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
        _unpack_into(gpu, fh2, f, img)
        seen["img"] = img.copy()

    t = threading.Thread(target=other); t.start(); t.join()
    assert np.array_equal(seen["img"], f)
    mine = np.full((h, w), 0xABCD, np.uint16)
    _unpack_into(gpu, fh, f, mine)
    assert (mine == 0xABCD).all()
    assert gpu.mlvfs_amd_frame_end() == 0 and np.array_equal(mine, f)
    # MLVFS_AMD_DEFER=0: the bracket calls do nothing (a child process: read once)
    code = ("import ctypes as C, numpy as np\nfrom mlvfs_amd import abi, lib, synth\nL = lib.load()\n"
            "f = synth.normal_frame(416, 264)\nfh = abi.make_frame_headers(416, 264, black=synth.BLACK, white=synth.WHITE)\n"
            "img = np.full((264, 416), 0xABCD, np.uint16)\np = np.ascontiguousarray(synth.pack_bits(f), np.uint16)\n"
            "L.mlvfs_amd_frame_begin()\nL.dng_get_image_data(C.byref(fh), lib.ptr(p), lib.ptr(img), 0, img.nbytes)\n"
            "assert np.array_equal(img, f)\nassert L.mlvfs_amd_frame_end() == 0\nprint('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, MLVFS_AMD_DEFER="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


def test_bracketed_result_is_fetched_by_symbols_that_read_the_host_frame(gpu, oracle):
    """A stage called out of process_frame's order and the symbols that read the host frame themselves first bring the host buffer
    up to date.  Pattern noise is a stage of the sequence (right behind the unpack, main.c:946-949): inside a bracket it works on the
    device copy and its result stays there like any other stage's."""
    w, h = 416, 264
    f = synth.normal_frame(w, h)
    fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
    img = np.full((h, w), 0xABCD, np.uint16)
    gpu.mlvfs_amd_frame_begin()
    _unpack_into(gpu, fh, f, img)
    gpu.fix_pattern_noise(lib.ptr(img), w, h, WHITE, 0)                  # main.c:946-949: right after the unpack
    assert (img == 0xABCD).all()                                         # nothing has crossed the link for it
    want = oracle.fix_pattern_noise(f, WHITE)
    assert gpu.mlvfs_amd_frame_sync(lib.ptr(img)) == 0
    assert np.array_equal(img, want)
    gpu.chroma_smooth(C.byref(fh), lib.ptr(img), 3)                      # takes up the device copy, result deferred
    gpu.chroma_smooth(C.byref(fh), lib.ptr(img), 2)                      # not "the next stage": fetched, uploaded again
    assert gpu.mlvfs_amd_frame_end() == 0
    assert np.array_equal(img, oracle.chroma_smooth(oracle.chroma_smooth(want, BLACK, 3), BLACK, 2))
    # deflicker (main.c:895-906) reads the frame through hist_add right after the unpack
    img2 = np.full((h, w), 0xABCD, np.uint16)
    gpu.mlvfs_amd_frame_begin()
    _unpack_into(gpu, fh, f, img2)
    assert (img2 == 0xABCD).all()
    # -- inside a bracket the samples are counted where the pixels are, the host frame stays as it was; the counts are the host loop's
    hist = gpu.hist_create(1 << 14)
    gpu.hist_add(hist, C.c_void_p(img2.ctypes.data + 2), (img2.size - 1) // 2, 1)
    assert (img2 == 0xABCD).all()
    ref_hist = gpu.hist_create(1 << 14)
    flat = np.ascontiguousarray(f.reshape(-1))
    gpu.hist_add(ref_hist, C.c_void_p(flat.ctypes.data + 2), (flat.size - 1) // 2, 1)      # a buffer the library knows nothing about: the host loop
    class Hist(C.Structure):                                          # histogram.h / include/mlvfs_amd.h
        _fields_ = [("white", C.c_uint16), ("count", C.c_uint32), ("data", C.POINTER(C.c_uint16))]
    a, b = C.cast(hist, C.POINTER(Hist)).contents, C.cast(ref_hist, C.POINTER(Hist)).contents
    assert a.count == b.count == ((img2.size - 1) // 2) // 2
    assert np.array_equal(np.ctypeslib.as_array(a.data, ((1 << 14) + 1,)), np.ctypeslib.as_array(b.data, ((1 << 14) + 1,)))
    assert gpu.hist_median(hist) == gpu.hist_median(ref_hist)
    gpu.hist_destroy(hist); gpu.hist_destroy(ref_hist)
    # a window that is not the whole frame is written at once (dng.c:815-826 arithmetic), bracket or not
    part = pipeline.get_image_data(fh, synth.pack_bits(f)[512 * 14 // 16:], offset=1024, max_size=4096)      # from the first pixel's word on
    assert np.array_equal(part, f.reshape(-1)[512:512 + 2048])
    assert gpu.mlvfs_amd_frame_end() == 0
    # an LJ92 clip: get_image_data decodes on the host and writes the frame itself (main.c:617-681), no unpack call at all;
    # the stages that follow upload it, keep their result on the GPU and the bracket's end delivers it
    img3 = f.copy()
    gpu.mlvfs_amd_frame_begin()
    gpu.fix_bad_pixels(C.byref(fh), lib.ptr(img3), 0, 0)
    gpu.chroma_smooth(C.byref(fh), lib.ptr(img3), 5)
    assert np.array_equal(img3, f)
    assert gpu.mlvfs_amd_frame_end() == 0
    assert np.array_equal(img3, oracle.chroma_smooth(oracle.fix_bad_pixels(f, BLACK, 0, 0), BLACK, 5))
    gpu.free_focus_pixel_maps()


def test_bracketed_process_frame_mirror_and_threads(gpu, oracle, monkeypatch):
    """pipeline.process_frame (the mirror of main.c's, chunk calls included) with the wrap shim's behaviour, from several threads."""
    import threading
    monkeypatch.setattr(pipeline, "WRAPPED", True)
    w, h = 416, 264
    frames = [synth.normal_frame(w, h, frame=k, hot=50, cold=50) for k in range(4)]
    opt = pipeline.MlvfsOptions(chroma_smooth=5, fix_bad_pixels=1, fix_stripes=0)
    want = [oracle.chroma_smooth(oracle.fix_bad_pixels(f, BLACK, 0, 0), BLACK, 5) for f in frames]
    got = {}

    def worker(i):
        for rep in range(3):
            fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
            fh.file_hdr.fileGuid = 0x7000 + i
            got[i] = pipeline.process_frame(synth.pack_bits(frames[i]), fh, opt, mlv_filename=f"t{i}.MLV")

    stats0 = (C.c_longlong * 2)()
    gpu.mlvfs_amd_dropin_stats(stats0)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(4):
        assert np.array_equal(got[i], want[i]), i
    # every thread's first frame detects its clip's bad pixels (that needs the pixels: the recorded unpack runs early); the two
    # that follow are recorded from the unpack to the chroma smoothing and run as one launch of the fused kernel at the bracket's end
    stats = (C.c_longlong * 2)()
    gpu.mlvfs_amd_dropin_stats(stats)
    assert stats[0] - stats0[0] == 8 and stats[1] - stats0[1] == 4, (list(stats0), list(stats))
    gpu.free_focus_pixel_maps()
