"""MLVFS_AMD_RESIDENT=1: the drop-in symbols keep the frame they just handed back on the device, and the next symbol called on the
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


@pytest.mark.skipif(RESIDENT, reason="this is the child")
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
