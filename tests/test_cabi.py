"""The C-ABI library loads and exports every symbol include/mlvfs_amd.h declares; the
host-only entry points work without a GPU (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from mlvfs_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mlvfs_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\(", text))
    return {n for n in names if n not in {"defined", "sizeof"}}


def test_every_declared_symbol_is_exported(amd):
    declared = declared_functions() - {"get_raw2ev", "get_ev2raw"}      # imported (weak) from the caller
    out = subprocess.run(["nm", "-D", "--defined-only", lib.SO_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert declared <= exported, sorted(declared - exported)
    assert set(lib.DROPIN_SYMBOLS + lib.DEVICE_SYMBOLS) == declared
    und = subprocess.run(["nm", "-D", "--undefined-only", lib.SO_PATH], capture_output=True, text=True).stdout
    weak = {l.split()[-1] for l in und.splitlines() if " w " in l}
    assert {"get_raw2ev", "get_ev2raw"} <= weak


def test_no_oracle_or_reference_linkage(amd):
    """The product must not route through the checkers."""
    deps = subprocess.run(["ldd", lib.SO_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in deps and "libmlvfs_ref" not in deps
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mlvfs_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle-", ""), os.path.join(dirpath, f)


def test_host_selftest(amd):
    assert amd.mlvfs_amd_selftest_host() == 0


def test_size_helpers(amd):
    from mlvfs_amd import abi
    fh = abi.make_frame_headers(3584, 1320)
    assert amd.dng_get_header_size() == 65536                        # dng.c:797-800
    assert amd.dng_get_image_size(C.byref(fh)) == 3584 * 1320 * 2
    assert amd.dng_get_size(C.byref(fh)) == 65536 + 3584 * 1320 * 2


def test_rand_stream_is_glibc(amd):
    libc = C.CDLL(None)
    libc.srand(1)
    want = np.array([libc.rand() % 1024 for _ in range(5000)], np.uint16)
    got = np.zeros(5000, np.uint16)
    amd.mlvfs_amd_rand_stream(lib.ptr(got), 5000, 0, 1)
    assert np.array_equal(got, want)
    part = np.zeros(100, np.uint16)
    amd.mlvfs_amd_rand_stream(lib.ptr(part), 100, 3210, 1)
    assert np.array_equal(part, want[3210:3310])


def test_stripes_solve_matches_oracle(amd, oracle):
    from mlvfs_amd import synth
    f = synth.normal_frame(256, 130)
    needed, co, hist, num = oracle.stripes_compute(f, synth.BLACK, synth.WHITE, want_hist=True)
    got = np.zeros(8, np.int32)
    h = np.ascontiguousarray(hist.reshape(-1))
    assert amd.mlvfs_amd_stripes_solve(lib.ptr(h), lib.ptr(num), 256 * 130 * 14 // 8, lib.ptr(got)) == needed
    assert list(got) == list(co)


def test_histogram_helpers(amd):
    data = np.full(70000, 100, np.uint16)
    data[:3000] = 50
    h = amd.hist_create(1000)
    amd.hist_add(h, lib.ptr(data), data.size, 0)
    # 16-bit counters wrap (histogram.h:30): bin 100 holds 67000 % 65536 = 1464, the running sum
    # never exceeds count/2 and the reference's loop falls through to 0 (histogram.c:74)
    assert amd.hist_median(h) == 0
    small = np.array([5, 7, 7, 9, 1000, 2000], np.uint16)
    h2 = amd.hist_create(1000)
    amd.hist_add(h2, lib.ptr(small), small.size, 0)
    assert amd.hist_median(h2) == 9                                  # first bin whose running sum exceeds count/2
    amd.hist_destroy(h2)
    amd.hist_destroy(h)


def test_stripes_correction_list(amd):
    a = amd.stripes_new_correction(b"/x/a.MLV")
    b = amd.stripes_new_correction(b"/x/b.MLV")
    assert a and b and amd.stripes_get_correction(b"/x/b.MLV").contents.mlv_filename == b"/x/b.MLV"
    assert not amd.stripes_get_correction(b"/x/c.MLV")
    assert a.contents.correction_needed == 0 and list(a.contents.coeffficients) == [0] * 8
    amd.stripes_free_corrections()
    assert not amd.stripes_get_correction(b"/x/a.MLV")
