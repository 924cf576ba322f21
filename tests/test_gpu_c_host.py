"""The drop-in boundary from a plain C program: tests/c_host.c is compiled with gcc against include/*.h, linked with
libmlvfs_amd.so and the system HIP runtime, and run as its own process (no Python, no torch in it) through
process_frame's call sequence.  Its output must equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

from mlvfs_amd import lib, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BLACK, WHITE = synth.BLACK, synth.WHITE


def build_c_host(exe, variant):
    so_dir = os.path.dirname(lib.SO_PATH)
    srcs = [os.path.join(HERE, "c_host.c"), os.path.join(HERE, "c_host_chunks.c")]
    flags = []
    if "caller" in variant:
        srcs.append(os.path.join(ROOT, "oracle", "ref_luts.c"))
    if "wrap" in variant:                                           # one more object, two linker flags
        srcs.append(os.path.join(ROOT, "integration", "mlvfs_amd_wrap.c"))
        flags = ["-Wl,--wrap=mlvfs_load_chunks", "-Wl,--wrap=mlvfs_close_chunks"]
    cmd = ["gcc", "-std=gnu99", "-O1", "-rdynamic", "-I", os.path.join(ROOT, "include"), *srcs, *flags, "-o", str(exe),
           "-L", so_dir, "-lmlvfs_amd", "-Wl,-rpath," + so_dir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lm"]
    subprocess.run(cmd, check=True)
    return str(exe)


@pytest.fixture(scope="module", params=["library builds the tables", "caller provides get_raw2ev / get_ev2raw",
                                        "wrap: mlvfs_load_chunks / mlvfs_close_chunks are the frame bracket",
                                        "wrap + caller provides the tables"])
def c_host(request, tmp_path_factory, gpu):
    """Four links of ONE source (tests/c_host.c, which calls the reference's symbols and nothing else): a bare one; one that --
    like MLVFS's main.c (mlvfs.h:90-92) -- exports the table accessors the library imports weakly (their definitions come from
    oracle/ref_luts.c, the caller's part of the reference build); and both again with integration/mlvfs_amd_wrap.c and
    -Wl,--wrap=..., where the stages between the two chunk calls are recorded and the frame is fetched inside mlvfs_close_chunks."""
    return build_c_host(tmp_path_factory.mktemp("c_host") / "c_host", request.param)


@pytest.mark.parametrize("cs,bad,stripes", [(5, 1, 1), (2, 0, 0), (3, 2, 1)])
def test_c_program_through_the_drop_in_symbols(c_host, oracle, tmp_path, cs, bad, stripes):
    w, h = 416, 264
    f = synth.normal_frame(w, h, seed=21)
    packed = synth.pack_bits(f)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    np.ascontiguousarray(packed, "<u2").tofile(fin)
    res = subprocess.run([c_host, str(fin), str(fout), str(w), str(h), str(BLACK), str(WHITE), str(cs), str(bad), str(stripes)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.fromfile(fout, "<u2").reshape(h, w)
    # The C process starts from srand(1) like a fresh MLVFS and the oracle's stripes_compute does the same.  The HIP runtime
    # consumes rand() values of its own while it initialises: the library parks the caller's generator state around its HIP
    # work (LibcRandGuard), otherwise the stripe coefficients here would differ from the reference's.
    want, _ = oracle.process_frame(packed, w, h, BLACK, WHITE, cs, bad, stripes)
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ (column phases {np.unique(np.nonzero(got != want)[1] % 8)})"


@pytest.mark.parametrize("mode", [1, 2])
def test_c_program_dual_iso(c_host, oracle, tmp_path, mode):
    """hdr_convert_data / cr2hdr20_convert_data from C: pixels and the levels written back into frame_headers."""
    w, h = 416, 264
    f = synth.dual_iso_frame(w, h)
    packed = synth.pack_bits(f)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    np.ascontiguousarray(packed, "<u2").tofile(fin)
    res = subprocess.run([c_host, str(fin), str(fout), str(w), str(h), str(BLACK), str(WHITE), "0", "0", "0", str(mode)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.fromfile(fout, "<u2").reshape(h, w)
    if mode == 1:
        ok, want, lv = oracle.hdr_preview(f, BLACK, WHITE)
    else:
        ok, want, lv = oracle.cr2hdr20(f, BLACK, WHITE, 0, 1, 1, 0, reset=True)
    assert ok == 1 and f"levels {lv[0]} {lv[1]} dual_iso 1" in res.stderr
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ"
