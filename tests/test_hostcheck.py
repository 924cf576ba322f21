"""Memory safety of the host-only code that parses untrusted files: the container walk (index.c:216-341's
replacement, csrc/mlvreader.cpp), the LZMA decoder (csrc/lzma.cpp), the LJ92 marker / Huffman parser (csrc/lj92.cpp), the DNG header
writer (csrc/dngheader.cpp), the histogram helpers, the rand() stream and the stripes solve -- compiled with
g++ -fsanitize=address,undefined (`make -C mlvfs_amd/csrc hostcheck`) and put through the CPU suite's own vector and damage tests in
a child process (LD_PRELOAD=libasan.so: python itself is not instrumented).  The GPU box cannot run sanitizers; this runs here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "mlvfs_amd", "libmlvfs_amd_hostcheck.so")
# the CPU tests that exercise host-only code of the library (no kernel launcher is reached: those are aborting stubs in this build)
SUITES = ["tests/test_lzma_gif.py", "tests/test_header.py", "tests/test_mlv_reader.py", "tests/test_lj92.py", "tests/test_lj92_encode.py", "tests/test_cabi.py",
          "tests/test_golden.py", "tests/test_failure_policy.py"]


def _asan():
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(os.environ.get("MLVFS_AMD_LIB") is not None, reason="this is the child")
def test_host_only_code_under_address_and_undefined_behaviour_sanitizers():
    asan = _asan()
    if asan is None:
        pytest.skip("gcc's libasan.so not found")
    b = subprocess.run(["make", "-C", os.path.join(ROOT, "mlvfs_amd", "csrc"), "hostcheck", "-j8"], capture_output=True, text=True)
    assert b.returncode == 0 and os.path.exists(SO), b.stdout[-2000:] + b.stderr[-2000:]
    env = dict(os.environ, MLVFS_AMD_LIB=SO, LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider", *SUITES],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    assert "device code called" not in r.stderr, tail
    assert r.returncode == 0 and " passed" in r.stdout and " failed" not in r.stdout, tail
