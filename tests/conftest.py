"""pytest configuration.

Markers
  gpu      needs a real MI355X (run by the driver on the GPU box: pytest -m gpu)
Everything else runs on CPU: the oracle against the reference build / golden
vectors, the host logic, and the C-ABI export table (no compute calls).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a HIP device (MI355X)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own code (oracle/_ref); only where it has been built."""
    from oracle import bindings
    if not bindings.have_ref():
        if os.path.isdir("/root/reference/mlvfs"):
            bindings.build()
        else:
            pytest.skip("oracle/_ref/libmlvfs_ref.so not present (needs /root/reference to build)")
    return bindings.Reference()


@pytest.fixture(scope="session")
def amd():
    """libmlvfs_amd.so through ctypes; fails (not skips) when the library is missing."""
    from mlvfs_amd import lib
    return lib.load()


@pytest.fixture(scope="session")
def gpu(amd):
    n = amd.mlvfs_amd_device_count()
    assert n > 0, "no HIP device visible: -m gpu tests must run on the GPU box"
    assert amd.mlvfs_amd_init(0) == 0, amd.mlvfs_amd_last_error()
    return amd


from mlvfs_amd.synth import fnv1a  # noqa: E402,F401  (kept importable from conftest: tests/golden/make_golden.py)
