import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _built(path):
    return os.path.exists(path)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(HERE, "golden", "scenes.npz"))


@pytest.fixture(scope="session")
def built_libs():
    """Build the CPU-side pieces once (oracle, probe). HIP libs are built by __graft_entry__.build()."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return True


@pytest.fixture(scope="session")
def oracle(built_libs):
    import b2harness as bh
    return bh.Harness(bh.ORACLE_LIB)


@pytest.fixture(scope="session")
def ref():
    import b2harness as bh
    if not bh.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return bh.Harness(bh.REF_LIB)


@pytest.fixture(scope="session")
def amd():
    """The product: drop-in Box2D API on the HIP C ABI. No fallback: a missing library is a failure."""
    import b2harness as bh
    if not bh.have_amd():
        pytest.fail("box2d-mt_amd/libb2amd_harness.so missing: run __graft_entry__.build() (there is no CPU fallback)")
    return bh.Harness(bh.AMD_LIB)
