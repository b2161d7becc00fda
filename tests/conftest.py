import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the built libraries are git-ignored: a fresh checkout builds them once (hipcc cross-compiles without a GPU)
    need = [os.path.join(ROOT, "dspsr_amd", "libdspsr_amd.so"), os.path.join(ROOT, "oracle", "liboracle_c.so")]
    if not all(os.path.exists(p) for p in need):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "dspsr_amd", "csrc")])
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def oracle():
    import oracle.dspsr_oracle as o
    return o
