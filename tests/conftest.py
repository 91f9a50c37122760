import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The libraries are build products (git-ignored).  On a tree where they are missing, build them once
    # (hipcc cross-compiles gfx950 without a GPU; the oracle's C port needs only gcc).
    import subprocess
    needed = [os.path.join(ROOT, "ga3c_amd", "libga3c_hip.so"), os.path.join(ROOT, "ga3c_amd", "libga3c_host.so"),
              os.path.join(ROOT, "oracle", "libga3c_oracle.so")]
    if not all(os.path.exists(p) for p in needed):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "ga3c_amd", "csrc")])
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
