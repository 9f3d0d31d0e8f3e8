"""pytest setup: `gpu` marker, import paths, one-time build of the C libraries."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "daily-ray-trace_amd")
for p in (PKG, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(args):
    r = subprocess.run(["make"] + args, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """The .so files normally travel with the repo; (re)build what is missing."""
    if not os.path.exists(os.path.join(REPO, "oracle", "libdrt_oracle.so")):
        _make(["-C", os.path.join(REPO, "oracle"), "all"])
    if not os.path.exists(os.path.join(PKG, "libdrt_host.so")):
        _make(["-C", PKG, "host"])
    if not os.path.exists(os.path.join(PKG, "libdrt_hip.so")) and os.path.exists("/opt/rocm/bin/hipcc"):
        _make(["-C", PKG, "hip"])
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
