import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A tree without the built library (a fresh checkout: *.so files are not in the history) gets it
    built once, exactly as __graft_entry__.build() does; the tests themselves never build the product."""
    lib = os.path.join(ROOT, "rays_amd", "lib", "librays_hip.so")
    if os.path.exists(lib) or os.environ.get("RAYS_TESTS_NO_BUILD"):
        return
    import shutil
    import subprocess

    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        print("[tests] librays_hip.so missing: building it (make -C rays_amd/csrc, a few minutes)", file=sys.stderr)
        subprocess.run(["make", "-C", os.path.join(ROOT, "rays_amd", "csrc"), "-j8"], check=False,
                       stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
