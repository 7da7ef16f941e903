import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The kernel library first, PyTorch after it. Some test modules import torch (device memory for resident inputs, gloo ranks), and torch ships its own copy of
# the HIP runtime (ROCm 7.0.2 next to this image's 7.2): whichever libamdhip64.so.7 is loaded first serves the whole process. With torch's copy first, two of
# nineteen full GPU-suite runs of round 6 ended in glibc's "double free or corruption (!prev)" inside test_gpu_error_isolation's create-a-tracker /
# destroy-it cycles; with the image's runtime first (what every run of a single test file had always used) none of ~150 runs and 9 000 stress cycles did
# (DESIGN.md section 9). bench.py imports torch first — it creates its trackers once.
try:
    from motcpp_amd import _lib as _motlib
    if os.path.exists(_motlib.HIP_LIB):
        _motlib.hip()
except Exception:  # (a tree without the built library: the tests that need it say so themselves)
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from tests import orclib
    return orclib.load()


def pytest_runtest_logfinish(nodeid, location):
    """MOT_TEST_STOP_AFTER=<substring of a test id>: end the session after that test (tools/repro_loop.sh: a run that is the full suite up to there —
    same collection, same imports, same heap — without paying for the rest)"""
    stop = os.environ.get("MOT_TEST_STOP_AFTER")
    if stop and stop in nodeid:
        pytest.exit("MOT_TEST_STOP_AFTER", returncode=0)
