import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("GATO_VERBOSE", "0")
# The in-process multi-rank tests (tests/test_gpu_cluster.py: up to 8 ranks as 8 streams of ONE process, because a 1-GPU
# box cannot host 8 GPU processes) need every rank's launch on a hardware queue of its own, or the launches that share
# one run one after the other and wait for each other until the time-out.  HIP gives a process 4 by default.  Must be
# set before HIP initialises; the product configuration (one process per GPU) does not need it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _dump_f32_parity_log():
    """Every measured fp32 pair of the session (tests/f32_parity.py) -> gpurun_out/f32_parity.json."""
    yield
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import f32_parity
    f32_parity.dump()
