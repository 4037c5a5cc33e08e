import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # built artefacts are git-ignored: a fresh checkout builds them once (hipcc cross-compiles gfx950 without a GPU)
    need = [os.path.join(ROOT, "mrs_optic_flow_amd", "libmof_hip.so"), os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "tests", "cpp", "test_processors")]
    if os.path.exists("/root/reference/include/OpticFlowCalc.h"):  # the adapter compile test needs the reference's header
        need.append(os.path.join(ROOT, "tests", "cpp", "test_adapter"))
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    return torch.device("cuda:0")
