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


def pytest_sessionfinish(session, exitstatus):
    """The f32-limited patches this session met (tests/tolerances.py): label, patch, oracle-to-oracle distance, bar used, distances."""
    try:
        import json

        import tolerances
    except Exception:
        return
    if not tolerances.RECORDS:
        return
    path = os.environ.get("MOF_F32_LIMITED_JSON", os.path.join(ROOT, "gpurun_out", "f32_limited.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"tol_px": tolerances.TOL, "factor": tolerances.F32_LIMITED_FACTOR, "ceiling_px": tolerances.CEILING,
                       "count": len(tolerances.RECORDS), "unpinned_from_px": tolerances.UNPINNED_FROM,
                       "unpinned": sum(r["bar_px"] is None for r in tolerances.RECORDS),
                       "worst_bar_px": max([r["bar_px"] for r in tolerances.RECORDS if r["bar_px"] is not None], default=None),
                       "patches": tolerances.RECORDS}, f, indent=1)
    except OSError:
        pass
