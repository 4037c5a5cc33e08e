"""Sanitizer builds of the host side: AddressSanitizer + UndefinedBehaviorSanitizer over
(1) the CPU oracle's whole C surface and (2) the product library's host code -- C-ABI argument / geometry validation,
every entry point on a null engine, the scale/rotation estimator's host-built remap tables (compared with the oracle's
inside the driver) and the geometry tail's host forms -- run WITHOUT a device (GPU sanitizers are not available on this
pool; device code is compiled as usual, -Xarch_host keeps the instrumentation on the host pass). tests/san/Makefile."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build_san")


@pytest.fixture(scope="module")
def san_build():
    if not os.path.exists(os.path.join(ROOT, "tests", "san", "Makefile")):
        pytest.skip("tests/san/ is not shipped to the GPU box (.gpurunignore): sanitizer builds are a CPU-suite job")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "san"), "-s"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return OUT


def _run(path, env_extra=None):
    env = dict(os.environ, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", **(env_extra or {}))
    r = subprocess.run([path], capture_output=True, text=True, timeout=600, env=env)
    report = r.stdout + r.stderr
    assert r.returncode == 0, report[-4000:]
    assert "runtime error" not in report and "AddressSanitizer" not in report and "LeakSanitizer" not in report, report[-4000:]
    return report


def test_oracle_under_asan_ubsan(san_build):
    assert "oracle sanitizer driver: ok" in _run(os.path.join(san_build, "oracle_san_driver"))


def test_capi_host_side_under_asan_ubsan(san_build):
    # leak detection stays ON for our own code; the HIP runtime's start-up allocations (made while it looks for a
    # device) are its own and are suppressed by library name
    supp = os.path.join(ROOT, "tests", "san", "lsan.supp")
    out = _run(os.path.join(san_build, "capi_san_driver"), {"LSAN_OPTIONS": f"suppressions={supp}:print_suppressions=0"})
    assert "C-ABI sanitizer driver: ok" in out
