"""Collected last (file name): bounds the relaxed / unpinned patches of the WHOLE session (tests/tolerances.py RECORDS, every module's;
conftest.py's pytest_sessionfinish checks the same bounds again and writes the JSON record)."""
import pytest

import tolerances


@pytest.mark.gpu
def test_relaxed_and_unpinned_patches_are_rare(gpu):
    assert tolerances.violations() == [], tolerances.summary()
    assert all(r["bar_px"] is None or r["bar_px"] <= tolerances.CEILING for r in tolerances.RECORDS)
