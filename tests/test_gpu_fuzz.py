"""A short run of the randomised differential test (tools/fuzz_parity.py): random frame sizes, feature and
level counts, scale factors, thresholds, presets and image kinds; extraction and both windowed searches must
equal the oracle bit for bit.  Longer runs: PYTHONPATH=. python tools/fuzz_parity.py 200 <seed>."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_configurations_equal_oracle():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    failed = [it for it in range(16) if not fz.one(it, 11)]
    assert not failed, "iterations %s differ from the oracle" % failed
