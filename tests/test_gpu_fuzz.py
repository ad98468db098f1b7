"""-m gpu: a slice of tools/fuzz_gpu.py (random data kinds x lengths x block sizes) inside the test suite: every compressed
stream equals the oracle's byte for byte and decodes back to its input."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomized_streams_match_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    assert fuzz_gpu.run(150, 20261004, verbose=False) == 0
