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


def test_randomized_streams_match_oracle_lds_table_kernel_alone(monkeypatch):
    """The same slice with every block on the LDS-table kernel (the stream form of the parse, snappy_k1_stream.hpp): small
    inputs would otherwise only see the global-table kernel."""
    monkeypatch.setenv("SNAPPY_HIP_COMPRESS_VARIANT", "1")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    assert fuzz_gpu.run(120, 20261005, verbose=False) == 0


def test_randomized_streams_match_oracle_global_table_kernel_alone(monkeypatch):
    """The same slice with every block on the global-table kernel: behind its slot cache and in the stream form for blocks of
    more than 8 KiB, plain (slot filter, bulk form) below; inputs this small would otherwise go to the LDS-table kernel."""
    monkeypatch.setenv("SNAPPY_HIP_LDS_WAVES", "0")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    assert fuzz_gpu.run(120, 20261006, verbose=False) == 0
