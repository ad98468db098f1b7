"""Opt-in GPU test (-m gpu, SNAPPY_TEST_ABLATION=1): the non-default kernel forms of csrc/ablation/ -- compiled only into
libsnappy_hip_ablation.so (tools/build_ablation.py) -- are bit-exact with the oracle.  Off by default: the forms are lab
notes, not product, and their build + 23-configuration matrix takes a few minutes."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get("SNAPPY_TEST_ABLATION") != "1", reason="set SNAPPY_TEST_ABLATION=1 to build and check the ablation kernels")
def test_ablation_kernels_bit_exact():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ablation_check.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
