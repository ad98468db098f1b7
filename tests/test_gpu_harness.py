"""GPU tests (-m gpu) of the callers either side of the hot path: the reference-style sweep
harness over the dpu_snappy CLI, and the raw-Snappy converter on streams the GPU produced."""
import csv
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

import datagen
import oracle_lib as oracle
from conftest import ROOT, golden_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shb():
    import torch
    import __graft_entry__ as entry
    entry.build_hip()
    entry.build_cli()
    import snappy_hip_binding as binding
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return binding


def _rows(path):
    with open(path, newline="") as f:
        return list(csv.reader(f))


def test_sweep_harness_writes_reference_layout_csvs(shb, tmp_path, monkeypatch):
    """tools/run_sweeps.py (counterpart of snappy/scripts/asplos21/run_tests.py): device-count sweep with the sharded path
    (2 and 4 shards, mapped onto the available device), the breakdown in the reference's column layout
    (run_tests.py:134), the block-size sweep with its LDS occupancy column; the harness checks parity of every run itself
    (CLI GPU output == CLI CPU output, byte for byte)."""
    monkeypatch.setenv("SNAPPY_HIP_OVERSUBSCRIBE", "1")
    out = tmp_path / "sweep"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sweeps.py"), "--out", str(out), "--gpus", "1,2,4",
                        "--mix-mib", "16", "--block-sizes", "4096,32768"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ref_header = ["prepare", "alloc", "load", "copy_in", "run", "copy_out", "free", "dpus"]          # run_tests.py:134
    for stem in ("terror2", "plrabn12", "world192", "silesia_mix_16MiB"):
        for direction in ("compression", "decompression"):
            rows = _rows(out / f"{stem}_{direction}_breakdown.csv")
            assert rows[0] == ref_header
            assert [r_[-1] for r_ in rows[1:]] == ["1", "2", "4"]
            assert all(len(r_) == 8 and all(float(v) >= 0 for v in r_[:-1]) for r_ in rows[1:])
            assert all(float(r_[4]) > 0 for r_ in rows[1:])                                          # run
    for direction in ("compression", "decompression"):
        rows = _rows(out / f"{direction}_speedup_dpu.csv")
        assert rows[0] == ["version", "time", "dpus"] and rows[1] == ["host", "1", "0"]            # run_tests.py:86-88
        assert len(rows) == 2 + 4 * 3 and all(float(r_[1]) > 0 for r_ in rows[2:])
    rows = _rows(out / "speedup.csv")
    assert rows[0] == ["file", "bytes", "direction", "gpus", "host_s", "gpu_kernel_s", "gpu_total_s", "speedup_kernel", "speedup_total"]
    assert len(rows) == 1 + 4 * 3 * 2
    rows = _rows(out / "breakdown.csv")
    assert rows[0] == ["file", "direction"] + ref_header[:-1] + ["gpus"]
    rows = _rows(out / "blocksize.csv")
    assert rows[0] == ["file", "block_size", "compressed_bytes", "space_saving", "host_compress_s", "gpu_run_s", "lds_waves_per_cu"]
    by_bs = {(r_[0], int(r_[1])): r_ for r_ in rows[1:]}
    mix = "silesia_mix_16MiB.bin"
    assert int(by_bs[(mix, 4096)][6]) > int(by_bs[(mix, 32768)][6]) >= 1          # smaller table, more LDS-table wavefronts
    assert int(by_bs[(mix, 4096)][2]) > int(by_bs[(mix, 32768)][2])                # and a worse ratio
    for name in ("terror2", "plrabn12", "world192"):                              # ratio column == the reference's goldens
        assert int(by_bs[(name + ".txt", 32768)][2]) == len(golden_bytes(name + ".snappy"))


def test_gpu_streams_through_the_raw_snappy_converter(shb):
    """SURVEY 8f row 4 on GPU output: a stream compressed on the GPU, converted to the original Snappy framing by
    tools/to_raw_snappy.py, is decoded to the plaintext by the converter's own reader and by libsnappy (pyarrow); and a
    stream written by libsnappy, re-framed through `dpu_snappy -d -c` (the GPU), is the oracle's stream."""
    import torch
    spec = importlib.util.spec_from_file_location("to_raw_snappy", os.path.join(ROOT, "tools", "to_raw_snappy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        import pyarrow as pa
        have_lib = pa.Codec.is_available("snappy")
    except ImportError:
        have_lib = False
    text = golden_bytes("plrabn12.txt")
    for data, bs in ((golden_bytes("world192.txt"), 32768), (datagen.text_random_interleave(text, 400_000), 4096),
                     (datagen.records(300_000), 65535), (datagen.zeros(200_000), 32768)):
        d = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
        d[:len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        stream = bytes(shb.compress_resident(d, bs, n=len(data)).cpu().numpy())
        assert stream == oracle.compress(data, bs)
        raw = mod.convert(stream)
        assert mod.decode_raw(raw) == data
        if have_lib:
            assert pa.decompress(raw, decompressed_size=len(data), codec="snappy").to_pybytes() == data
            theirs = pa.compress(data, codec="snappy").to_pybytes()
            assert mod.reframe(theirs, bs, gpu=True) == stream
