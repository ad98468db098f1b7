"""The dpu_snappy CLI (pim-compression_amd/host): same flags and stdout contract as the reference's tool
(snappy/dpu_snappy.c:12,152-227), whose output lines the reference's scripts scrape
(snappy/scripts/asplos21/parse_output_file.py)."""
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, GOLDEN_PAIRS, ROOT, golden_bytes

HOST_DIR = os.path.join(ROOT, "pim-compression_amd", "host")
CLI = os.path.join(HOST_DIR, "dpu_snappy")

LINES = [r"Using input file .+", r"Using output file .+", r"(Compressed|Decompressed) \d+ bytes to: .+",
         r"Compression ratio: -?\d+\.\d+", r"Pre-processing time: \d+\.\d+", r"Alloc time: \d+\.\d+",
         r"Load time: \d+\.\d+", r"Copy in time: \d+\.\d+", r"Host time: \d+\.\d+", r"Copy out time: \d+\.\d+",
         r"Free time: \d+\.\d+"]


@pytest.fixture(scope="module")
def cli():
    import __graft_entry__ as entry
    entry.build_hip()
    subprocess.check_call(["make", "-s", "-C", HOST_DIR])
    return CLI


def run(cli, *args):
    return subprocess.run([cli, *args], capture_output=True, text=True)


def check_stdout_contract(text, gpu=False):
    lines = [l for l in text.strip().splitlines() if not l.startswith("GPU ")]
    assert len(lines) == len(LINES), text
    for pat, line in zip(LINES, lines):
        assert re.fullmatch(pat, line), (pat, line)
    if gpu:
        assert re.search(r"^GPU \d+: \d+\.\d+ s, \d+ bytes$", text, re.M)


@pytest.mark.parametrize("name", GOLDEN_PAIRS)
def test_host_mode_goldens(cli, name, tmp_path):
    out = tmp_path / "o.snappy"
    r = run(cli, "-c", "-i", os.path.join(GOLDEN, name + ".txt"), "-o", str(out))
    assert r.returncode == 0, r.stderr
    check_stdout_contract(r.stdout)
    assert out.read_bytes() == golden_bytes(name + ".snappy")
    back = tmp_path / "o.txt"
    r = run(cli, "-i", str(out), "-o", str(back))
    assert r.returncode == 0, r.stderr
    check_stdout_contract(r.stdout)
    assert back.read_bytes() == golden_bytes(name + ".txt")


def test_host_mode_block_size_flag_and_errors(cli, tmp_path):
    import oracle_lib as oracle
    src = os.path.join(GOLDEN, "coding.txt")
    out = tmp_path / "o.snappy"
    r = run(cli, "-c", "-b", "4096", "-i", src, "-o", str(out))
    assert r.returncode == 0
    assert out.read_bytes() == oracle.compress(golden_bytes("coding.txt"), 4096)
    r = run(cli)
    assert r.returncode != 0 and "usage:" in r.stderr
    r = run(cli, "-i", str(tmp_path / "missing"))
    assert r.returncode != 0 and "Invalid input file" in r.stderr
    bad = tmp_path / "bad.snappy"
    bad.write_bytes(golden_bytes("alice.snappy")[:-5])
    r = run(cli, "-i", str(bad), "-o", str(tmp_path / "x"))
    assert r.returncode != 0 and "Encountered Snappy error" in r.stderr


def test_gpu_flag_without_gpu_fails_loudly(cli, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = run(cli, "-d", "-c", "-i", os.path.join(GOLDEN, "alice.txt"), "-o", str(tmp_path / "x"))
    assert r.returncode != 0
    assert "no HIP device" in r.stderr and "Encountered Snappy error" in r.stderr
    assert not (tmp_path / "x").exists()


@pytest.mark.gpu
def test_gpu_mode_goldens_and_contract(cli, tmp_path):
    for name in GOLDEN_PAIRS:
        out = tmp_path / (name + ".snappy")
        r = run(cli, "-d", "-c", "-i", os.path.join(GOLDEN, name + ".txt"), "-o", str(out))
        assert r.returncode == 0, r.stderr
        check_stdout_contract(r.stdout, gpu=True)
        assert out.read_bytes() == golden_bytes(name + ".snappy")
        back = tmp_path / (name + ".txt")
        r = run(cli, "-d", "-i", str(out), "-o", str(back))
        assert r.returncode == 0, r.stderr
        check_stdout_contract(r.stdout, gpu=True)
        assert back.read_bytes() == golden_bytes(name + ".txt")
    # the reference-style regression target (snappy/Makefile:54-60)
    subprocess.check_call(["make", "-s", "-C", HOST_DIR, "test_gpu"])
