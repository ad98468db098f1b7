"""GPU test: bench.py's one JSON line keeps the driver's contract (small run: 2 containers of 64 MiB), and the multi-rank path
runs with two ranks on this one device (gloo for the control plane, as the CPU tests do)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_json_line_keeps_the_contract():
    d = _run([sys.executable, "bench.py", "--containers", "2", "--container-mib", "64", "--steps", "2", "--warmup", "1"])
    assert REQUIRED <= set(d), REQUIRED - set(d)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert d["metric"] == json.load(f)["metric"]
    assert d["unit"] == "GB/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "u8" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = bytes of all containers / step time
    assert abs(d["value"] - 2 * 64 * 2**20 / (d["ms_per_step"] * 1e-3) / 1e9) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and 0 < r["frac"] < 1
    assert r["traffic"] is None or (r["traffic"] > 0 and "pmc" in r["traffic_source"])
    # the same steps with the block index walked from the streams alone: a second timed region, slower by the walk
    assert 0 < d["value_from_stream_alone"] <= d["value"] * 1.05 and d["ms_per_step_from_stream_alone"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "GB/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert c["gpu_stream_equals_oracle_stream"] is True


def test_bench_two_ranks_on_one_device():
    import socket
    with socket.socket() as sock:                                 # a free port (a lingering run would hold a fixed one)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), "bench.py", "--gpus", "2", "--containers", "2", "--container-mib", "64", "--steps", "2",
              "--warmup", "1", "--no-cpu-baseline"],
             env={"SNAPPY_BENCH_BACKEND": "gloo", "SNAPPY_BENCH_SINGLE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["containers_this_rank"] == 1               # the two containers were dealt one per rank


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torchrun around it (how the driver runs the N = 1 line) must BE a two-rank job: the
    parent starts one child per GPU and relays rank 0's line (the reference's entry point allocates its
    own N devices, snappy_compress.c:535).  Both ranks share this box's one device; gloo carries the control plane."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"SNAPPY_BENCH_BACKEND": "gloo", "SNAPPY_BENCH_SINGLE_DEVICE": "1"})
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--containers", "2", "--container-mib", "64", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["roundtrip_bit_exact"] is True
    assert d["config"]["containers_this_rank"] == 1
    # a rank that fails takes the job down with a non-zero exit code instead of hanging the others at a barrier
    # (SNAPPY_HIP_GT_CACHE=256 is an ablation-only value: the product library refuses the first K1 launch of every rank)
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--containers", "2", "--container-mib", "64", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT, env=dict(env, SNAPPY_HIP_GT_CACHE="256"),
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_bench_rccl_path_with_one_rank():
    """The N > 1 runs of the driver use RCCL (backend "nccl") for the barrier and the MAX / SUM reductions.  One rank is all
    this box can give it, but the calls are the same: process group with a device id, barrier on the device, all_reduce of
    device tensors (SNAPPY_BENCH_FORCE_DIST=1)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    d = _run([sys.executable, "bench.py", "--containers", "2", "--container-mib", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
             env={"SNAPPY_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                  "MASTER_PORT": str(port)})
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roundtrip_bit_exact"] is True
