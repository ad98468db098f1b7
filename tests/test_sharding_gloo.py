"""world_size-2 gloo tests (CPU) of the multi-rank paths.

1. bench.py's rank logic: container sharding (strong and weak) and the MAX-time / SUM-bytes aggregation.
2. The data path itself, multi-process: every rank COMPRESSES its shard with the product's kernel source (compiled for
   the wave emulator, tests/emu) and rank 0 does the host-side concat -- (a) the drop-in pair's partition of ONE file
   into contiguous block ranges per device (reference snappy_compress.c:494-520) with the concat of per-device outputs
   (:697-704), checked against the oracle's stream of the whole file, and the decode direction with per-rank output
   ranges (snappy_decompress.c:306-341, :463); (b) bench.py's whole-container plan, every container's stream checked
   against the oracle.  A wrong concat order or a shard boundary off by one block fails the byte comparison.
No collective is on the data path (blocks are independent); gloo only carries the gather of the results for checking.
"""
import os
import socket
import sys

import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden_bytes

sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _logic_worker(rank, world, port, q):
    _init(rank, world, port)
    import bench
    assert bench.dist_env() == (rank, world, rank)
    strong = bench.shard_plan(rank, world, 8, "strong")
    weak = bench.shard_plan(rank, world, 8, "weak")
    # every rank times its own containers; rank 1 is slower
    secs, tot, comp = bench.reduce_results(1.0 + rank, len(strong) * 100, len(strong) * 40 + rank, dist, device="cpu")
    q.put((rank, strong, weak, secs, tot, comp))
    dist.barrier()
    dist.destroy_process_group()


def _run(target, world=2, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_two_rank_sharding_and_reduction():
    got = _run(_logic_worker)
    assert [g[1] for g in got] == [[0, 1, 2, 3], [4, 5, 6, 7]]                     # strong: the fixed 8 containers, dealt
    assert [g[2] for g in got] == [list(range(0, 8)), list(range(8, 16))]          # weak: 8 per rank
    for _, _, _, secs, tot, comp in got:
        assert secs == 2.0                      # MAX over ranks
        assert tot == 800 and comp == 321       # SUM over ranks


def test_single_process_passthrough():
    import bench
    assert bench.reduce_results(0.5, 10, 4, None) == (0.5, 10, 4)
    assert bench.shard_plan(0, 1, 8) == list(range(8))
    assert bench.shard_plan(0, 1, 8, "weak") == list(range(8))
    for world in (1, 2, 3, 4, 8, 16):                       # every container exactly once, in order, whatever the world size
        assert sum((bench.shard_plan(r, world, 8, "strong") for r in range(world)), []) == list(range(8))


def _file():
    import datagen
    return datagen.text_random_interleave(golden_bytes("plrabn12.txt"), 9 * 4096 + 1234, seed=11)


def _datapath_worker(rank, world, port, q):
    _init(rank, world, port)
    sys.path.insert(0, os.path.join(ROOT, "pim-compression_amd"))
    import bench
    import emu_lib as emu
    import oracle_lib as oracle
    import snappy_hip_binding as shb
    bs = 4096
    data = _file()                                          # 10 blocks, the last one short
    nb = (len(data) + bs - 1) // bs

    # ---- (a) one file, contiguous block ranges per rank; each rank frames its own slice, rank 0 concatenates ----
    first, count = shb.shard_block_range(nb, world, rank)
    lo, hi = first * bs, min(len(data), (first + count) * bs)
    local = emu.compress(data[lo:hi], bs)                   # a framed stream of the slice: own header + blocks
    _, _, local_hdr = oracle.read_header(local)
    body = local[local_hdr:]
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, first, count, body))
    whole = None
    if rank == 0:
        parts = sorted(gathered)                            # rank order == block order (contiguous ranges)
        assert [p[1] for p in parts] == [shb.shard_block_range(nb, world, r)[0] for r in range(world)]
        whole = shb.write_header(len(data), bs) + b"".join(p[3] for p in parts)
        assert whole == oracle.compress(data, bs)
    # decode direction: rank 0 walks the size chain (host pre-scan) and hands each rank its slice of the stream
    box = [whole]
    dist.broadcast_object_list(box, src=0)
    whole = box[0]
    offs = [int(v) for v in oracle.index_blocks(whole)] + [len(whole)]
    s_lo, s_hi = offs[first], offs[first + count]
    out_len = hi - lo
    piece = shb.write_header(out_len, bs) + whole[s_lo:s_hi]
    _, _, ph = oracle.read_header(piece)
    st, plain = emu.decompress(piece, out_len, bs, ph)
    assert st == 0
    outs = [None] * world
    dist.all_gather_object(outs, (rank, plain))
    if rank == 0:
        assert b"".join(p for _, p in sorted(outs)) == data

    # ---- (b) bench.py's strong plan: whole containers per rank, no concat across containers ----
    containers = [data[k * 5000:k * 5000 + 9000] for k in range(3)]
    mine = bench.shard_plan(rank, world, len(containers), "strong")
    streams = [(cid, emu.compress(containers[cid], bs)) for cid in mine]
    allst = [None] * world
    dist.all_gather_object(allst, streams)
    if rank == 0:
        flat = sorted(sum(allst, []))
        assert [cid for cid, _ in flat] == list(range(len(containers)))
        for cid, s in flat:
            assert s == oracle.compress(containers[cid], bs), cid
    q.put((rank, count))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_compress_their_shards_and_rank0_concatenates():
    got = _run(_datapath_worker)
    assert sum(c for _, c in got) == 10                     # every block of the file was compressed by exactly one rank


def test_shard_to_device_mapping_and_stream_cache_key(tmp_path):
    """csrc/shard_devices.hpp (the drop-in pair's per-call device map), compiled on the CPU: shard 0 runs on the caller's
    current device, shard g on (base + g) % devices, and the cached stream sets are keyed by (device, shard) -- so a second
    call entered with another current device never picks up streams created on the first call's device, and two shards that
    share a device (oversubscription) never share a set."""
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <cstdio>
#include <set>
#include "shard_devices.hpp"
int main() {
    for (int physical = 1; physical <= 8; ++physical)
        for (int base = 0; base < physical; ++base) {
            ShardDevices d;
            d.physical = physical;
            d.base = base;
            if (d.device_of(0) != base) return 1;                       // shard 0 on the caller's device
            std::set<int> devs, keys;
            for (int g = 0; g < physical; ++g) {
                const int dev = d.device_of(g);
                if (dev < 0 || dev >= physical || dev != (base + g) % physical) return 2;
                devs.insert(dev);
            }
            if ((int)devs.size() != physical) return 3;                   // `physical` shards cover every device once
            for (int g = 0; g < 64; ++g) keys.insert(pipeline_stream_key(d.device_of(g), g));
            if (keys.size() != 64) return 4;                              // oversubscribed shards never share a stream set
        }
    // the same shard number on two different devices (two calls with different current devices) -> different sets
    ShardDevices a, b;
    a.physical = b.physical = 8;
    a.base = 0;
    b.base = 1;
    for (int g = 0; g < 8; ++g)
        if (pipeline_stream_key(a.device_of(g), g) == pipeline_stream_key(b.device_of(g), g)) return 5;
    std::puts("ok");
    return 0;
}
''')
    exe = tmp_path / "t"
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pim-compression_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-I", csrc, str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.returncode


def test_bench_child_environment_and_launcher_detection():
    """bench.py --gpus N without a launcher starts its own ranks (bench.launch_ranks); the pieces that need no GPU: the
    environment each child gets, and when the script considers itself launched already."""
    import bench
    base = {"PATH": "/usr/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    envs = [bench.child_environment(r, 4, 29517, base) for r in range(4)]
    for r, e in enumerate(envs):
        assert e["RANK"] == e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == e["LOCAL_WORLD_SIZE"] == "4"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29517"
        assert e["PATH"] == "/usr/bin" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"      # the parent's environment travels
        assert bench.launched_by_a_launcher(e)                                          # a child never starts children
    assert base == {"PATH": "/usr/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}              # (not modified in place)
    assert not bench.launched_by_a_launcher(base)
    assert bench.launched_by_a_launcher({"WORLD_SIZE": "1"}) and bench.launched_by_a_launcher({"RANK": "0"})
