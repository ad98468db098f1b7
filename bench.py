#!/usr/bin/env python3
"""bench.py -- end-to-end GB/s (compress + decompress) on the synthetic Silesia-mix, MI355X.

One "step" = one pass of the hot path over one batch: every container of this rank is compressed
(K1 per-block compress + scan/gather into the framed stream), then every stream is indexed (size-chain
walk) and decompressed (K2).  Inputs are resident in HBM when the timed region starts.  One process per
GPU; blocks/containers are independent, so ranks share nothing on the data path (no collective): each
rank processes its own 8 GiB batch (4 containers x 2 GiB: the format's length field is a uint32) ("weak" scaling) and rank 0 reports the whole-job aggregate.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "end-to-end GB/s (compress + decompress) on Silesia-mix; bit-exact ratio parity"
BLOCK_SIZE = 32768              # reference default, snappy/dpu_snappy.c:100


# ---------------------------------------------------------------------------------------------------
# sharding / aggregation (pure logic + torch.distributed; covered by the gloo CPU test)
# ---------------------------------------------------------------------------------------------------

def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


def shard_plan(rank, world, containers_per_gpu):
    """Container ids (global numbering) owned by `rank`.  Weak scaling: a fixed count per rank,
    whole containers per GPU ('for the 8 GiB batch: whole containers per GPU')."""
    return [rank * containers_per_gpu + i for i in range(containers_per_gpu)]


def reduce_results(local_seconds, local_bytes, local_comp_bytes, dist=None, device="cpu"):
    """MAX of the elapsed time over ranks, SUM of the bytes.  Returns (seconds, bytes, comp_bytes)."""
    import torch
    if dist is None or not dist.is_initialized():
        return local_seconds, local_bytes, local_comp_bytes
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    b = torch.tensor([float(local_bytes), float(local_comp_bytes)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return float(t.item()), int(b[0].item()), int(b[1].item())


# ---------------------------------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------------------------------

class Batch:
    """Device-resident containers + the buffers one step needs."""

    def __init__(self, shb, torch, container_ids, container_len):
        import numpy as np
        import silesia_mix
        self.shb, self.torch = shb, torch
        self.n = container_len
        self.count = len(container_ids)
        self.nb = shb.num_blocks(container_len, BLOCK_SIZE)
        self.hdr = len(shb.write_header(container_len, BLOCK_SIZE))
        # the xml plaintext comes from the product's own decoder, checked by digest
        with open(os.path.join(ROOT, "tests", "golden", "xml.snappy"), "rb") as f:
            xml_snappy = np.frombuffer(f.read(), dtype=np.uint8).copy()
        st, d_xml = shb.decompress_resident(torch.from_numpy(xml_snappy).cuda())
        xml = d_xml.cpu().numpy()
        if st != 0 or hashlib.sha256(xml.tobytes()).hexdigest() != silesia_mix.XML_TXT_SHA256:
            raise RuntimeError("xml.snappy did not decode to the expected plaintext")
        self.inputs = []
        for cid in container_ids:
            unit = silesia_mix.build_unit(xml, seed=cid)
            self.inputs.append(silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), container_len))
        self.ws = shb.CompressWorkspace(container_len, BLOCK_SIZE)
        cap = self.ws.stream_capacity(container_len) + 16
        self.streams = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in container_ids]
        self.stream_lens = [0] * self.count
        self.boffs = [torch.empty(self.nb, dtype=torch.int64, device="cuda") for _ in container_ids]
        self.results = [torch.zeros(2, dtype=torch.int32, device="cuda") for _ in container_ids]
        self.status = torch.empty(self.nb, dtype=torch.int32, device="cuda")
        self.out = torch.empty(container_len + 16, dtype=torch.uint8, device="cuda")
        self.kernel_events = {"compress": [], "decompress": []}
        # One side stream per container for its size-chain walk (a single latency-bound wavefront, ~30-45 ms for 65536
        # hops): the walks of different containers overlap each other and the compression of the following containers.
        # Their descriptors live on the device and take the stream length from the compressor's device-side result, so
        # the host never waits inside the compress phase.
        self.side_streams = [torch.cuda.Stream() for _ in container_ids]
        self.index_done = [torch.cuda.Event() for _ in container_ids]
        self.descs = [shb.make_stream_descs([
            dict(stream=self.streams[i], stream_len=0, block_offsets=self.boffs[i], result=self.results[i],
                 total_len=self.n, block_size=BLOCK_SIZE, header_len=self.hdr, num_blocks=self.nb)])
            for i in range(self.count)]
        self.d_stream_lens = torch.zeros(self.count, dtype=torch.int64, device="cuda")

    def _timed(self, key, record, fn):
        if not record:
            fn()
            return
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()          # torch's current stream == the stream handed to the C ABI
        fn()
        e1.record()
        self.kernel_events[key].append((e0, e1))

    def step(self, record=False):
        shb, torch = self.shb, self.torch
        main = torch.cuda.current_stream()
        # ---- compress every container; the size-chain walk of container i runs on its own side stream underneath
        #      the compression of the containers after it ----
        for i, d_in in enumerate(self.inputs):
            self._timed("compress", record, lambda: shb.compress_blocks(d_in, self.n, self.ws))
            shb.compact(self.n, self.ws, self.streams[i])
            # device-side hand-over of the stream length (descriptor field at byte 8, and the list the host reads later)
            self.descs[i][8:16].copy_(self.ws.stream_len.view(torch.uint8), non_blocking=True)
            self.d_stream_lens[i:i + 1].copy_(self.ws.stream_len, non_blocking=True)
            side = self.side_streams[i]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                shb.index_streams(self.descs[i], 1)
                self.index_done[i].record(side)
        self.stream_lens = [int(v) for v in self.d_stream_lens.cpu().tolist()]   # the decoder's launch needs the lengths
        # ---- decompress each stream as soon as its index is ready ----
        for i in range(self.count):
            main.wait_event(self.index_done[i])
            self._timed("decompress", record,
                        lambda: shb.decompress_blocks(self.streams[i], self.stream_lens[i], self.boffs[i], self.n,
                                                      BLOCK_SIZE, self.out, self.status))

    def verify(self):
        """Outside the timed region: every container round-trips bit-exactly and every block decoded OK."""
        torch, shb = self.torch, self.shb
        for i, d_in in enumerate(self.inputs):
            shb.compress_blocks(d_in, self.n, self.ws)
            shb.compact(self.n, self.ws, self.streams[i])
            slen = int(self.ws.stream_len.item())
            self.stream_lens[i] = slen
            st, d_out = shb.decompress_resident(self.streams[i][:slen])
            if st != 0 or not torch.equal(d_out[:self.n], d_in[:self.n]):
                return False
        return True

    def lds_share(self):
        """Fraction of the last container's blocks compressed by the LDS-table wavefronts of the concurrent K1 launch."""
        return self.ws.lds_form_blocks() / max(1, self.nb)

    def kernel_ms(self, key):
        self.torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.kernel_events[key]]
        return (sum(ms) / len(ms)) if ms else None


def cpu_baseline(batch, torch, gpu_stream_bytes):
    """Oracle (CPU restatement of the reference host path) timed on this box's host cores, on a bounded
    sample of the same workload: container 0 of rank 0 (all cores, pthreads over block ranges) and its
    first 64 MiB on one core (the reference's actual single-threaded mode)."""
    import numpy as np
    import oracle_lib as oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    host = batch.inputs[0][:batch.n].cpu().numpy()
    reps = 3 if batch.n >= (1 << 28) else 10
    oracle.compress(host[:1 << 20], BLOCK_SIZE)                       # warm the library
    t_c = t_d = 0.0
    comp = None
    for _ in range(reps):
        t0 = time.perf_counter()
        comp = oracle.compress(host, BLOCK_SIZE, threads=threads)
        t1 = time.perf_counter()
        st, plain = oracle.decompress(comp, threads=threads)
        t2 = time.perf_counter()
        assert st == 0 and len(plain) == batch.n
        t_c += t1 - t0
        t_d += t2 - t1
    parity = (hashlib.sha256(comp).hexdigest() == hashlib.sha256(gpu_stream_bytes).hexdigest())
    one = host[:min(batch.n, 64 << 20)]
    t0 = time.perf_counter()
    c1 = oracle.compress(one, BLOCK_SIZE)
    t1 = time.perf_counter()
    oracle.decompress(c1)
    t2 = time.perf_counter()
    gb = batch.n / 1e9
    return {
        "value": round(reps * gb / (t_c + t_d), 4), "unit": "GB/s", "cores": threads, "kind": "port",
        "sample": f"container 0 ({batch.n} B of the same Silesia-mix), {reps} reps, oracle/snappy_oracle.c, "
                  f"{threads} pthreads over contiguous block ranges",
        "compress_GBps": round(reps * gb / t_c, 4), "decompress_GBps": round(reps * gb / t_d, 4),
        "single_core_value": round(one.size / 1e9 / (t2 - t0), 4),
        "single_core_sample": f"first {one.size} B of container 0, 1 thread (the reference's own mode)",
        "gpu_stream_equals_oracle_stream": bool(parity),
    }


def load_pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes, if present."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--containers", type=int, default=4, help="containers per GPU")
    ap.add_argument("--container-mib", type=int, default=2048, help="container size in MiB (format limit: < 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import snappy_hip_binding as shb
    rank, world, local = dist_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    shb.lib()                                   # fails loudly if libsnappy_hip.so is missing
    # SNAPPY_BENCH_BACKEND=gloo + SNAPPY_BENCH_SINGLE_DEVICE=1 rehearse the N>1 path on a one-GPU box
    backend = os.environ.get("SNAPPY_BENCH_BACKEND", "nccl")
    if os.environ.get("SNAPPY_BENCH_SINGLE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    n = args.container_mib << 20
    batch = Batch(shb, torch, shard_plan(rank, world, args.containers), n)
    ok = batch.verify()
    gpu_stream0 = bytes(batch.streams[0][:batch.stream_lens[0]].cpu().numpy()) \
        if (rank == 0 and world == 1 and not args.no_cpu_baseline) else b""

    for _ in range(args.warmup):
        batch.step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.step(record=True)
    fence()
    elapsed = time.perf_counter() - t0

    local_bytes = args.steps * batch.count * n
    local_comp = args.steps * sum(batch.stream_lens)
    secs, tot_bytes, tot_comp = reduce_results(elapsed, local_bytes, local_comp, dist,
                                               device="cuda" if backend == "nccl" else "cpu")

    if rank == 0:
        c_ms = batch.kernel_ms("compress")
        d_ms = batch.kernel_ms("decompress")
        u = n
        c = sum(batch.stream_lens) / batch.count
        algo_bytes = u + c                                    # read plaintext once, write compressed once
        achieved = algo_bytes / (c_ms * 1e-3) / 1e9
        pmc = load_pmc_traffic()
        share = batch.lds_share()
        traffic = None
        if pmc and "k1_global_table_bytes_per_input_byte" in pmc:
            # rocprofv3 --pmc serialises the two co-running K1 kernels, so HBM bytes were measured for each form
            # running alone and are combined here with the block share the LDS-table form actually took
            traffic = int(u * ((1.0 - share) * pmc["k1_global_table_bytes_per_input_byte"] +
                               share * pmc["k1_lds_table_bytes_per_input_byte"]))
        line = {
            "metric": METRIC,
            "value": round(tot_bytes / secs / 1e9, 4),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(secs / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"silesia_mix {args.containers} x {args.container_mib} MiB containers per GPU "
                                   f"(BASELINE configs[4]), block_size {BLOCK_SIZE}",
                       "containers_per_gpu": args.containers, "container_bytes": n, "block_size": BLOCK_SIZE,
                       "parallelism": f"containers sharded over {world} GPU(s), no collective"},
            "roundtrip_bit_exact": bool(ok),
            "space_saving": round(1.0 - tot_comp / tot_bytes, 6),
            "compress_kernel_GBps": round(u / (c_ms * 1e-3) / 1e9, 3),
            "decompress_kernel_GBps": round(u / (d_ms * 1e-3) / 1e9, 3),
            "roofline": {"bound": "hbm", "kernel": "compress_blocks_global_table_kernel + compress_blocks_lds_table_kernel (co-running pair)", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6),
                         "traffic": traffic, "lds_table_block_share": round(share, 4),
                         "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": round(c_ms, 4),
                         "decompress_kernel": {"achieved": round(algo_bytes / (d_ms * 1e-3) / 1e9, 3),
                                               "avg_launch_ms": round(d_ms, 4)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(batch, torch, gpu_stream0)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("round trip mismatch")


if __name__ == "__main__":
    main()
