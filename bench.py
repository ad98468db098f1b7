#!/usr/bin/env python3
"""bench.py -- end-to-end GB/s (compress + decompress) on the synthetic Silesia-mix, MI355X.

One "step" = one pass of the hot path over one batch: every container of this rank is compressed (K1 per-block
compress, ONE launch over all the rank's containers, + scan/gather into the framed streams), every stream's index is
checked against its size chain (each link in parallel), and every stream is decompressed (K2, ONE launch too).  Inputs are resident in
HBM when the timed region starts.  One process per GPU; blocks / containers are independent, so ranks share nothing on
the data path (no collective) and rank 0 reports the whole-job aggregate.

Workload = BASELINE.json configs[4], a fixed 8 GiB Silesia-mix as 8 containers of 1 GiB (the format's length field is a
uint32).  --scaling strong (default): the 8 containers are dealt to the ranks, 8/N each -- the headline
"8 GiB at 1/2/4/8 GPUs".  --scaling weak: every rank gets its own 8 containers.
--workload dickens_like|mozilla_like|spamfile_like times one file of BASELINE configs[2]/[3] on one GPU instead.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus N --steps K --warmup W          (no launcher: starts its own N ranks, one per GPU)
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
TOTAL_CONTAINERS = 8            # the 8 GiB batch: 8 x 1 GiB
STREAM_DESC_BYTES = 48          # sizeof(snappy_hip_stream_desc)


# ---------------------------------------------------------------------------------------------------
# sharding / aggregation (pure logic + torch.distributed; covered by the gloo CPU tests)
# ---------------------------------------------------------------------------------------------------

def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


def child_environment(rank, world, port, base=None):
    """Environment of rank `rank` of a job this script starts itself (`python bench.py --gpus N` without a launcher): what
    torch.distributed.run would have set, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    return env


def launched_by_a_launcher(environ=None):
    """True under torch.distributed.run (or any launcher that sets the rank environment): then this process IS a rank."""
    environ = os.environ if environ is None else environ
    return "WORLD_SIZE" in environ or "RANK" in environ


def launch_ranks(world, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher: like the reference's offload entry, which allocates its own N
    devices (snappy_compress.c:535), start the N ranks here -- one child process per GPU, each a plain `python bench.py` with
    the rank environment set.  The parent never touches the GPU (no torch.cuda / HIP call is made before this point and none
    after), nothing re-execs; rank 0's stdout (the ONE JSON line) is relayed, every rank's stderr is inherited, and the exit
    code is non-zero if any rank's is.  A rank that dies takes the others down (they would wait at a barrier for ever)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = os.path.abspath(__file__)
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=child_environment(r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True if r == 0 else None))
    out0 = []
    import threading
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    codes = [None] * world
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):                         # exactly the processes started above, by handle
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    bad = [c for c in codes if c != 0]
    return (bad[0] if bad[0] and bad[0] > 0 else 1) if bad else 0


def shard_plan(rank, world, containers, scaling="strong"):
    """Container ids (global numbering) owned by `rank`: whole containers per GPU.
    strong: `containers` is the size of the whole job, dealt in contiguous ranges of ceil(containers / world) -- the
            partitioning of snappy_compress.c:494-520 at container granularity (a rank may get none);
    weak:   `containers` per rank, whatever the world size."""
    if scaling == "weak":
        return [rank * containers + i for i in range(containers)]
    per = (containers + world - 1) // world
    return list(range(min(containers, rank * per), min(containers, (rank + 1) * per)))


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

def xml_plaintext(shb, torch):
    """The xml plaintext comes from the product's own decoder (tests/golden/xml.snappy), checked by digest."""
    import numpy as np
    import silesia_mix
    with open(os.path.join(ROOT, "tests", "golden", "xml.snappy"), "rb") as f:
        xml_snappy = np.frombuffer(f.read(), dtype=np.uint8).copy()
    st, d_xml = shb.decompress_resident(torch.from_numpy(xml_snappy).cuda())
    xml = d_xml.cpu().numpy()
    if st != 0 or hashlib.sha256(xml.tobytes()).hexdigest() != silesia_mix.XML_TXT_SHA256:
        raise RuntimeError("xml.snappy did not decode to the expected plaintext")
    return xml


def build_inputs(shb, torch, workload, container_ids, container_len):
    """-> list of (device tensor with >= 16 bytes of slack, length)."""
    import numpy as np
    import silesia_mix
    import standins
    if workload == "silesia_mix":
        xml = xml_plaintext(shb, torch)
        out = []
        for cid in container_ids:
            unit = silesia_mix.build_unit(xml, seed=cid)
            out.append((silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), container_len), container_len))
        return out
    if workload == "dickens_like":
        data = standins.dickens_like(standins.prose_texts())
    elif workload == "mozilla_like":
        data = standins.mozilla_like(xml_plaintext(shb, torch).tobytes())
    elif workload == "spamfile_like":
        data = standins.spamfile_like(standins.prose_texts())
    else:
        raise SystemExit(f"unknown workload {workload}")
    t = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
    t[:len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    return [(t, len(data))]


class Batch:
    """Device-resident containers + the buffers one step needs."""

    def __init__(self, shb, torch, inputs, groups=1):
        self.shb, self.torch = shb, torch
        self.inputs = inputs
        self.count = len(inputs)
        self.n = [ln for _, ln in inputs]
        self.nb = [shb.num_blocks(ln, BLOCK_SIZE) for ln in self.n]
        self.hdr = [len(shb.write_header(ln, BLOCK_SIZE)) for ln in self.n]
        self.wss = [shb.CompressWorkspace(ln, BLOCK_SIZE, scratch=(i == 0)) for i, ln in enumerate(self.n)]
        self.streams = [torch.empty(ws.stream_capacity(ln) + 16, dtype=torch.uint8, device="cuda")
                        for ws, ln in zip(self.wss, self.n)]
        self.stream_lens = [0] * self.count
        self.results = torch.zeros(2 * max(1, self.count), dtype=torch.int32, device="cuda")
        self.status = [torch.empty(max(nb, 1), dtype=torch.int32, device="cuda") for nb in self.nb]
        self.outs = [torch.empty(ln + 16, dtype=torch.uint8, device="cuda") for ln in self.n]
        self.kernel_events = {"compress": [], "decompress": []}
        # Stream descriptors, one device array for all containers of the rank.  block_offsets points at the offsets the
        # compressor's own scan writes (num_blocks + 1 entries): the candidate index that snappy_hip_verify_index checks
        # against the stream's size chain, every link in parallel; stream_len is patched on the device from the
        # compressor's device-side result, so the host never waits inside the compress phase.
        self.descs = shb.make_stream_descs([
            dict(stream=self.streams[i], stream_len=0, block_offsets=self.wss[i].offsets,
                 result=self.results[2 * i:2 * i + 2], total_len=self.n[i], block_size=BLOCK_SIZE,
                 header_len=self.hdr[i], num_blocks=self.nb[i]) for i in range(self.count)]) if self.count else None
        self.d_meta = torch.zeros(3 * max(1, self.count), dtype=torch.int64, device="cuda")   # stream_len, result[0..1]
        # "stream alone" mode (a decoder that is handed nothing but the framed streams, snappy_decompress.c:306-341): the
        # size chains are walked on the device, all streams of the rank in ONE launch, into offsets of their own
        self.walk_offsets = [torch.zeros(nb + 1, dtype=torch.int64, device="cuda") for nb in self.nb]
        self.walk_results = torch.zeros(2 * max(1, self.count), dtype=torch.int32, device="cuda")
        self.walk_descs = shb.make_stream_descs([
            dict(stream=self.streams[i], stream_len=0, block_offsets=self.walk_offsets[i],
                 result=self.walk_results[2 * i:2 * i + 2], total_len=self.n[i], block_size=BLOCK_SIZE,
                 header_len=self.hdr[i], num_blocks=self.nb[i]) for i in range(self.count)]) if self.count else None
        self.fallback_walks = 0
        self.steps_run = 0
        self.groups = max(1, min(groups, self.count))
        self.side_streams = [torch.cuda.Stream() for _ in range(2)]
        self.group_done = [torch.cuda.Event() for _ in range(self.groups)]

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

    def step(self, record=False, stream_alone=False):
        """Everything is enqueued without a host round trip: the stream lengths stay on the device (the decoder reads them
        there), and the containers go through in `groups` launches so that the decode of group g (second stream) runs
        underneath the compression of group g+1.  The host reads the index check results once, at the end.
        stream_alone: the decoder gets nothing from the compressor but the framed streams -- the block index comes from the
        streams' own size chains (ONE snappy_hip_index_streams call over the group's streams: parallel segments, the serial
        walk for what they leave) instead of the compressor's offsets checked link by link."""
        shb, torch = self.shb, self.torch
        if not self.count:
            return
        self.steps_run += 1
        main = torch.cuda.current_stream()
        bounds = [(self.count * g) // self.groups for g in range(self.groups + 1)]
        for g in range(self.groups):
            lo, hi = bounds[g], bounds[g + 1]
            if lo == hi:
                continue
            # ---- compress: ONE K1 launch over the group's containers, then framing per container ----
            jobs = [(self.inputs[i][0], self.n[i], self.wss[i]) for i in range(lo, hi)]
            self._timed("compress", record, lambda: shb.compress_blocks_batch(jobs, scratch_ws=self.wss[0]))
            for i in range(lo, hi):
                shb.compact(self.n[i], self.wss[i], self.streams[i])
                at = STREAM_DESC_BYTES * i + 8                   # descriptor field stream_len
                (self.walk_descs if stream_alone else self.descs)[at:at + 8].copy_(self.wss[i].stream_len.view(torch.uint8),
                                                                                   non_blocking=True)
                self.d_meta[3 * i:3 * i + 1].copy_(self.wss[i].stream_len, non_blocking=True)
            if stream_alone:
                # ---- index: the size chains of the group's streams walked on the device, one launch (:317-340) ----
                shb.index_streams(self.walk_descs[STREAM_DESC_BYTES * lo:STREAM_DESC_BYTES * hi], hi - lo)
                offs = self.walk_offsets
            else:
                # ---- index: the compressor's offsets, checked link by link against the size chain of each stream ----
                shb.verify_index(self.descs[STREAM_DESC_BYTES * lo:STREAM_DESC_BYTES * hi], hi - lo)
                offs = [ws.offsets for ws in self.wss]
            # ---- decompress: ONE K2 launch over the group's streams, beside the next group's K1 ----
            djobs = [(self.streams[i], self.wss[i].stream_len, offs[i], self.n[i], self.outs[i], self.status[i])
                     for i in range(lo, hi)]
            last = g == self.groups - 1
            side = main if last else self.side_streams[g % len(self.side_streams)]
            if not last:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                self._timed("decompress", record, lambda: shb.decompress_blocks_batch(djobs, BLOCK_SIZE))
            if not last:
                self.group_done[g].record(side)
        for g in range(self.groups - 1):
            main.wait_event(self.group_done[g])
        self.d_meta.view(self.count, 3)[:, 1:3].copy_((self.walk_results if stream_alone else self.results).view(-1, 2)[:self.count])
        meta = self.d_meta.cpu().tolist()                         # the only host read of the step, after everything is enqueued
        self.stream_lens = [int(meta[3 * i]) for i in range(self.count)]
        for i in range(self.count):
            if stream_alone and (meta[3 * i + 1] != 0 or meta[3 * i + 2] != self.nb[i]):
                raise RuntimeError(f"container {i}: the size chain of the compressed stream is broken")
            if meta[3 * i + 1] != 0 or meta[3 * i + 2] != self.nb[i]:
                # not expected for our own streams: walk the chain serially and decode that stream again
                self.fallback_walks += 1
                one = self.descs[STREAM_DESC_BYTES * i:STREAM_DESC_BYTES * (i + 1)]
                shb.index_streams(one, 1)
                if self.results[2 * i:2 * i + 2].cpu().tolist() != [0, self.nb[i]]:
                    raise RuntimeError(f"container {i}: the size chain of the compressed stream is broken")
                shb.decompress_blocks(self.streams[i], self.stream_lens[i], self.wss[i].offsets, self.n[i], BLOCK_SIZE, self.outs[i],
                                      self.status[i])

    def verify(self):
        """Outside the timed region: every container round-trips bit-exactly through the stream alone (serial walk of the
        size chain, no side index) and every block decoded OK."""
        torch, shb = self.torch, self.shb
        for i, (d_in, ln) in enumerate(self.inputs):
            shb.compress_blocks(d_in, ln, self.wss[0])
            shb.compact(ln, self.wss[0], self.streams[i])
            slen = int(self.wss[0].stream_len.item())
            self.stream_lens[i] = slen
            st, d_out = shb.decompress_resident(self.streams[i][:slen])
            if st != 0 or not torch.equal(d_out[:ln], d_in[:ln]):
                return False
        return True

    def verify_last_step(self):
        """After the timed steps: the plaintext the last step decoded is the input, for every container; every block status
        is OK and no step had to fall back to the serial walk."""
        torch = self.torch
        for i in range(self.count):
            if not torch.equal(self.outs[i][:self.n[i]], self.inputs[i][0][:self.n[i]]):
                return False
            if int((self.status[i][:self.nb[i]] != 0).sum()) != 0:
                return False
        return self.fallback_walks == 0

    def lds_share(self):
        """Fraction of the last K1 launch's blocks compressed by wavefronts whose hash table lives in LDS."""
        return self.wss[0].lds_form_blocks() / max(1, sum(self.nb)) if self.count else 0.0

    def kernel_ms(self, key):
        self.torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.kernel_events[key]]
        return (sum(ms) / len(ms)) if ms else None

    def walk_ms(self, serial=False):
        """snappy_hip_index_streams on container 0's stream, timed outside the step for the record: the size chain resolved in
        parallel segments (default), or -- serial=True, SNAPPY_HIP_INDEX_PARALLEL=0 -- by the serial walk alone."""
        torch, shb = self.torch, self.shb
        if not self.count:
            return None
        if serial:
            prev = os.environ.get("SNAPPY_HIP_INDEX_PARALLEL")
            os.environ["SNAPPY_HIP_INDEX_PARALLEL"] = "0"
            try:
                return self.walk_ms()
            finally:
                if prev is None:
                    os.environ.pop("SNAPPY_HIP_INDEX_PARALLEL", None)
                else:
                    os.environ["SNAPPY_HIP_INDEX_PARALLEL"] = prev
        boff = torch.empty(self.nb[0] + 1, dtype=torch.int64, device="cuda")
        res = torch.zeros(2, dtype=torch.int32, device="cuda")
        d = shb.make_stream_descs([dict(stream=self.streams[0], stream_len=self.stream_lens[0], block_offsets=boff, result=res,
                                        total_len=self.n[0], block_size=BLOCK_SIZE, header_len=self.hdr[0],
                                        num_blocks=self.nb[0])])
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            shb.index_streams(d, 1)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None else min(best, t)
        assert res.cpu().tolist() == [0, self.nb[0]]
        if self.steps_run:                                    # the walk finds what the verified side index holds
            assert torch.equal(boff[:self.nb[0]], self.wss[0].offsets[:self.nb[0]])
        return best


def host_cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        pass
    return model, affinity, quota


def cpu_baseline(batch, torch, gpu_stream_bytes):
    """Oracle (CPU restatement of the reference host path) timed on this box's host cores, on a bounded sample of the same
    workload: container 0 of rank 0.  All buffers are allocated and touched before the clock starts; a timed call is a
    pthread launch per phase -- per-block codec over contiguous block ranges, then a parallel concat (oracle_mt_*).  Thread counts
    tried: every core the process may run on, and (when a cgroup CPU quota is set) the quota; the better one is `value`.
    Plus the first 64 MiB on one core, the reference's own (single-threaded) mode."""
    import numpy as np
    import oracle_lib as oracle
    model, affinity, quota = host_cpu_info()
    n = batch.n[0]
    host = np.ascontiguousarray(batch.inputs[0][0][:n].cpu().numpy())
    reps = 3 if n >= (1 << 28) else 10
    candidates = sorted({max(1, min(256, affinity))} | ({max(1, min(256, int(round(quota))))} if quota else set()))
    runs = {}
    comp_bytes = None
    for threads in candidates:
        ctx = oracle.MtContext(n, BLOCK_SIZE, threads)
        dst = np.empty(ctx.bound, dtype=np.uint8)
        dst.fill(1)                                           # pre-fault
        out = np.empty(max(n, 1), dtype=np.uint8)
        out.fill(1)
        clen = ctx.compress_into(host, dst)                   # warm-up (page tables, caches)
        t_c = t_d = 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            clen = ctx.compress_into(host, dst)
            t1 = time.perf_counter()
            st = ctx.decompress_into(dst, clen, out)
            t2 = time.perf_counter()
            assert st == 0
            t_c += t1 - t0
            t_d += t2 - t1
        assert out[:n].tobytes() == host.tobytes() if n < (1 << 26) else bool((out[:n] == host).all())
        comp_bytes = dst[:clen].tobytes() if comp_bytes is None else comp_bytes
        ctx.close()
        gb = n / 1e9
        runs[threads] = {"e2e": reps * gb / (t_c + t_d), "compress": reps * gb / t_c, "decompress": reps * gb / t_d}
    best = max(runs, key=lambda t: runs[t]["e2e"])
    parity = hashlib.sha256(comp_bytes).hexdigest() == hashlib.sha256(gpu_stream_bytes).hexdigest()
    one = host[:min(n, 64 << 20)]
    t0 = time.perf_counter()
    c1 = oracle.compress(one, BLOCK_SIZE)
    t1 = time.perf_counter()
    oracle.decompress(c1)
    t2 = time.perf_counter()
    single = one.size / 1e9 / (t2 - t0)
    return {
        "value": round(runs[best]["e2e"], 4), "unit": "GB/s", "cores": best, "kind": "port",
        "sample": f"container 0 ({n} B of the same workload), {reps} reps after a warm-up, oracle/snappy_oracle.c "
                  f"(oracle_mt_*: pre-faulted buffers, {best} pthreads over contiguous block ranges, parallel concat)",
        "compress_GBps": round(runs[best]["compress"], 4), "decompress_GBps": round(runs[best]["decompress"], 4),
        "cpu_model": model, "nproc_affinity": affinity, "cgroup_cpu_quota": quota,
        "by_threads": {str(t): {k: round(v, 4) for k, v in r.items()} for t, r in runs.items()},
        "single_core_value": round(single, 4),
        "single_core_sample": f"first {one.size} B of container 0, 1 thread (the reference's own mode)",
        "speedup_over_single_core": round(runs[best]["e2e"] / single, 2),
        "gpu_stream_equals_oracle_stream": bool(parity),
    }


def load_pmc_traffic():
    """HBM bytes per input byte of the K1 forms from the committed rocprofv3 --pmc passes, if present."""
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
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: the fixed 8 GiB batch dealt over the ranks (headline); weak: 8 GiB per rank")
    ap.add_argument("--containers", type=int, default=TOTAL_CONTAINERS, help="containers in the job (strong) / per GPU (weak)")
    ap.add_argument("--container-mib", type=int, default=1024, help="container size in MiB (format limit: < 4096)")
    ap.add_argument("--workload", default="silesia_mix",
                    choices=("silesia_mix", "dickens_like", "mozilla_like", "spamfile_like"),
                    help="silesia_mix = the 8 GiB batch (BASELINE configs[4]); the others time one stand-in file of "
                         "configs[2]/[3] on one GPU")
    ap.add_argument("--groups", type=int, default=1,
                    help="launch groups per step: the decode of group g runs on a second stream underneath the compression of "
                         "group g+1 (1 = strictly compress-all then decompress-all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-alone", action="store_true",
                    help="skip the second timed region (the same steps with the block index walked from the streams alone)")
    ap.add_argument("--no-preverify", action="store_true",
                    help="skip the extra round trip through the serial size-chain walk before the timed steps (its per-container "
                         "launches would mix with the batched ones in a rocprofv3 --stats average); the timed path itself is "
                         "still checked bit for bit after the last step")
    args = ap.parse_args()

    # N > 1 without a launcher: start the N ranks from here (before anything touches the GPU) and relay rank 0's line
    if args.gpus > 1 and not launched_by_a_launcher():
        if args.workload != "silesia_mix":
            raise SystemExit("--workload of one file runs on one GPU (its blocks shard inside snappy_compress_gpu, not here)")
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

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
    if world > 1 or os.environ.get("SNAPPY_BENCH_FORCE_DIST") == "1":     # FORCE_DIST: exercise the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    single_file = args.workload != "silesia_mix"
    if single_file and world > 1:
        raise SystemExit("--workload of one file runs on one GPU (its blocks shard inside snappy_compress_gpu, not here)")

    n = args.container_mib << 20
    plan = [0] if single_file else shard_plan(rank, world, args.containers, args.scaling)
    batch = Batch(shb, torch, build_inputs(shb, torch, args.workload, plan, n), groups=args.groups)
    ok = True if args.no_preverify else batch.verify()
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and batch.count

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
    ok = ok and batch.verify_last_step()
    # ---- the same K steps with a decoder that starts from the streams alone (its own timed region, reported beside `value`):
    #      the block index comes from snappy_hip_index_streams inside the step ----
    elapsed_alone = None
    if not args.no_stream_alone:
        batch.step(stream_alone=True)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            batch.step(stream_alone=True)
        fence()
        elapsed_alone = time.perf_counter() - t0
        ok = ok and batch.verify_last_step()
        for i in range(batch.count):                              # the walk found what the compressor's scan wrote
            ok = ok and torch.equal(batch.walk_offsets[i][:batch.nb[i]], batch.wss[i].offsets[:batch.nb[i]])
    gpu_stream0 = bytes(batch.streams[0][:batch.stream_lens[0]].cpu().numpy()) if want_cpu else b""   # as the timed steps left it

    local_bytes = args.steps * sum(batch.n)
    local_comp = args.steps * sum(batch.stream_lens)
    secs, tot_bytes, tot_comp = reduce_results(elapsed, local_bytes, local_comp, dist,
                                               device="cuda" if backend == "nccl" else "cpu")
    secs_alone = None
    if elapsed_alone is not None:
        secs_alone, _, _ = reduce_results(elapsed_alone, local_bytes, local_comp, dist,
                                          device="cuda" if backend == "nccl" else "cpu")

    if rank == 0:
        c_ms = batch.kernel_ms("compress")
        d_ms = batch.kernel_ms("decompress")
        launches = max(1, batch.groups)                       # K1 / K2 launches per step (equal groups of containers)
        u = sum(batch.n) / launches                           # uncompressed bytes per launch
        c = sum(batch.stream_lens) / launches
        algo_bytes = u + c                                    # read plaintext once, write compressed once
        achieved = algo_bytes / (c_ms * 1e-3) / 1e9
        d_algo = u + c
        pmc = load_pmc_traffic()
        share = batch.lds_share()
        traffic = None
        if pmc and "k1_global_table_bytes_per_input_byte" in pmc:
            # rocprofv3 --pmc serialises the co-running K1 kernels, so HBM bytes were measured for each form running
            # alone and are combined here with the block share the LDS-table form actually took
            traffic = int(u * ((1.0 - share) * pmc["k1_global_table_bytes_per_input_byte"] +
                               share * pmc["k1_lds_table_bytes_per_input_byte"]))
        if single_file:
            workload = (f"{args.workload} {batch.n[0]} B (stand-in for BASELINE configs[2]/[3], "
                        f"pim-compression_amd/standins.py), block_size {BLOCK_SIZE}")
        else:
            workload = (f"silesia_mix {args.containers} x {args.container_mib} MiB containers "
                        f"{'in all' if args.scaling == 'strong' else 'per GPU'} (BASELINE configs[4]), block_size {BLOCK_SIZE}")
        line = {
            "metric": METRIC,
            "value": round(tot_bytes / secs / 1e9, 4),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(secs / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak" if (args.scaling == "weak" and not single_file) else "strong",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": workload,
                       "containers_this_rank": batch.count, "container_bytes": batch.n[0] if batch.count else 0,
                       "block_size": BLOCK_SIZE, "launch_groups_per_step": batch.groups,
                       "parallelism": f"whole containers dealt over {world} GPU(s), no collective"},
            "roundtrip_bit_exact": bool(ok),
            "space_saving": round(1.0 - tot_comp / tot_bytes, 6),
            "compress_kernel_GBps": round(u / (c_ms * 1e-3) / 1e9, 3),
            "decompress_kernel_GBps": round(u / (d_ms * 1e-3) / 1e9, 3),
            # the same steps with the block index taken from the streams alone: ONE snappy_hip_index_streams launch walks the
            # size chains of all the rank's streams inside the step (snappy_decompress.c:306-341), nothing but the framed
            # streams crosses from the compressor to the decoder
            "value_from_stream_alone": round(tot_bytes / secs_alone / 1e9, 4) if secs_alone else None,
            "ms_per_step_from_stream_alone": round(secs_alone / args.steps * 1e3, 3) if secs_alone else None,
            "index": {"mode": "value: the compressor's offsets verified against the size chain, every link in parallel "
                              "(snappy_hip_verify_index); value_from_stream_alone: the chain found on the device from the "
                              "stream's bytes alone, all streams of the rank in one call (snappy_hip_index_streams: 256 "
                              "walkers per stream over segments of the chain, laid end to end iff every segment ends exactly "
                              "on the next one's start; the serial walk for any stream that leaves unresolved)",
                      "fallback_serial_walks": batch.fallback_walks,
                      "index_ms_container0": round(batch.walk_ms(), 3),
                      "serial_walk_ms_container0": round(batch.walk_ms(serial=True), 3)},
            "roofline": {"bound": "hbm", "kernel": "K1 = compress_blocks_global_table_kernel + compress_blocks_lds_table_kernel "
                                                   "(co-running pair, one launch over the rank's containers)",
                         "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6),
                         "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json (offline rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE of each K1 "
                                           "form alone, bytes per input byte) x this run's block share of the two forms",
                         "lds_table_block_share": round(share, 4),
                         # what the kernel IS bound by (offline counter passes, like `traffic`): instruction issue, not HBM bytes
                         "issue": (pmc or {}).get("k1_issue"),
                         "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": round(c_ms, 4),
                         "decompress_kernel": {"achieved": round(d_algo / (d_ms * 1e-3) / 1e9, 3),
                                               "frac": round(d_algo / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                                               "traffic": int(u * pmc["k2_bytes_per_output_byte"])
                                               if pmc and "k2_bytes_per_output_byte" in pmc else None,
                                               "algorithmic_bytes_per_launch": int(d_algo),
                                               "avg_launch_ms": round(d_ms, 4)}},
        }
        if want_cpu:
            line["cpu_baseline"] = cpu_baseline(batch, torch, gpu_stream0)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("round trip mismatch")


if __name__ == "__main__":
    main()
