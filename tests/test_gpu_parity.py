"""GPU parity tests (-m gpu): the HIP path, called through the C ABI of libsnappy_hip.so, against the oracle
and the reference's golden vectors.  Bit-exact: identical .snappy bytes on compress, identical plaintext on
decompress."""
import hashlib

import numpy as np
import os

import pytest

import datagen
import oracle_lib as oracle
from conftest import GOLDEN_PAIRS, XML_TXT_LEN, XML_TXT_SHA256, golden_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shb():
    import torch
    import __graft_entry__ as entry
    entry.build_hip()
    import snappy_hip_binding as binding
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert binding.lib().snappy_hip_device_count() >= 1
    return binding


def to_dev(data):
    import torch
    a = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(0, dtype=np.uint8)
    t = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
    if len(data):
        t[:len(data)] = torch.from_numpy(a).cuda()
    return t


def gpu_compress(shb, data, bs):
    d = to_dev(data)
    return bytes(shb.compress_resident(d, bs, n=len(data)).cpu().numpy())


def gpu_decompress(shb, stream):
    d = to_dev(stream)
    st, out = shb.decompress_resident(d, stream_len=len(stream))
    return st, bytes(out.cpu().numpy())


# ---- reference golden vectors (reference snappy/Makefile:54-56 does the decompress half) -------------

@pytest.mark.parametrize("name", GOLDEN_PAIRS)
def test_decompress_golden(shb, name):
    st, out = gpu_decompress(shb, golden_bytes(name + ".snappy"))
    assert st == 0
    assert out == golden_bytes(name + ".txt")


@pytest.mark.parametrize("name", GOLDEN_PAIRS)
def test_compress_golden(shb, name):
    assert gpu_compress(shb, golden_bytes(name + ".txt"), 32768) == golden_bytes(name + ".snappy")


def test_xml_golden_both_directions(shb):
    s = golden_bytes("xml.snappy")
    st, out = gpu_decompress(shb, s)
    assert st == 0 and len(out) == XML_TXT_LEN
    assert hashlib.sha256(out).hexdigest() == XML_TXT_SHA256
    assert gpu_compress(shb, out, 32768) == s


# ---- drop-in pair with host buffers (what dpu_snappy.c's main() calls) --------------------------------

@pytest.mark.parametrize("name", ["alice", "terror2", "world192"])
def test_dropin_pair_host_buffers(shb, name):
    txt, snp = golden_bytes(name + ".txt"), golden_bytes(name + ".snappy")
    st, stream, rt = shb.compress_host(txt, 32768)
    assert st == 0 and stream == snp
    assert all(v >= 0 for v in rt.values()) and rt["run"] > 0 and rt["copy_in"] > 0
    st, plain, rt = shb.decompress_host(snp)
    assert st == 0 and plain == txt
    assert rt["run"] > 0


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_dropin_pair_sharding_and_concat(shb, shards, monkeypatch):
    """Contiguous block ranges per device + host-side concat (reference snappy_compress.c:494-520, :697-704):
    more shards than devices are mapped round-robin onto the available GPU(s), output must not change."""
    monkeypatch.setenv("SNAPPY_HIP_NUM_GPUS", str(shards))
    monkeypatch.setenv("SNAPPY_HIP_OVERSUBSCRIBE", "1")
    for name in ("terror2", "plrabn12", "world192"):
        txt, snp = golden_bytes(name + ".txt"), golden_bytes(name + ".snappy")
        st, stream, _ = shb.compress_host(txt, 32768)
        assert st == 0 and stream == snp, (name, shards)
        st, plain, _ = shb.decompress_host(snp)
        assert st == 0 and plain == txt, (name, shards)
    data = datagen.text_random_interleave(golden_bytes("plrabn12.txt"), 1_000_003)
    for bs in (4097, 65535):
        ref = oracle.compress(data, bs, threads=8)
        st, stream, _ = shb.compress_host(data, bs)
        assert st == 0 and stream == ref, (bs, shards)
        st, plain, _ = shb.decompress_host(ref)
        assert st == 0 and plain == data, (bs, shards)


@pytest.mark.parametrize("chunk_blocks,shards", [(16, 1), (8, 3), (32, 3), (1024, 1), (0, 1)])
def test_dropin_pair_overlapped_pipeline(shb, chunk_blocks, shards, monkeypatch):
    """copy-in / kernels / copy-out overlapped chunk by chunk.  Chunks are whole blocks and the host
    concatenates them like per-device outputs (snappy_compress.c:697-704), so the stream is the oracle's whatever the
    chunking; 0 = the strictly phased form."""
    monkeypatch.setenv("SNAPPY_HIP_PIPELINE_BLOCKS", str(chunk_blocks))
    monkeypatch.setenv("SNAPPY_HIP_NUM_GPUS", str(shards))
    monkeypatch.setenv("SNAPPY_HIP_OVERSUBSCRIBE", "1")
    data = datagen.text_random_interleave(golden_bytes("world192.txt"), 3_000_017)
    for bs in (32768, 1000, 65535):
        ref = oracle.compress(data, bs, threads=8)
        st, stream, rt = shb.compress_host(data, bs)
        assert st == 0 and stream == ref, (bs, chunk_blocks, shards)
        assert all(v >= 0 for v in rt.values()) and rt["run"] > 0
        st, plain, rt = shb.decompress_host(ref)
        assert st == 0 and plain == data, (bs, chunk_blocks, shards)
        assert all(v >= 0 for v in rt.values()) and rt["run"] > 0
    # caller-owned output buffer of exactly the right size, and one byte short
    ref = oracle.compress(data, 32768, threads=8)
    st, stream, _ = shb.compress_host(data, 32768, out_capacity=len(ref))
    assert st == 0 and stream == ref
    st, _, _ = shb.compress_host(data, 32768, out_capacity=len(ref) - 1)
    assert st == 2                                                   # SNAPPY_BUFFER_TOO_SMALL (dpu_snappy.h:21-25)
    # a stream that outgrows the reference's output bound (tiny blocks: 9 bytes per 4): the callee grows its buffer
    small = data[:100_000]
    ref = oracle.compress(small, 4, threads=8)
    assert len(ref) > 32 + len(small) + len(small) // 6
    st, stream, _ = shb.compress_host(small, 4)
    assert st == 0 and stream == ref
    st, plain, _ = shb.decompress_host(ref)
    assert st == 0 and plain == small
    # malformed block somewhere in a late chunk; a size chain that leaves the stream or stops short of its end
    good = oracle.compress(data, 32768, threads=8)
    ref = bytearray(good)
    ref[len(ref) - 20000] ^= 0xFF
    st, plain, _ = shb.decompress_host(bytes(ref))
    assert st != 0 or plain != data
    st, _, _ = shb.decompress_host(good[:-3])
    assert st != 0
    st, _, _ = shb.decompress_host(good + b"\0\0\0\0\0")
    assert st != 0
    st, plain, _ = shb.decompress_host(good)
    assert st == 0 and plain == data


@pytest.mark.parametrize("bs", [1, 7, 16])
@pytest.mark.parametrize("shards", [2, 5])
def test_dropin_multi_shard_output_growth_tiny_blocks(shb, bs, shards, monkeypatch):
    """Tiny blocks outgrow the reference's 32+n+n/6 output bound (snappy_compress.c:446-449: 4-byte prefix + literal header
    per block), so the callee-owned buffer is re-allocated in the middle of the multi-shard copy-out (post-join loop):
    every shard's pending device-to-host copies must have landed first.  Several chunks per shard, shards oversubscribed
    onto the available device(s)."""
    monkeypatch.setenv("SNAPPY_HIP_NUM_GPUS", str(shards))
    monkeypatch.setenv("SNAPPY_HIP_OVERSUBSCRIBE", "1")
    monkeypatch.setenv("SNAPPY_HIP_PIPELINE_BLOCKS", "4096")
    data = datagen.text_random_interleave(golden_bytes("plrabn12.txt"), 200_003 if bs == 1 else 600_011)
    ref = oracle.compress(data, bs, threads=8)
    assert len(ref) > 32 + len(data) + len(data) // 6
    st, stream, rt = shb.compress_host(data, bs)
    assert st == 0 and stream == ref, (bs, shards)
    assert rt["run"] > 0 and rt["copy_in"] >= 0
    st, plain, _ = shb.decompress_host(ref)
    assert st == 0 and plain == data, (bs, shards)


def test_dropin_hostile_header_is_rejected_not_fatal(shb):
    """A header that promises 4 Gi blocks of one byte (total 0xffffffff, block size 1) on a 30-byte stream: every block needs
    its u32 prefix, so the stream cannot hold them -- SNAPPY_INVALID_INPUT, no allocation sized by the header."""
    hostile = bytes([0xff, 0xff, 0xff, 0xff, 0x0f, 0x01]) + bytes(24)
    st, _, _ = shb.decompress_host(hostile, out_len_override=16)
    assert st == 1


def test_concurrent_decode_launches_on_many_streams(shb):
    """Every K2 launch owns its work counter: 96 launches on 12 streams in flight at once (more than the 64 slots the
    counter ring of round 1 had) decode every block, and a status word never reads back as OK for a block nobody decoded."""
    import torch
    data = datagen.text_random_interleave(golden_bytes("world192.txt"), 2_000_003)
    stream = oracle.compress(data, 4096, threads=8)
    total, bs, hdr = shb.parse_header(stream[:10])
    nb = shb.num_blocks(total, bs)
    d_stream = to_dev(stream)
    st, plain = shb.decompress_resident(d_stream, stream_len=len(stream))
    assert st == 0 and bytes(plain.cpu().numpy()) == data
    offs = np.zeros(nb, dtype=np.int64)
    at = hdr
    for i in range(nb):
        offs[i] = at
        at += 4 + int.from_bytes(stream[at:at + 4], "little")
    d_offs = torch.from_numpy(offs).cuda()
    streams = [torch.cuda.Stream() for _ in range(12)]
    outs = [torch.zeros(total + 16, dtype=torch.uint8, device="cuda") for _ in range(12)]
    stats = [torch.zeros(nb, dtype=torch.int32, device="cuda") for _ in range(12)]
    torch.cuda.synchronize()
    for rep in range(8):
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                shb.decompress_blocks(d_stream, len(stream), d_offs, total, bs, outs[k], stats[k])
    torch.cuda.synchronize()
    want = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    for k in range(12):
        assert int((stats[k] != 0).sum()) == 0, k
        assert torch.equal(outs[k][:total], want), k


# ---- BASELINE.json configs 3 and 4 (stand-ins of the absent Silesia files, tests/datagen.py) -------------------------

def _prose():
    return [golden_bytes(n + ".txt") for n in ("plrabn12", "world192", "terror2", "alice")]


def test_baseline_config3_dickens_like(shb):
    """"dickens.txt (10 MB) compress+decompress on 1 MI355X": 10,192,446 bytes of shuffled prose, 312 blocks."""
    data = datagen.dickens_like(_prose())
    assert len(data) == datagen.DICKENS_LIKE_BYTES and shb.num_blocks(len(data), 32768) == 312
    ref = oracle.compress(data, 32768, threads=8)
    assert gpu_compress(shb, data, 32768) == ref
    st, out = gpu_decompress(shb, ref)
    assert st == 0 and out == data


@pytest.mark.parametrize("gpus", [1, 2, 4, 8])
def test_baseline_config4_block_sharded(shb, gpus, monkeypatch):
    """"mozilla (51 MB) + spamfile (84 MB) block-sharded across 1/2/4/8 MI355X, host-side concat": contiguous block
    ranges per device through the drop-in pair (more shards than devices are mapped round-robin onto the GPU(s) present)."""
    monkeypatch.setenv("SNAPPY_HIP_NUM_GPUS", str(gpus))
    monkeypatch.setenv("SNAPPY_HIP_OVERSUBSCRIBE", "1")
    st, xml = oracle.decompress(golden_bytes("xml.snappy"))
    assert st == 0
    mozilla = datagen.mozilla_like(xml)
    spamfile = datagen.spamfile_like(_prose()[:2] + [golden_bytes("coding.txt")])
    assert (shb.num_blocks(len(mozilla), 32768), shb.num_blocks(len(spamfile), 32768)) == (1564, 2571)
    for name, data in (("mozilla_like", mozilla), ("spamfile_like", spamfile)):
        ref = oracle.compress(data, 32768, threads=8)
        st, stream, rt = shb.compress_host(data, 32768)
        assert st == 0 and stream == ref, (name, gpus)
        st, plain, rt = shb.decompress_host(ref)
        assert st == 0 and plain == data, (name, gpus)


def test_dropin_rejects_bad_block_size_and_streams(shb):
    st, _, _ = shb.compress_host(b"x" * 100, 0)
    assert st != 0
    st, _, _ = shb.compress_host(b"x" * 100, 65536)
    assert st != 0
    good = golden_bytes("alice.snappy")
    st, _, _ = shb.decompress_host(good[:-3])
    assert st != 0
    bad = bytearray(good)
    bad[9] = 0xFE   # corrupt the first tag into a copy-2 with nothing before it
    st, _, _ = shb.decompress_host(bytes(bad))
    assert st != 0
    st, out, _ = shb.decompress_host(oracle.compress(b"", 32768))
    assert st == 0 and out == b""


# ---- oracle parity on edge cases and every block size -------------------------------------------------

def test_edges_all_block_sizes_vs_oracle(shb):
    text = golden_bytes("plrabn12.txt")
    for name, data in datagen.edge_cases(text):
        for bs in datagen.BLOCK_SIZES:
            if len(data) > 80_000 and bs < 1000:
                continue
            ref = oracle.compress(data, bs)
            got = gpu_compress(shb, data, bs) if len(data) else ref
            assert got == ref, (name, bs)
            st, out = gpu_decompress(shb, ref)
            assert st == 0 and out == data, (name, bs)


def test_unaligned_and_odd_block_sizes(shb):
    data = golden_bytes("world192.txt")[:300_001]
    for bs in (65535, 32767, 12345, 4097, 999, 77):
        ref = oracle.compress(data, bs, threads=8)
        assert gpu_compress(shb, data, bs) == ref, bs
        st, out = gpu_decompress(shb, ref)
        assert st == 0 and out == data, bs


def test_decoder_strictness(shb):
    body = bytes([0x00, 0x41, (3 << 2) | 2, 9, 0])                      # offset before block start
    s1 = bytes([5, 0x80, 0x80, 0x02]) + len(body).to_bytes(4, "little") + body
    body = bytes([0x10, 0x41])                                          # literal overruns the block
    s2 = bytes([5, 0x80, 0x80, 0x02]) + len(body).to_bytes(4, "little") + body
    body = bytes([0x00, 0x41, (3 << 2) | 2, 0, 0])                      # zero offset
    s3 = bytes([5, 0x80, 0x80, 0x02]) + len(body).to_bytes(4, "little") + body
    for s in (s1, s2, s3):
        st, _ = gpu_decompress(shb, s)
        assert st == 1
    # COPY_4 elements are accepted (the compressor never emits them, snappy_decompress.c:278-283)
    body = bytes([0x0C, 0x61, 0x62, 0x63, 0x64, (3 << 2) | 3, 4, 0, 0, 0])
    s4 = bytes([8, 0x80, 0x80, 0x02]) + len(body).to_bytes(4, "little") + body
    st, out = gpu_decompress(shb, s4)
    assert st == 0 and out == b"abcdabcd"
    assert oracle.decompress(s4) == (0, b"abcdabcd")


def test_batched_index_matches_scan_offsets(shb):
    """Two independent computations of the same thing: the compressor's exclusive scan of block sizes and
    the decompressor's walk of the u32 size chain."""
    import torch
    items = [golden_bytes("plrabn12.txt"), golden_bytes("world192.txt"), datagen.random_bytes(100_000)]
    entries, keep = [], []
    for data in items:
        d_in = to_dev(data)
        ws = shb.CompressWorkspace(len(data), 32768)
        d_stream = torch.empty(ws.stream_capacity(len(data)) + 16, dtype=torch.uint8, device="cuda")
        shb.compress_blocks(d_in, len(data), ws)
        shb.compact(len(data), ws, d_stream)
        slen = int(ws.stream_len.item())
        nb = shb.num_blocks(len(data), 32768)
        boff = torch.zeros(nb, dtype=torch.int64, device="cuda")
        res = torch.full((2,), 7, dtype=torch.int32, device="cuda")
        hdr = len(shb.write_header(len(data), 32768))
        entries.append(dict(stream=d_stream, stream_len=slen, block_offsets=boff, result=res, total_len=len(data),
                            block_size=32768, header_len=hdr, num_blocks=nb))
        keep.append((ws, d_stream, boff, res, nb, slen))
    descs = shb.make_stream_descs(entries)
    shb.index_streams(descs, len(entries))
    for ws, d_stream, boff, res, nb, slen in keep:
        assert res.cpu().tolist() == [0, nb]
        assert torch.equal(boff, ws.offsets[:nb])
        assert int(ws.offsets[nb].item()) == slen
        assert int(ws.block_bytes[:nb].sum().item()) + int(ws.offsets[0].item()) == slen


def test_index_streams_parallel_segments_agree_with_the_serial_walk(shb, monkeypatch):
    """snappy_hip_index_streams resolves the size chain in parallel segments and leaves to the serial walk what they cannot
    resolve.  Both ways must give the chain -- on a 300 MB stream (256 segments), with 16-byte blocks (segment
    buffers overflow: serial walk), on a short stream (one segment), on literal payloads full of zero bytes -- and both must
    call a truncated or corrupted stream invalid.  SNAPPY_HIP_INDEX_PARALLEL=0 is the serial walk alone."""
    import torch
    prose = datagen.dickens_like(_prose())
    zero_literals = b"".join(bytes([i & 0xff, 0, 0, 0, 0, 0, (i >> 3) & 0xff, 0]) for i in range(300_000))
    cases = [((prose * 30)[:300_000_000], 32768), (prose[:3_000_000], 16), (golden_bytes("terror2.txt"), 32768), (zero_literals, 32768),
             (datagen.random_bytes(5_000_000), 65535)]
    entries, keep = [], []
    for data, bs in cases:
        stream = oracle.compress(data, bs, threads=8)
        total, got_bs, hdr = oracle.read_header(stream)
        nb = shb.num_blocks(total, got_bs)
        offs, at = [], hdr
        for _ in range(nb):
            offs.append(at)
            at += 4 + int.from_bytes(stream[at:at + 4], "little")
        d_stream = to_dev(stream)
        for damage in (None, "truncate", "size"):
            slen, d = len(stream), d_stream
            if damage == "truncate":
                slen = offs[nb // 2] + 3
            elif damage == "size":
                bad = bytearray(stream)
                bad[offs[nb // 2]] ^= 0x10
                d = to_dev(bytes(bad))
            boff = torch.zeros(nb, dtype=torch.int64, device="cuda")
            res = torch.full((2,), 7, dtype=torch.int32, device="cuda")
            entries.append(dict(stream=d, stream_len=slen, block_offsets=boff, result=res, total_len=total, block_size=got_bs,
                                header_len=hdr, num_blocks=nb))
            keep.append((damage, offs, boff, res, nb))
    for parallel in ("1", "0"):
        monkeypatch.setenv("SNAPPY_HIP_INDEX_PARALLEL", parallel)
        for _, _, boff, res, _ in keep:
            boff.zero_()
            res.fill_(7)
        for lo in range(0, len(entries), 8):                     # (8 streams per call, as bench.py's step)
            shb.index_streams(shb.make_stream_descs(entries[lo:lo + 8]), len(entries[lo:lo + 8]))
        torch.cuda.synchronize()
        for damage, offs, boff, res, nb in keep:
            st, found = res.cpu().tolist()
            if damage is None:
                assert (st, found) == (0, nb) and boff.cpu().tolist() == offs, (parallel, nb)
            else:
                assert st != 0, (parallel, damage, nb)


def test_seeded_fuzz_vs_oracle(shb):
    r = np.random.default_rng(2026)
    for seed in range(48):
        n = int(r.integers(1, 260_000))
        bs = int(r.choice([64, 100, 4096, 32768, 32769, 65535]))
        if bs < 1000:
            n = min(n, 30_000)
        data = datagen.lz_structured(n, seed)
        ref = oracle.compress(data, bs)
        assert gpu_compress(shb, data, bs) == ref, (seed, n, bs)
        st, out = gpu_decompress(shb, ref)
        assert st == 0 and out == data, (seed, n, bs)


@pytest.mark.parametrize("flavour", [0, 1, 2, 3])
def test_decoder_on_random_element_streams(shb, flavour):
    """Valid streams that no greedy compressor writes (datagen.element_stream: copies shorter than 4, 4-byte offsets,
    non-minimal literal headers, chains of short-offset copies, maximal expansion): K2 must decode them exactly as the oracle
    -- the reference's decoder restated -- does."""
    r = np.random.default_rng(77 + flavour)
    for seed in range(10):
        bs = int(r.choice([64, 700, 4097, 20000, 32768, 65535]))
        n = int(r.integers(1, 400_000)) if bs >= 4097 else int(r.integers(1, 40_000))
        stream, plain = datagen.element_stream(n, bs, 5000 * flavour + seed, flavour)
        st, ref = oracle.decompress(stream)
        assert st == 0 and ref == plain, (flavour, seed)
        st, out = gpu_decompress(shb, stream)
        assert st == 0 and out == plain, (flavour, seed, n, bs)


def test_decoder_agrees_with_oracle_on_damaged_element_streams(shb):
    """The same streams with bytes overwritten: whatever the damage, K2 never faults, and where both it and the oracle accept
    the stream they produce the same bytes.  (K2 is strict where the reference would read outside the block; acceptance may
    differ only in that direction.)"""
    r = np.random.default_rng(4242)
    accepted = 0
    for seed in range(60):
        bs = int(r.choice([700, 4097, 32768]))
        stream, _ = datagen.element_stream(int(r.integers(2_000, 60_000)), bs, 9000 + seed, seed % 4)
        _, _, hdr = oracle.read_header(stream)
        b = bytearray(stream)
        for _ in range(int(r.integers(1, 4))):
            at = int(r.integers(hdr + 4, len(b)))         # element bytes (and later size words), never the two header varints
            b[at] = int(r.integers(0, 256))
        try:
            st_ref, ref = oracle.decompress(bytes(b))
        except ValueError:
            continue
        st, out = gpu_decompress(shb, bytes(b))
        assert st in (0, 1)
        if st == 0:
            assert st_ref == 0 and out == ref, seed      # K2 accepted: the oracle must have, with the same bytes
            accepted += 1
    assert accepted > 0                                   # damage inside a literal's payload leaves a valid stream


# ---- every kernel variant produces the same bytes ------------------------------------------------------

# The shipped K1 / K2 set: the concurrent launch (global-table + LDS-table kernels), each kernel alone, tiny grids, the bulk
# and the stream form of the parse, with and without the slot cache.  (The other kernel forms of rounds 1-3 were removed in
# round 4; profiles/HISTORY.md.)
TINY_HYBRID = {"SNAPPY_HIP_LDS_WAVES": "5", "SNAPPY_HIP_GT_WAVES": "11", "SNAPPY_HIP_HYBRID_MIN_BLOCKS": "1"}


@pytest.mark.parametrize("env", [{"SNAPPY_HIP_COMPRESS_VARIANT": "1"},
                                 {"SNAPPY_HIP_COMPRESS_VARIANT": "3", "SNAPPY_HIP_GT_WAVES": "7"},
                                 {"SNAPPY_HIP_LDS_WAVES": "0"}, {"SNAPPY_HIP_LDS_WAVES": "1024"}, TINY_HYBRID,
                                 {"SNAPPY_HIP_K2_WAVES": "5"},
                                 # the stream form (snappy_k1_stream.hpp) and the bulk form, for either kind of table
                                 {"SNAPPY_HIP_K1_STREAM": "0"}, {"SNAPPY_HIP_K1_STREAM": "1"}, {"SNAPPY_HIP_K1_STREAM": "2"},
                                 {"SNAPPY_HIP_K1_STREAM": "3"}, {"SNAPPY_HIP_K1_STREAM": "3", "SNAPPY_HIP_COMPRESS_VARIANT": "1"},
                                 {"SNAPPY_HIP_K1_STREAM": "3", "SNAPPY_HIP_LDS_WAVES": "0"}, dict(TINY_HYBRID, SNAPPY_HIP_K1_STREAM="3"),
                                 # the global-table kernel behind its slot cache (default for blocks of more than 8 KiB) in the
                                 # bulk form, and without the cache in either form
                                 {"SNAPPY_HIP_LDS_WAVES": "0", "SNAPPY_HIP_K1_STREAM": "1"},
                                 {"SNAPPY_HIP_LDS_WAVES": "0", "SNAPPY_HIP_GT_CACHE": "0"},
                                 {"SNAPPY_HIP_LDS_WAVES": "0", "SNAPPY_HIP_GT_CACHE": "0", "SNAPPY_HIP_K1_STREAM": "3"},
                                 dict(TINY_HYBRID, SNAPPY_HIP_GT_CACHE="0")])
def test_kernel_variants_bit_exact(shb, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    text = golden_bytes("plrabn12.txt")
    inputs = [golden_bytes("world192.txt"), datagen.text_random_interleave(text, 200_000), datagen.zeros(100_000),
              datagen.periodic(70_000, 7), datagen.random_bytes(150_000), datagen.records(120_000)]
    for data in inputs:
        for bs in (32768, 65535, 4097, 600):
            ref = oracle.compress(data, bs, threads=8)
            assert gpu_compress(shb, data, bs) == ref, (env, bs)
            st, out = gpu_decompress(shb, ref)
            assert st == 0 and out == data, (env, bs)


def test_batched_launch_matches_per_container_oracle(shb):
    """snappy_hip_compress_blocks_batch: several containers (ragged, one empty, one tiny) in one launch, each stream == oracle."""
    import torch
    text = golden_bytes("plrabn12.txt")
    datas = [golden_bytes("world192.txt"), datagen.text_random_interleave(text, 300_000), b"", datagen.records(77_777),
             b"abc", datagen.lz_structured(200_000, 5), golden_bytes("coding.txt"),
             datagen.zeros(131072), datagen.random_bytes(65_537), datagen.periodic(50_000, 9), datagen.records(12_345)]
    for bs in (32768, 4097):
        jobs, keep = [], []
        for data in datas:
            d = to_dev(data) if len(data) else torch.empty(16, dtype=torch.uint8, device="cuda")
            ws = shb.CompressWorkspace(max(len(data), 1), bs)
            jobs.append((d, len(data), ws))
            keep.append(d)
        for env in ({}, {"SNAPPY_HIP_HYBRID_MIN_BLOCKS": "1", "SNAPPY_HIP_LDS_WAVES": "5", "SNAPPY_HIP_GT_WAVES": "11"}):
            for k, v in env.items():
                os.environ[k] = v
            try:
                shb.compress_blocks_batch(jobs)
            finally:
                for k in env:
                    os.environ.pop(k, None)
            for data, (d, n, ws) in zip(datas, jobs):
                if n == 0:
                    continue
                out = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device="cuda")
                shb.compact(n, ws, out)
                slen = int(ws.stream_len.item())
                assert bytes(out[:slen].cpu().numpy()) == oracle.compress(data, bs, threads=4), (bs, n)


def test_verify_index_on_device(shb):
    """snappy_hip_verify_index on the GPU: the offsets snappy_hip_compact produced pass for every stream of a batch; a
    candidate with one wrong entry, a candidate for a stream whose size field was changed, and a wrong stream length are
    rejected -- and snappy_hip_index_streams then finds what verification accepted."""
    import torch
    text = golden_bytes("plrabn12.txt")
    datas = [datagen.text_random_interleave(text, 700_001), golden_bytes("world192.txt"), datagen.records(300_000)]
    bs = 4096
    entries, keep = [], []
    for d in datas:
        n = len(d)
        ws = shb.CompressWorkspace(n, bs)
        d_stream = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device="cuda")
        shb.compress_blocks(to_dev(d), n, ws)
        shb.compact(n, ws, d_stream)
        slen = int(ws.stream_len.item())
        assert bytes(d_stream[:slen].cpu().numpy()) == oracle.compress(d, bs)
        nb = shb.num_blocks(n, bs)
        res = torch.full((2,), 7, dtype=torch.int32, device="cuda")
        hdr = len(shb.write_header(n, bs))
        entries.append(dict(stream=d_stream, stream_len=slen, block_offsets=ws.offsets, result=res, total_len=n, block_size=bs,
                            header_len=hdr, num_blocks=nb))
        keep.append((ws, d_stream, res, nb, slen))
    descs = shb.make_stream_descs(entries)
    shb.verify_index(descs, len(entries))
    for ws, _, res, nb, _ in keep:
        assert res.cpu().tolist() == [0, nb]
    # one wrong entry in stream 1's candidate; a changed size field in stream 2; a wrong length for stream 0
    keep[1][0].offsets[keep[1][3] // 2] += 1
    at = int(keep[2][0].offsets[3].item())
    keep[2][1][at] ^= 1
    entries[0]["stream_len"] = keep[0][4] - 1
    descs = shb.make_stream_descs(entries)
    shb.verify_index(descs, len(entries))
    for ws, _, res, nb, _ in keep:
        st, links = res.cpu().tolist()
        assert st != 0 and links < nb
    # the serial walk agrees with the accepted candidate (stream 1, whose candidate was only perturbed after verification)
    boff = torch.zeros(keep[1][3] + 1, dtype=torch.int64, device="cuda")
    entries[1]["block_offsets"] = boff
    shb.index_streams(shb.make_stream_descs([entries[1]]), 1)
    good = keep[1][0].offsets.clone()
    good[keep[1][3] // 2] -= 1
    assert torch.equal(boff[:keep[1][3]], good[:keep[1][3]]) and keep[1][2].cpu().tolist() == [0, keep[1][3]]


def test_batched_decode_launch_matches_per_stream(shb):
    """snappy_hip_decompress_blocks_batch: several streams of different lengths (one empty) decoded by one launch, each
    into its own output and status arrays, equal the plaintexts; a corrupt block in one stream is reported in that stream's
    status only."""
    import torch
    text = golden_bytes("plrabn12.txt")
    datas = [golden_bytes("world192.txt"), b"", datagen.text_random_interleave(text, 500_003), datagen.records(70_001),
             datagen.random_bytes(33_000)]
    bs = 4096
    streams = [oracle.compress(d, bs) for d in datas]
    bad = bytearray(streams[3])
    at = list(oracle.index_blocks(streams[3]))[5] + 4          # first element of block 5 becomes "copy 64 bytes from 65535 back"
    bad[at:at + 3] = bytes([0xFE, 0xFF, 0xFF])
    streams.append(bytes(bad))
    datas.append(datas[3])
    jobs, keep = [], []
    for d, s in zip(datas, streams):
        total, got_bs, hdr = shb.parse_header(s[:10])
        nb = shb.num_blocks(total, got_bs)
        offs = np.zeros(max(nb, 1), dtype=np.int64)
        at = hdr
        for i in range(nb):
            offs[i] = at
            at += 4 + int.from_bytes(s[at:at + 4], "little")
        d_stream, d_offs = to_dev(s), torch.from_numpy(offs).cuda()
        d_out = torch.zeros(total + 16, dtype=torch.uint8, device="cuda")
        d_status = torch.full((max(nb, 1),), 7, dtype=torch.int32, device="cuda")
        jobs.append((d_stream, len(s), d_offs, total, d_out, d_status))
        keep.append((d, nb))
    shb.decompress_blocks_batch(jobs, bs)
    torch.cuda.synchronize()
    for k, ((d, nb), job) in enumerate(zip(keep, jobs)):
        out, status = bytes(job[4][:len(d)].cpu().numpy()), job[5][:nb].cpu().numpy()
        if k < len(jobs) - 1:
            assert out == d and (status == 0).all(), k
        else:
            assert status[5] != 0 and (np.delete(status, 5) == 0).all()              # the corrupt block, and only that one


def test_compress_without_scratch_uses_lds_table_kernel(shb):
    import torch
    data = golden_bytes("world192.txt")
    d = to_dev(data)
    ws = shb.CompressWorkspace(len(data), 32768)
    ws.scratch_ptr, ws.scratch_bytes = 0, 0        # no scratch: the library must still compress on the GPU
    d_stream = torch.empty(ws.stream_capacity(len(data)) + 16, dtype=torch.uint8, device="cuda")
    shb.compress_blocks(d, len(data), ws)
    shb.compact(len(data), ws, d_stream)
    n = int(ws.stream_len.item())
    assert bytes(d_stream[:n].cpu().numpy()) == golden_bytes("world192.snappy")


def test_lds_form_block_count_is_written_by_every_launch_shape(shb, monkeypatch):
    """The statistics word of the scratch (include/snappy_hip.h: blocks compressed by LDS-table wavefronts) must be written by
    EVERY launch shape -- bench.py prices roofline.traffic with it.  A small input goes to the LDS-table kernel alone
    (share 1.0); the global-table kernel alone has share 0; the mix lies in between; and a small launch after a large one must
    not read back the large one's count."""
    import torch
    if os.environ.get("SNAPPY_HIP_TEST_DEVICE_CUS"):
        pytest.skip("the expectations below are those of a whole MI355X (312 blocks <= 4 x 256 LDS-table wavefronts)")
    prose = datagen.dickens_like(_prose())                        # 312 blocks: the small-input rule applies
    big = (prose * 17)[:5000 * 32768]                              # 5000 blocks: the concurrent launch
    ws = shb.CompressWorkspace(len(big), 32768)
    d_big, d_small = to_dev(big), to_dev(prose)

    def run(d, n):
        shb.compress_blocks(d, n, ws)
        torch.cuda.synchronize()
        return ws.lds_form_blocks(), shb.num_blocks(n, 32768)

    got, nb = run(d_big, len(big))
    assert nb == 5000 and 0 < got < nb                            # both kernels took blocks
    got, nb = run(d_small, len(prose))
    assert nb == 312 and got == 312                               # every block on an LDS-table wavefront, and not 5000's count
    monkeypatch.setenv("SNAPPY_HIP_LDS_WAVES", "0")
    got, nb = run(d_big, len(big))
    assert got == 0                                               # the global-table kernel alone
    got, nb = run(d_small, len(prose))                            # (SNAPPY_HIP_LDS_WAVES set: no small-input shortcut)
    assert got == 0
    monkeypatch.delenv("SNAPPY_HIP_LDS_WAVES")
    monkeypatch.setenv("SNAPPY_HIP_COMPRESS_VARIANT", "1")
    got, nb = run(d_big, len(big))
    assert got == nb == 5000


def test_block_with_three_lanes_in_one_slot_cache_word(shb, monkeypatch):
    """The benchmark block in which three lanes of one commit meet in one word of the slot cache (see the emulator test of the
    same fixture): the cached global-table kernel alone, both forms, must produce the oracle's bytes."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "mix_block_three_lanes_in_a_cache_word.bin"), "rb") as f:
        data = f.read() * 3
    ref = oracle.compress(data, 32768)
    monkeypatch.setenv("SNAPPY_HIP_LDS_WAVES", "0")
    for stream_form in ("3", "1"):
        monkeypatch.setenv("SNAPPY_HIP_K1_STREAM", stream_form)
        assert gpu_compress(shb, data, 32768) == ref, stream_form


def test_launches_sized_for_a_partition_of_the_chip():
    """The launches and the hash-table scratch are sized from the device's properties (csrc/launch_shape.hpp).  With
    SNAPPY_HIP_TEST_DEVICE_CUS=32 the library believes it runs on a 32-CU partition (CPX mode): the scratch it asks for is an
    eighth, its grids are an eighth, and the bytes are the oracle's -- in a process of its own, because the shape is read once
    per device."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "pim-compression_amd")); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
import oracle_lib as oracle, snappy_hip_binding as shb, datagen
from conftest import golden_bytes
assert shb.lib().snappy_hip_compress_scratch_bytes() == 256 + 32 * 32 * 65536, shb.lib().snappy_hip_compress_scratch_bytes()
prose = datagen.dickens_like([golden_bytes(n + ".txt") for n in ("plrabn12", "world192", "terror2", "alice")])
data = (prose * 7)[:2000 * 32768 + 777]                      # 2,001 blocks: more than the 1,024 wavefront slots of 32 CUs
ref = oracle.compress(data, 32768, threads=8)
t = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
t[:len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
for env in ({}, {"SNAPPY_HIP_HYBRID_MIN_BLOCKS": "1"}, {"SNAPPY_HIP_LDS_WAVES": "0"}, {"SNAPPY_HIP_COMPRESS_VARIANT": "1"}):
    os.environ.update(env)
    got = bytes(shb.compress_resident(t, 32768, n=len(data)).cpu().numpy())
    assert got == ref, env
    for k in env: os.environ.pop(k)
st, out = shb.decompress_resident(torch.from_numpy(np.frombuffer(ref, dtype=np.uint8).copy()).cuda())
assert st == 0 and bytes(out.cpu().numpy()) == data
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SNAPPY_HIP_TEST_DEVICE_CUS="32")
    r = subprocess.run([sys.executable, "-c", code, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]


def test_product_library_refuses_ablation_only_knob_values(shb, monkeypatch):
    """SNAPPY_HIP_GT_CACHE=256 / 1024 and SNAPPY_HIP_K1_STREAM bit 2 exist in the ablation build only; the product library
    must fail loudly instead of running its default under that label."""
    data = golden_bytes("terror2.txt")
    for name, value in (("SNAPPY_HIP_GT_CACHE", "256"), ("SNAPPY_HIP_GT_CACHE", "1024"), ("SNAPPY_HIP_K1_STREAM", "4"),
                        ("SNAPPY_HIP_K1_STREAM", "7")):
        monkeypatch.setenv(name, value)
        with pytest.raises(shb.SnappyHipError):
            gpu_compress(shb, data, 32768)
        monkeypatch.delenv(name)
    for name, value in (("SNAPPY_HIP_GT_CACHE", "0"), ("SNAPPY_HIP_GT_CACHE", "512"), ("SNAPPY_HIP_K1_STREAM", "3")):
        monkeypatch.setenv(name, value)
        assert gpu_compress(shb, data, 32768) == golden_bytes("terror2.snappy")
        monkeypatch.delenv(name)


# ---- BASELINE.json full size: one 1 GiB Silesia-mix container -----------------------------------------

def test_full_size_container_roundtrip_and_oracle(shb):
    import torch
    import silesia_mix
    st, xml = gpu_decompress(shb, golden_bytes("xml.snappy"))
    assert st == 0 and hashlib.sha256(xml).hexdigest() == XML_TXT_SHA256
    unit = silesia_mix.build_unit(np.frombuffer(xml, dtype=np.uint8), seed=0)
    n = 1 << 30
    d_in = silesia_mix.container_from_unit(torch.from_numpy(unit.copy()).cuda(), n)
    d_stream = shb.compress_resident(d_in, 32768, n=n)
    st, d_out = shb.decompress_resident(d_stream)
    assert st == 0
    assert torch.equal(d_out[:n], d_in[:n])                      # encode -> decode round trip, on device
    saving = 1.0 - d_stream.numel() / n
    assert 0.3 < saving < 0.75, saving
    # whole-container oracle comparison (multi-threaded oracle finishes in seconds)
    host_in = d_in[:n].cpu().numpy()
    ref = oracle.compress(host_in, 32768, threads=16)
    got = d_stream.cpu().numpy()
    assert got.size == len(ref)
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(ref).hexdigest()
