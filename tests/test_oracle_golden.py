"""CPU tests: the oracle against the reference's committed golden vectors (both directions).

The reference's own regression (snappy/Makefile:54-56) decompresses every test/*.snappy and
`cmp`s with test/*.txt.  The same vectors also pin the compressor, because the committed
.snappy files were produced by the reference host compressor (README.md:25).
"""
import hashlib

import pytest

import datagen
import oracle_lib as oracle
from conftest import GOLDEN_PAIRS, GOLDEN_SHA256, XML_TXT_LEN, XML_TXT_SHA256, golden_bytes


def test_fixture_digests():
    for name, digest in GOLDEN_SHA256.items():
        assert hashlib.sha256(golden_bytes(name)).hexdigest() == digest, name


@pytest.mark.parametrize("name", GOLDEN_PAIRS)
def test_oracle_decompress_golden(name):
    st, out = oracle.decompress(golden_bytes(name + ".snappy"))
    assert st == 0
    assert out == golden_bytes(name + ".txt")


@pytest.mark.parametrize("name", GOLDEN_PAIRS)
def test_oracle_compress_golden(name):
    assert oracle.compress(golden_bytes(name + ".txt"), 32768) == golden_bytes(name + ".snappy")


def test_oracle_xml_both_directions():
    s = golden_bytes("xml.snappy")
    st, out = oracle.decompress(s)
    assert st == 0 and len(out) == XML_TXT_LEN
    assert hashlib.sha256(out).hexdigest() == XML_TXT_SHA256
    assert oracle.compress(out, 32768) == s


@pytest.mark.parametrize("name", GOLDEN_PAIRS)
def test_oracle_mt_drivers_match(name):
    txt, snp = golden_bytes(name + ".txt"), golden_bytes(name + ".snappy")
    assert oracle.compress(txt, 32768, threads=3) == snp
    st, out = oracle.decompress(snp, threads=5)
    assert st == 0 and out == txt


def test_header_and_known_small_streams():
    # SURVEY Appendix A examples
    assert oracle.compress(b"a", 32768) == bytes.fromhex("01808002" "02000000" "0061")
    assert oracle.compress(b"", 32768) == bytes.fromhex("00808002")
    blk = datagen.random_bytes(32768, seed=9)
    c = oracle.compress(blk, 32768)
    total, bs, hdr = oracle.read_header(c)
    assert (total, bs) == (32768, 32768)
    # incompressible block -> one long literal: size = 3 + 32768
    assert c[hdr:hdr + 7] == bytes.fromhex("03800000" "f4ff7f")


def test_table_size_rule():
    L = oracle.lib()
    assert L.oracle_table_size(312) == 512
    assert L.oracle_table_size(7134) == 8192
    assert L.oracle_table_size(32768) == 16384
    assert L.oracle_table_size(65535) == 16384
    assert L.oracle_table_size(1) == 256
    assert L.oracle_table_size(256) == 256
    assert L.oracle_table_size(257) == 512


def test_oracle_roundtrip_edges_and_block_sizes():
    text = golden_bytes("plrabn12.txt")
    for name, data in datagen.edge_cases(text):
        for bs in datagen.BLOCK_SIZES:
            if len(data) > 80_000 and bs < 1000:
                continue
            c = oracle.compress(data, bs)
            total, got_bs, _ = oracle.read_header(c)
            assert (total, got_bs) == (len(data), bs)
            st, out = oracle.decompress(c)
            assert st == 0 and out == data, (name, bs)
            st, out = oracle.decompress(c, threads=4)
            assert st == 0 and out == data, (name, bs)
            assert oracle.compress(data, bs, threads=4) == c


def test_oracle_rejects_bad_offset():
    # copy-2 of length 4 at offset 9 with only 1 byte of output so far
    body = bytes([0x00, 0x41, (3 << 2) | 2, 9, 0])
    stream = bytes([5, 0x80, 0x80, 0x02]) + len(body).to_bytes(4, "little") + body
    st, _ = oracle.decompress(stream)
    assert st == 1


def _decode_raw_snappy(buf):
    """Minimal decoder of the ORIGINAL raw Snappy block format: varint(uncompressed length) + elements."""
    n, shift, i = 0, 0, 0
    while True:
        c = buf[i]
        i += 1
        n |= (c & 0x7f) << shift
        if c < 0x80:
            break
        shift += 7
    out = bytearray()
    while i < len(buf):
        tag = buf[i]
        i += 1
        t = tag & 3
        if t == 0:
            ln = (tag >> 2) + 1
            if ln > 60:
                nb = ln - 60
                ln = int.from_bytes(buf[i:i + nb], "little") + 1
                i += nb
            out += buf[i:i + ln]
            i += ln
            continue
        if t == 1:
            ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | buf[i]
            i += 1
        elif t == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(buf[i:i + 2], "little")
            i += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(buf[i:i + 4], "little")
            i += 4
        for _ in range(ln):
            out.append(out[-off])
    assert len(out) == n
    return bytes(out)


def test_blocks_are_raw_snappy_compatible():
    """Interop (reference snappy/README.md:9-18): a block body prefixed with varint(block length) is a valid stream of
    the original Snappy format, so standard decoders can read individual blocks."""
    data = golden_bytes("terror2.txt")
    stream = oracle.compress(data, 32768)
    total, bs, _ = oracle.read_header(stream)
    offs = oracle.index_blocks(stream)
    out = bytearray()
    for k, at in enumerate(offs):
        at = int(at)
        csz = int.from_bytes(stream[at:at + 4], "little")
        blen = min(bs, total - k * bs)
        varint = bytearray()
        v = blen
        while v >= 0x80:
            varint.append((v & 0x7f) | 0x80)
            v >>= 7
        varint.append(v)
        out += _decode_raw_snappy(bytes(varint) + stream[at + 4:at + 4 + csz])
    assert bytes(out) == data


def test_converter_to_original_snappy_framing():
    """tools/to_raw_snappy.py: the whole framed file becomes one stream of the original format."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("to_raw_snappy", os.path.join(ROOT, "tools", "to_raw_snappy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name in ("alice", "terror2", "plrabn12"):
        raw = mod.convert(golden_bytes(name + ".snappy"))
        assert _decode_raw_snappy(raw) == golden_bytes(name + ".txt")
    for bs in (64, 4096, 65535):
        data = golden_bytes("coding.txt")
        assert _decode_raw_snappy(mod.convert(oracle.compress(data, bs))) == data


def test_converter_from_original_snappy_framing():
    """The opposite direction: a raw stream (here: made from the goldens by the converter above, plus one with elements no
    block-framed file can hold -- a back-reference across 100 KB and an overlapping run) is decoded and re-framed by the
    dpu_snappy tool; for the goldens' plaintext at 32 KiB blocks that must reproduce the reference's own .snappy files."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("to_raw_snappy", os.path.join(ROOT, "tools", "to_raw_snappy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.CLI):
        pytest.skip("dpu_snappy CLI not built")
    for name in ("alice", "terror2", "world192"):
        raw = mod.convert(golden_bytes(name + ".snappy"))
        assert mod.decode_raw(raw) == golden_bytes(name + ".txt")
        assert mod.reframe(raw, 32768) == golden_bytes(name + ".snappy")
    # hand-made raw stream: 100000-byte literal, a COPY_4 reaching back over all of it, an overlapping COPY_1 run
    lit = golden_bytes("plrabn12.txt")[:100000]
    n = len(lit) + 64 + 11
    raw = bytearray(mod.varint(n))
    raw += bytes([62 << 2]) + (len(lit) - 1).to_bytes(3, "little") + lit           # literal with a 3-byte length
    raw += bytes([(63 << 2) | 3]) + (100000).to_bytes(4, "little")                  # COPY_4: 64 bytes from offset 100000
    raw += bytes([((11 - 4) << 2) | 1 | (0 << 5), 3])                               # COPY_1: 11 bytes from offset 3 (overlap)
    plain = mod.decode_raw(bytes(raw))
    assert plain == lit + lit[:64] + ((lit[:64][-3:] * 4)[:11])
    framed = mod.reframe(bytes(raw), 4096)
    assert framed == oracle.compress(plain, 4096)
    st, back = oracle.decompress(framed)
    assert st == 0 and back == plain


def test_baseline_standins_shapes_and_oracle_roundtrip():
    """The stand-ins of BASELINE.json configs 3 and 4 have the named sizes / block counts, are deterministic, and the
    oracle round-trips them (the GPU tests compare against these oracle streams)."""
    prose = [golden_bytes(n + ".txt") for n in ("plrabn12", "world192", "terror2", "alice")]
    d = datagen.dickens_like(prose)
    assert len(d) == 10_192_446 and d == datagen.dickens_like(prose)
    st, xml = oracle.decompress(golden_bytes("xml.snappy"))
    assert st == 0
    m = datagen.mozilla_like(xml)
    sp = datagen.spamfile_like(prose[:2] + [golden_bytes("coding.txt")])
    assert (len(m), len(sp)) == (51_220_480, 84_217_482)
    assert hashlib.sha256(m).hexdigest() == hashlib.sha256(datagen.mozilla_like(xml)).hexdigest()
    for data in (d, m[:8 << 20], sp[:8 << 20]):
        stream = oracle.compress(data, 32768, threads=8)
        st, back = oracle.decompress(stream)
        assert st == 0 and back == data


def _libsnappy():
    """Google's Snappy as bundled with pyarrow (the image's offline wheelhouse): a real implementation of the ORIGINAL
    format (snappy/README.md:9-18) that shares no code with this repository."""
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("snappy"):
        pytest.skip("pyarrow was built without snappy")
    return pa


def test_interop_with_a_real_snappy_library_both_directions():
    """SURVEY 8f rank 4, pinned: the converter's output is decoded by libsnappy (pyarrow) to the plaintext -- for the
    reference's own golden .snappy files and for oracle streams at other block sizes -- and a stream written BY libsnappy
    (whose back-references cross our block boundaries freely) is decoded by the converter's reader and re-framed by the
    dpu_snappy tool into exactly the oracle's bytes."""
    import importlib.util
    import os
    import datagen
    from conftest import ROOT
    pa = _libsnappy()
    spec = importlib.util.spec_from_file_location("to_raw_snappy", os.path.join(ROOT, "tools", "to_raw_snappy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name in ("alice", "coding", "terror2", "plrabn12", "world192"):
        txt = golden_bytes(name + ".txt")
        raw = mod.convert(golden_bytes(name + ".snappy"))
        assert pa.decompress(raw, decompressed_size=len(txt), codec="snappy").to_pybytes() == txt, name
    text = golden_bytes("plrabn12.txt")
    for data in (datagen.text_random_interleave(text, 300_000), datagen.records(200_000), datagen.zeros(100_000),
                 datagen.lz_structured(150_000, 5), b"", b"a"):
        for bs in (64, 4096, 32768, 65535):
            if bs == 64 and len(data) > 50_000:
                continue
            raw = mod.convert(oracle.compress(data, bs))
            got = pa.decompress(raw, decompressed_size=len(data), codec="snappy").to_pybytes() if data else b""
            assert got == data, (len(data), bs)
        theirs = pa.compress(data, codec="snappy").to_pybytes()
        assert mod.decode_raw(theirs) == data
        if os.path.exists(mod.CLI) and data:
            assert mod.reframe(theirs, 4096) == oracle.compress(data, 4096)
