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
