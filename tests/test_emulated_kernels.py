"""CPU tests: the UNMODIFIED HIP kernel source (pim-compression_amd/csrc/snappy_kernels.hpp) compiled
for the lockstep wave emulator in tests/emu and compared with the oracle.  This is logic coverage for the
GPU-less container; the real parity tests are the -m gpu ones through the C ABI."""
import os

import pytest

import datagen
import emu_lib as emu
import oracle_lib as oracle
from conftest import golden_bytes


@pytest.mark.parametrize("name", ["alice", "coding", "terror2"])
def test_emulated_golden_both_directions(name):
    txt, snp = golden_bytes(name + ".txt"), golden_bytes(name + ".snappy")
    assert emu.compress(txt, 32768) == snp
    total, bs, hdr = oracle.read_header(snp)
    st, out = emu.decompress(snp, total, bs, hdr)
    assert st == 0 and out == txt


def test_emulated_edges_vs_oracle():
    text = golden_bytes("plrabn12.txt")
    for name, data in datagen.edge_cases(text):
        data = data[:12000]
        for bs in (64, 257, 4096, 32768, 65535):
            if len(data) > 3000 and bs < 4096:
                continue
            ref = oracle.compress(data, bs)
            assert emu.compress(data, bs) == ref, (name, bs)
            total, got_bs, hdr = oracle.read_header(ref)
            st, out = emu.decompress(ref, total, got_bs, hdr)
            assert st == 0 and out == data, (name, bs)


def test_emulated_long_runs_and_split_copies():
    # long matches exercise the 64/60/rest copy split (snappy_compress.c:254-272) and 256-byte extension rounds
    for n in (67, 68, 69, 131, 132, 1000, 40000):
        data = b"abcd" + bytes(n)
        ref = oracle.compress(data, 65535)
        assert emu.compress(data, 65535) == ref, n
        total, bs, hdr = oracle.read_header(ref)
        st, out = emu.decompress(ref, total, bs, hdr)
        assert st == 0 and out == data


# compress variant = table kind * 10000 + form * 1000 + 500 + kernel (tests/emu/emu_runtime.cpp): kernel 1 = LDS table,
# 3 = global table; form 2 = bulk, 3 = stream; table kind 1 = slot filter, 4 / 5 = slot cache of 512 / 256 slots
@pytest.mark.parametrize("cv,dv", [(2501, 3), (12503, 3), (3501, 3), (13503, 3), (43503, 3), (53503, 3), (42503, 3), (52503, 3)])
def test_emulated_other_variants(cv, dv):
    """Every shipped K1 form (and the 256-slot cache, which only the emulator instantiates) produces the oracle's bytes."""
    text = golden_bytes("plrabn12.txt")
    cases = [golden_bytes("coding.txt"), datagen.text_random_interleave(text, 40_000), datagen.periodic(9000, 5),
             datagen.zeros(5000), datagen.random_bytes(20_000)]
    for data in cases:
        for bs in (32768, 4097, 65535):
            ref = oracle.compress(data, bs)
            assert emu.compress(data, bs, cv) == ref, (cv, bs)
            total, got_bs, hdr = oracle.read_header(ref)
            st, out = emu.decompress(ref, total, got_bs, hdr, dv)
            assert st == 0 and out == data, (dv, bs)


def test_emulated_seeded_fuzz():
    for seed in range(10):
        data = datagen.lz_structured(6000 + 900 * seed, seed)
        for bs in (32768, 2048):
            ref = oracle.compress(data, bs)
            assert emu.compress(data, bs) == ref, (seed, bs)
            total, got_bs, hdr = oracle.read_header(ref)
            st, out = emu.decompress(ref, total, got_bs, hdr)
            assert st == 0 and out == data, (seed, bs)


def test_emulated_decoder_is_strict():
    # copy reaching before the block start
    body = bytes([0x00, 0x41, (3 << 2) | 2, 9, 0])
    stream = bytes([5, 0x80, 0x80, 0x02]) + len(body).to_bytes(4, "little") + body
    st, _ = emu.decompress(stream, 5, 32768, 4)
    assert st == 1
    # literal longer than the block's compressed size
    body = bytes([0x10, 0x41])
    stream = bytes([5, 0x80, 0x80, 0x02]) + len(body).to_bytes(4, "little") + body
    st, _ = emu.decompress(stream, 5, 32768, 4)
    assert st == 1
    # truncated chain
    good = oracle.compress(b"hello hello hello hello", 32768)
    st, _ = emu.decompress(good[:-1], 23, 32768, 4)
    assert st == 1


def test_emulated_verify_index_accepts_the_chain_and_nothing_else():
    """snappy_hip_verify_index's kernels: a candidate index passes iff it is exactly the walk of the size chain
    (snappy_decompress.c:317-340); every kind of wrong candidate is rejected as a whole."""
    import numpy as np
    data = datagen.text_random_interleave(golden_bytes("plrabn12.txt"), 300_000)
    for bs in (32768, 1000, 65535, 7):
        stream = oracle.compress(data if bs > 7 else data[:20_000], bs)
        total, got_bs, hdr = oracle.read_header(stream)
        nb = (total + got_bs - 1) // got_bs
        good = np.concatenate([oracle.index_blocks(stream), np.array([len(stream)], dtype=np.uint64)])
        assert emu.verify_index(stream, good, total, got_bs, hdr) == (0, nb)
        for k in sorted({0, 1, nb // 2, nb - 1, nb}):
            for delta in (1, -1, 4, 1 << 33):
                bad = good.copy()
                bad[k] = np.uint64((int(bad[k]) + delta) & ((1 << 64) - 1))
                st, links = emu.verify_index(stream, bad, total, got_bs, hdr)
                assert st != 0 and links < nb, (bs, k, delta)
        # a stream cut short, or with bytes appended, does not match the index of the intact one
        assert emu.verify_index(stream[:-1], good, total, got_bs, hdr)[0] != 0
        assert emu.verify_index(stream + b"\0", good, total, got_bs, hdr)[0] != 0
        # a corrupted size field breaks exactly the links around it
        if nb >= 3:
            broken = bytearray(stream)
            broken[int(good[1])] ^= 0x01
            st, links = emu.verify_index(bytes(broken), good, total, got_bs, hdr)
            assert st != 0 and links == nb - 1
    empty = oracle.compress(b"", 32768)
    total, got_bs, hdr = oracle.read_header(empty)
    assert emu.verify_index(empty, np.array([hdr], dtype=np.uint64), total, got_bs, hdr)[0] == 0
    assert emu.verify_index(empty + b"x", np.array([hdr], dtype=np.uint64), total, got_bs, hdr)[0] != 0


@pytest.mark.parametrize("flavour", [0, 1, 2, 3])
def test_emulated_decoder_on_random_element_streams(flavour):
    """Streams no compressor of ours would write (datagen.element_stream): the decoder kernel under the emulator must decode
    them exactly as the oracle (= the reference's decoder restated) does."""
    for seed in range(6):
        bs = (32768, 4097, 700, 65535, 64, 20000)[seed]
        stream, plain = datagen.element_stream(9000 + 3777 * seed, bs, 1000 * flavour + seed, flavour)
        st, ref = oracle.decompress(stream)
        assert st == 0 and ref == plain, (flavour, seed)
        total, got_bs, hdr = oracle.read_header(stream)
        st, out = emu.decompress(stream, total, got_bs, hdr)
        assert st == 0 and out == plain, (flavour, seed, bs)


_STREAM_STRESS = """
import sys
sys.path.insert(0, sys.argv[1])
import datagen, emu_lib as emu, oracle_lib as oracle
from conftest import golden_bytes
text = golden_bytes("plrabn12.txt")
big = [golden_bytes("terror2.txt")[:36000], datagen.text_random_interleave(text, 34000), datagen.records(34000), datagen.low_entropy(20000),
       datagen.lz_structured(34000, 7), b"abcd" + bytes(34000)]
small = [datagen.zeros(9000), datagen.periodic(9000, 5)] + [d[:6000] for _, d in datagen.edge_cases(text)]
for data, sizes in [(d, (32768, 4097)) for d in big] + [(d, (65535, 700)) for d in small]:
    for bs in sizes:
        ref = oracle.compress(data, bs)
        for cv in (3501, 43503):
            assert emu.compress(data, bs, cv) == ref, (len(data), bs, cv)
print("ok")
"""


@pytest.mark.parametrize("unordered", [False, True])
def test_emulated_stream_form(unordered):
    """K1's stream form (snappy_k1_stream.hpp), LDS-table and global-table kernels, against the oracle: windows taken by the
    pipeline, sent back to the bulk form (stride > 1, block tails), copies of 64+ bytes taken in place, chains of equal
    hashes.  EMU_LDS_UNORDERED=1 lets the emulator's fibers run analyse()'s LDS atomics in no particular order, which the
    kernel must detect and answer with the race tables of the bulk form."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    if unordered:
        env["EMU_LDS_UNORDERED"] = "1"
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-c", _STREAM_STRESS, here], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


_SHUFFLE_STRESS = """
import sys
sys.path.insert(0, sys.argv[1])
import datagen, emu_lib as emu, oracle_lib as oracle
from conftest import golden_bytes
text = golden_bytes("plrabn12.txt")
cases = [golden_bytes("terror2.txt")[:40000], datagen.text_random_interleave(text, 40000), datagen.periodic(20000, 7), datagen.records(40000),
         datagen.zeros(9000)]
for data in cases:
    for bs in (32768, 65535):
        ref = oracle.compress(data, bs)
        for cv in (43503, 42503):
            assert emu.compress(data, bs, cv) == ref, (len(data), bs, cv)
print("ok")
"""


@pytest.mark.parametrize("seed", [5])
def test_emulated_slot_cache_under_shuffled_lane_schedules(seed):
    """The slot cache's store protocol (CachedGlobalTable::store_masked) lets the lanes of one call race for a cache word and
    reads back who won.  EMU_SHUFFLE runs the emulator's fibers -- the lanes -- in random order between two collectives, so
    the winner of every such race changes from run to run; the bytes must not."""
    import subprocess
    import sys
    env = dict(os.environ, EMU_SHUFFLE=str(seed))
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-c", _SHUFFLE_STRESS, here], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_emulated_slot_cache_block_with_three_lanes_in_one_cache_word():
    """tests/fixtures/mix_block_three_lanes_in_a_cache_word.bin: block 392 of the benchmark container.  In one commit of its
    parse three lanes meet in one word of the slot cache while the word's old content belongs to the slot one of them inserts
    -- the case in which round 4's one-exchange store protocol put a slot's old and new position into ONE store instruction
    (2 bytes differed on the GPU; the emulator now aborts on two lanes of a table-store instruction sharing an address,
    snappy_kernels.hpp: emu_check_distinct_stores).  Every cached-table form must produce the oracle's bytes."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "mix_block_three_lanes_in_a_cache_word.bin"), "rb") as f:
        data = f.read()
    assert len(data) == 32768
    ref = oracle.compress(data, 32768)
    for cv in (43503, 42503, 53503):
        assert emu.compress(data, 32768, variant=cv) == ref, cv


def _chain(stream):
    """(total_len, block_size, header_len, offsets of the chain) by the plain walk of snappy_decompress.c:317-340."""
    total, bs, hdr = oracle.read_header(stream)
    nb = (total + bs - 1) // bs
    offs, at = [], hdr
    for _ in range(nb):
        offs.append(at)
        at += 4 + int.from_bytes(stream[at:at + 4], "little")
    assert at == len(stream)
    return total, bs, hdr, offs


def test_emulated_size_chain_in_parallel_segments():
    """snappy_hip_index_streams' parallel form (chain_anchor / chain_walk / chain_finish kernels, csrc/snappy_kernels.hpp):
    walkers start at recognised block boundaries and their segments are laid end to end iff every one ends exactly on the next
    one's start.  Whatever the walkers make of the bytes, the result must be the chain of snappy_decompress.c:317-340 -- from
    the segments when they fit, from the serial walk when they do not (tiny blocks overflow a segment's buffer; literal
    payloads full of zero bytes look like size fields)."""
    text = golden_bytes("plrabn12.txt")
    big = datagen.text_random_interleave(text, 1_500_000) + datagen.records(700_000) + datagen.zeros(300_000) + datagen.random_bytes(600_000)
    sparse = bytes(2_000_000)                                   # all zero: the stream is copies of zeros
    zero_literals = b"".join(bytes([i & 0xff, 0, 0, 0, 0, 0, (i >> 3) & 0xff, 0]) for i in range(150_000))
    parallel = serial = 0
    for data, sizes in ((big, (32768, 65535, 4096, 700)), (golden_bytes("world192.txt"), (32768, 64, 16)), (sparse, (32768,)),
                        (zero_literals, (32768, 8192)), (golden_bytes("alice.txt"), (32768,))):
        for bs in sizes:
            stream = oracle.compress(data, bs)
            total, got_bs, hdr, ref = _chain(stream)
            resolved, st, nb, offs = emu.index_parallel(stream, total, got_bs, hdr)
            assert st == 0 and nb == len(ref) and offs.tolist() == ref, (len(data), bs, resolved)
            parallel += resolved
            serial += not resolved
    assert parallel >= 8                                        # the segments do resolve the ordinary cases themselves
    assert serial >= 1                                          # 16-byte blocks: ~6,500 hops per segment do not fit its 2,048


def test_emulated_size_chain_segments_on_damaged_streams():
    """Truncated, extended and corrupted streams: the parallel segments must not resolve what is not a chain, must stay inside
    the stream, and the serial walk behind them reports the stream as the reference's pre-scan would (invalid)."""
    data = datagen.text_random_interleave(golden_bytes("plrabn12.txt"), 1_200_000)
    stream = oracle.compress(data, 32768)
    total, bs, hdr, ref = _chain(stream)
    # cut inside a block / right behind a size field / one byte short
    for cut in (len(stream) - 1, ref[-1] + 4, ref[len(ref) // 2] + 100, ref[3] + 2):
        resolved, st, nb, _ = emu.index_parallel(stream[:cut], total, bs, hdr)
        assert not resolved and st != 0, cut
    # bytes appended: the last hop no longer ends on the stream's end
    resolved, st, nb, _ = emu.index_parallel(stream + b"\x00" * 7, total, bs, hdr)
    assert not resolved and st != 0
    # a size field made one larger / made huge / zeroed, in the first, a middle and the last block
    for b in (0, len(ref) // 2, len(ref) - 1):
        for new in (lambda v: v + 1, lambda v: 0xfffffff0, lambda v: 0):
            bad = bytearray(stream)
            v = int.from_bytes(bad[ref[b]:ref[b] + 4], "little")
            bad[ref[b]:ref[b] + 4] = (new(v) & 0xffffffff).to_bytes(4, "little")
            resolved, st, nb, _ = emu.index_parallel(bytes(bad), total, bs, hdr)
            assert not resolved and st != 0, (b, v)
    # a header that promises one block more or fewer than the chain has
    for wrong_total in (total + 32768, total - 32768):
        resolved, st, nb, _ = emu.index_parallel(stream, wrong_total, bs, hdr)
        assert not resolved and st != 0


def test_emulated_size_chain_segments_on_arbitrary_bytes():
    """Property: for ANY bytes, header fields and block count, the parallel segments + serial walk give what the plain walk of
    snappy_decompress.c:317-340 gives (the chain, or invalid) -- and read nothing beyond the stream: the emulator places the
    stream right in front of an inaccessible page.  Seeded: random bytes, valid streams with random spans overwritten, streams
    cut at random places, sparse bytes (mostly zero: plausible size fields everywhere)."""
    import random
    rnd = random.Random(20261005)
    base = oracle.compress(datagen.text_random_interleave(golden_bytes("plrabn12.txt"), 700_000) + bytes(200_000), 8192)
    total, bs, hdr, ref = _chain(base)

    def plain_walk(stream, hdr, nb):
        at = hdr
        for _ in range(nb):
            if at + 4 > len(stream):
                return None
            at += 4 + int.from_bytes(stream[at:at + 4], "little")
        return at == len(stream)

    for case in range(60):
        kind = case % 4
        if kind == 0:
            stream = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40_000)))
        elif kind == 1:
            b = bytearray(base)
            for _ in range(rnd.randrange(1, 4)):
                lo = rnd.randrange(0, len(b) - 64)
                b[lo:lo + rnd.randrange(1, 64)] = bytes(rnd.getrandbits(8) for _ in range(8))[:1] * 1
            stream = bytes(b)
        elif kind == 2:
            stream = base[:rnd.randrange(0, len(base))]
        else:
            stream = bytes(rnd.getrandbits(8) if rnd.random() < 0.03 else 0 for _ in range(rnd.randrange(1, 300_000)))
        h = hdr if kind in (1, 2) else rnd.randrange(0, 12)
        nb = len(ref) if kind in (1, 2) else rnd.randrange(0, 50)
        t = total if kind in (1, 2) else nb * 8192
        resolved, st, found, _ = emu.index_parallel(stream, t, 8192, h)
        want = plain_walk(stream, h, nb) if len(stream) >= h else None
        assert (st == 0) == bool(want), (case, kind, len(stream), h, nb, resolved, st, found)
