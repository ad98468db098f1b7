"""csrc/host_chain.hpp -- the drop-in pair's host pre-scan of the size chain in parallel shares (plain C++, compiled here
with g++): it must return exactly the offsets of the plain walk of snappy_decompress.c:317-340 or decline (the caller then
walks serially), whatever the bytes are, and never read beyond the stream (the driver below ends the stream at an
inaccessible page)."""
import os
import subprocess

import numpy as np

import datagen
import oracle_lib as oracle
from conftest import golden_bytes

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pim-compression_amd", "csrc")

DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#include "host_chain.hpp"
// usage: driver <file> <first> <num_blocks> <block_size> <threads> <min share bytes>  ->  "declined" | "ok <fnv of the offsets>" ; exit 3 on a wrong chain
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    const size_t page = 4096, mapped = ((n + page - 1) / page + 1) * page;
    uint8_t* base = (uint8_t*)mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    mprotect(base + mapped - page, page, PROT_NONE);
    uint8_t* buf = base + mapped - page - n;
    if (n && fread(buf, 1, n, f) != (size_t)n) return 2;
    const uint64_t first = strtoull(argv[2], 0, 10), nb = strtoull(argv[3], 0, 10);
    const uint32_t bs = (uint32_t)strtoul(argv[4], 0, 10);
    std::vector<uint64_t> off;
    if (!host_chain::parallel_walk(buf, (uint64_t)n, first, nb, bs, (unsigned)atoi(argv[5]), off, strtoull(argv[6], 0, 10))) { puts("declined"); return 0; }
    // whatever it accepted must be the plain walk
    uint64_t at = first;
    for (uint64_t i = 0; i < nb; ++i) {
        if (off[i] != at || at + 4 > (uint64_t)n) return 3;
        at += 4 + (uint64_t)host_chain::le32(buf + at);
    }
    if (at != (uint64_t)n || off[nb] != (uint64_t)n) return 3;
    unsigned long long h = 1469598103934665603ull;
    for (uint64_t v : off) { h ^= v; h *= 1099511628211ull; }
    printf("ok %llx\n", h);
    return 0;
}
'''


def _build(tmp_path):
    src = tmp_path / "d.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "d"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", CSRC, str(src), "-o", str(exe)])
    return str(exe)


def _run(exe, tmp_path, stream, first, nb, bs, threads=8, min_share=1 << 20):
    """(shares of 1 MiB here, so that streams of a few MB exercise the machinery; the library's default is 16 MiB)"""
    p = tmp_path / "s.bin"
    p.write_bytes(stream)
    r = subprocess.run([exe, str(p), str(first), str(nb), str(bs), str(threads), str(min_share)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])
    return r.stdout.strip()


def test_host_parallel_size_chain(tmp_path):
    exe = _build(tmp_path)
    text = golden_bytes("plrabn12.txt")
    data = (datagen.text_random_interleave(text, 6_000_000) + datagen.records(3_000_000) + bytes(1_000_000) + datagen.random_bytes(4_000_000)) * 4
    zero_literals = b"".join(bytes([i & 0xff, 0, 0, 0, 0, 0, (i >> 3) & 0xff, 0]) for i in range(4_000_000))
    accepted = 0
    for payload, bs in ((data, 32768), (data, 65535), (data[:20_000_000], 4096), (zero_literals, 32768), (data[:9_000_000], 64)):
        stream = oracle.compress(payload, bs, threads=8)
        total, got_bs, hdr = oracle.read_header(stream)
        nb = (total + got_bs - 1) // got_bs
        out = _run(exe, tmp_path, stream, hdr, nb, got_bs)
        accepted += out.startswith("ok")
        assert out.startswith("ok") or out == "declined"
        assert _run(exe, tmp_path, stream, hdr, nb, got_bs, threads=1) == "declined"      # one thread: the caller's serial walk
        # damaged: truncated, a size field changed, a wrong block count -> never a wrong chain (exit 3), normally declined
        for bad in (stream[:len(stream) * 2 // 3], stream[:-1], stream + b"\0\0\0"):
            assert _run(exe, tmp_path, bad, hdr, nb, got_bs) == "declined"
        at = hdr
        for _ in range(nb // 2):
            at += 4 + int.from_bytes(stream[at:at + 4], "little")
        b = bytearray(stream)
        b[at] ^= 0x20
        assert _run(exe, tmp_path, bytes(b), hdr, nb, got_bs) == "declined"
        assert _run(exe, tmp_path, stream, hdr, nb + 1, got_bs) == "declined"
        assert _run(exe, tmp_path, stream, hdr, nb - 1, got_bs) == "declined"
    assert accepted >= 3                                         # the shares do fit together on ordinary streams
    # short streams are left to the serial walk
    small = oracle.compress(golden_bytes("world192.txt"), 32768)
    total, bs, hdr = oracle.read_header(small)
    assert _run(exe, tmp_path, small, hdr, (total + bs - 1) // bs, bs) == "declined"
    stream = oracle.compress(data[:20_000_000], 32768, threads=8)    # ~9 MB of stream: below two shares of the default 16 MiB
    total, bs, hdr = oracle.read_header(stream)
    assert _run(exe, tmp_path, stream, hdr, (total + bs - 1) // bs, bs, min_share=16 << 20) == "declined"
    # arbitrary bytes: whatever comes back is never a wrong chain (the driver exits 3 on one)
    rnd = np.random.default_rng(7)
    for k in range(6):
        junk = rnd.integers(0, 256 if k % 2 else 3, size=12_000_000, dtype=np.uint8).tobytes()
        _run(exe, tmp_path, junk, int(rnd.integers(0, 10)), int(rnd.integers(1, 500)), 32768)
