#!/usr/bin/env python3
"""Convert a block-framed .snappy file (this repository / the reference's format, snappy/README.md:19-33)
into ONE stream of the original Snappy format (snappy/README.md:9-18): varint(total length) + elements.

Possible without re-compressing because every block is a self-contained element stream whose back-references
stay inside the block, so the concatenation of the block bodies is a valid raw Snappy stream that any standard
decoder accepts.  (The opposite direction needs re-framing at 64 KiB chunk boundaries, which the reference's u16
block geometry -- block_size <= 65535 -- cannot express; use `dpu_snappy -c` on the plaintext instead.)

Usage: python tools/to_raw_snappy.py in.snappy out.raw_snappy
"""
import sys


def varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def read_varint(buf, i):
    v, shift = 0, 0
    for _ in range(5):
        c = buf[i]
        i += 1
        v |= (c & 0x7f) << shift
        if c < 0x80:
            return v, i
        shift += 7
    raise ValueError("malformed varint")


def convert(stream):
    total, i = read_varint(stream, 0)
    _block_size, i = read_varint(stream, i)
    out = bytearray(varint(total))
    while i < len(stream):
        size = int.from_bytes(stream[i:i + 4], "little")
        i += 4
        if i + size > len(stream):
            raise ValueError("truncated block")
        out += stream[i:i + size]
        i += size
    return bytes(out)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    with open(sys.argv[1], "rb") as f:
        data = f.read()
    with open(sys.argv[2], "wb") as f:
        f.write(convert(data))
