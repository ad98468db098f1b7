#!/usr/bin/env python3
"""Convert a block-framed .snappy file (this repository / the reference's format, snappy/README.md:19-33)
into ONE stream of the original Snappy format (snappy/README.md:9-18): varint(total length) + elements.

Possible without re-compressing because every block is a self-contained element stream whose back-references
stay inside the block, so the concatenation of the block bodies is a valid raw Snappy stream that any standard
decoder accepts.

The opposite direction (--from-raw) cannot keep the elements: a raw stream's back-references may reach across any
block boundary.  It decodes the raw stream here and hands the plaintext to this repository's `dpu_snappy -c`
(CPU mode, or the GPU with --gpu), so the result is exactly what the tool writes for that plaintext and block size.

Usage: python tools/to_raw_snappy.py in.snappy out.raw_snappy
       python tools/to_raw_snappy.py --from-raw [--gpu] [-b BLOCK_SIZE] in.raw_snappy out.snappy
"""
import os
import subprocess
import sys
import tempfile

CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pim-compression_amd", "host", "dpu_snappy")


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


def decode_raw(buf):
    """Decoder of the ORIGINAL Snappy format: varint(uncompressed length) + elements (all four tag types)."""
    n, i = read_varint(buf, 0)
    out = bytearray()
    while i < len(buf):
        tag = buf[i]
        i += 1
        kind = tag & 3
        if kind == 0:
            ln = (tag >> 2) + 1
            if ln > 60:
                extra = ln - 60
                ln = int.from_bytes(buf[i:i + extra], "little") + 1
                i += extra
            if i + ln > len(buf):
                raise ValueError("literal runs past the end of the stream")
            out += buf[i:i + ln]
            i += ln
            continue
        if kind == 1:
            ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | buf[i]
            i += 1
        elif kind == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(buf[i:i + 2], "little")
            i += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(buf[i:i + 4], "little")
            i += 4
        if off == 0 or off > len(out):
            raise ValueError("back-reference outside the decoded data")
        if off >= ln:
            out += out[len(out) - off:len(out) - off + ln]
        else:                                           # overlapping copy: the last `off` bytes repeat
            pattern = bytes(out[len(out) - off:])
            out += (pattern * (ln // off + 1))[:ln]
    if len(out) != n:
        raise ValueError(f"stream decodes to {len(out)} bytes, its header says {n}")
    return bytes(out)


def reframe(raw, block_size=32768, gpu=False):
    """raw Snappy stream -> block-framed stream, through the dpu_snappy tool of this repository."""
    plain = decode_raw(raw)
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "plain"), os.path.join(tmp, "framed")
        with open(src, "wb") as f:
            f.write(plain)
        cmd = [CLI] + (["-d"] if gpu else []) + ["-c", "-b", str(block_size), "-i", src, "-o", dst]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)} failed: {r.stderr}")
        with open(dst, "rb") as f:
            return f.read()


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--from-raw":
        args = args[1:]
        gpu = "--gpu" in args
        args = [a for a in args if a != "--gpu"]
        bs = 32768
        if "-b" in args:
            k = args.index("-b")
            bs = int(args[k + 1])
            del args[k:k + 2]
        if len(args) != 2:
            sys.exit(__doc__)
        with open(args[0], "rb") as f:
            data = f.read()
        with open(args[1], "wb") as f:
            f.write(reframe(data, bs, gpu))
        sys.exit(0)
    if len(args) != 2:
        sys.exit(__doc__)
    with open(args[0], "rb") as f:
        data = f.read()
    with open(args[1], "wb") as f:
        f.write(convert(data))
