"""Deterministic synthetic inputs shared by the CPU and GPU parity tests (no reference files needed)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pim-compression_amd"))
from standins import (DICKENS_LIKE_BYTES, MOZILLA_LIKE_BYTES, SPAMFILE_LIKE_BYTES, dickens_like, mozilla_like,  # noqa: E402,F401
                      records, spamfile_like)


def rng(seed):
    return np.random.default_rng(seed)


def random_bytes(n, seed=1):
    return rng(seed).integers(0, 256, size=n, dtype=np.uint8).tobytes()


def zeros(n):
    return bytes(n)


def periodic(n, period, seed=2):
    base = rng(seed).integers(0, 256, size=period, dtype=np.uint8)
    reps = (n + period - 1) // period
    return np.tile(base, reps)[:n].tobytes()


def low_entropy(n, alphabet=4, seed=3):
    return rng(seed).integers(0, alphabet, size=n, dtype=np.uint8).tobytes()


def text_random_interleave(text, n, seed=4, chunk=3000):
    """Alternating slices of real text and random bytes: ramps the skip heuristic up and down."""
    r = rng(seed)
    out = bytearray()
    pos = 0
    while len(out) < n:
        k = int(r.integers(200, chunk))
        if r.random() < 0.5:
            out += text[pos % len(text):pos % len(text) + k]
            pos += k
        else:
            out += r.integers(0, 256, size=k, dtype=np.uint8).tobytes()
    return bytes(out[:n])


def edge_cases(text):
    """(name, data) pairs covering SURVEY Appendix E's edge list."""
    cases = [("empty", b"")]
    for n in (1, 2, 3, 4, 5, 14, 15, 16, 17, 59, 60, 61, 64, 255, 256, 257):
        cases.append((f"text{n}", text[:n]))
        cases.append((f"zeros{n}", zeros(n)))
    cases.append(("zeros100k", zeros(100_000)))
    for p in (1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 2047, 2048, 2049):
        cases.append((f"period{p}", periodic(70_000, p)))
    cases.append(("random200k", random_bytes(200_000)))
    cases.append(("lowent", low_entropy(90_000)))
    cases.append(("interleave", text_random_interleave(text, 150_000)))
    cases.append(("records", records(120_000)))
    return cases


BLOCK_SIZES = (64, 255, 256, 257, 1000, 4096, 16384, 32768, 65535)


def lz_structured(n, seed):
    """Random LZ-like data: literals, back-references at random distances/lengths, runs -- exercises every element
    type, offset class (1-byte / 2-byte), overlapping copies and the skip heuristic with randomised geometry."""
    r = rng(seed)
    out = bytearray(r.integers(0, 256, size=8, dtype=np.uint8).tobytes())
    alphabet = int(r.choice([2, 4, 16, 64, 256]))
    while len(out) < n:
        k = r.random()
        if k < 0.35:
            out += r.integers(0, alphabet, size=int(r.integers(1, 40)), dtype=np.uint8).tobytes()
        elif k < 0.85:
            dist = int(r.choice([1, 2, 3, 4, 7, 8, 60, 64, 2047, 2048, 2049, 5000, 40000, 70000]))
            dist = min(dist, len(out))
            ln = int(r.choice([4, 5, 11, 12, 13, 59, 60, 64, 65, 67, 68, 69, 130, 1000]))
            start = len(out) - dist
            for i in range(ln):
                out.append(out[start + i])
        else:
            out += bytes([int(r.integers(0, 256))]) * int(r.integers(1, 300))
    return bytes(out[:n])


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def element_stream(total_len, block_size, seed, flavour=0):
    """A random but VALID framed stream built element by element -- not the output of any compressor -- and the plaintext it
    encodes.  Decoders must accept everything the format allows (snappy_decompress.c:232-285), and a greedy compressor only
    ever produces a narrow subset: no copies shorter than 4, no 4-byte offsets, no non-minimal literal headers, few
    overlapping copies, literal / copy sizes tied to its parse.  Elements here: literals of 1..~5000 bytes (every header
    size, sometimes longer than needed), copies of 1..64 bytes with 1-, 2- and 4-byte offsets, offsets from 1 (run length)
    to the start of the block, overlapping and adjacent ones.  flavour: 0 = mixed, 1 = copy-heavy with short offsets (chains
    of dependent copies inside one 64-byte window of compressed data), 2 = long literals, 3 = maximal expansion (3-byte copies
    of 64)."""
    r = rng(seed)
    plain = bytearray()
    stream = bytearray(_varint(total_len) + _varint(block_size))
    done = 0
    while done < total_len:
        want = min(block_size, total_len - done)
        out = bytearray()
        body = bytearray()
        while len(out) < want:
            left = want - len(out)
            k = r.random()
            lit_p = (0.35, 0.12, 0.8, 0.03)[flavour]
            if not out or k < lit_p:
                c = r.random()
                if flavour == 2:
                    ln = int(r.choice([61, 64, 100, 128, 129, 200, 256, 257, 1000, 5000]))
                elif c < 0.6:
                    ln = int(r.integers(1, 9))
                elif c < 0.9:
                    ln = int(r.integers(1, 70))
                else:
                    ln = int(r.choice([60, 61, 62, 63, 64, 65, 127, 128, 129, 190, 256, 257, 300, 3000]))
                ln = min(ln, left)
                n = ln - 1
                minimal = 0 if n < 60 else (1 if n < 256 else 2)
                nb = minimal if r.random() < 0.85 else min(4, minimal + int(r.integers(1, 3)))   # length bytes; 0 = in the tag
                if nb == 0:
                    body.append(n << 2)
                else:
                    body.append((59 + nb) << 2)
                    body += n.to_bytes(nb, "little")
                payload = r.integers(0, int(r.choice([2, 16, 256])), size=ln, dtype=np.uint8).tobytes()
                body += payload
                out += payload
            else:
                c = r.random()
                near_p = (0.4, 0.8, 0.3, 0.1)[flavour]
                if c < near_p:
                    off = int(r.choice([1, 1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 63, 64]))
                elif c < near_p + 0.3:
                    off = int(r.integers(1, 400))
                else:
                    off = int(r.integers(1, len(out) + 1))
                off = max(1, min(off, len(out)))
                ln = 64 if flavour == 3 else int(r.choice([1, 2, 3, 4, 4, 5, 7, 8, 11, 12, 16, 33, 60, 63, 64]))
                ln = min(ln, left)
                c = r.random()
                if 4 <= ln <= 11 and off < 2048 and c < 0.5:
                    body.append(1 | ((ln - 4) << 2) | ((off >> 8) << 5))
                    body.append(off & 0xff)
                elif off < 65536 and c < 0.93:
                    body.append(2 | ((ln - 1) << 2))
                    body += off.to_bytes(2, "little")
                else:
                    body.append(3 | ((ln - 1) << 2))
                    body += off.to_bytes(4, "little")
                start = len(out) - off
                for i in range(ln):
                    out.append(out[start + i])
        stream += len(body).to_bytes(4, "little") + body
        plain += out
        done += want
    return bytes(stream), bytes(plain)
