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
