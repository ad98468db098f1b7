"""Deterministic synthetic "Silesia-mix" (SURVEY.md section 8d, config 5).

The Silesia corpus files named in BASELINE.json (dickens, mozilla, ...) are absent from the reference
checkout (.MISSING_LARGE_BLOBS), so the workload is built from the reference's committed test texts
(tests/golden/: plrabn12, world192, terror2, coding, alice, and the plaintext of xml.snappy) plus seeded
generators, in Silesia-like proportions: text ~45 %, xml ~25 %, structured binary ~20 %, incompressible ~10 %.
One "unit" is ~21 MB; a container is the repeated concatenation of its unit cut to the requested length
(the unit is not a multiple of the block size, so block contents differ between repetitions).
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
XML_TXT_SHA256 = "0e82e54e695c1938e4193448022543845b33020c8be6bf3bf3ead2224903e08c"


def _read(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return np.frombuffer(f.read(), dtype=np.uint8)


def _records(n, seed):
    """Structured binary: runs of 16-64 byte records with a few mutated fields (tarred-executable-like)."""
    r = np.random.default_rng(seed)
    parts, have = [], 0
    while have < n:
        rec_len = int(r.integers(16, 65))
        reps = int(r.integers(40, 400))
        rec = r.integers(0, 256, size=rec_len, dtype=np.uint8)
        blk = np.tile(rec, reps).reshape(reps, rec_len)
        nmut = reps * 2
        blk[r.integers(0, reps, size=nmut), r.integers(0, rec_len, size=nmut)] = r.integers(0, 256, size=nmut, dtype=np.uint8)
        # a little-endian counter field, as in symbol tables
        if rec_len >= 20:
            blk[:, 4] = (np.arange(reps) & 0xff).astype(np.uint8)
            blk[:, 5] = ((np.arange(reps) >> 8) & 0xff).astype(np.uint8)
        parts.append(blk.reshape(-1))
        have += blk.size
    return np.concatenate(parts)[:n]


def build_unit(xml_plain, seed=0):
    """xml_plain: uint8 array holding the 5,345,280-byte plaintext of tests/golden/xml.snappy."""
    r = np.random.default_rng(1000 + seed)
    texts = [_read("world192.txt"), _read("plrabn12.txt"), _read("terror2.txt"), _read("coding.txt"), _read("alice.txt")]
    text_target = int(xml_plain.size * 45 / 25)
    chunks, have, i = [], 0, 0
    while have < text_target:
        t = texts[i % 3] if i % 7 else texts[3 + (i // 7) % 2]
        # rotate each repetition so repeats never align with block boundaries the same way
        k = int(r.integers(0, t.size))
        chunks.append(np.concatenate([t[k:], t[:k]]))
        have += t.size
        i += 1
    text = np.concatenate(chunks)[:text_target]
    binary = _records(int(xml_plain.size * 20 / 25), 2000 + seed)
    noise = r.integers(0, 256, size=int(xml_plain.size * 10 / 25), dtype=np.uint8)
    # interleave in ~1 MiB slices so every region of a container sees every data class
    srcs = [text, xml_plain, binary, noise]
    pos = [0, 0, 0, 0]
    piece = [int(s.size // 5) + 1 for s in srcs]
    out = []
    for _ in range(5):
        for j, s in enumerate(srcs):
            out.append(s[pos[j]:pos[j] + piece[j]])
            pos[j] += piece[j]
    return np.concatenate(out)


def container_from_unit(unit_dev, length):
    """Tile a device-resident unit (torch uint8 tensor) to `length` bytes (+16 bytes of slack)."""
    import torch
    reps = (length + unit_dev.numel() - 1) // unit_dev.numel()
    buf = torch.empty(length + 16, dtype=torch.uint8, device=unit_dev.device)
    for k in range(reps):
        lo = k * unit_dev.numel()
        hi = min(length, lo + unit_dev.numel())
        buf[lo:hi] = unit_dev[:hi - lo]
    buf[length:] = 0
    return buf
