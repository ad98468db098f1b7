"""Stand-ins for the Silesia files BASELINE.json names in configs 3 and 4 (dickens, mozilla, spamfile).

The files themselves are absent from the reference checkout (.MISSING_LARGE_BLOBS); the shapes follow SURVEY.md 8(d):
exact byte counts, English prose shuffled at paragraph granularity so that repeats fall outside the 32 KiB block window,
a tarred-executables-like mix, a text-heavy mail-like mix.  Deterministic (seeded); built from the reference's committed
test texts (tests/golden/).  Used by the parity tests (through tests/datagen.py) and by `bench.py --workload ...`.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def rng(seed):
    return np.random.default_rng(seed)


def golden_text(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def prose_texts():
    return [golden_text(n + ".txt") for n in ("plrabn12", "world192", "terror2", "alice")]


def records(n, seed=5):
    """Structured binary: repeating 16-64 B records with a few mutated fields (mozilla-like)."""
    r = rng(seed)
    out = bytearray()
    while len(out) < n:
        rec_len = int(r.integers(16, 65))
        rec = bytearray(r.integers(0, 256, size=rec_len, dtype=np.uint8).tobytes())
        for _ in range(int(r.integers(20, 200))):
            for _ in range(int(r.integers(0, 4))):
                rec[int(r.integers(0, rec_len))] = int(r.integers(0, 256))
            out += rec
    return bytes(out[:n])


DICKENS_LIKE_BYTES = 10_192_446      # 312 blocks of 32 KiB
MOZILLA_LIKE_BYTES = 51_220_480      # 1,564 blocks
SPAMFILE_LIKE_BYTES = 84_217_482     # 2,571 blocks


def _paragraph_shuffle(texts, n, seed):
    r = rng(seed)
    paras = []
    for t in texts:
        paras += [p + b"\n\n" for p in t.replace(b"\r\n", b"\n").split(b"\n\n") if p]
    out, have = [], 0
    while have < n:
        for k in r.permutation(len(paras)):
            out.append(paras[int(k)])
            have += len(paras[int(k)])
            if have >= n:
                break
    return b"".join(out)[:n]


def dickens_like(texts):
    """texts: plrabn12, world192, terror2, alice (bytes)."""
    return _paragraph_shuffle(texts, DICKENS_LIKE_BYTES, 0xD1C3)


def mozilla_like(xml_plain):
    """40 % xml-derived text, 35 % structured binary records, 25 % incompressible, interleaved in 256 KiB slices."""
    n = MOZILLA_LIKE_BYTES
    r = rng(0x4D4F5A)
    parts = {"xml": xml_plain, "rec": records(4 << 20, seed=0x4D4F), "rnd": None}
    out = bytearray()
    pos = {"xml": 0, "rec": 0}
    while len(out) < n:
        kind = r.choice(["xml", "rec", "rnd"], p=[0.40, 0.35, 0.25])
        k = 256 << 10
        if kind == "rnd":
            out += r.integers(0, 256, size=k, dtype=np.uint8).tobytes()
        else:
            src = parts[kind]
            p = pos[kind] % max(1, len(src) - k)
            out += src[p:p + k]
            pos[kind] += k + int(r.integers(0, 4096))
    return bytes(out[:n])


def spamfile_like(texts):
    """Text-heavy: shuffled paragraphs with repeated header-like lines in front of every 'message'."""
    n = SPAMFILE_LIKE_BYTES
    r = rng(0x5A4D)
    body = _paragraph_shuffle(texts, 24 << 20, 0x5A4E)
    out = bytearray()
    pos = 0
    msg = 0
    while len(out) < n:
        msg += 1
        out += (b"From user%05d@example.org  Mon Jan  1 00:00:%02d 2001\nReceived: from mail.example.org (10.0.%d.%d)\n"
                b"X-Spam-Flag: YES\nX-Spam-Level: ********\nSubject: offer %d\n\n"
                % (int(r.integers(0, 50000)), msg % 60, int(r.integers(0, 256)), int(r.integers(0, 256)), msg))
        k = int(r.integers(2000, 30000))
        p = pos % (len(body) - k)
        out += body[p:p + k]
        pos += k + int(r.integers(0, 100000))
    return bytes(out[:n])
