"""Deterministic synthetic inputs shared by the CPU and GPU parity tests (no reference files needed)."""
import numpy as np


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


# ---------------------------------------------------------------------------
# Stand-ins for the Silesia files BASELINE.json names (configs 3 and 4).  The files themselves are absent from the
# reference checkout; the shapes follow the survey: exact byte counts, English prose shuffled at paragraph granularity so
# that repeats fall outside the 32 KiB block window, a tarred-executables-like mix, a text-heavy mail-like mix.
# ---------------------------------------------------------------------------
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
