"""Seeded synthetic inputs for tests and bench.

Nothing here is downloaded: every stream is generated from a numpy PCG64 seed, so the GPU
box, this container and the golden-minting script all see the same bytes.

The small named inputs restate the reference's own test inputs
(/root/reference/test/Main.hs:44-45, :60-78, :114-119).
"""
from __future__ import annotations

import numpy as np

KIB = 1024
MIB = 1024 * 1024
GIB = 1024 * MIB


def hello20() -> bytes:
    """test/Main.hs:62 -- "hellohellohellohello"."""
    return b"hello" * 4


def ints_100000() -> bytes:
    """test/Main.hs:66-67 with `prepare` (test/Main.hs:44-45): "BEGIN 1 2 ... 100000 END" (588 904 B)."""
    return b" ".join([b"BEGIN"] + [str(i).encode() for i in range(1, 100001)] + [b"END"])


def hello_100000() -> bytes:
    """test/Main.hs:71-72: "BEGIN hello x100000 END" (600 009 B)."""
    return b" ".join([b"BEGIN"] + [b"hello"] * 100000 + [b"END"])


def rep42(n: int = 100000) -> bytes:
    """test/Main.hs:76-77: BS.replicate 100000 42 (the test titled "1MB ByteString")."""
    return bytes([42]) * n


def random_bytes(n: int, seed: int = 7) -> np.ndarray:
    """BASELINE configs[0]: uniform random bytes, numpy default_rng(seed)."""
    return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8)


def synth50(n: int, seed: int = 1234) -> np.ndarray:
    """~50 %-compressible stream: 512-byte rows, even rows random, odd row r is a copy of the even
    row r-(2k+1), k in [1,60) -- i.e. a match at most ~60 KiB back (inside LZ4's 64 KiB window)."""
    assert n % 1024 == 0
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, n, dtype=np.uint8).reshape(-1, 512)
    rows = a.shape[0]
    odd = np.arange(1, rows, 2)
    back = rng.integers(1, 60, odd.size) * 2 + 1
    src = np.maximum(odd - back, 0)
    src -= src % 2
    a[odd] = a[src]
    return a.reshape(-1)


_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)


def synth_text(n: int, seed: int = 99, vocab: int = 65536, zipf_a: float = 1.05) -> np.ndarray:
    """"enwik-style" text (enwik itself is not available offline): a `vocab`-word dictionary
    (word length U[2,10], letters drawn with probability ~ 1/rank over etaoin...), words drawn
    Zipf(a) truncated to the vocabulary, joined by single spaces."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(2, 11, vocab)
    p = 1.0 / np.arange(1, _LETTERS.size + 1)
    p /= p.sum()
    letters = _LETTERS[rng.choice(_LETTERS.size, size=int(lens.sum()), p=p)]
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    out = np.empty(n + 16, dtype=np.uint8)
    pos = 0
    # draw in batches; mean word length ~6 (+1 space)
    while pos < n:
        need = (n - pos) // 6 + 64
        ranks = rng.zipf(zipf_a, need)
        ranks = ranks[ranks <= vocab] - 1
        wl = lens[ranks]
        tot = wl + 1
        ends = np.cumsum(tot)
        k = int(np.searchsorted(ends, n + 16 - pos, side="right"))
        if k == 0:
            break
        ranks, wl, ends, tot = ranks[:k], wl[:k], ends[:k], tot[:k]
        begin = ends - tot + pos
        # scatter words: build index arrays
        idx_src = np.repeat(starts[ranks], wl) + (np.arange(int(wl.sum())) - np.repeat(np.cumsum(wl) - wl, wl))
        idx_dst = np.repeat(begin, wl) + (np.arange(int(wl.sum())) - np.repeat(np.cumsum(wl) - wl, wl))
        out[idx_dst] = letters[idx_src]
        out[begin + wl] = 32
        pos = int(ends[-1]) + pos
    if pos < n:
        out[pos:n] = 32
    return out[:n].copy()


def mutate(frame: bytes, pos: int, xor: int = 0xFF) -> bytes:
    b = bytearray(frame)
    b[pos] ^= xor
    return bytes(b)


def structured(n: int, seed: int) -> bytes:
    """Test-input generator with the sequence shapes an LZ4 codec has to survive: literal runs and matches of every
    length class (below 16, a few hundred, beyond 1 KiB, tens of KiB), self-overlapping matches (period 1..63 and
    64..1023), far and near offsets, and matches that run into the end of the buffer.  Deterministic in (n, seed)."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    alphabets = [256, 256, 16, 4, 2]
    while len(out) < n:
        kind = int(rng.integers(0, 8))
        if kind <= 1 or len(out) < 8:                                   # literals
            ln = int(rng.choice([1, 3, 14, 15, 16, 40, 270, 271, 600, 5000]))
            out += rng.integers(0, int(rng.choice(alphabets)), ln, dtype=np.uint8).tobytes()
        elif kind == 2:                                                 # run / short period (overlapping match)
            period = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 33, 63]))
            ln = int(rng.choice([5, 19, 64, 300, 1025, 4000, 70000]))
            pat = bytes(out[-period:]) if len(out) >= period else b"\x00" * period
            out += (pat * (ln // period + 1))[:ln]
        elif kind == 3:                                                 # medium period (overlap, 64 <= period < 1024)
            period = int(rng.integers(64, 1024))
            if len(out) >= period:
                ln = int(rng.integers(period, 6 * period))
                pat = bytes(out[-period:])
                out += (pat * (ln // period + 1))[:ln]
        else:                                                           # copy from earlier: near, far, beyond the 64 KiB window
            back = int(rng.choice([4, 17, 100, 1000, 5000, 30000, 65535, 65536, 200000]))
            ln = int(rng.choice([4, 5, 8, 15, 18, 19, 20, 100, 273, 274, 512, 1023, 1024, 1025, 3000, 20000]))
            if back <= len(out):
                start = len(out) - back
                ln = min(ln, back)
                out += out[start:start + ln]
    return bytes(out[:n])
