"""Seeded synthetic inputs shared by tests, smoke() and bench.py (no reference data ships offline)."""
import numpy as np


def zipf_bytes(n: int, seed: int = 1234, alpha: float = 1.1) -> np.ndarray:
    """i.i.d. bytes, P(byte = k) ~ (k+1)^-alpha  (BASELINE.json configs[1]; order-0 entropy 5.766 bits/byte)."""
    p = (np.arange(1, 257, dtype=np.float64)) ** -alpha
    cdf = np.cumsum(p / p.sum())
    u = np.random.default_rng(seed).random(n)
    return np.minimum(np.searchsorted(cdf, u), 255).astype(np.uint8)


def text_like(n: int, seed: int = 7) -> np.ndarray:
    """Zipf-distributed words from a 4096-word vocabulary: a declared stand-in for Silesia 'dickens' (absent offline).
    (Vectorised: the bytes are those of b"".join(vocab[i] + seps[s] ...) over the same random draws, 64 MiB in ~3 s.)"""
    rng = np.random.default_rng(seed)
    vocab = [bytes(rng.integers(97, 123, int(rng.integers(2, 11)), dtype=np.uint8)) for _ in range(4096)]
    p = (np.arange(1, 4097, dtype=np.float64)) ** -1.0
    idx = rng.choice(4096, size=n // 4 + 16, p=p / p.sum())
    seps = [b" ", b" ", b" ", b", ", b". ", b"\n"]
    sep = rng.integers(0, len(seps), size=len(idx))
    tab = np.zeros((4096 * 6, 12), dtype=np.uint8)              # token = word + separator, at most 10 + 2 bytes
    tl = np.zeros(4096 * 6, dtype=np.int32)
    for w, v in enumerate(vocab):
        for s, sp in enumerate(seps):
            t = v + sp
            tab[w * 6 + s, :len(t)] = np.frombuffer(t, dtype=np.uint8)
            tl[w * 6 + s] = len(t)
    tok = (idx * 6 + sep).astype(np.int32)
    ln = tl[tok]
    ends = np.cumsum(ln, dtype=np.int64)
    k = min(len(tok), int(np.searchsorted(ends, n, side="left")) + 1)      # tokens that cover n bytes
    tok, ln, ends = tok[:k], ln[:k], ends[:k]
    starts = ends - ln
    out = np.zeros(int(ends[-1]) + 16, dtype=np.uint8)
    for j in range(12):
        m = ln > j
        out[starts[m] + j] = tab[tok[m], j]
    return out[:n].copy()


def gen(kind: str, n: int, seed: int = 0) -> bytes:
    rng = np.random.default_rng(seed)
    if kind == "zipf":
        return zipf_bytes(n, seed).tobytes()
    if kind == "rand":
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == "zeros":
        return bytes(n)
    if kind == "bytei":                       # GenerateBuffer, T/ZstdNetTests.cs:617-622
        return (np.arange(n, dtype=np.uint32) & 255).astype(np.uint8).tobytes()
    if kind == "text":
        return text_like(n, seed).tobytes()
    if kind == "runs":
        out = bytearray()
        while len(out) < n:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 300))
        return bytes(out[:n])
    if kind == "period":
        pat = rng.integers(0, 256, int(rng.integers(2, 40)), dtype=np.uint8).tobytes()
        return (pat * (n // len(pat) + 1))[:n]
    if kind == "mixed":
        kinds = ["text", "zipf", "runs", "rand", "period"]
        return b"".join(gen(k, n // 5 + 1, seed + i) for i, k in enumerate(kinds))[:n]
    raise ValueError(kind)


KINDS = ["zipf", "rand", "zeros", "bytei", "text", "runs", "period", "mixed"]
