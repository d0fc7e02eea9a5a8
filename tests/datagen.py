"""Seeded synthetic inputs shared by tests, smoke() and bench.py (no reference data ships offline)."""
import numpy as np


def zipf_bytes(n: int, seed: int = 1234, alpha: float = 1.1) -> np.ndarray:
    """i.i.d. bytes, P(byte = k) ~ (k+1)^-alpha  (BASELINE.json configs[1]; order-0 entropy 5.766 bits/byte)."""
    p = (np.arange(1, 257, dtype=np.float64)) ** -alpha
    cdf = np.cumsum(p / p.sum())
    u = np.random.default_rng(seed).random(n)
    return np.minimum(np.searchsorted(cdf, u), 255).astype(np.uint8)


def text_like(n: int, seed: int = 7) -> np.ndarray:
    """Zipf-distributed words from a 4096-word vocabulary: a declared stand-in for Silesia 'dickens' (absent offline)."""
    rng = np.random.default_rng(seed)
    vocab = [bytes(rng.integers(97, 123, int(rng.integers(2, 11)), dtype=np.uint8)) for _ in range(4096)]
    p = (np.arange(1, 4097, dtype=np.float64)) ** -1.0
    idx = rng.choice(4096, size=n // 4 + 16, p=p / p.sum())
    seps = [b" ", b" ", b" ", b", ", b". ", b"\n"]
    sep = rng.integers(0, len(seps), size=len(idx))
    out = b"".join(vocab[i] + seps[s] for i, s in zip(idx, sep))
    return np.frombuffer(out[:n], dtype=np.uint8).copy()


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
