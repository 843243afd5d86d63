"""Deterministic synthetic FASTQ-shaped read batches.

Used by the tests, the golden-vector generator and bench.py.  numpy only; the
output is the SoA layout of include/snk_filter.h (uint8 [n, pitch] per array).

Distributions: iid uniform ACGT fragments; insert size ~ N(350,100) clipped to
[40,900] so that ~2-3 % of PE150 pairs read through into the adapter (adapter +
random tail appended, a fraction of them with 1-3 substitutions inside the
adapter to exercise the mismatch budget); 1 % of reads get 10 % N; 1 % of R2 get
a 30-60 nt poly-G tail; per-read quality profile 85 % good (Q~N(36,4)) and 5 %
each with mean 30 / 12 / 4, clipped to [2,41] (reference quirk Q4: Q<=41).
"""
import numpy as np

ADAPTER1 = "AAGTCGGAGGCCAAGCGGTCTTAGGAAGACAA"            # README example, 32 nt
ADAPTER2 = "AAGTCGGATCGTAGCCATGTCGTTCTGTGAGCCAAGGAGTTG"  # README example, 42 nt
SEED = 20260928
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def pitch_for(L):
    return (L + 15) // 16 * 16


def _mate(rng, n, L, pitch, ins, adapter, polyg_frac, var_len):
    seq = _BASES[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
    ada = np.frombuffer(adapter.encode(), dtype=np.uint8)
    pos = np.arange(L, dtype=np.int32)[None, :]
    j = pos - ins[:, None]                       # index into the adapter
    inside = (j >= 0) & (j < len(ada))
    a_chars = ada[np.clip(j, 0, len(ada) - 1)]
    # substitutions inside the adapter for ~30 % of read-through reads
    mut = inside & (rng.random((n, L)) < 0.04) & (rng.random(n) < 0.3)[:, None]
    a_chars = np.where(mut, _BASES[rng.integers(0, 4, size=(n, L), dtype=np.uint8)], a_chars)
    seq = np.where(inside, a_chars, seq)
    # N-rich reads
    nrich = rng.random(n) < 0.01
    seq = np.where(nrich[:, None] & (rng.random((n, L)) < 0.10), np.uint8(ord("N")), seq)
    # poly-G tails
    if polyg_frac > 0:
        pg = rng.random(n) < polyg_frac
        glen = rng.integers(30, 61, size=n)
        seq = np.where(pg[:, None] & (pos >= (L - glen)[:, None]), np.uint8(ord("G")), seq)
    # qualities
    prof = rng.random(n)
    mean = np.where(prof < 0.85, 36.0, np.where(prof < 0.90, 30.0, np.where(prof < 0.95, 12.0, 4.0)))
    q = np.rint(rng.normal(mean[:, None], 4.0, size=(n, L))).astype(np.int32)
    # a few reads with bad ends (exercise trimBadHead/Tail)
    bad_tail = rng.random(n) < 0.05
    tl = rng.integers(1, 25, size=n)
    q = np.where(bad_tail[:, None] & (pos >= (L - tl)[:, None]), q - 25, q)
    bad_head = rng.random(n) < 0.02
    hl = rng.integers(1, 12, size=n)
    q = np.where(bad_head[:, None] & (pos < hl[:, None]), q - 25, q)
    q = np.clip(q, 2, 41).astype(np.uint8) + 33
    lens = None
    if var_len:
        lens = rng.integers(max(len(ada) + 8, L // 2), L + 1, size=n).astype(np.uint16)
    S = np.full((n, pitch), 0xEE, dtype=np.uint8)   # garbage padding on purpose
    Q = np.full((n, pitch), 0xEE, dtype=np.uint8)
    S[:, :L] = seq
    Q[:, :L] = q
    if lens is not None:
        beyond = pos >= lens[:, None].astype(np.int32)
        S[:, :L][beyond] = 0xEE
        Q[:, :L][beyond] = 0xEE
    return S, Q, lens


def make_batch(n, L=150, paired=True, seed=SEED, adapters=(ADAPTER1, ADAPTER2),
               var_len=False, pitch=None):
    """Returns dict(seq=[S1,S2], qual=[Q1,Q2], len=[l1,l2] or None, L, pitch, n)."""
    rng = np.random.default_rng(seed)
    pitch = pitch or pitch_for(L)
    ins = np.clip(np.rint(rng.normal(350, 100, size=n)), 40, 900).astype(np.int32)
    if L >= 200:                                     # keep ~3 % read-through at PE250 too
        ins = np.clip(np.rint(rng.normal(350 * L / 150, 100 * L / 150, size=n)), 40, 2000).astype(np.int32)
    out = {"seq": [], "qual": [], "len": [], "L": L, "pitch": pitch, "n": n, "paired": paired}
    for m in range(2 if paired else 1):
        S, Q, lens = _mate(rng, n, L, pitch, ins, adapters[m], 0.01 if m == 1 else 0.0, var_len)
        out["seq"].append(S)
        out["qual"].append(Q)
        out["len"].append(lens)
    return out
