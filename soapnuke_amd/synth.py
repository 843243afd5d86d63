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
    # read-through rows only: adapter (+ substitutions for ~30 % of them) then random tail
    rt = np.nonzero(ins < L)[0]
    if len(rt):
        j = pos - ins[rt, None]
        inside = (j >= 0) & (j < len(ada))
        a_chars = ada[np.clip(j, 0, len(ada) - 1)]
        mut = inside & (rng.random((len(rt), L), dtype=np.float32) < 0.04) & (rng.random(len(rt)) < 0.3)[:, None]
        a_chars = np.where(mut, _BASES[rng.integers(0, 4, size=(len(rt), L), dtype=np.uint8)], a_chars)
        seq[rt] = np.where(inside, a_chars, seq[rt])
    # N-rich reads (1 %, 10 % N)
    nr = np.nonzero(rng.random(n) < 0.01)[0]
    if len(nr):
        seq[nr] = np.where(rng.random((len(nr), L), dtype=np.float32) < 0.10, np.uint8(ord("N")), seq[nr])
    # poly-G tails
    if polyg_frac > 0:
        pg = np.nonzero(rng.random(n) < polyg_frac)[0]
        if len(pg):
            glen = rng.integers(30, 61, size=len(pg))
            seq[pg] = np.where(pos >= (L - glen)[:, None], np.uint8(ord("G")), seq[pg])
    # qualities: per-read profile
    prof = rng.random(n)
    mean = np.where(prof < 0.85, 36.0, np.where(prof < 0.90, 30.0, np.where(prof < 0.95, 12.0, 4.0))).astype(np.float32)
    q = rng.standard_normal((n, L), dtype=np.float32)
    q *= 4.0
    q += mean[:, None]
    np.rint(q, out=q)
    # a few reads with bad ends (exercise trimBadHead/Tail)
    bt = np.nonzero(rng.random(n) < 0.05)[0]
    if len(bt):
        tl = rng.integers(1, 25, size=len(bt))
        q[bt] -= np.where(pos >= (L - tl)[:, None], np.float32(25), np.float32(0))
    bh = np.nonzero(rng.random(n) < 0.02)[0]
    if len(bh):
        hl = rng.integers(1, 12, size=len(bh))
        q[bh] -= np.where(pos < hl[:, None], np.float32(25), np.float32(0))
    np.clip(q, 2, 41, out=q)
    q = q.astype(np.uint8)
    q += 33
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
               var_len=False, pitch=None, dimer_frac=0.0):
    """Returns dict(seq=[S1,S2], qual=[Q1,Q2], len=[l1,l2] or None, L, pitch, n).
    dimer_frac > 0: that fraction of the pairs are adapter dimers / very short inserts -- insert size in
    [-12, 40), a negative insert meaning the read starts |insert| characters INTO the adapter (phase A of
    adapter_pos, src/read_filter.cpp:720-742).  The default stream of random numbers is unchanged by it."""
    rng = np.random.default_rng(seed)
    pitch = pitch or pitch_for(L)
    ins = np.clip(np.rint(rng.normal(350, 100, size=n)), 40, 900).astype(np.int32)
    if L >= 200:                                     # keep ~3 % read-through at PE250 too
        ins = np.clip(np.rint(rng.normal(350 * L / 150, 100 * L / 150, size=n)), 40, 2000).astype(np.int32)
    if dimer_frac > 0:
        rng2 = np.random.default_rng([seed, 0xD1AE])
        sel = np.nonzero(rng2.random(n) < dimer_frac)[0]
        ins[sel] = rng2.integers(-12, 40, size=len(sel))
    out = {"seq": [], "qual": [], "len": [], "L": L, "pitch": pitch, "n": n, "paired": paired}
    for m in range(2 if paired else 1):
        S, Q, lens = _mate(rng, n, L, pitch, ins, adapters[m], 0.01 if m == 1 else 0.0, var_len)
        out["seq"].append(S)
        out["qual"].append(Q)
        out["len"].append(lens)
    return out


def write_fastq(path, S, Q, L, mate, lens=None, first_index=0):
    """Plain 4-line FASTQ with fixed-width IDs `@SNK:1:1101:<idx>/<mate>` (vectorised)."""
    n = S.shape[0]
    if lens is not None:
        with open(path, "wb") as f:
            for i in range(n):
                f.write(b"@SNK:1:1101:%09d/%d\n" % (first_index + i, mate))
                f.write(S[i, :lens[i]].tobytes() + b"\n+\n" + Q[i, :lens[i]].tobytes() + b"\n")
        return
    idw = len(b"@SNK:1:1101:000000000/1\n")
    rec = np.empty((n, idw + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :12] = np.frombuffer(b"@SNK:1:1101:", dtype=np.uint8)
    idx = np.arange(first_index, first_index + n, dtype=np.int64)
    for k in range(9):
        rec[:, 12 + 8 - k] = (idx // 10 ** k % 10 + 48).astype(np.uint8)
    rec[:, 21] = ord("/")
    rec[:, 22] = 48 + mate
    rec[:, 23] = 10
    rec[:, idw:idw + L] = S[:, :L]
    rec[:, idw + L:idw + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, idw + L + 3:idw + 2 * L + 3] = Q[:, :L]
    rec[:, -1] = 10
    rec.tofile(path)
