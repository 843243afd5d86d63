"""Parameter sets shared by the oracle-vs-reference, golden and GPU parity tests.
Names follow BASELINE.json's configs (C1..C3) plus edge cases."""
from soapnuke_amd import synth

A1, A2 = synth.ADAPTER1, synth.ADAPTER2

PE_CASES = {
    # C1-like defaults, no adapter
    "defaults": dict(),
    # C2: -f A1 -r A2 -J -l 10 -q 0.1
    "C2_adatrim_lowq": dict(adapters1=[A1], adapters2=[A2], ada_trim=1, low_qual=10, low_qual_ratio=0.1),
    # same in discard mode (default adapter_discard_or_trim)
    "C2_adadiscard": dict(adapters1=[A1], adapters2=[A2], low_qual=10, low_qual_ratio=0.1),
    # C3: C2 + -n 0.01 -m 20 -g 10 -X 50 -p 0.8 + trimBadTail=20,30
    "C3_full": dict(adapters1=[A1], adapters2=[A2], ada_trim=1, low_qual=10, low_qual_ratio=0.1,
                    n_ratio=0.01, mean_quality=20, polyG_tail=10, polyX_num=50, highA_ratio=0.8,
                    trim_bad_tail=(20, 30)),
    "hard_lq_trim": dict(adapters1=[A1], adapters2=[A2], ada_trim=1, hard_trim=[3, 5, 2, 7],
                         trim_bad_head=(15, 10), trim_bad_tail=(20, 30)),
    "polyG_only": dict(polyG_tail=20),
    "all_off": dict(low_qual_ratio=-1, n_ratio=-1, min_read_length=-1),
    "meanq_polyx": dict(mean_quality=30, polyX_num=12, low_qual_ratio=-1, highA_ratio=0.4),
    "multi_adapter_params2": dict(adapters1=["GGGGGGGGGGGGTTTTACGT", A1], adapters2=[A2, A1], ada_trim=1,
                                  ada_mis=(1, 3), ada_mr=(0.6, 0.4), ada_edge=(8, 5), max_read_length=140,
                                  min_read_length=50),
    "short_adapter_edge": dict(adapters1=["AAGTCGG"], adapters2=["AAGTCGGATC"], ada_trim=1, ada_edge=(6, 4)),
}


def se_kwargs(kw):
    kw = dict(kw)
    if "hard_trim" in kw:
        kw["hard_trim"] = kw["hard_trim"][:2]
    return kw


# contaminant screening: config strings exactly as the reference takes them
CT1, CT2 = "ACGTTGCAAGGCTTAACCGGTTAGCATGCAAT", "TTGGCCAAGGTTCCAAGGTTAACCGGTT"
GC1 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
CONTAM_CASES = {
    "single": dict(contam1=CT1, contam2=CT2, ct_match_r="0.5"),
    "single_default_mr": dict(contam1=CT1 + CT1[:25], ct_match_r="0.2"),
    "list": dict(contam1=CT1 + ",GGGGGGGGGGGGGGGGGGGGGGGG", contam2=CT2 + ",CCCCCCCCCCCCCCCCCCCCCC", ct_match_r="0.6,0.7"),
    "global": dict(global_contams=CT1 + "," + GC1, g_mrs="0.5,0.4", g_mms="1,2"),
    # mate 2 of a pair is screened with adaMis2 / adaEdge2 (gp2, src/sequence.cpp:182-189)
    "single_params2": dict(contam1=CT1, contam2=CT2, ct_match_r="0.5", ada_mis=(2, 0), ada_edge=(6, 12)),
    "both_trim": dict(contam1=CT1, ct_match_r="0.4", global_contams=GC1, g_mrs="0.4", g_mms="1", contam_trim=1),
    "both_discard": dict(contam1=CT1, contam2=CT1, ct_match_r="0.4", global_contams=GC1, g_mrs="0.4", g_mms="1",
                         adapters1=[A1], adapters2=[A2], ada_trim=1),
}


def plant_contams(d, kw, seed=4):
    """writes whole / head-truncated / tail-truncated copies of the configured contaminants into 5 % of the reads each"""
    import numpy as np
    rng = np.random.default_rng(seed)
    seqs = []
    for k in ("contam1", "contam2", "global_contams"):
        if k in kw:
            seqs += kw[k].split(",")
    n = d["n"]
    for m in range(len(d["seq"])):
        for s in seqs:
            b = np.frombuffer(s.encode(), dtype=np.uint8)
            for r in rng.choice(n, n // 20, replace=False):
                L = int(d["len"][m][r]) if d["len"][m] is not None else d["L"]
                k, mode = int(rng.integers(min(8, len(b)), len(b) + 1)), int(rng.integers(0, 3))
                k = min(k, L)                                 # (a contaminant longer than the read: as much of it as fits)
                if mode == 0 and L >= len(b):
                    p = int(rng.integers(0, L - len(b) + 1))
                    d["seq"][m][r, p:p + len(b)] = b
                elif mode == 1:
                    d["seq"][m][r, :k] = b[len(b) - k:]
                else:
                    d["seq"][m][r, L - k:L] = b[:k]


def contam_kwargs(kw, paired):
    return {k: v for k, v in kw.items() if paired or k not in ("contam2", "adapters2")}


# ---- random parameter contexts (VERDICT r2 task 1): oracle vs compiled reference, HIP vs oracle

def rebase_quality(d, phred=33, qmax=41, seed=7):
    """moves the qualities of a synth batch (Phred-33, values 2..41) to another offset / value range: values above
    `qmax` are cut, and when `qmax` > 41 a tenth of the positions is redrawn uniformly from 0..qmax"""
    import numpy as np
    rng = np.random.default_rng(seed)
    L = d["L"]
    for m in range(len(d["qual"])):
        q = d["qual"][m][:, :L].astype(np.int16) - 33
        if qmax > 41:
            q = np.where(rng.random(q.shape) < 0.1, rng.integers(0, qmax + 1, q.shape), q)
        q = np.clip(q, 0, qmax)
        d["qual"][m][:, :L] = (q + phred).astype(np.uint8)
        if d["len"][m] is not None:                      # keep the garbage past the read ends
            beyond = np.arange(L)[None, :] >= d["len"][m][:, None].astype(np.int32)
            d["qual"][m][:, :L][beyond] = 0xEE
    return d


def random_context(rng, L, paired, min_len):
    """one random parameter set the reference defines (no UB zone): trimBadHead/Tail limits stay <= the shortest read
    (quirk Q11: beyond it src/read_filter.cpp:411-426 reads outside the quality string), qualities stay inside
    [0, maxBaseQuality] (Q4), adapters are not longer than the shortest read (Q6)"""
    import numpy as np
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    kw = {}
    nada = int(rng.integers(0, 4))
    if nada:
        for k in ("adapters1", "adapters2")[:2 if paired else 1]:
            kw[k] = [bytes(B[rng.integers(0, 4, int(rng.integers(8, min(60, min_len - 2))))]).decode() for _ in range(nada)]
        kw["ada_trim"] = int(rng.integers(0, 2))
        kw["ada_mis"] = (int(rng.integers(0, 5)), int(rng.integers(0, 5)))
        kw["ada_mr"] = (float(rng.choice([0.2, 0.4, 0.5, 0.7, 1.0])), float(rng.choice([0.3, 0.5, 0.8])))
        kw["ada_edge"] = (int(rng.integers(1, 8)), int(rng.integers(1, 8)))      # <= the shortest adapter
    kw["low_qual"] = int(rng.integers(0, 30))
    kw["low_qual_ratio"] = float(rng.choice([-1, 0.05, 0.1, 0.3, 0.5, 0.9]))
    kw["n_ratio"] = float(rng.choice([-1, 0.0, 0.01, 0.05, 0.2]))
    if rng.random() < 0.5:
        kw["mean_quality"] = int(rng.integers(5, 38))
    if rng.random() < 0.4:
        kw["polyG_tail"] = float(rng.choice([5, 10, 20.5, 40]))
    if rng.random() < 0.4:
        kw["polyX_num"] = int(rng.integers(3, 60))
    if rng.random() < 0.4:
        kw["highA_ratio"] = float(rng.choice([0.2, 0.3, 0.5, 0.8]))
    kw["min_read_length"] = int(rng.choice([-1, 0, 30, 60, L - 10]))
    if rng.random() < 0.3:
        kw["max_read_length"] = int(rng.integers(L // 2, L + 5))
    if rng.random() < 0.4:
        kw["hard_trim"] = [int(x) for x in rng.integers(0, L + 20 if rng.random() < 0.2 else 12, 4 if paired else 2)]
    if rng.random() < 0.4:
        kw["trim_bad_head"] = (int(rng.integers(3, 35)), int(rng.integers(1, min_len + 1)))
    if rng.random() < 0.5:
        kw["trim_bad_tail"] = (int(rng.integers(3, 35)), int(rng.integers(1, min_len + 1)))
    return kw
