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
                k, mode = int(rng.integers(8, len(b) + 1)), int(rng.integers(0, 3))
                if mode == 0 and L >= len(b):
                    p = int(rng.integers(0, L - len(b) + 1))
                    d["seq"][m][r, p:p + len(b)] = b
                elif mode == 1:
                    d["seq"][m][r, :k] = b[len(b) - k:]
                else:
                    d["seq"][m][r, L - k:L] = b[:k]


def contam_kwargs(kw, paired):
    return {k: v for k, v in kw.items() if paired or k not in ("contam2", "adapters2")}
