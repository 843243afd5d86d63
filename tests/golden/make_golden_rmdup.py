"""Generates tests/golden/rmdup.npz from the REFERENCE (oracle/_ref/libsnkref.so: std::hash<std::string>
as src/peprocess.cpp:3680 calls it, and rmdup::markDup of src/rmdup.cpp compiled where it lies).
Run in the build container:  python tests/golden/make_golden_rmdup.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import snk_testlib as T  # noqa: E402
from soapnuke_amd import synth  # noqa: E402

ref = T.ref_lib()
out = {}
# 1. a PE batch with variable lengths (seam of the two mates at every residue mod 8) and an SE batch
for name, paired, L, var in (("pe", True, 150, True), ("se", False, 100, True), ("pe_fixed", True, 150, False)):
    d = synth.make_batch(1500, L, paired=paired, var_len=var, seed=424242)
    n = d["n"]
    h = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        l1 = int(d["len"][0][i]) if d["len"][0] is not None else L
        s = bytes(d["seq"][0][i, :l1])
        if paired:
            l2 = int(d["len"][1][i]) if d["len"][1] is not None else L
            s += bytes(d["seq"][1][i, :l2])
        h[i] = ref.snkref_hash(s, len(s))
    out[name + "_hash"] = h
# 2. markDup on crafted hash arrays: heavy duplication, the 2^64-1 sentinel with and without bucket mates
rng = np.random.default_rng(99)
for k, n in enumerate((1, 9, 10, 1000, 50000)):
    h = rng.integers(0, max(2, n // 3), n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    if n >= 10:
        h[rng.integers(0, n, 3)] = np.uint64(0xFFFFFFFFFFFFFFFF)
    out[f"mark{k}_hash"] = h
    out[f"mark{k}_dup"] = T.ref_markdup(h)
h = rng.integers(0, 2**63, 1000, dtype=np.uint64)
h[500] = np.uint64(0xFFFFFFFFFFFFFFFF)        # a lone sentinel: flagged only if its bucket has company
out["mark_lone_hash"] = h
out["mark_lone_dup"] = T.ref_markdup(h)
np.savez_compressed(os.path.join(HERE, "rmdup.npz"), **out)
print({k: (v.shape, int(v.sum()) if v.dtype == np.uint8 else None) for k, v in out.items()})
