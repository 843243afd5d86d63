#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ from the COMPILED REFERENCE
(oracle/_ref/libsnkref.so = /root/reference/src objects + oracle/ref_shim.cpp).

Run in the build container (needs /root/reference to have been compiled by
`make -C oracle ref`).  Only DATA is written: seeded synthetic inputs (as the
generator arguments that reproduce them, plus a checksum), the reference's
per-read records and its stats block in sparse form.  No reference source or
script text is stored.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import snk_testlib as T  # noqa: E402
from cases import PE_CASES, se_kwargs  # noqa: E402
from soapnuke_amd import abi, synth  # noqa: E402

N = 3000
SHAPES = [  # (tag, paired, L, var_len, seed)
    ("pe150", True, 150, False, 101),
    ("pe150var", True, 150, True, 102),
    ("se100", False, 100, False, 103),
    ("pe250", True, 250, False, 104),
]


def main():
    assert T.have_ref(), "build the reference first: make -C oracle ref"
    index = []
    for tag, paired, L, var_len, seed in SHAPES:
        d = synth.make_batch(N, L, paired=paired, var_len=var_len, seed=seed)
        digest = hashlib.sha256(b"".join(np.ascontiguousarray(x).tobytes() for x in d["seq"] + d["qual"])).hexdigest()
        for name in sorted(PE_CASES):
            if tag == "pe250" and name not in ("C3_full", "C2_adatrim_lowq"):
                continue
            kw = PE_CASES[name] if paired else se_kwargs(PE_CASES[name])
            p = abi.default_params(paired=paired, max_read_len=L, **kw)
            r = T.run_ref(p, d)
            assert r["rc"] == 0
            nz = np.nonzero(r["sum"])[0].astype(np.int64)
            fn = f"{tag}__{name}.npz"
            np.savez_compressed(os.path.join(HERE, fn),
                                rec1=r["rec"][0].view(np.uint8).reshape(-1, 16),
                                rec2=r["rec"][1].view(np.uint8).reshape(-1, 16),
                                sum_idx=nz, sum_val=r["sum"][nz], max=r["max"])
            index.append(dict(file=fn, shape=tag, paired=paired, L=L, var_len=var_len, seed=seed, n=N,
                              case=name, input_sha256=digest))
    # the inputs themselves for one shape, so the vectors stay usable if the generator changes
    d = synth.make_batch(N, 150, paired=True, var_len=False, seed=101)
    np.savez_compressed(os.path.join(HERE, "input_pe150.npz"), seq1=d["seq"][0], seq2=d["seq"][1],
                        qual1=d["qual"][0], qual2=d["qual"][1])
    json.dump(dict(generator="tests/golden/make_golden.py", reference="BGI-flexlab/SOAPnuke v2.1.9 (oracle/_ref/libsnkref.so)",
                   vectors=index), open(os.path.join(HERE, "index.json"), "w"), indent=1)
    print(len(index), "golden vectors written")


if __name__ == "__main__":
    main()
