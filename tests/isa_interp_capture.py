"""Child process of tests/test_simt_isa_interp.py: one batch through the tiled kernel of the EMULATED library with SIMT_DUMP_DIR
set (tests/simt/simt_runtime.cpp: kernarg segment and all device memory before and after every launch of the kernels whose
library offsets SIMT_DUMP_OFFSETS names).  argv[1]: a JSON object -- case (tests/cases.py), n, L, var_len, paired, pitch, seed,
lower (fraction of the reads that get lower-case letters or many N: the sequential fall-back inside the kernel),
first (the batch starts at this row: planes that are not 16-byte aligned), contam (a CONTAM_CASES name), kernel (the C ABI's
kernel selector: 2 tiled, 0 auto -- long reads take the long path)."""
import ctypes as C
import json
import sys

import numpy as np

import simt_lib as S
import snk_testlib as T
from cases import CONTAM_CASES, PE_CASES, contam_kwargs, plant_contams
from soapnuke_amd import abi, synth


def main():
    spec = json.loads(sys.argv[1])
    n, L, paired = int(spec["n"]), int(spec.get("L", 150)), bool(spec.get("paired", True))
    lib = S.lib()
    lib.simt_dump_register.argtypes = [C.c_void_p, C.c_size_t]
    if "any_length" in spec or "long_any_length" in spec:
        # the contexts of the GPU tier's first-contact tests (adapters of 1..255 characters, adaEdge beyond the adapter), smaller
        if "any_length" in spec:
            from test_adapter_fuzz_gpu import any_length_context
            p, d = any_length_context(int(spec["any_length"]), n)
        else:
            from test_long_reads_gpu import long_any_length_context
            p, d = long_any_length_context(*spec["long_any_length"], n=n)
        return run_and_check(lib, spec, p, d, True)
    d = synth.make_batch(n, L, paired=paired, seed=int(spec.get("seed", 5)), var_len=bool(spec.get("var_len", False)), pitch=spec.get("pitch"),
                         dimer_frac=float(spec.get("dimer_frac", 0.0)))
    if spec.get("lower"):
        rng = np.random.default_rng(99)
        for m in range(len(d["seq"])):
            rows = rng.choice(n, max(1, int(n * float(spec["lower"]))), replace=False)
            for r in rows:
                ln = int(d["len"][m][r]) if d["len"][m] is not None else L
                k = int(rng.integers(0, 3))
                if k == 0:
                    d["seq"][m][r, :ln] |= 0x20                                  # all lower case
                elif k == 1:
                    d["seq"][m][r, int(rng.integers(0, ln)):] |= 0x20            # lower case from somewhere on
                else:
                    d["seq"][m][r, :ln][rng.random(ln) < 0.3] = ord("N")
    first = int(spec.get("first", 0))
    if first:
        d = {"n": n - first, "L": d["L"], "pitch": d["pitch"], "seq": [x[first:] for x in d["seq"]], "qual": [x[first:] for x in d["qual"]],
             "len": [None if x is None else x[first:] for x in d["len"]]}
    kw = dict(PE_CASES[spec["case"]])
    if not paired:
        kw = {k: v for k, v in kw.items() if not k.endswith("2")}
    if spec.get("contam"):                               # contaminant lists on top (tests/cases.py CONTAM_CASES), copies planted in the reads
        ck = CONTAM_CASES[spec["contam"]]
        plant_contams(d, ck)
        kw.update(contam_kwargs(ck, paired))
    p = abi.default_params(paired=paired, max_read_len=L, **kw)
    return run_and_check(lib, spec, p, d, paired)


def run_and_check(lib, spec, p, d, paired):
    keep = []
    for key in ("seq", "qual", "len"):
        for a in d[key]:
            if a is not None:
                base = a.base if a.base is not None else a
                lib.simt_dump_register(base.ctypes.data, base.nbytes)
                keep.append(a)
    real_zeros = np.zeros

    def zeros(*a, **k):                                  # the record arrays run_device() makes: device memory too
        x = real_zeros(*a, **k)
        if x.dtype == abi.record_dtype():
            lib.simt_dump_register(x.ctypes.data, x.nbytes)
            keep.append(x)
        return x

    np.zeros = zeros
    try:
        got = S.run_device(p, d, kernel=int(spec.get("kernel", 2)))
    finally:
        np.zeros = real_zeros
    want = T.run_oracle(p, d)
    for m in range(2 if paired else 1):
        assert np.array_equal(got["rec"][m], want["rec"][m]), "emulated records differ from the oracle"
    assert np.array_equal(got["sum"], want["sum"])
    print("captured")


if __name__ == "__main__":
    main()
