"""Child process of tests/test_simt_isa_interp.py: one batch through the tiled kernel of the EMULATED library with SIMT_DUMP_DIR
set (tests/simt/simt_runtime.cpp: kernarg segment and all device memory before and after every launch of the kernels whose
library offsets SIMT_DUMP_OFFSETS names).  argv[1]: a JSON object -- case (tests/cases.py), n, L, var_len, paired, pitch, seed,
lower (fraction of the reads that get lower-case letters or many N: the sequential fall-back inside the kernel),
first (the batch starts at this row: planes that are not 16-byte aligned), errors ([["seq"|"qual", mate, row, position, byte], ...]: offending
characters -- the kernels' error paths; only the error word is compared with the oracle then), kw (parameters on top of the case's), plant (fraction of the reads that get whole / truncated / mutated copies of the configured adapters), dup (host
verdict bits), first_index, contam (a CONTAM_CASES name), kernel (the C ABI's
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
    extra = {}
    if spec.get("plant"):                               # the case's own first adapters as the read-through material, and copies of every adapter planted
        ak = dict(PE_CASES[spec["case"]], **spec.get("kw", {}))
        extra["adapters"] = (ak["adapters1"][0], ak.get("adapters2", ak["adapters1"])[0])
    d = synth.make_batch(n, L, paired=paired, seed=int(spec.get("seed", 5)), var_len=bool(spec.get("var_len", False)), pitch=spec.get("pitch"),
                         dimer_frac=float(spec.get("dimer_frac", 0.0)), **extra)
    if spec.get("plant"):                               # whole / truncated / mutated copies in a fraction of the reads (tests/test_adapter_fuzz_gpu.py)
        from test_adapter_fuzz_gpu import plant
        prng = np.random.default_rng(int(spec.get("seed", 5)) + 1000)
        for m in range(len(d["seq"])):
            plant(prng, d["seq"][m], d["len"][m], L, ak["adapters%d" % (m + 1)] if ("adapters%d" % (m + 1)) in ak else ak["adapters1"], float(spec["plant"]))
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
    for which, mate, row, pos, ch in spec.get("errors", []):      # offending characters (the error paths: only `err` is defined afterwards)
        d[which][int(mate)][int(row), int(pos)] = int(ch)
    first = int(spec.get("first", 0))
    if first:
        d = {"n": n - first, "L": d["L"], "pitch": d["pitch"], "seq": [x[first:] for x in d["seq"]], "qual": [x[first:] for x in d["qual"]],
             "len": [None if x is None else x[first:] for x in d["len"]]}
    kw = dict(PE_CASES[spec["case"]])
    if not paired:
        kw = {k: v for k, v in kw.items() if not k.endswith("2")}
    if spec.get("contam"):                               # contaminant lists on top (tests/cases.py CONTAM_CASES), copies planted in the reads
        ck = CONTAM_CASES[spec["contam"]] if isinstance(spec["contam"], str) else {k: (tuple(v) if isinstance(v, list) else v) for k, v in spec["contam"].items()}
        plant_contams(d, ck)
        kw.update(contam_kwargs(ck, paired))
    kw.update(spec.get("kw", {}))                        # any parameter of abi.default_params on top of the case
    kw = {k: (tuple(v) if isinstance(v, list) and k in ("trim_bad_head", "trim_bad_tail", "ada_mis", "ada_mr", "ada_edge") else v) for k, v in kw.items()}
    p = abi.default_params(paired=paired, max_read_len=L, **kw)
    for i, v in enumerate(spec.get("hard_trim_without_flag", [])):      # values in hard_trim[] while has_hard_trim stays 0: the flag gates them (include/snk_filter.h)
        p.hard_trim[i] = int(v)
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
        dup = None
        if spec.get("dup"):                              # the host's verdict bits (bit 0 duplicate, 1 tile, 2 fov) on a tenth of the rows each
            dup = np.random.default_rng(2).choice(np.arange(8, dtype=np.uint8), d["n"], p=[.72, .04, .04, .04, .04, .04, .04, .04])
            lib.simt_dump_register(dup.ctypes.data, dup.nbytes)
        got = S.run_device(p, d, kernel=int(spec.get("kernel", 2)), dup=dup, first_index=int(spec.get("first_index", 0)))
    finally:
        np.zeros = real_zeros
    want = T.run_oracle(p, d, dup=dup, first_index=int(spec.get("first_index", 0)))
    if spec.get("errors"):
        assert tuple(got["err"]) == tuple(want["err"]) and want["err"][0] != 0, (got["err"], want["err"])
        print("captured")
        return
    for m in range(2 if paired else 1):
        assert np.array_equal(got["rec"][m], want["rec"][m]), "emulated records differ from the oracle"
    assert np.array_equal(got["sum"], want["sum"])
    print("captured")


if __name__ == "__main__":
    main()
