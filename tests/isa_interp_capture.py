"""Child process of tests/test_isa_interp.py: one batch through the tiled kernel of the EMULATED library with SIMT_DUMP_DIR set
(tests/simt/simt_runtime.cpp: kernarg segment and all device memory before and after every launch of the kernels whose library
offsets SIMT_DUMP_OFFSETS names).  argv: case, pairs, read length, variable lengths (0 / 1), paired (0 / 1) [, seed]."""
import ctypes as C
import sys

import numpy as np

import simt_lib as S
import snk_testlib as T
from cases import PE_CASES
from soapnuke_amd import abi, synth


def main():
    case, n, L, var_len, paired = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1", sys.argv[5] == "1"
    seed = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    lib = S.lib()
    lib.simt_dump_register.argtypes = [C.c_void_p, C.c_size_t]
    d = synth.make_batch(n, L, paired=paired, seed=seed, var_len=var_len)
    p = abi.default_params(paired=paired, max_read_len=L, **PE_CASES[case])
    keep = []
    for key in ("seq", "qual", "len"):
        for a in d[key]:
            if a is not None:
                lib.simt_dump_register(a.ctypes.data, a.nbytes)
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
        got = S.run_device(p, d, kernel=2)
    finally:
        np.zeros = real_zeros
    want = T.run_oracle(p, d)
    for m in range(2 if paired else 1):
        assert np.array_equal(got["rec"][m], want["rec"][m]), "emulated records differ from the oracle"
    assert np.array_equal(got["sum"], want["sum"])
    print("captured")


if __name__ == "__main__":
    main()
