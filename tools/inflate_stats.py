"""What the device inflate's decoder meets in gzip'ed FASTQ (CPU emulation, tests/host_emul): literals and matches per read, the
length of the matches, how often the match queue runs because it is full and how often because a source reaches into it.
    python tools/inflate_stats.py [reads=20000]"""
import ctypes as C
import gzip
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_inflate import _fastq_bytes  # noqa: E402

HERE = os.path.join(ROOT, "tests", "host_emul")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "soapnuke_amd", "csrc"), "-x", "c++", "inflate_emul.cpp",
                       "-o", "/tmp/libsnk_inflate_stats.so", "-lz", "-pthread"], cwd=HERE)
lib = C.CDLL("/tmp/libsnk_inflate_stats.so")
lib.snk_emul_gunzip.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_long), C.c_long]
lib.snk_emul_gunzip.restype = C.c_long
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
raw = _fastq_bytes(n)
for level in (1, 2, 4, 6, 9):
    blob = gzip.compress(raw, level)
    lib.snk_emul_set_coop(1)
    st = (C.c_ulonglong * 14)()
    lib.snk_emul_stats(st, 1)
    out = np.zeros(len(raw) + 64, dtype=np.uint8)
    info = (C.c_long * 4)()
    r = lib.snk_emul_gunzip(blob, len(blob), 1 << 17, out.ctypes.data, len(raw) + 32, info, 64)
    assert r == len(raw)
    lib.snk_emul_stats(st, 1)
    lit, mat, msym, fc, ff = [int(x) for x in st][:5]
    near = [int(x) / max(mat, 1) for x in st][5:13]
    nearq = int(st[13])
    print(f"level {level}: {len(blob) / len(raw):.3f} of the text; per read {lit / n:.0f} literals + {mat / n:.1f} matches of {msym / max(mat, 1):.1f} symbols; "
          f"queue runs: full / long {ff}, a near match needs a queued symbol {fc} (one match in {mat / max(fc, 1):.0f}); symbols per queue run {(lit + msym) / max(ff + fc, 1):.0f}; "
          f"copied in LDS at once {nearq / max(mat, 1):.2f}; matches nearer than 256 / 1 K / 4 K / 16 K: " + " / ".join(f"{near[k]:.2f}" for k in (0, 2, 4, 6)))
