"""PCIe-inclusive rate of the host-pointer entry snk_filter_batch() (pageable numpy buffers in, records out):
python tools/bench_host_entry.py [pairs]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import snk_testlib as T  # noqa: E402
from cases import PE_CASES  # noqa: E402
from soapnuke_amd import abi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
lib = abi.load_library()
d = synth.make_batch(n, 150, paired=True, seed=5)
p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C2_adatrim_lowq"])
ctx = lib.snk_create(C.byref(p), 0)
assert ctx, lib.snk_last_error()
b = T.host_batch(d, 0, None)
r1 = np.zeros(n, dtype=abi.record_dtype())
r2 = np.zeros(n, dtype=abi.record_dtype())
for rep in range(3):
    t0 = time.time()
    assert lib.snk_filter_batch(ctx, C.byref(b), r1.ctypes.data, r2.ctypes.data) == 0, lib.snk_last_error()
    dt = time.time() - t0
    print(f"snk_filter_batch (host buffers, {n} PE150 pairs): {dt * 1e3:.1f} ms  {2 * n / dt / 1e6:.1f} Mreads/s  "
          f"{n * (4 * d['pitch'] + 32) / dt / 1e9:.1f} GB/s over the bus", flush=True)
lib.snk_destroy(ctx)
