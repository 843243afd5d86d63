"""Throughput of the device inflate (include/snk_gunzip.h) on synthetic FASTQ: python tools/bench_gunzip.py [Mpairs=4] [level=1] [chunk_kb=128]
One window = the whole file; prints the time of snk_gunzip_decode (upload + search + marker decode) and snk_gunzip_resolve
(chain + resolve + download), the text rate, the chunk statistics, and checks the text against zlib."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from soapnuke_amd import abi, synth  # noqa: E402
from test_gunzip_gpu import Chunk, _first_block_bit  # noqa: E402


def main():
    mp = float(sys.argv[1]) if len(sys.argv) > 1 else 4
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    chunk = (int(sys.argv[3]) if len(sys.argv) > 3 else 128) << 10
    n = int(mp * 1e6)
    tmp = tempfile.mkdtemp(prefix="snkgz_", dir="/dev/shm")
    try:
        u = min(n, 1_000_000)
        d = synth.make_batch(u, 150, paired=False)
        path = os.path.join(tmp, "a.fq")
        for k in range((n + u - 1) // u):
            synth.write_fastq(path + ".p", d["seq"][0][:min(u, n - k * u)], d["qual"][0][:min(u, n - k * u)], 150, 1, first_index=k * u)
            subprocess.check_call(f"cat {path}.p >> {path}", shell=True)
        subprocess.check_call(["gzip", f"-{level}", "-k", path])
        blob = open(path + ".gz", "rb").read()
        raw_len = os.path.getsize(path)
    finally:
        pass
    lib = abi.load_library()
    lib.snk_gunzip_create.restype = C.c_void_p
    lib.snk_gunzip_create.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.snk_gunzip_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    lib.snk_gunzip_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.snk_gunzip_destroy.argtypes = [C.c_void_p]
    nb = len(blob)
    epc = 8
    g = lib.snk_gunzip_create(0, nb, chunk, 12 * chunk, epc)
    assert g, lib.snk_last_error()
    nc = (nb + chunk - 1) // chunk
    chunks = (Chunk * nc)()
    ends = np.zeros((nc * epc, 4), dtype=np.uint32)
    first = _first_block_bit(blob)
    for rep in range(2):
        t0 = time.time()
        assert lib.snk_gunzip_decode(g, blob, nb, first, 1, chunks, ends.ctypes.data) == 0, lib.snk_last_error()
        t1 = time.time()
        order, expect, total = [], first, 0
        for c in range(nc):
            ck = chunks[c]
            if ck.start_bit == 2**64 - 1 or ck.start_bit < expect:
                continue
            if ck.start_bit != expect or ck.status != 0:
                break
            order.append(c)
            total += ck.n_syms
            expect = ck.end_bit
            if ck.stream_end:
                break
        text = np.zeros(total + 64, dtype=np.uint8)
        wout = np.zeros(32768, dtype=np.uint8)
        o = np.array(order, dtype=np.uint32)
        t2 = time.time()
        assert lib.snk_gunzip_resolve(g, o.ctypes.data, len(order), None, text.ctypes.data, total, wout.ctypes.data) == 0, lib.snk_last_error()
        t3 = time.time()
        print(f"run {rep}: {nb / 1e6:.0f} MB compressed (level {level}) -> {total / 1e6:.0f} MB text of {raw_len / 1e6:.0f}; {nc} chunks of {chunk >> 10} KB, "
              f"{len(order)} chained, found starts {sum(1 for c in chunks if c.start_bit != 2**64 - 1)}; decode {t1 - t0:.3f} s ({total / (t1 - t0) / 1e9:.2f} GB/s of text), "
              f"resolve + download {t3 - t2:.3f} s; statuses {sorted(set(c.status for c in chunks))}", flush=True)
    want = zlib.decompress(blob, 47)
    print("identical to zlib:", bytes(text[:total]) == want)
    lib.snk_gunzip_destroy(g)
    subprocess.call(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
