"""The device inflate on the GPU (include/snk_gunzip.h, csrc/snk_inflate.hip) against zlib's bytes: the kernels through the C ABI
(search + marker decode, chain + resolve) on the streams of tests/test_inflate_emul.py, and the CLI with SNK_DEVICE_INFLATE=1 against
the reference binary.  The decoding core itself is pinned on zlib without a GPU by tests/test_inflate_emul.py."""
import ctypes as C
import filecmp
import gzip
import os
import zlib

import numpy as np
import pytest

import report_util as R
import snk_testlib as T
from soapnuke_amd import abi
from test_inflate_emul import _vectors

pytestmark = [pytest.mark.gpu, T.first_contact]


class Chunk(C.Structure):
    _fields_ = [("start_bit", C.c_uint64), ("stop_bit", C.c_uint64), ("out_off", C.c_uint64), ("out_cap", C.c_uint32),
                ("first_of_member", C.c_uint32), ("n_syms", C.c_uint32), ("status", C.c_uint32), ("end_bit", C.c_uint64),
                ("known_from", C.c_uint32), ("n_ends", C.c_uint32), ("stream_end", C.c_uint32), ("ends_off", C.c_uint32),
                ("ends_cap", C.c_uint32), ("pad_", C.c_uint32 * 3)]


def _first_block_bit(blob):
    flg, p = blob[3], 10
    if flg & 4:
        p += 2 + blob[p] + (blob[p + 1] << 8)
    if flg & 8:
        p = blob.index(0, p) + 1
    if flg & 16:
        p = blob.index(0, p) + 1
    if flg & 2:
        p += 2
    return p * 8


def _device_gunzip(lib, blob, chunk, spc, epc=64):
    """one window = the whole file; the chain walked as host/snk_dgunzip.h walks it"""
    assert C.sizeof(Chunk) == 80
    lib.snk_gunzip_create.restype = C.c_void_p
    lib.snk_gunzip_create.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.snk_gunzip_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    lib.snk_gunzip_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.snk_gunzip_destroy.argtypes = [C.c_void_p]
    n = len(blob)
    g = lib.snk_gunzip_create(0, max(n, chunk), chunk, spc, epc)
    assert g, lib.snk_last_error()
    try:
        nc = (n + chunk - 1) // chunk
        chunks = (Chunk * nc)()
        ends = np.zeros((nc * epc, 4), dtype=np.uint32)
        first = _first_block_bit(blob)
        assert lib.snk_gunzip_decode(g, blob, n, first, 1, chunks, ends.ctypes.data) == 0, lib.snk_last_error()
        order, expect, total, end_seen = [], first, 0, False
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
                end_seen = True
                break
        if not end_seen:
            return None, chunks
        text = np.zeros(total + 16, dtype=np.uint8)
        wout = np.zeros(32768, dtype=np.uint8)
        o = np.array(order, dtype=np.uint32)
        assert lib.snk_gunzip_resolve(g, o.ctypes.data, len(order), None, text.ctypes.data, total, wout.ctypes.data) == 0, lib.snk_last_error()
        got = bytes(text[:total])
        assert bytes(wout) == (b"\0" * 32768 + got)[-32768:]       # the window handed to the next call
        return got, chunks
    finally:
        lib.snk_gunzip_destroy(g)


@pytest.mark.parametrize("chunk", [1 << 16, 1 << 18])
def test_device_inflate_kernels_produce_zlibs_bytes(snk_lib, chunk):
    for name, (blob, raw) in _vectors().items():
        got, chunks = _device_gunzip(snk_lib, blob, chunk, 12 * chunk)
        assert got is not None and got == raw, (name, chunk, [(c.start_bit, c.end_bit, c.status) for c in chunks][:6])


def test_device_inflate_refuses_what_does_not_fit(snk_lib):
    raw = b"A" * 4_000_000                                          # 1000 : 1 -- far more symbols than a chunk's slots
    got, chunks = _device_gunzip(snk_lib, gzip.compress(raw, 6), 1 << 16, 1 << 18)
    assert got is None and chunks[0].status == 1                    # SNK_GZ_FULL: the host decoder takes over (host/snk_dgunzip.h)


def test_cli_with_device_inflate_matches_the_reference_binary(tmp_path):
    from test_cli_gpu import _cat, _run_ours
    case = R.REPORT_CASES[0]
    d, p = R.case_inputs(case)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    for env in ({"SNK_DEVICE_INFLATE": "1"}, {"SNK_DEVICE_INFLATE": "1", "SNK_DGZ_WINDOW_MB": "1", "SNK_DGZ_CHUNK_KB": "64"}):
        ours = _run_ours(case, work, gz=True, env=env)
        for f in R.REPORT_FILES_PE:
            assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
        for c in ("c1.fq", "c2.fq"):
            assert _cat(os.path.join(ours, c + ".gz")) == _cat(os.path.join(ref, c)), c
