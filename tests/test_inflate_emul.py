"""The chunk decoder of the DEVICE inflate (soapnuke_amd/csrc/snk_inflate_core.hip.h: block-start probe, marker-mode DEFLATE decoding,
gzip framing) compiled for the host and run chunk by chunk the way the device path runs it -- against zlib's bytes on every kind of
stream the reader can meet (the vectors of tests/test_inflate.py), and errors on damaged ones.  No GPU needed."""
import ctypes as C
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

import snk_testlib as T
from test_inflate import _fastq_bytes

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
LIB = os.path.join(HERE, "libsnk_inflate_emul.so")


@pytest.fixture(scope="module")
def emul():
    srcs = [os.path.join(HERE, "inflate_emul.cpp"), os.path.join(T.ROOT, "soapnuke_amd", "csrc", "snk_inflate_core.hip.h"),
            os.path.join(T.ROOT, "soapnuke_amd", "host", "snk_dgunzip.h"), os.path.join(T.ROOT, "soapnuke_amd", "host", "snk_inflate.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-fPIC", "-shared", "-I" + os.path.join(T.ROOT, "soapnuke_amd", "csrc"),
                               "-x", "c++", "inflate_emul.cpp", "-o", LIB, "-lz", "-pthread"], cwd=HERE)
    lib = C.CDLL(LIB)
    lib.snk_emul_gunzip.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_long), C.c_long]
    lib.snk_emul_gunzip.restype = C.c_long
    lib.snk_emul_probe.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
    lib.snk_emul_set_coop.argtypes = [C.c_int]
    return lib


@pytest.fixture(params=[0, 1], ids=["one_thread_io", "wave_cooperative_io"])
def io_mode(emul, request):
    """the decoder's two I/O forms: plain loads / stores of one thread, and the wavefront-cooperative one (LDS input ring, LDS
    symbol buffer, lane-parallel flushes and match copies; the 64 lanes emulated in turn)"""
    emul.snk_emul_set_coop(request.param)
    yield request.param
    emul.snk_emul_set_coop(0)


def _gunzip(lib, blob, chunk, cap, ends_cap=64):
    out = np.zeros(cap + 16, dtype=np.uint8)
    info = (C.c_long * 4)()
    r = lib.snk_emul_gunzip(blob, len(blob), chunk, out.ctypes.data, cap, info, ends_cap)
    return r, bytes(out[:max(r, 0)]), list(info)


def _vectors():
    raw = _fastq_bytes(5000)                                    # 1.6 MB of FASTQ
    files = {f"l{lvl}": (gzip.compress(raw, compresslevel=lvl), raw) for lvl in (1, 2, 6, 9)}
    co = zlib.compressobj(0, zlib.DEFLATED, 31)
    files["stored"] = (co.compress(raw[:300000]) + co.flush(), raw[:300000])
    files["multi"] = (gzip.compress(raw[:1000], 1) + gzip.compress(b"", 6) + gzip.compress(raw[1000:700_000], 2) + gzip.compress(raw[700_000:], 9), raw)
    files["empty"] = (gzip.compress(b""), b"")
    files["tiny"] = (gzip.compress(b"@r\nACGT\n+\nIIII\n"), b"@r\nACGT\n+\nIIII\n")
    runs = b"A" * 300_000 + b"ACGT" * 50000
    files["runs"] = (gzip.compress(runs, 9), runs)
    rnd = np.random.default_rng(1).integers(0, 256, 200_000, dtype=np.uint8).tobytes()
    files["random"] = (gzip.compress(rnd, 6), rnd)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)                  # fixed Huffman blocks
    files["fixed"] = (co.compress(raw[:300000]) + co.flush(), raw[:300000])
    files["named_header"] = (b"\x1f\x8b\x08\x08\x00\x00\x00\x00\x00\x03name.fq\x00" + gzip.compress(raw[:50000])[10:], raw[:50000])
    many = b"".join(gzip.compress(raw[i:i + 20000], 1) for i in range(0, 400000, 20000))   # BGZF-like: a member per few KB
    files["many_members"] = (many, raw[:400000])
    return files


def test_chunked_marker_decoding_is_zlibs_bytes(emul, io_mode):
    total_markers = 0
    for name, (blob, raw) in _vectors().items():
        assert zlib.decompress(blob, 47) == raw[:len(zlib.decompress(blob, 47))] or name in ("multi", "many_members")
        for chunk in (1 << 30, 1 << 16, 20000):
            r, got, info = _gunzip(emul, blob, chunk, len(raw) + 1000)
            assert r == len(raw) and got == raw, (name, chunk, r, info)
            if name == "many_members" and chunk > 400000:       # more member ends in one chunk than it was given slots for: refused, never wrong
                assert _gunzip(emul, blob, chunk, len(raw) + 1000, ends_cap=8)[0] == -(1000 + 10 * 3 + 3)
            total_markers += info[2]
    assert total_markers > 300_000                              # the unknown-window path really ran


def test_block_start_probe_finds_zlibs_blocks_and_little_else(emul):
    """every dynamic block of a level-1 stream (found by decoding with zlib's Z_BLOCK-free cousin: the chain of the emulation
    itself) answers the probe; random offsets practically never do"""
    raw = _fastq_bytes(8000)
    blob = gzip.compress(raw, 1)
    rng = np.random.default_rng(5)
    res = [emul.snk_emul_probe(blob, len(blob), int(b)) for b in rng.integers(100, len(blob) * 8 - 100, 100000)]
    assert min(res) >= 0                                       # the lanes' quick screen never rejects what the full check accepts
    assert sum(1 for r in res if r & 1) <= 2                   # (a true block start may be hit by chance: ~100 of 20 M offsets)
    assert sum(1 for r in res if r & 2) < len(res) // 200      # the quick screen passes well under 1 % of the offsets
    r, got, info = _gunzip(emul, blob, 1 << 16, len(raw) + 100)
    assert got == raw and info[1] >= len(blob) // (1 << 16) - 1   # a start was found in (nearly) every 64 KiB chunk


def test_damaged_streams_are_errors_or_zlibs_bytes(emul, io_mode):
    """bit flips anywhere in the deflate data: the decoder refuses exactly what zlib's inflate refuses, and where zlib decodes
    (the damage only changed bytes: CRC-32 / ISIZE, which the caller checks, would catch it) it produces zlib's bytes"""
    raw = _fastq_bytes(3000)
    blob = bytearray(gzip.compress(raw, 6))
    rng = np.random.default_rng(9)
    refused = same = 0
    for _ in range(120):
        b = bytearray(blob)
        for pos in rng.integers(12, len(b) - 8, int(rng.integers(1, 4))):
            b[int(pos)] ^= 1 << int(rng.integers(0, 8))
        b = bytes(b)
        d = zlib.decompressobj(-15)
        try:
            want = d.decompress(b[10:])
            ok = d.eof                                          # (not eof: the damaged stream runs off the end of the file)
        except zlib.error:
            want, ok = None, False
        for chunk in (1 << 30, 1 << 16):
            r, got, info = _gunzip(emul, b, chunk, len(raw) * 3)
            if ok:
                if r >= 0:                                      # (a chunked run may also refuse: a damaged block start nobody can find)
                    assert got == want
                    same += 1
                else:
                    assert chunk < len(b)
                    refused += 1
            else:
                assert r < 0, (r, info)
                refused += 1
    assert refused >= 4 and same > 100
    for cut in (20, len(blob) // 2, len(blob) - 4):            # truncated files
        r, got, info = _gunzip(emul, bytes(blob[:cut]), 1 << 16, len(raw) * 2)
        assert r < 0


# ---- the host orchestration (soapnuke_amd/host/snk_dgunzip.h) over a CPU backend made of the same core functions
def _dgunzip(lib, blob, window, chunk, spc, cap, epc=16):
    out = np.zeros(cap + 16, dtype=np.uint8)
    info = (C.c_long * 4)()
    eb = C.create_string_buffer(256)
    r = lib.snk_emul_dgunzip(blob, len(blob), window, chunk, spc, epc, out.ctypes.data, cap, info, eb, 256)
    return r, bytes(out[:max(r, 0)]), list(info), eb.value.decode()


@pytest.fixture(scope="module")
def dg(emul):
    emul.snk_emul_dgunzip.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_long), C.c_char_p, C.c_size_t]
    emul.snk_emul_dgunzip.restype = C.c_long
    return emul


def test_device_gunzip_orchestration_windows_chains_and_fallbacks(dg, io_mode):
    """DeviceGunzip: windows of several sizes (a window's last chunk is cut by the window and decoded again by the next), chunks
    whose symbol slots overflow and BGZF-like files with more members per chunk than slots (both: sequential host decoder from the
    last good block), empty and tiny files -- always zlib's bytes"""
    vec = _vectors()
    device_only = 0
    for name, (blob, raw) in vec.items():
        for window, chunk, spc, epc in ((1 << 22, 1 << 16, 1 << 20, 64), (200000, 1 << 16, 1 << 20, 64), (1 << 22, 1 << 17, 150000, 2)):
            r, got, info, err = _dgunzip(dg, blob, window, chunk, spc, len(raw) + 1000, epc)
            assert r == len(raw) and got == raw, (name, window, chunk, spc, r, info, err)
            device_only += info[1] == -1
    assert device_only >= 2 * len(vec)                         # most runs never needed the fallback


def test_device_gunzip_checks_crc_and_length(dg):
    raw = _fastq_bytes(3000)
    co = zlib.compressobj(0, zlib.DEFLATED, 31)                 # stored blocks: a flipped payload bit decodes fine and must fail the CRC
    blob = bytearray(co.compress(raw) + co.flush())
    blob[5000] ^= 4
    r, got, info, err = _dgunzip(dg, bytes(blob), 1 << 22, 1 << 16, 1 << 20, len(raw) + 100)
    assert r == -1 and "CRC" in err
    blob = bytearray(gzip.compress(raw, 6))
    blob[-2] ^= 1                                               # ISIZE
    r, got, info, err = _dgunzip(dg, bytes(blob), 1 << 22, 1 << 16, 1 << 20, len(raw) + 100)
    assert r == -1
    for cut in (30, len(blob) // 2, len(blob) - 3):             # truncated files: an error, like zlib's gzread
        r, got, info, err = _dgunzip(dg, bytes(gzip.compress(raw, 6)[:cut]), 1 << 22, 1 << 16, 1 << 20, len(raw) + 100)
        assert r == -1, (cut, r, err)


def test_device_gunzip_member_ending_on_a_window_edge(dg):
    """a member whose trailer ends exactly where a window ends is not the end of the file -- the chunk's
    `stream_end` only says "the input given to the kernel ends here".  Windows swept around the seams of a three-member file (every
    offset of the 4-byte window alignment), BGZF-like small members, trailing garbage behind a seam on the edge (zlib's gzread stops
    quietly), and a truncated member header there (an error)."""
    raw = [_fastq_bytes(2500 + 300 * k) for k in range(3)]
    members = [gzip.compress(r, 6) for r in raw]
    blob, want = b"".join(members), b"".join(raw)
    seams = [len(members[0]), len(members[0]) + len(members[1])]
    exact = 0
    for seam in seams:
        for window in range(seam - 12, seam + 13):
            for chunk in (1 << 16, 1 << 20):
                r, got, info, err = _dgunzip(dg, blob, window, chunk, 1 << 21, len(want) + 1000)
                assert r == len(want) and got == want, (seam, window, chunk, r, info, err)
                exact += window == seam
    assert exact == 4
    # small members (BGZF-like) and windows that are multiples of the member size: many seams on edges
    small = [gzip.compress(want[k:k + 4096], 1) for k in range(0, 65536, 4096)]
    sblob, swant = b"".join(small), want[:65536]
    for window in sorted({len(small[0]), len(small[0]) + len(small[1]), sum(len(m) for m in small[:5]), 4096, 10000}):
        r, got, info, err = _dgunzip(dg, sblob, window, 1 << 12, 1 << 17, len(swant) + 1000, 8)
        assert r == len(swant) and got == swant, (window, r, info, err)
    # trailing garbage right behind a member that ends on the window's edge: the text of the members, no error
    r, got, info, err = _dgunzip(dg, members[0] + b"\0" * 64, len(members[0]), 1 << 16, 1 << 21, len(raw[0]) + 1000)
    assert r == len(raw[0]) and got == raw[0], (r, info, err)
    # ... of any length (fewer than 18 bytes used to be "truncated" whatever they were; zlib looks at the magic first)
    for pad in (b"\0", b"\0" * 5, b"\x1f", b"\x1f\x00" * 8, b"garbage!" * 2 + b"x"):
        r, got, info, err = _dgunzip(dg, members[0] + pad, len(members[0]), 1 << 16, 1 << 21, len(raw[0]) + 1000)
        assert r == len(raw[0]) and got == raw[0], (pad, r, info, err)
    # ... but a member that BEGINS there (the magic) and cannot be complete is truncated (as the sequential decoder says)
    r, got, info, err = _dgunzip(dg, members[0] + members[1][:9], len(members[0]), 1 << 16, 1 << 21, len(want))
    assert r == -1 and "truncated" in err, (r, err)


def test_device_gunzip_resumes_behind_a_host_spell(dg):
    """a region the device path refuses -- here 3 MB of one letter, whose chunk would need more symbol slots than
    it has -- is decoded by the host for ONE spell; the device windows resume at the next block header instead of leaving the rest
    of the file to one core.  Bytes are zlib's, CRC / ISIZE of the member that contains the region are still checked (one member:
    the CRC runs through device text, host text, device text), and a member boundary inside a spell hands over cleanly."""
    rng = np.random.default_rng(77)

    def fastq_like(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n)) + b"\n" + bytes(rng.integers(35, 75, n, dtype=np.uint8)) + b"\n"
    head, tail = b"".join(fastq_like(150) for _ in range(4000)), b"".join(fastq_like(150) for _ in range(9000))
    poly = b"N" * 3_000_000
    for level in (1, 6):
        want = head + poly + tail
        blob = gzip.compress(want, level)
        spc = 400_000                                    # symbol slots per chunk: the poly region's chunk overflows them
        r, got, info, err = _dgunzip(dg, blob, 1 << 20, 1 << 14, spc, len(want) + 1000)
        assert r == len(want) and got == want, (level, r, info, err)
        spells, resumes = info[3] & 0xFFFF, info[3] >> 16
        assert info[1] != -1 and spells >= 1 and resumes >= 1, (level, info)          # the host took over, and gave back
        # a damaged trailer behind all of that is still caught (the CRC ran through both engines)
        bad = bytearray(blob)
        bad[-6] ^= 0x40
        r, got, info, err = _dgunzip(dg, bytes(bad), 1 << 20, 1 << 14, spc, len(want) + 1000)
        assert r == -1 and "CRC" in err, (level, r, err)
        # two members, the seam inside the host's spell
        want2 = head + poly
        blob2 = gzip.compress(want2, level) + gzip.compress(tail, level)
        r, got, info, err = _dgunzip(dg, blob2, 1 << 20, 1 << 14, spc, len(want) + 1000)
        assert r == len(want) and got == want, (level, r, info, err)
