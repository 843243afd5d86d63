"""The CLI's own gzip encoder (soapnuke_amd/host/snk_deflate.h): whatever it writes must decompress -- with zlib and with
this repo's decoder -- to exactly the input, for every kind of data and every slice size (one gzip member per slice, as
the writer threads produce them).  Pure host code, no GPU."""
import os
import subprocess

import numpy as np
import pytest

import snk_testlib as T
from test_inflate import _fastq_bytes

SRC = os.path.join(T.ROOT, "tools", "micro", "deflate_test.cpp")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("defl") / "deflate_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", out, SRC, "-lz"])
    return out


def test_round_trip_all_kinds(exe, tmp_path):
    raw = _fastq_bytes(40000)
    rng = np.random.default_rng(4)
    binned = bytearray(raw)
    files = {
        "fastq": raw,
        "random": rng.integers(0, 256, 1_500_000, dtype=np.uint8).tobytes(),
        "runs": b"A" * 700_000 + b"ACGT" * 100_000 + bytes(range(256)) * 300 + b"F" * 300_000,
        "empty": b"",
        "one": b"x",
        "tiny": b"@r\nACGT\n+\nIIII\n",
        "two_symbols": b"ab" * 5000,                       # a distance code with a single symbol
        "skewed": bytes(rng.choice(256, 600_000, p=np.array([0.5 ** min(i + 1, 40) for i in range(256)]) / sum(0.5 ** min(i + 1, 40) for i in range(256))).astype(np.uint8)),  # code lengths past 15: the limiter
        "zeros": bytes(2_000_000),
    }
    for name, blob in files.items():
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        for slice_bytes in ("0", "1000003", "65536", "4099"):
            if slice_bytes != "0" and len(blob) > 3_000_000 and slice_bytes == "4099":
                continue
            r = subprocess.run([exe, p] + ([slice_bytes] if slice_bytes != "0" else []), capture_output=True)
            assert r.returncode == 0 and b"ROUNDTRIP_OK" in r.stdout, (name, slice_bytes, r.stdout[-300:])


def test_fastq_mode_round_trips(exe, tmp_path):
    """the record-aware front end (deflate_fastq: what the CLI uses): FASTQ of every shape, text that only looks like
    FASTQ, names repeated on the '+' line, names longer than a match, a truncated last record, CRLF"""
    rng = np.random.default_rng(9)
    raw = _fastq_bytes(30000)

    def recs(n, name, plus_name=False, L=100):
        out = bytearray()
        for i in range(n):
            nm = name(i)
            seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L))
            q = bytes(rng.integers(33, 74, L, dtype=np.uint8))
            out += b"@" + nm + b"\n" + seq + b"\n+" + (nm if plus_name else b"") + b"\n" + q + b"\n"
        return bytes(out)

    files = {
        "fastq": raw,
        "plus_name": recs(20000, lambda i: b"SRR1234567.%d %d/1" % (i, i), plus_name=True),
        "long_names": recs(3000, lambda i: b"x" * 700 + b"%d" % i + b"y" * 300, plus_name=True, L=30),
        "changing_names": recs(20000, lambda i: bytes(rng.integers(48, 123, int(rng.integers(1, 40)), dtype=np.uint8)).replace(b"\n", b"_")),
        "huge_name": b"@" + b"ab" * 300000 + b"\nACGT\n+\nIIII\n" + b"@" + b"ab" * 300000 + b"\nACGT\n+\nIIII\n",
        "truncated": raw[:len(raw) - 77],
        "crlf": raw[:400000].replace(b"\n", b"\r\n"),
        "at_random": b"@" + rng.integers(0, 256, 1_000_000, dtype=np.uint8).tobytes(),
        "at_newlines": b"@" + b"\n" * 500000,
        "at_only": b"@",
    }
    for name, blob in files.items():
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        for slice_bytes in ("0", "1000003"):
            r = subprocess.run([exe, p] + ([slice_bytes] if slice_bytes != "0" else []), capture_output=True, env=dict(os.environ, FASTQ="1"))
            assert r.returncode == 0 and b"ROUNDTRIP_OK" in r.stdout, (name, slice_bytes, r.stdout[-300:])
    # ... and it is not worse than the hash search on FASTQ
    sizes = {}
    for mode in ("0", "1"):
        env = dict(os.environ)
        if mode == "1":
            env["FASTQ"] = "1"
        r = subprocess.run([exe, str(tmp_path / "fastq")], capture_output=True, env=env)
        sizes[mode] = int(r.stdout.decode().split("->")[1].split("bytes")[0])
    assert sizes["1"] <= 1.01 * sizes["0"], sizes


def test_ratio_is_in_zlib_low_level_territory(exe, tmp_path):
    raw = _fastq_bytes(60000)
    p = str(tmp_path / "fq")
    open(p, "wb").write(raw)
    r = subprocess.run([exe, p, "2000000"], capture_output=True)
    assert r.returncode == 0, r.stdout[-300:]
    txt = r.stdout.decode()
    ours = int(txt.split("->")[1].split("bytes")[0])
    zl = int(txt.split("zlib -2:")[1].split(")")[0])
    assert ours < 1.03 * zl, (ours, zl)


def test_crc32_matches_zlib(tmp_path):
    """snk_crc32.h (PCLMULQDQ folding) is the same function as zlib's crc32_z: 20 000 random (offset, length, seed) cases
    and a 64 MB buffer (tools/micro/crc_test.cpp)."""
    exe = str(tmp_path / "crc_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(T.ROOT, "tools", "micro", "crc_test.cpp"), "-lz"])
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 0 and b"CRC_OK" in r.stdout, r.stdout[-300:]


def test_symbol_buffer_growth_under_asan(tmp_path):
    """one name line can append far more symbols than the slack behind BLOCK_SYMS (its literal runs plus one match
    piece per 258 equal bytes, for every equal run of the line, behind a capacity check per run).  12-byte records fill the
    symbol buffer to just below the block limit; the name line behind them repeats itself at distance 12 -- a few short
    equal runs, then one of 32 000 bytes.  Built with AddressSanitizer (the round-2 header writes past the buffer here:
    heap-buffer-overflow at ntiny = 32728), must round-trip."""
    exe = str(tmp_path / "deflate_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address", "-std=c++17", "-o", exe, SRC, "-lz"])

    def giant(nshort, nlong):
        c = bytearray(b"@BCDEFGHIJKL")
        for _ in range(nshort):
            for _ in range(8):
                c.append(c[len(c) - 12])
            c.append((c[len(c) - 12] + 1 - 65) % 26 + 65)         # differs
        for _ in range(nlong):
            c.append(c[len(c) - 12])
        return bytes(c)

    for ntiny in (32728, 32731, 32735):
        for nshort in (20, 28):
            blob = b"@aaaa\nA\n+\nI\n" * ntiny + giant(nshort, 32000) + b"\nACGT\n+\nIIII\n"
            p = str(tmp_path / "blob")
            open(p, "wb").write(blob)
            r = subprocess.run([exe, p], capture_output=True, env=dict(os.environ, FASTQ="1", ASAN_OPTIONS="detect_leaks=0"))
            assert r.returncode == 0 and b"ROUNDTRIP_OK" in r.stdout, (ntiny, nshort, r.stdout[-200:], r.stderr[-600:])
