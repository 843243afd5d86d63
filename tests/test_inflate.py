"""The CLI's own gzip/DEFLATE decoder (soapnuke_amd/host/snk_inflate.h) against zlib: byte-identical output on
every kind of stream the reader can meet, errors on damaged ones.  Pure host code (g++ + zlib), no GPU."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

import snk_testlib as T
from soapnuke_amd import synth

SRC = os.path.join(T.ROOT, "tools", "micro", "inflate_test.cpp")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("inf") / "inflate_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", out, SRC, "-lz"])
    return out


def _fastq_bytes(n=20000):
    d = synth.make_batch(n, 150, paired=False, seed=3)
    rows = []
    for i in range(n):
        rows.append(b"@SNK:1:1101:%09d/1\n" % i + d["seq"][0][i, :150].tobytes() + b"\n+\n" + d["qual"][0][i, :150].tobytes() + b"\n")
    return b"".join(rows)


def test_identical_to_zlib(exe, tmp_path):
    raw = _fastq_bytes()
    files = {}
    for lvl in (1, 2, 6, 9):
        files[f"l{lvl}"] = gzip.compress(raw, compresslevel=lvl)
    co = zlib.compressobj(0, zlib.DEFLATED, 31)
    files["stored"] = co.compress(raw[:1500000]) + co.flush()
    files["multi"] = gzip.compress(raw[:1000], 1) + gzip.compress(b"", 6) + gzip.compress(raw[1000:2_000_000], 2) + gzip.compress(raw[2_000_000:], 9)
    files["empty"] = gzip.compress(b"")
    files["tiny"] = gzip.compress(b"@r\nACGT\n+\nIIII\n")
    files["runs"] = gzip.compress(b"A" * 1_000_000 + b"ACGT" * 100000, 9)
    files["random"] = gzip.compress(np.random.default_rng(1).integers(0, 256, 1_000_000, dtype=np.uint8).tobytes(), 6)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)                  # fixed Huffman blocks
    files["fixed"] = co.compress(raw[:300000]) + co.flush()
    # trailing garbage behind the last member, of any length (zlib's gzread looks at the magic first and stops quietly; ADVICE r5: fewer
    # than 18 bytes used to be "truncated gzip member" here)
    for k, pad in enumerate((b"\0", b"\0" * 7, b"\x1f", b"\x1f\x00pad", b"\0" * 64)):
        files[f"garbage{k}"] = gzip.compress(raw[:40000], 6) + pad
    files["named_header"] = b"\x1f\x8b\x08\x08\x00\x00\x00\x00\x00\x03name.fq\x00" + gzip.compress(raw[:50000])[10:]
    for name, blob in files.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        for chunk in ("4194304", "777"):                                          # big blocks; tiny blocks = careful path + cut matches
            r = subprocess.run([exe, p, chunk], capture_output=True)
            assert r.returncode == 0 and b"IDENTICAL" in r.stdout, (name, chunk, r.stdout[-200:])
        # the parallel decoder (snk_pgunzip.h): 64 KiB chunks = dozens of block-start searches / marker chunks per file
        for threads, cb in (("4", "65536"), ("3", "300000")):
            r = subprocess.run([exe, p, "1000000", "par", threads, cb], capture_output=True)
            assert r.returncode == 0 and b"IDENTICAL" in r.stdout, (name, threads, cb, r.stdout[-200:])


def test_damaged_streams_are_errors(exe, tmp_path):
    raw = _fastq_bytes(3000)
    good = gzip.compress(raw, 6)
    bad_crc = bytearray(good)
    bad_crc[-6] ^= 0x55
    truncated = good[: len(good) // 2]
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0xFF
    for name, blob in (("crc", bytes(bad_crc)), ("trunc", truncated), ("flip", bytes(flipped)), ("notgz", b"@r\nACGT\n+\nIIII\n" * 10)):
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        r = subprocess.run([exe, p, "65536"], capture_output=True)
        assert r.returncode == 2 and b"ERROR" in r.stdout, (name, r.stdout[-200:])
        r = subprocess.run([exe, p, "65536", "par", "4", "65536"], capture_output=True)
        assert r.returncode == 2 and b"ERROR" in r.stdout, (name, "par", r.stdout[-200:])


def test_parallel_decoder_on_a_long_stream(exe, tmp_path):
    """one 60 MB FASTQ stream (levels 1 / 6 / 9, and cut into three members): every chunk size, more chunks than threads,
    member ends inside chunks; the output is what zlib produces, CRC-32 and ISIZE of every member hold"""
    raw = _fastq_bytes(180000 if os.environ.get("SNK_SIMT_FULL") == "1" else 60000)       # (an ordinary run: 20 MB)
    blobs = {"l1": gzip.compress(raw, 1), "l6": gzip.compress(raw, 6), "l9": gzip.compress(raw, 9),
             "three": gzip.compress(raw[:7_000_000], 6) + gzip.compress(raw[7_000_000:7_000_100], 9) + gzip.compress(raw[7_000_100:], 2)}
    for name, blob in blobs.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        for threads, cb in (("8", "65536"), ("4", "1000000"), ("2", "4194304")):
            r = subprocess.run([exe, p, "4194304", "par", threads, cb], capture_output=True)
            assert r.returncode == 0 and b"IDENTICAL" in r.stdout, (name, threads, cb, r.stdout[-200:])
    # a flipped bit in the middle of the long stream is an error (CRC at the latest), never silent garbage
    bad = bytearray(blobs["l6"])
    bad[len(bad) // 2] ^= 0x10
    p = str(tmp_path / "bad.gz")
    open(p, "wb").write(bytes(bad))
    r = subprocess.run([exe, p, "4194304", "par", "4", "1000000"], capture_output=True)
    assert r.returncode == 2 and b"ERROR" in r.stdout, r.stdout[-200:]


def test_parallel_decoder_between_members_without_block_starts(exe, tmp_path):
    """a long stretch of single-final-block members (what libdeflate's bgzip writes: nothing the block-start search
    accepts) between two ordinary streams.  The round-2 scheduler dead-locked here -- the chain thread waited for a chunk no
    worker was allowed to take -- and one thread searched the whole stretch for block starts.  Small thread counts, default and
    small chunk sizes; zlib's bytes, within seconds."""
    raw = _fastq_bytes(60000)
    mid = b"".join(gzip.compress(raw[k:k + 6000], 6) for k in range(0, 9_000_000, 6000))       # ~ 1500 one-block members
    stored = zlib.compressobj(0, zlib.DEFLATED, 31)
    blobs = {
        "members": gzip.compress(raw[:4_000_000], 6) + mid + gzip.compress(raw[4_000_000:9_000_000], 6),
        "stored_run": gzip.compress(raw[:3_000_000], 6) + stored.compress(raw[:6_000_000]) + stored.flush() + gzip.compress(raw[:3_000_000], 1),
        "members_only": mid,
    }
    for name, blob in blobs.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        for threads, cb in (("2", "2097152"), ("6", "2097152"), ("2", "65536"), ("3", "131072")):
            r = subprocess.run([exe, p, "4194304", "par", threads, cb], capture_output=True, timeout=120)
            assert r.returncode == 0 and b"IDENTICAL" in r.stdout, (name, threads, cb, r.stdout[-200:])
