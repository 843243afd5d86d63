"""Sanitizer runs of the threaded host code that has no HIP in it: the parallel gzip decoder -- reader /
worker / chain threads, chunk hand-over, fallbacks -- and the device-inflate orchestration with its CPU backend, built with
ThreadSanitizer and with AddressSanitizer + UndefinedBehaviorSanitizer, on single-member, multi-member and damaged streams.
(The CLI itself links the HIP runtime and only runs on the GPU box: tests/test_cli_gpu.py::test_cli_under_sanitizers.)"""
import gzip
import os
import subprocess

import numpy as np
import pytest

import snk_testlib as T
from test_inflate import _fastq_bytes

SRC = os.path.join(T.ROOT, "tools", "micro", "inflate_test.cpp")
EMUL = os.path.join(T.ROOT, "tests", "host_emul")


def _files(tmp_path):
    raw = _fastq_bytes(6000)
    out = {"single": gzip.compress(raw, 2),
           "multi": gzip.compress(raw[:300000], 1) + gzip.compress(b"", 6) + gzip.compress(raw[300000:], 6),
           "damaged": bytes(bytearray(gzip.compress(raw, 6))[:600000]) + b"\x55" * 64}
    paths = {}
    for k, v in out.items():
        paths[k] = str(tmp_path / (k + ".gz"))
        open(paths[k], "wb").write(v)
    return paths


@pytest.mark.parametrize("san,env", [("thread", {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"}),
                                     ("address,undefined", {"ASAN_OPTIONS": "detect_leaks=0", "UBSAN_OPTIONS": "halt_on_error=1 print_stacktrace=1"})])
def test_parallel_gunzip_under_sanitizers(san, env, tmp_path):
    exe = str(tmp_path / "inflate_san")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=" + san, "-fno-sanitize-recover=all", "-std=c++17", "-pthread", "-o", exe, SRC, "-lz"])
    paths = _files(tmp_path)
    for name in ("single", "multi"):
        for threads, cb in (("4", "65536"), ("3", "200000")):
            r = subprocess.run([exe, paths[name], "500000", "par", threads, cb], capture_output=True, env=dict(os.environ, **env))
            assert r.returncode == 0 and b"IDENTICAL" in r.stdout, (san, name, threads, r.stdout[-200:], r.stderr[-1500:])
    r = subprocess.run([exe, paths["damaged"], "500000", "par", "4", "65536"], capture_output=True, env=dict(os.environ, **env))
    assert r.returncode == 2 and b"ERROR" in r.stdout and b"Sanitizer" not in r.stderr, (san, r.stdout[-200:], r.stderr[-1500:])


def test_device_inflate_orchestration_under_asan_ubsan(tmp_path):
    """host/snk_dgunzip.h + csrc/snk_inflate_core.hip.h (CPU backend) with AddressSanitizer + UBSan: windows, chains, the threaded CRC,
    the fallback to the sequential decoder"""
    import ctypes as C
    lib_path = str(tmp_path / "libemul_asan.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-w", "-fPIC", "-shared",
                           "-I" + os.path.join(T.ROOT, "soapnuke_amd", "csrc"), "-x", "c++", "inflate_emul.cpp", "-o", lib_path, "-lz", "-pthread"], cwd=EMUL)
    raw = _fastq_bytes(6000)
    blob = gzip.compress(raw[:500000], 1) + gzip.compress(raw[500000:], 6)
    code = f"""
import ctypes as C, sys
lib = C.CDLL({lib_path!r})
lib.snk_emul_dgunzip.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_long), C.c_char_p, C.c_size_t]
lib.snk_emul_dgunzip.restype = C.c_long
blob = open({str(tmp_path / 'b.gz')!r}, 'rb').read()
want = open({str(tmp_path / 'b.raw')!r}, 'rb').read()
for window, chunk, spc in ((1 << 22, 1 << 16, 1 << 20), (300000, 1 << 16, 1 << 20), (1 << 22, 1 << 16, 90000)):
    out = C.create_string_buffer(len(want) + 64)
    info = (C.c_long * 4)()
    err = C.create_string_buffer(200)
    r = lib.snk_emul_dgunzip(blob, len(blob), window, chunk, spc, 16, out, len(want) + 32, info, err, 200)
    assert r == len(want) and out.raw[:r] == want, (r, err.value)
print('ASAN_RUN_OK')
"""
    open(str(tmp_path / "b.gz"), "wb").write(blob)
    open(str(tmp_path / "b.raw"), "wb").write(raw)
    asan = subprocess.check_output(["g++", "-print-file-name=libasan.so"]).decode().strip()
    r = subprocess.run(["python", "-c", code], capture_output=True,
                       env=dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1 print_stacktrace=1"))
    assert r.returncode == 0 and b"ASAN_RUN_OK" in r.stdout, (r.stdout[-300:], r.stderr[-2500:])


def test_device_inflate_orchestration_under_tsan(tmp_path):
    """The producer thread of host/snk_dgunzip.h (window k + 1 decoded while window k is read; two text slots, the threaded CRC,
    the hand-over to the sequential decoder, a reader that walks away early) with ThreadSanitizer, CPU backend"""
    exe = str(tmp_path / "dgz_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-sanitize-recover=all", "-w", "-DSNK_EMUL_MAIN",
                           "-I" + os.path.join(T.ROOT, "soapnuke_amd", "csrc"), "-x", "c++", "inflate_emul.cpp", "-o", exe, "-lz", "-pthread"], cwd=EMUL)
    raw = _fastq_bytes(6000)
    blob = gzip.compress(raw[:500000], 1) + gzip.compress(raw[500000:], 6)
    open(str(tmp_path / "b.gz"), "wb").write(blob)
    open(str(tmp_path / "b.raw"), "wb").write(raw)
    open(str(tmp_path / "d.gz"), "wb").write(blob[:len(blob) * 2 // 3] + b"\x55" * 64)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1")
    for window, chunk, spc in ((1 << 22, 1 << 16, 1 << 20), (200000, 1 << 16, 1 << 20), (1 << 22, 1 << 16, 90000)):
        r = subprocess.run([exe, str(tmp_path / "b.gz"), str(tmp_path / "b.raw"), str(window), str(chunk), str(spc)], capture_output=True, env=env)
        assert r.returncode == 0 and b"IDENTICAL" in r.stdout, (window, spc, r.stdout[-300:], r.stderr[-2500:])
    r = subprocess.run([exe, str(tmp_path / "d.gz"), str(tmp_path / "b.raw"), "200000", "65536", "1048576"], capture_output=True, env=env)
    assert r.returncode == 2 and b"ERROR" in r.stdout and b"Sanitizer" not in r.stderr, (r.stdout[-300:], r.stderr[-2500:])
