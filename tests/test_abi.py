"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/snk_filter.h declares, and the ctypes mirror matches
the C structs.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import snk_testlib as T
from soapnuke_amd import abi


def test_library_exports_every_declared_symbol():
    from soapnuke_amd import build
    build.build()
    lib = C.CDLL(abi.LIB_PATH)
    hdr = "".join(open(os.path.join(T.ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(T.ROOT, "include")))
                  if h.endswith(".h"))                     # every public header: snk_filter.h, snk_rmdup.h
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)      # drop comments
    declared = set(re.findall(r"\b(snk_[a-z_]+)\s*\(", hdr)) - {"snk_file_block_u64", "snk_stats_u64",
                                                               "snk_file_off", "snk_bs_off", "snk_qs_off", "snk_ts_off", "snk_adapter_at"}
    assert declared == set(abi.EXPORTS), declared ^ set(abi.EXPORTS)
    for s in declared:
        assert hasattr(lib, s), s


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "snk_filter.h"\n#include "snk_fastq.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(snk_params),sizeof(snk_batch),sizeof(snk_read_result),sizeof(snk_error),'
                   'offsetof(snk_params,adapters),offsetof(snk_batch,first_index));'
                   'printf("%zu %zu\\n",offsetof(snk_params,adapter_list),sizeof(snk_fastq_format));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(T.ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.Params), C.sizeof(abi.Batch), C.sizeof(abi.ReadResult), C.sizeof(abi.Error),
            abi.Params.adapters.offset, abi.Batch.first_index.offset, abi.Params.adapter_list.offset, C.sizeof(abi.FastqFormat)]
    assert got == want
    assert C.sizeof(abi.ReadResult) == 16 and abi.record_dtype().itemsize == 16


def test_create_without_device_fails_loudly():
    """No silent CPU fallback: without a HIP device snk_create() must fail with a message."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    lib = abi.load_library()
    p = abi.default_params()
    assert not lib.snk_create(C.byref(p), 0)
    assert lib.snk_last_error()


def test_default_params_match_reference_defaults():
    lib = abi.load_library()
    p = abi.Params()
    lib.snk_params_default(C.byref(p))
    q = abi.default_params()
    for name, _ in abi.Params._fields_:
        if name in ("adapters", "adapter_list"):
            continue
        a, b = getattr(p, name), getattr(q, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name
