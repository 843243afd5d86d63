"""ctypes mirror of include/snk_filter.h (the C ABI of the hot path).

Plumbing only: no algorithm lives here.  The structs must stay byte-identical
to the header; tests/test_abi.py checks sizes and every exported symbol.
"""
import ctypes as C
import os

SNK_READ_MAX_LEN = 1000
SNK_MAX_ADAPTERS = 16
SNK_FS_N, SNK_GS_N, SNK_TS_N, SNK_MAX_N = 64, 16, 5000, 8
# enum snk_fs_index (families of 4: total, fq1 only, fq2 only, both)
SNK_FS_DUP, SNK_FS_SHORT, SNK_FS_LONG, SNK_FS_GCONTAM, SNK_FS_CONTAM = 0, 4, 8, 12, 16
SNK_FS_NRATE, SNK_FS_HIGHA, SNK_FS_POLYX, SNK_FS_LOWQUAL, SNK_FS_MEANQ, SNK_FS_ADAPTER = 20, 24, 28, 32, 36, 40

# enum snk_reason
KEEP, R_DUP, R_TILE, R_FOV, R_SHORT, R_LONG, R_GCONTAM, R_CONTAM, R_NRATE, R_HIGHA, \
    R_POLYX, R_LOWQUAL, R_MEANQ, R_OVERLAP, R_ADAPTER, R_EMPTY = range(16)
# enum snk_fs_index
FS_DUP, FS_TILE, FS_FOV, FS_OVERLAP = 0, 1, 2, 3
FS_SHORT, FS_LONG, FS_GCONTAM, FS_CONTAM, FS_NRATE, FS_HIGHA, FS_POLYX, FS_LOWQUAL, \
    FS_MEANQ, FS_ADAPTER = 4, 8, 12, 16, 20, 24, 28, 32, 36, 40
GS_READS, GS_BASES, GS_A, GS_C, GS_G, GS_T, GS_N, GS_Q20, GS_Q30 = range(9)
TS_HLQ, TS_HT, TS_TA, TS_TLQ, TS_TT = 0, 1000, 2000, 3000, 4000
# enum snk_error_code
OK, E_BAD_BASE, E_EMPTY_SEQ, E_QUAL_RANGE, E_TOO_LONG = 0, 1, 2, 3, 4


class Params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("paired", C.c_int32),
        ("quality_phred", C.c_int32), ("output_quality_phred", C.c_int32),
        ("max_base_quality", C.c_int32), ("low_qual", C.c_int32),
        ("low_qual_ratio", C.c_float), ("n_ratio", C.c_float),
        ("highA_ratio", C.c_float), ("polyG_tail", C.c_float),
        ("polyX_num", C.c_int32), ("mean_quality", C.c_int32),
        ("min_read_length", C.c_int32), ("max_read_length", C.c_int32),
        ("ada_trim", C.c_int32), ("contam_trim", C.c_int32),
        ("has_hard_trim", C.c_int32), ("hard_trim", C.c_int32 * 4),
        ("has_lq_trim", C.c_int32),
        ("lq_head_qual", C.c_int32), ("lq_head_len", C.c_int32),
        ("lq_tail_qual", C.c_int32), ("lq_tail_len", C.c_int32),
        ("ada_mis", C.c_int32 * 2), ("ada_mr", C.c_float * 2), ("ada_edge", C.c_int32 * 2),
        ("n_adapters", C.c_int32 * 2),
        ("adapters", (C.c_char_p * SNK_MAX_ADAPTERS) * 2),
        ("rmdup", C.c_int32), ("max_read_len", C.c_int32),
        ("contam", C.c_char_p * 2), ("ct_match_r", C.c_char_p),
        ("global_contams", C.c_char_p), ("g_mrs", C.c_char_p), ("g_mms", C.c_char_p),
        ("adapter_list", C.POINTER(C.c_char_p) * 2),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("pitch", C.c_int32), ("fixed_len", C.c_int32 * 2),
        ("seq", C.c_void_p * 2), ("qual", C.c_void_p * 2), ("len", C.c_void_p * 2),
        ("dup", C.c_void_p), ("first_index", C.c_uint64),
    ]


class ReadResult(C.Structure):
    _fields_ = [
        ("head_hdcut", C.c_int16), ("head_lqcut", C.c_int16),
        ("tail_hdcut", C.c_int16), ("tail_lqcut", C.c_int16),
        ("adacut_pos", C.c_int16), ("clean_start", C.c_uint16), ("clean_len", C.c_uint16),
        ("reason", C.c_uint8), ("flags", C.c_uint8),
    ]


class Error(C.Structure):
    _fields_ = [("code", C.c_int32), ("mate", C.c_int32), ("index", C.c_uint64)]


def record_dtype():
    import numpy as np
    return np.dtype([("head_hdcut", "<i2"), ("head_lqcut", "<i2"), ("tail_hdcut", "<i2"),
                     ("tail_lqcut", "<i2"), ("adacut_pos", "<i2"), ("clean_start", "<u2"),
                     ("clean_len", "<u2"), ("reason", "u1"), ("flags", "u1")])


def file_block_u64(lcap, nq):
    return SNK_GS_N + lcap * 5 + lcap * nq + SNK_TS_N


def stats_u64(lcap, nq):
    return SNK_FS_N + 4 * file_block_u64(lcap, nq)


def file_off(lcap, nq, k):
    return SNK_FS_N + k * file_block_u64(lcap, nq)


def bs_off(lcap, nq):
    return SNK_GS_N


def qs_off(lcap, nq):
    return SNK_GS_N + lcap * 5


def ts_off(lcap, nq):
    return SNK_GS_N + lcap * 5 + lcap * nq


def default_params(paired=True, max_read_len=150, **kw):
    """Reference defaults (src/global_parameter.h:20-83) + keyword overrides.

    adapters1/adapters2: lists of str; hard_trim: 4 (PE) / 2 (SE) ints;
    trim_bad_head / trim_bad_tail: (qual, maxlen) tuples; contam1/contam2/ct_match_r/
    global_contams/g_mrs/g_mms: the config file's comma-separated strings.
    """
    p = Params()
    p.struct_size = C.sizeof(Params)
    p.paired = 1 if paired else 0
    p.quality_phred = 33
    p.output_quality_phred = 33
    p.max_base_quality = 42
    p.low_qual = 5
    p.low_qual_ratio = 0.5
    p.n_ratio = 0.05
    p.highA_ratio = -1
    p.polyG_tail = -1
    p.polyX_num = -1
    p.mean_quality = -1
    p.min_read_length = 30
    p.max_read_length = -1
    for m in range(2):
        p.ada_mis[m], p.ada_mr[m], p.ada_edge[m] = 2, 0.5, 6
    p.max_read_len = max_read_len
    keep = []
    for k, v in kw.items():
        if k in ("adapters1", "adapters2"):
            m = 0 if k == "adapters1" else 1
            p.n_adapters[m] = len(v)
            bs = [a.encode() if isinstance(a, str) else bytes(a) for a in v]
            keep += bs
            if len(v) > SNK_MAX_ADAPTERS:                     # longer lists: snk_params.adapter_list
                arr = (C.c_char_p * len(v))(*bs)
                keep.append(arr)
                p.adapter_list[m] = C.cast(arr, C.POINTER(C.c_char_p))
            else:
                for i, b in enumerate(bs):
                    p.adapters[m][i] = b
        elif k == "hard_trim":
            p.has_hard_trim = 1
            for i, x in enumerate(v):
                p.hard_trim[i] = x
        elif k == "trim_bad_head":
            p.has_lq_trim = 1
            p.lq_head_qual, p.lq_head_len = v
        elif k == "trim_bad_tail":
            p.has_lq_trim = 1
            p.lq_tail_qual, p.lq_tail_len = v
        elif k in ("contam1", "contam2", "ct_match_r", "global_contams", "g_mrs", "g_mms"):
            b = v.encode() if isinstance(v, str) else bytes(v)      # comma-separated lists, as in the config file
            keep.append(b)
            if k == "contam1":
                p.contam[0] = b
            elif k == "contam2":
                p.contam[1] = b
            else:
                setattr(p, k, b)
        elif k in ("ada_mis", "ada_mr", "ada_edge"):
            arr = getattr(p, k)
            arr[0], arr[1] = (v if isinstance(v, (tuple, list)) else (v, v))
        else:
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
    p._keepalive = keep
    return p


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsnk_filter.so")

# every symbol include/snk_filter.h declares
EXPORTS = [
    "snk_params_default", "snk_create", "snk_destroy", "snk_last_error",
    "snk_stats_geometry", "snk_bind_stats", "snk_stats_clear",
    "snk_filter_batch_device", "snk_filter_batch", "snk_stats_finalize",
    "snk_stats_fetch", "snk_error_peek_async", "snk_error_decode", "snk_stats_allreduce", "snk_set_timing", "snk_last_kernel_ms", "snk_reserve",
    # include/snk_rmdup.h
    "snk_rmdup_hash_device", "snk_rmdup_bucket_count_device", "snk_rmdup_mark_device", "snk_rmdup_prime",
    "snk_rmdup_stream_create", "snk_rmdup_stream_mark_device", "snk_rmdup_stream_mark_se_device", "snk_rmdup_stream_stats", "snk_rmdup_stream_destroy", "snk_rmdup_stream_bytes",
    "snk_rmdup_partition_device", "snk_rmdup_flags_home_device",
    # include/snk_selftest.h
    "snk_selftest_bit_transpose",
    # include/snk_fastq.h
    "snk_fastq_tmp_bytes", "snk_fastq_parse_device", "snk_fastq_format_device",
    "snk_fastq_deflate_tmp_bytes", "snk_fastq_deflate_device",
    # include/snk_gunzip.h
    "snk_gunzip_create", "snk_gunzip_destroy", "snk_gunzip_decode", "snk_gunzip_resolve",
]


class FastqFormat(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("space_num", C.c_int32), ("qual_delta", C.c_int32), ("id_suffix_times", C.c_int32),
                ("id_suffix", C.c_char * 4), ("base_from", C.c_uint8), ("base_to", C.c_uint8), ("select_reason", C.c_uint8), ("whole_read", C.c_uint8)]


FQ_STATUS_N, FQ_F_LEN_MISMATCH, FQ_F_TOO_LONG, FQ_F_TRUNCATED = 4, 1, 2, 4


def load_library(path=None):
    """dlopen the HIP extension.  Fails loudly: there is no CPU fallback."""
    path = path or os.environ.get("SNK_LIB") or LIB_PATH      # SNK_LIB: profiling builds (tools/ablate.sh)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The hot path has no CPU fallback.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, i32 = C.c_void_p, C.c_int
    if os.environ.get("SNK_LIB") and path == os.environ["SNK_LIB"]:
        # an A/B library of an OLDER build (tools/variant_build.sh, abl/<tag>/soapnuke_amd/libsnk_filter.so under this tree's Python): entry
        # points it does not have yet become stubs that say so when called, instead of failing the load
        class _Tolerant:
            def __init__(self, real):
                object.__setattr__(self, "_real", real)

            def __getattr__(self, name):
                try:
                    return getattr(self._real, name)
                except AttributeError:
                    def missing(*a, **k):
                        raise RuntimeError(f"{path} (SNK_LIB) has no {name}(): a library of an older build")
                    missing.argtypes = missing.restype = None
                    return missing

            def __setattr__(self, name, value):
                setattr(self._real, name, value)
        lib = _Tolerant(lib)
    lib.snk_params_default.argtypes = [C.POINTER(Params)]
    lib.snk_params_default.restype = None
    lib.snk_create.argtypes = [C.POINTER(Params), i32]
    lib.snk_create.restype = vp
    lib.snk_destroy.argtypes = [vp]
    lib.snk_destroy.restype = None
    lib.snk_last_error.restype = C.c_char_p
    lib.snk_stats_geometry.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.snk_bind_stats.argtypes = [vp, vp, vp]
    lib.snk_stats_clear.argtypes = [vp, vp]
    lib.snk_filter_batch_device.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, i32]
    lib.snk_filter_batch.argtypes = [vp, C.POINTER(Batch), vp, vp]
    lib.snk_stats_finalize.argtypes = [vp, vp]
    lib.snk_stats_fetch.argtypes = [vp, vp, vp, C.POINTER(Error), vp]
    lib.snk_stats_allreduce.argtypes = [vp, vp, vp]
    lib.snk_set_timing.argtypes = [vp, i32]
    lib.snk_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.snk_rmdup_hash_device.argtypes = [vp, C.POINTER(Batch), vp, vp]
    lib.snk_rmdup_bucket_count_device.argtypes = [vp, vp, C.c_int64, C.c_uint64, vp, vp]
    lib.snk_rmdup_mark_device.argtypes = [vp, vp, vp, C.c_int64, C.c_uint64, C.c_int64, vp, vp]
    lib.snk_rmdup_prime.argtypes = [C.c_uint64]
    lib.snk_rmdup_prime.restype = C.c_uint32
    lib.snk_rmdup_stream_create.argtypes = [vp, C.c_uint64]
    lib.snk_rmdup_stream_create.restype = vp
    lib.snk_rmdup_stream_mark_device.argtypes = [vp, vp, C.c_uint64, C.c_int64, vp, vp]
    lib.snk_rmdup_stream_mark_se_device.argtypes = [vp, vp, C.c_uint64, C.c_int64, C.c_int64, vp, vp]
    lib.snk_rmdup_stream_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    lib.snk_rmdup_stream_destroy.argtypes = [vp]
    lib.snk_rmdup_stream_destroy.restype = None
    lib.snk_rmdup_partition_device.argtypes = [vp, vp, C.c_int64, C.c_uint64, C.c_int32, vp, vp, vp, vp, vp]
    lib.snk_rmdup_flags_home_device.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
    lib.snk_selftest_bit_transpose.argtypes = [i32, vp, i32, vp, vp]
    lib.snk_fastq_tmp_bytes.argtypes = [C.c_uint64, C.c_int64]
    lib.snk_fastq_tmp_bytes.restype = C.c_size_t
    lib.snk_fastq_parse_device.argtypes = [vp, C.c_uint64, C.c_int64, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.snk_fastq_format_device.argtypes = [vp, vp, vp, vp, C.c_int64, C.POINTER(FastqFormat), vp, vp, vp, C.c_size_t, vp]
    lib.snk_fastq_deflate_tmp_bytes.argtypes = [C.c_int64, i32]
    lib.snk_fastq_deflate_tmp_bytes.restype = C.c_size_t
    lib.snk_fastq_deflate_device.argtypes = [vp, vp, C.c_int64, i32, vp, C.c_uint64, vp, vp, C.c_size_t, vp]
    return lib
