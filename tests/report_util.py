"""Drives the report writer (soapnuke_amd/libsnk_report.so) from per-virtual-thread stats and,
where the compiled reference binary is available, the reference CLI on the same FASTQ."""
import ctypes as C
import os
import subprocess

import numpy as np

import snk_testlib as T
from soapnuke_amd import abi, synth

REPORT_SO = os.path.join(T.ROOT, "soapnuke_amd", "libsnk_report.so")
REPORT_SRC = os.path.join(T.ROOT, "soapnuke_amd", "host", "snk_report.cpp")

# (name, paired, L, n, threads, patch, synth kwargs, library params, reference CLI args, config lines)
REPORT_CASES = [
    ("pe_adatrim_T4", True, 150, 60000, 4, 250, dict(seed=55),
     dict(adapters1=[synth.ADAPTER1], adapters2=[synth.ADAPTER2], ada_trim=1, low_qual=10, low_qual_ratio=0.1),
     ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-l", "10", "-q", "0.1"], []),
    ("pe_full_T3", True, 150, 50000, 3, 200, dict(seed=56),
     dict(adapters1=[synth.ADAPTER1], adapters2=[synth.ADAPTER2], ada_trim=1, low_qual=10, low_qual_ratio=0.1, n_ratio=0.01,
          mean_quality=20, polyG_tail=10, polyX_num=50, highA_ratio=0.8, trim_bad_tail=(20, 30), hard_trim=[2, 0, 0, 3]),
     ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-l", "10", "-q", "0.1", "-n", "0.01", "-m", "20", "-g", "10",
      "-X", "50", "-p", "0.8", "-t", "2,0,0,3"], ["trimBadTail=20,30"]),
    ("pe_defaults_T1", True, 100, 20000, 1, 0, dict(seed=57), dict(), [], []),
    ("se_adadiscard_T2", False, 100, 40000, 2, 300, dict(seed=58),
     dict(adapters1=[synth.ADAPTER1], low_qual=10, low_qual_ratio=0.2),
     ["-f", synth.ADAPTER1, "-l", "10", "-q", "0.2"], []),
    ("se_trim_T2", False, 150, 30000, 2, 300, dict(seed=59),
     dict(adapters1=[synth.ADAPTER1], ada_trim=1, hard_trim=[3, 4], polyG_tail=15),
     ["-f", synth.ADAPTER1, "-J", "-t", "3,4", "-g", "15"], []),
]
REPORT_FILES_PE = ["Statistics_of_Filtered_Reads.txt", "Basic_Statistics_of_Sequencing_Quality.txt"] + [
    f"{n}_{m}.txt" for n in ("Base_distributions_by_read_position", "Base_quality_value_distribution_by_read_position",
                             "Distribution_of_Q20_Q30_bases_by_read_position", "Statistics_of_Trimming_Position_of_Reads")
    for m in (1, 2)]
REPORT_FILES_SE = [f for f in REPORT_FILES_PE if not f.endswith("_2.txt")]


def report_lib():
    if (not os.path.exists(REPORT_SO)) or os.path.getmtime(REPORT_SO) < os.path.getmtime(REPORT_SRC):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", REPORT_SO, REPORT_SRC])
    lib = C.CDLL(REPORT_SO)
    lib.snk_vthread_block.restype = C.c_int64
    lib.snk_vthread_block.argtypes = [C.c_int, C.c_int]
    return lib


def case_inputs(case):
    name, paired, L, n, threads, patch, skw, pkw, cli, cfg = case
    d = synth.make_batch(n, L, paired=paired, **skw)
    p = abi.default_params(paired=paired, max_read_len=L, **pkw)
    return d, p


def vthread_stats(case, d, p, run):
    """Per-virtual-thread accumulators: `run(params, sub_batch, first_index, stats)` adds one block."""
    name, paired, L, n, threads, patch, *_ = case
    block = report_lib().snk_vthread_block(threads, patch)
    stats = [T.new_stats(p) for _ in range(threads)]
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        sub = dict(n=hi - lo, L=L, pitch=d["pitch"], paired=paired, seq=[x[lo:hi] for x in d["seq"]],
                   qual=[x[lo:hi] for x in d["qual"]], len=[None if x is None else x[lo:hi] for x in d["len"]])
        run(p, sub, lo, stats[(lo // block) % threads])
    return stats


def write_reports(p, stats, out_dir):
    lib = report_lib()
    os.makedirs(out_dir, exist_ok=True)
    n = len(stats)
    sp = (C.c_void_p * n)(*[s[0].ctypes.data for s in stats])
    mp = (C.c_void_p * n)(*[s[1].ctypes.data for s in stats])
    err = C.create_string_buffer(512)
    rc = lib.snk_write_reports(C.byref(p), n, sp, mp, out_dir.encode(), err, 512)
    assert rc == 0, err.value


def _gzip_copy(path):
    import gzip
    with open(path, "rb") as src, gzip.open(path + ".gz", "wb", compresslevel=1) as dst:
        dst.write(src.read())


def run_reference_cli(case, d, work, gz_input=False):
    """`SOAPnuke filter` of the compiled reference on the same reads.  Its report files are right
    either way; its *clean FASTQ* is only complete/in order for .gz input (or -T 1, or < 1 cycle of
    reads): SURVEY quirk Q10 -- pass gz_input=True when the clean bytes are compared."""
    name, paired, L, n, threads, patch, skw, pkw, cli, cfg = case
    os.makedirs(work, exist_ok=True)
    ext = ".fq.gz" if gz_input else ".fq"
    synth.write_fastq(os.path.join(work, "r1.fq"), d["seq"][0], d["qual"][0], L, 1)
    if gz_input:
        _gzip_copy(os.path.join(work, "r1.fq"))
    cmd = [T.REF_BIN, "filter", "-1", os.path.join(work, "r1" + ext), "-C", "c1.fq", "-o", os.path.join(work, "ref"), "-T", str(threads)]
    if paired:
        synth.write_fastq(os.path.join(work, "r2.fq"), d["seq"][1], d["qual"][1], L, 2)
        if gz_input:
            _gzip_copy(os.path.join(work, "r2.fq"))
        cmd += ["-2", os.path.join(work, "r2" + ext), "-D", "c2.fq"]
    lines = list(cfg) + ([f"patch={patch}"] if patch else [])
    if lines:
        with open(os.path.join(work, "cfg"), "w") as f:
            f.write("\n".join(lines) + "\n")
        cmd += ["-c", os.path.join(work, "cfg")]
    r = subprocess.run(cmd + cli, capture_output=True)
    assert r.returncode == 0, r.stderr[-500:]
    return os.path.join(work, "ref")
