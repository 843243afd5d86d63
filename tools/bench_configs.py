"""Kernel time of other parameter sets / read lengths than the bench.py headline (informational):
python tools/bench_configs.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import PE_CASES  # noqa: E402
from soapnuke_amd import abi, synth  # noqa: E402
from soapnuke_amd.filter import FilterContext  # noqa: E402


def run(name, L, paired, kw, n=10_000_000, var_len=False):
    uniq = 1_000_000 if L <= 150 else (500_000 if L <= 256 else 200_000)
    d = synth.make_batch(uniq, L, paired=paired, var_len=var_len)
    ctx = FilterContext(abi.default_params(paired=paired, max_read_len=L, **kw), device=0)
    dev = ctx.upload(d)
    reps = max(1, n // uniq)
    dev["seq"] = [x.repeat(reps, 1) for x in dev["seq"]]
    dev["qual"] = [x.repeat(reps, 1) for x in dev["qual"]]
    dev["len"] = [None if x is None else x.repeat(reps) for x in dev["len"]]
    dev["n"] = uniq * reps
    b = ctx.make_batch(dev)
    rec = ctx.alloc_records(dev["n"])
    kern = int(os.environ.get("SNK_BENCH_KERNEL", "0"))          # 1: force the generic kernel
    ctx.filter_batch(b, rec, kernel=kern)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ctx.filter_batch(b, rec, kernel=kern)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    reads = dev["n"] * (2 if paired else 1)
    print(f"{name:28s} L={L:4d} {'PE' if paired else 'SE'} var_len={var_len!s:5s} {dev['n']:9d} units: {ms:7.3f} ms  {reads / ms / 1e3:8.1f} Mreads/s  "
          f"{reads * (2 * L + 16) / ms / 1e6:7.0f} GB/s algorithmic", flush=True)


only = sys.argv[1] if len(sys.argv) > 1 else ""
_run = run


def run(name, *a, **k):  # noqa: F811
    if only in name:
        _run(name, *a, **k)


se = lambda kw: {k: (v[:2] if k == "hard_trim" else v) for k, v in kw.items() if k != "adapters2"}  # noqa: E731
run("C2 (bench)", 150, True, PE_CASES["C2_adatrim_lowq"])
run("C2 variable length", 150, True, PE_CASES["C2_adatrim_lowq"], var_len=True)
run("C3 full trim+filter", 150, True, PE_CASES["C3_full"])
run("defaults (no adapter)", 150, True, PE_CASES["defaults"])
run("C2 SE150", 150, False, se(PE_CASES["C2_adatrim_lowq"]))
run("C2 PE100", 100, True, PE_CASES["C2_adatrim_lowq"])
run("C5 PE250 (C2 params)", 250, True, PE_CASES["C2_adatrim_lowq"], n=6_000_000)
run("long C2 PE500", 500, True, PE_CASES["C2_adatrim_lowq"], n=2_000_000)
run("long C2 PE1000", 1000, True, PE_CASES["C2_adatrim_lowq"], n=1_000_000)
run("long C3 PE1000", 1000, True, PE_CASES["C3_full"], n=1_000_000)
run("long defaults PE1000", 1000, True, PE_CASES["defaults"], n=1_000_000)
run("C2 + contam1/2 + global", 150, True, dict(PE_CASES["C2_adatrim_lowq"], contam1="ACGTTGCAAGGCTTAACCGGTTAGCATGCAAT", contam2="TTGGCCAAGGTTCCAAGGTTAACCGGTT",
                                               ct_match_r="0.5", global_contams="AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", g_mrs="0.4", g_mms="1"), n=5_000_000)
run("C5 PE250 + contam1/2 + global", 250, True, dict(PE_CASES["C2_adatrim_lowq"], contam1="ACGTTGCAAGGCTTAACCGGTTAGCATGCAAT", contam2="TTGGCCAAGGTTCCAAGGTTAACCGGTT",
                                                     ct_match_r="0.5", global_contams="AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", g_mrs="0.4", g_mms="1"), n=3_000_000)
run("C2 + contam1/2 only", 150, True, dict(PE_CASES["C2_adatrim_lowq"], contam1="ACGTTGCAAGGCTTAACCGGTTAGCATGCAAT", contam2="TTGGCCAAGGTTCCAAGGTTAACCGGTT",
                                           ct_match_r="0.5"), n=5_000_000)
run("C2 + global contam only", 150, True, dict(PE_CASES["C2_adatrim_lowq"], global_contams="AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", g_mrs="0.4", g_mms="1"), n=5_000_000)
if only.startswith("sweep"):                    # which optional feature of C3 costs what
    only = ""
    base = dict(PE_CASES["C3_full"])
    for drop in ("n_ratio", "mean_quality", "polyG_tail", "polyX_num", "highA_ratio", "trim_bad_tail"):
        kw = {k: v for k, v in base.items() if k != drop}
        _run("C3 without " + drop, 150, True, kw)
    for add in ("n_ratio", "mean_quality", "polyG_tail", "polyX_num", "highA_ratio", "trim_bad_tail"):
        kw = dict(PE_CASES["C2_adatrim_lowq"])
        kw[add] = base[add]
        _run("C2 plus " + add, 150, True, kw)
