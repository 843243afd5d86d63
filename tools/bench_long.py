"""Kernel time of the long-read path (snk_long.hip) and of the fallback's histogram kernel, device-resident inputs:

    python tools/bench_long.py [L=1000] [pairs=1000000] [c2|c3|contam] [kernel=0]

Prints the average of 5 launches (whole path: decide kernel + histogram kernel); run under
`rocprofv3 --kernel-trace --stats` for the split between the two kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from soapnuke_amd import abi, synth  # noqa: E402
from soapnuke_amd.filter import FilterContext  # noqa: E402
import bench  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    wl = sys.argv[3] if len(sys.argv) > 3 else "c2"
    kw = bench.bench_params_kwargs("c3") if wl == "c3" else bench.bench_params_kwargs()
    if wl == "contam":                                      # bench.py's contaminant row
        kw = dict(kw, contam1="ACGTTGCAAGGCTTAACCGGTTAGCATGCAAT", contam2="TTGGCCAAGGTTCCAAGGTTAACCGGTT", ct_match_r="0.5",
                  global_contams="AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", g_mrs="0.4", g_mms="1")
    uniq = 100_000 if L > 150 else 500_000
    d = synth.make_batch(uniq, L, paired=True)
    ctx = FilterContext(abi.default_params(paired=True, max_read_len=L, **kw), device=0)
    dev = ctx.upload(d)
    reps = n // uniq
    dev["seq"] = [x.repeat(reps, 1) for x in dev["seq"]]
    dev["qual"] = [x.repeat(reps, 1) for x in dev["qual"]]
    dev["len"] = [None if x is None else x.repeat(reps) for x in dev["len"]]
    dev["n"] = uniq * reps
    b = ctx.make_batch(dev)
    rec = ctx.alloc_records(dev["n"])
    for kern in (int(sys.argv[4]) if len(sys.argv) > 4 else 0,):
        ctx.filter_batch(b, rec, kernel=kern)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.filter_batch(b, rec, kernel=kern)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        nbytes = 2 * dev["n"] * (2 * L + 16)
        print("L=%d pairs=%d %s kernel=%d: %.3f ms  %.1f Mreads/s  %.1f GB/s (%.4f of 8 TB/s)" % (
            L, dev["n"], wl, kern, ms, 2 * dev["n"] / ms / 1e3, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000), flush=True)


if __name__ == "__main__":
    main()
