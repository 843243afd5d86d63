#!/usr/bin/env python
"""bench.py -- throughput of the SOAPnuke-filter hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (filter + raw stats + clean stats, i.e. one
reference patch: filter_pe_fqs + stat_pe_fqs x2) over one device-resident batch of
synthetic PE150 reads, followed by the stats all-reduce (RCCL) when N > 1.
Workload = BASELINE.json configs[1]: PE 10M x 150 bp, adapter-trim + lowQual
(`-f A1 -r A2 -J -l 10 -q 0.1`), inputs resident in HBM when the timed region
starts.  Every rank processes its own 10M-pair shard (weak scaling).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against the
8 TB/s HBM peak with the algorithmic bytes of SURVEY 8(d): 2*L + 16 per read =
632 B per PE150 pair, timed with hipEvents on the launch stream inside the C ABI.
`cpu_baseline` times the compiled reference (oracle/_ref/SOAPnuke) on the host
cores on a bounded sample of the same workload (rank 0, N == 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PAIRS_TOTAL = 10_000_000
PAIRS_UNIQUE = 1_000_000
L = 150
BYTES_PER_PAIR = 2 * (2 * L + 16)      # bases + qualities read once + 16 B record, per mate
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# tests/test_simt_bench.py runs this file on the CPU emulator to check that every leg produces its keys: the row sizes of
# other_workloads() shrink by this factor there (1 on the GPU box -- never set it for a measurement)
TEST_DIVISOR = max(1, int(os.environ.get("SNK_BENCH_TEST_DIVISOR", "1")))


WORKLOADS = {
    # name: (description, extra parameters on top of configs[1]'s, variable read lengths)
    "c2": ("BASELINE configs[1]: PE 10Mx150bp, adapter-trim + lowQual (-f/-r README adapters, -J -l 10 -q 0.1)", {}, False),
    "c3": ("BASELINE configs[2] parameters on 10M pairs: configs[1] + -n 0.01 -m 20 -g 10 -X 50 -p 0.8 + trimBadTail=20,30 (full trim+filter)",
           dict(n_ratio=0.01, mean_quality=20, polyG_tail=10, polyX_num=50, highA_ratio=0.8, trim_bad_tail=(20, 30)), False),
    "c2var": ("configs[1] parameters, variable read lengths 75..150", {}, True),
    "c3var": ("configs[2] parameters, variable read lengths 75..150",
              dict(n_ratio=0.01, mean_quality=20, polyG_tail=10, polyX_num=50, highA_ratio=0.8, trim_bad_tail=(20, 30)), True),
}


def bench_params_kwargs(workload="c2"):
    from soapnuke_amd import synth
    return dict(dict(adapters1=[synth.ADAPTER1], adapters2=[synth.ADAPTER2], ada_trim=1, low_qual=10, low_qual_ratio=0.1),
                **WORKLOADS[workload][1])


def bench_params(workload="c2"):
    from soapnuke_amd import abi
    return abi.default_params(paired=True, max_read_len=L, **bench_params_kwargs(workload))


def cpu_baseline(data, n_sample):
    """The reference binary itself on the host cores (kind 'reference'), or the C port."""
    from soapnuke_amd import synth
    cores = min(16, os.cpu_count() or 1)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "SOAPnuke")
    tmp = tempfile.mkdtemp(prefix="snkbench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        if os.path.exists(ref_bin):
            f1, f2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
            copies = 4          # 4 x the unique pairs (< one 6.4M-pair merge cycle): the reference's 5 s merge-poll quantum stays a small part of the wall
            st = os.statvfs(tmp)
            if st.f_bavail * st.f_frsize < 8 * (1 << 30):      # inputs + clean outputs of 4 copies need ~5 GB
                copies = 2
            for k in range(copies):
                for m, f in ((0, f1), (1, f2)):
                    part = f + f".{k}"
                    synth.write_fastq(part, data["seq"][m][:n_sample], data["qual"][m][:n_sample], L, m + 1,
                                      first_index=k * n_sample)
                    with open(f, "ab") as out, open(part, "rb") as src:
                        while True:
                            blk = src.read(1 << 24)
                            if not blk:
                                break
                            out.write(blk)
                    os.unlink(part)
            n_sample *= copies
            cmd = [ref_bin, "filter", "-1", f1, "-2", f2, "-C", "c1.fq", "-D", "c2.fq", "-o", os.path.join(tmp, "out"),
                   "-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-l", "10", "-q", "0.1", "-T", str(cores)]
            t0 = time.time()
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            wall = time.time() - t0
            if r.returncode == 0:
                return {"value": round(2 * n_sample / wall / 1e6, 4), "unit": "Mreads/s", "cores": cores,
                        "kind": "reference",
                        "sample": f"{n_sample} PE150 pairs, plain FASTQ in /dev/shm, `SOAPnuke filter -J -l 10 -q 0.1 -T {cores}`, "
                                  f"whole-process wall {wall:.1f}s (includes its 5 s merge-poll quantum)"}
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import snk_testlib as T
        n = min(n_sample, 200_000)
        sub = {"n": n, "L": L, "pitch": data["pitch"], "paired": True,
               "seq": [x[:n] for x in data["seq"]], "qual": [x[:n] for x in data["qual"]], "len": [None, None]}
        t0 = time.time()
        T.run_oracle(bench_params(), sub)
        wall = time.time() - t0
        return {"value": round(2 * n / wall / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
                "sample": f"{n} PE150 pairs through oracle/snk_oracle.c (hot path only, no I/O), {wall:.1f}s"}
    finally:
        subprocess.call(["rm", "-rf", tmp])


def kernel_source_sha():
    """sha1 over the sources the tiled kernel is compiled from: a committed PMC figure is quoted only for the kernel it was taken on"""
    import glob
    import hashlib
    h = hashlib.sha1()
    cs = os.path.join(ROOT, "soapnuke_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(cs, "snk_tiled.hip")) + glob.glob(os.path.join(cs, "*.hip.h")) +
                    [os.path.join(cs, "snk_device.h")]):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(workload, timeout_s=240):
    """HBM bytes per launch of the tiled kernel, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need a pass each,
    MI355X_MICROARCH.md 'rocprofv3 PMC slots') around a short child run of this file, kernel trace only (no other trace domain next
    to --pmc); bytes = 2 x FETCH_SIZE + WRITE_SIZE, KB units (the guide's gfx950 correction for wide streaming reads, re-derived for
    this kernel's access patterns in profiles/r02_fetch_calibration.txt).  None when rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    if not shutil.which("rocprofv3"):
        return None
    tmp = tempfile.mkdtemp(prefix="snkpmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", ctr, "-d", d, "-o", "tiled", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-traffic",
                   "--workload", workload]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if "snk_tiled_kernel" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                            per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None
            vals[ctr] = sum(per.values()) / len(per)
        return {"bytes": int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024), "fetch_kb_raw": vals["FETCH_SIZE"],
                "write_kb_raw": vals["WRITE_SIZE"]}
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        subprocess.call(["rm", "-rf", tmp])


def end_to_end(n_pairs, threads=16, rmdup_pairs=4_000_000, deadline=None):
    """This repo's CLI and the reference binary on the same /dev/shm FASTQ, whole-process wall clock (tools/bench_e2e.py):
    configs[1] parameters .gz -> .gz and .gz -> plain, plain -> plain for this CLI only (the reference's plain-INPUT run is
    its 60-s remove_tmpDir stall, SURVEY Q10: the plain speed-up is quoted against its .gz -> plain time), BASELINE configs[2]'s
    parameters .gz -> .gz on the same files, and BASELINE configs[4]'s shape (PE250 + rmdup, 5 % duplicate pairs) on
    rmdup_pairs pairs.  The fastest reference leg doubles as the cpu_baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_e2e
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="snkbench_", dir=shm)
    try:
        res = bench_e2e.measure(tmp, n_pairs, threads, ["gz", "gz2plain", "plain_ours", "gz_c3"], deadline=deadline)
    finally:
        subprocess.call(["rm", "-rf", tmp])
    # (plain_ours stays an absolute number: the reference's plain-INPUT run is its 60-s remove_tmpDir stall, SURVEY Q10, and a ratio
    # against its .gz-input time would compare two different workloads -- ADVICE r4)
    rmdup_pairs //= TEST_DIVISOR
    if deadline is not None and time.time() > deadline:
        res["pe250_rmdup"] = {"skipped": "bench.py's time budget (--budget-s) was spent before this leg"}
    elif rmdup_pairs > 0:
        tmp = tempfile.mkdtemp(prefix="snkbench_", dir=shm)
        try:
            res["pe250_rmdup"] = bench_e2e.measure(tmp, rmdup_pairs, threads, ["gz"], c3=False, extra_cfg=["rmdup"], L=250, dup_frac=0.05)
        except Exception as ex:
            res["pe250_rmdup"] = {"error": repr(ex)[:200]}
        finally:
            subprocess.call(["rm", "-rf", tmp])
    return res


def rmdup_kernels(n=10_000_000, L=250):
    """the two kernels of the rmdup pre-pass on resident data: std::hash of mate1 ++ mate2 per pair, and the
    first-occurrence marking (5 % duplicate hashes)"""
    import torch
    from soapnuke_amd import abi, synth
    from soapnuke_amd.filter import FilterContext
    uniq = max(64, 500_000 // TEST_DIVISOR)
    n = max(uniq, n // TEST_DIVISOR)
    d = synth.make_batch(uniq, L, paired=True)
    ctx = FilterContext(abi.default_params(paired=True, max_read_len=L, rmdup=1), device=0)
    dev = ctx.upload(d)
    reps = n // uniq
    dev["seq"] = [x.repeat(reps, 1) for x in dev["seq"]]
    dev["qual"] = [x[:1] for x in dev["qual"]]          # (the hash does not look at the qualities)
    dev["n"] = uniq * reps
    b = ctx.make_batch(dev)
    h = ctx.hash_batch(b)
    torch.cuda.synchronize()

    def timed(f, k=5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f()
        e0.record()
        for _ in range(k):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k

    ms_h = timed(lambda: ctx.hash_batch(b, h))
    nn = dev["n"]
    hh = torch.randint(-2**62, 2**62, (nn,), dtype=torch.int64, device="cuda")
    k = nn // 20
    hh[torch.randint(0, nn, (k,), device="cuda")] = hh[torch.randint(0, nn, (k,), device="cuda")]
    ms_m = timed(lambda: ctx.mark_dups(hh))
    ctx.close()
    del dev, h, hh
    torch.cuda.empty_cache()
    return [{"workload": f"rmdup hash kernel (std::hash of mate1 ++ mate2), PE{L}, {nn / 1e6:g} M pairs", "ms": round(ms_h, 3),
             "Mreads_per_s": round(2 * nn / ms_h / 1e3, 1), "algorithmic_GBps": round((2 * L + 8) * nn / ms_h / 1e6, 1),
             "frac_of_hbm_peak": round((2 * L + 8) * nn / ms_h / 1e6 / HBM_PEAK_GBS, 4), "error": 0},
            {"workload": f"rmdup marking kernels (hash table in HBM: insert + look-up), {nn / 1e6:g} M hashes, 5 % duplicates", "ms": round(ms_m, 3),
             "Mreads_per_s": round(2 * nn / ms_m / 1e3, 1), "algorithmic_GBps": round(9 * nn / ms_m / 1e6, 1),
             "frac_of_hbm_peak": round(9 * nn / ms_m / 1e6 / HBM_PEAK_GBS, 4), "error": 0,
             "note": "random access: 8 B hash in + 1 B flag out per pair are the algorithmic bytes, the table traffic is not counted"}]


def other_workloads_child(timeout_s):
    """other_workloads() in a process of its own: these rows launch every kernel family of the library, and a fault or a hang in one
    of them (the process dies with the HIP runtime's abort) must not take the headline line down"""
    cmd = [sys.executable, os.path.abspath(__file__), "--child-other-workloads", "--no-cpu-baseline", "--no-traffic"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"the other_workloads child did not finish within {timeout_s:.0f} s (killed)"}
    lines = [x for x in r.stdout.decode(errors="replace").split("\n") if x.startswith("[") or x.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"the other_workloads child ended with rc {r.returncode}", "stderr_tail": r.stderr[-300:].decode(errors="replace")}
    return json.loads(lines[-1])


def other_workloads():
    """Kernel time of the widened rows (SURVEY 8f / VERDICT r1 #7, #8), device-resident like the headline: contaminant
    screening in front of the tiled kernel, reads of 1000 positions (snk_long.hip), and the fallback for configurations
    the fast paths refuse (generic decisions + LDS histogram kernel).  Informational; parity for each in tests/."""
    import torch
    from soapnuke_amd import abi, synth
    from soapnuke_amd.filter import FilterContext
    c2 = bench_params_kwargs()
    c3 = bench_params_kwargs("c3")
    rows = [
        ("BASELINE configs[2] parameters (full trim + filter: the FULL kernel variant), PE150, 10 M pairs", 150, 10_000_000, 0, c3, False),
        ("configs[1] parameters, variable read lengths 75..150, 10 M pairs", 150, 10_000_000, 0, c2, True),
        ("configs[2] parameters, variable read lengths 75..150, 10 M pairs", 150, 10_000_000, 0, c3, True),
        ("contaminants (contam1/2 32/28 nt + one 33-nt global), PE150, 5 M pairs", 150, 5_000_000, 0,
         dict(c2, contam1="ACGTTGCAAGGCTTAACCGGTTAGCATGCAAT", contam2="TTGGCCAAGGTTCCAAGGTTAACCGGTT", ct_match_r="0.5",
              global_contams="AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", g_mrs="0.4", g_mms="1"), False),
        ("BASELINE configs[4] read length: PE250, configs[1] parameters, 8 M pairs", 250, 8_000_000, 0, c2, False),
        ("PE250, configs[2] parameters, 8 M pairs", 250, 8_000_000, 0, c3, False),
        ("PE300 (first length past the tiled kernel's 256 positions), configs[1] parameters, 4 M pairs", 300, 4_000_000, 0, c2, False),
        ("long reads, PE1000, 1 M pairs", 1000, 1_000_000, 0, c2, False),
        ("fallback (kernel=1: generic decisions + LDS histograms), PE150, 2 M pairs", 150, 2_000_000, 1, c2, False),
    ]
    out = []
    for row in rows:
        try:
            out.append(_workload_row(*row))
        except Exception as ex:          # one row must not take the others down
            out.append({"workload": row[0], "error": repr(ex)[:200]})
        import torch
        torch.cuda.empty_cache()
    try:
        out += rmdup_kernels()
    except Exception as ex:
        out.append({"workload": "rmdup kernels", "error": repr(ex)[:200]})
    return out


def _workload_row(name, L, n, kern, kw, var_len):
    import torch
    from soapnuke_amd import abi, synth
    from soapnuke_amd.filter import FilterContext
    if True:
        uniq = max(64, (500_000 if L <= 150 else 100_000) // TEST_DIVISOR)
        n = max(uniq, n // TEST_DIVISOR)
        d = synth.make_batch(uniq, L, paired=True, var_len=var_len)
        ctx = FilterContext(abi.default_params(paired=True, max_read_len=L, **kw), device=0)
        dev = ctx.upload(d)
        reps = n // uniq
        dev["seq"] = [x.repeat(reps, 1) for x in dev["seq"]]
        dev["qual"] = [x.repeat(reps, 1) for x in dev["qual"]]
        dev["len"] = [None if x is None else x.repeat(reps) for x in dev["len"]]
        dev["n"] = uniq * reps
        b = ctx.make_batch(dev)
        rec = ctx.alloc_records(dev["n"])
        ctx.filter_batch(b, rec, kernel=kern)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ctx.filter_batch(b, rec, kernel=kern)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        _, _, err = ctx.fetch()
        nbytes = 2 * dev["n"] * (2 * L + 16)
        if var_len:      # SURVEY 8(d)'s per-read figure on the real lengths
            nbytes = reps * int(sum(2 * int(x.astype(np.int64).sum()) + 16 * len(x) for x in d["len"]))
        res = {"workload": name, "ms": round(ms, 3), "Mreads_per_s": round(2 * dev["n"] / ms / 1e3, 1),
               "algorithmic_GBps": round(nbytes / ms / 1e6, 1), "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
               "error": int(err[0])}
        ctx.close()
        del dev, rec
        return res


KERNEL_INSTANCE = {   # the tiled kernel's instance snk_filter.cpp picks for each bench workload (soapnuke_amd/csrc/snk_tiled.hip, PE150)
    "c2": "snk_tiled_kernel<5,false,true,16,TileShape<160,768,4>>", "c3": "snk_tiled_kernel<5,true,true,16,TileShape<160,768,4>>",
    "c2var": "snk_tiled_kernel<5,false,true,16,TileShape<0,0,0>>", "c3var": "snk_tiled_kernel<5,true,true,16,TileShape<0,0,0>>",
}


def headline_child(args):
    """the timed region in a process of its own (`--child-headline`): its JSON line, or an error record that names the kernel"""
    cmd = [sys.executable, os.path.abspath(__file__), "--child-headline", "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--pairs", str(args.pairs), "--kernel", str(args.kernel), "--workload", args.workload, "--no-cpu-baseline", "--no-traffic"]
    limit = float(os.environ.get("SNK_BENCH_HEADLINE_TIMEOUT_S", "420"))
    err = None
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit)
        lines = [x for x in r.stdout.decode(errors="replace").split("\n") if x.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        err = {"error": f"the headline child ended with rc {r.returncode}" + ("" if r.returncode >= 0 else f" (signal {-r.returncode})"),
               "stderr_tail": r.stderr[-600:].decode(errors="replace")}
    except subprocess.TimeoutExpired as ex:
        err = {"error": f"the headline child did not finish within {limit:.0f} s (killed): a kernel hangs",
               "stderr_tail": (ex.stderr or b"")[-600:].decode(errors="replace")}
    return dict({"metric": "Mreads/s PE150 `filter` (adapter+qual), bit-exact vs ref", "value": None, "unit": "Mreads/s", "n_gpus": 1,
                 "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                 "vs_baseline": None, "dtype": "u8", "data": "synthetic PE150",
                 "config": {"workload": WORKLOADS[args.workload][0], "pairs_per_gpu_per_step": args.pairs, "read_len": L},
                 "kernel_instance": KERNEL_INSTANCE.get(args.workload) if args.kernel in (0, 2) else f"kernel={args.kernel}",
                 "roofline": None}, **err)


def finish_single(args, out, data, t_start):
    """N == 1, after the timed region: roofline.traffic (two rocprofv3 --pmc child runs), cpu_baseline, end_to_end, other_workloads --
    every one of them in processes of their own -- then the ONE JSON line"""
    n = out["config"]["pairs_per_gpu_per_step"]
    if out.get("roofline") is not None:
        # HBM bytes per launch: measured in this run by two rocprofv3 --pmc passes around a short child run of this command
        # (measured_traffic), N == 1 only; failing that, the committed profile's figure -- but only while profiles/traffic.json
        # was taken on the kernel sources of this tree (kernel_source_sha), never a stale constant
        traffic, extra = None, {}
        if not args.no_traffic and not args.no_cpu_baseline and TEST_DIVISOR == 1 and n == 10_000_000 and args.kernel in (0, 2):
            mt = measured_traffic(args.workload)
            if mt is not None:
                traffic = mt["bytes"]
                extra = {"traffic_measured_in_run": True, "traffic_how": "2 x FETCH_SIZE + WRITE_SIZE (KB), one rocprofv3 --pmc pass each, "
                         "mean over the tiled kernel's launches of a 4-step child run", "fetch_kb_raw": round(mt["fetch_kb_raw"], 1),
                         "write_kb_raw": round(mt["write_kb_raw"], 1), "traffic_over_algorithmic": round(mt["bytes"] / out["roofline"]["bytes_per_launch"], 3)}
        if traffic is None and not args.no_traffic:
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                    tj = json.load(fh)
                if (n == 10_000_000 and args.kernel in (0, 2) and args.workload == "c2"
                        and tj.get("kernel_source_sha") == kernel_source_sha()):
                    traffic = int(tj["hbm_bytes_per_launch"])
                    extra = {"traffic_measured_in_run": False, "traffic_source": tj["source"]}
                else:
                    extra = {"traffic_measured_in_run": False, "traffic_note": "no counter figure for this build: profiles/traffic.json "
                             "was taken on other kernel sources and rocprofv3 did not produce one in this run"}
            except (OSError, KeyError, ValueError):
                pass

        out["roofline"]["traffic"] = traffic
        out["roofline"].update(extra)
    if not args.no_cpu_baseline and args.workload == "c2":
        e2e = None
        if args.e2e_pairs > 0 and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "SOAPnuke")):
            try:
                e2e = end_to_end(args.e2e_pairs, deadline=t_start + args.budget_s)
            except Exception as ex:          # the host legs must never take the kernel line down
                e2e = {"error": repr(ex)[:200]}
        # the reference's best leg of the three (plain -> plain carries its 60-s remove_tmpDir stall past one merge cycle,
        # SURVEY Q10; .gz input does not): the most favourable number for the baseline
        legs = [(m, v["reference"]) for m, v in (e2e or {}).get("modes", {}).items()
                if isinstance(v, dict) and v.get("reference", {}).get("rc") == 0]
        legs = [x for x in legs if x[0] in ("gz", "gz2plain")]       # configs[1] parameters only: the headline's workload
        if legs:
            m, best = max(legs, key=lambda x: x[1]["Mreads_per_s"])
            out["cpu_baseline"] = {"value": best["Mreads_per_s"], "unit": "Mreads/s", "cores": 16, "kind": "reference",
                                   "sample": f"{args.e2e_pairs} PE150 pairs in /dev/shm, `SOAPnuke filter -J -l 10 -q 0.1 -T 16`, leg `{m}` "
                                             f"(the fastest of the reference's legs in end_to_end), whole-process wall {best['wall_s']}s"}
        else:
            if data is None:
                from soapnuke_amd import synth
                data = synth.make_batch(min(PAIRS_UNIQUE, args.pairs), L, paired=True, seed=synth.SEED)
            out["cpu_baseline"] = cpu_baseline(data, min(data["n"], 1_000_000))
        if e2e is not None:
            out["end_to_end"] = e2e
            # the PIPELINE's rate next to the resident-batch `value`: this CLI's whole-process wall clock, FASTQ files
            # in /dev/shm -> clean FASTQ + the reports, per leg
            ev = {m: v["ours"]["Mreads_per_s"] for m, v in e2e.get("modes", {}).items()
                  if isinstance(v, dict) and isinstance(v.get("ours"), dict) and v["ours"].get("rc") == 0 and "Mreads_per_s" in v["ours"]}
            if ev:
                out["e2e_value"] = dict(ev, unit="Mreads/s", pairs=args.e2e_pairs,
                                        what="`soapnuke_amd/SOAPnuke filter` end to end (FASTQ in /dev/shm -> clean FASTQ + reports), whole-process wall")
        if time.time() > t_start + args.budget_s:
            out["other_workloads"] = {"skipped": "bench.py's time budget (--budget-s) was spent before this leg"}
        else:
            if os.environ.get("SNK_BENCH_INPROCESS") == "1":       # (tests/test_simt_bench.py: the emulated device lives in this process)
                try:
                    out["other_workloads"] = other_workloads()
                except Exception as ex:
                    out["other_workloads"] = {"error": repr(ex)[:200]}
            else:
                out["other_workloads"] = other_workloads_child(max(120.0, t_start + args.budget_s + 240.0 - time.time()))
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=PAIRS_TOTAL, help="pairs per GPU per step")
    ap.add_argument("--kernel", type=int, default=0, choices=[0, 1, 2, 3], help="0 auto, 1 generic decisions + LDS histograms, 2 fast paths only, 3 generic alone (anchor)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the host-side legs (cpu_baseline, end_to_end)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS), help="c2 = the headline (BASELINE configs[1]); the others are profiling workloads")
    ap.add_argument("--budget-s", type=float, default=600.0, help="wall-clock budget of the whole run: the optional legs behind the timed region "
                    "(end_to_end legs, other_workloads) that would start after it are recorded as skipped -- the headline line always comes out")
    ap.add_argument("--child-other-workloads", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--child-headline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--e2e-pairs", type=int, default=16_000_000, help="pairs of the end_to_end legs (0: only the cpu_baseline sample); 16 M: the reference needs "
                    "~50 s per .gz leg (three of them), ~40 s for the 4 M-pair PE250 + rmdup leg")
    args = ap.parse_args()
    t_start = time.time()
    if args.child_other_workloads:
        print(json.dumps(other_workloads()), flush=True)
        return

    forced = os.environ.get("SNK_BENCH_FORCE_LAUNCHER") == "1"
    if (args.gpus == 1 and not forced and "WORLD_SIZE" not in os.environ and not args.child_headline
            and os.environ.get("SNK_BENCH_INPROCESS") != "1"):
        # N == 1: the timed region runs in a child.  A fault inside the kernel ends the process with the HIP runtime's
        # abort and a hang ends at the child's time limit: either way this process still prints its ONE JSON line, with "error" and the
        # kernel instance instead of a number, and the host-side legs (which run in processes of their own) are still measured.
        out, data = headline_child(args), None
        finish_single(args, out, data, t_start)
        return

    # SNK_BENCH_FORCE_LAUNCHER=1: take the launcher branch, the nccl process group and the collective at world size 1 too
    # (tests/test_multirank_gpu.py drives the N > 1 code path on a one-GPU box this way)
    if (args.gpus > 1 or forced) and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher -- one rank per GPU under
        # torch.distributed.run (exactly the command line the driver uses), same arguments
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from soapnuke_amd import synth
    from soapnuke_amd.filter import FilterContext

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if torch.cuda.device_count() < world:
        sys.exit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} HIP device(s) visible")
    use_dist = world > 1 or forced
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    torch.cuda.set_device(local_rank)

    n_unique = min(PAIRS_UNIQUE, args.pairs)
    reps = max(1, args.pairs // n_unique)
    n = n_unique * reps
    var_len = WORKLOADS[args.workload][2]
    data = synth.make_batch(n_unique, L, paired=True, seed=synth.SEED + rank, var_len=var_len)
    ctx = FilterContext(bench_params(args.workload), device=local_rank)
    dev = ctx.upload(data)
    if reps > 1:   # distinct HBM copies: no cache reuse between replicas (6.4 GB >> 256 MB L3)
        dev["seq"] = [x.repeat(reps, 1) for x in dev["seq"]]
        dev["qual"] = [x.repeat(reps, 1) for x in dev["qual"]]
        dev["len"] = [None if x is None else x.repeat(reps) for x in dev["len"]]
        dev["n"] = n
    rec = ctx.alloc_records(n)
    batch = ctx.make_batch(dev, first_index=rank * n)
    ctx.set_timing(True)

    def step():
        ctx.filter_batch(batch, rec, kernel=args.kernel)

    def merge():
        # the path's one collective: the reference merges its per-thread statistics once per
        # run (merge_stat, src/peprocess.cpp:1994); here one sum + one max all-reduce over RCCL, inside the
        # timed region
        if use_dist:
            ctx.allreduce()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    merge()
    barrier()
    ctx.last_kernel_ms()
    ctx.clear()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    merge()
    barrier()
    elapsed = time.perf_counter() - t0
    # average launch duration over the K timed steps: one hipEvent pair per launch, recorded on
    # the launch stream inside the C ABI (warm-up pairs were drained before the timed region)
    kernel_ms.append(ctx.last_kernel_ms())
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    s, mx, err = ctx.fetch()
    assert err[0] == 0, err
    kept = int(s[(64 + 2 * (16 + L * 5 + L * 43 + 5000))])   # clean fq1 reads_number

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = 2.0 * n * world * args.steps / elapsed / 1e6
        k_ms = float(np.mean(kernel_ms))
        bytes_launch = BYTES_PER_PAIR * n
        if var_len:      # SURVEY 8(d)'s per-read figure on the real lengths: 2 * len + 16
            bytes_launch = reps * int(sum(2 * int(x.astype(np.int64).sum()) + 16 * len(x) for x in data["len"]))
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "Mreads/s PE150 `filter` (adapter+qual), bit-exact vs ref",
            "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if use_dist else 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": f"synthetic PE150 (seed {synth.SEED}+rank): {n_unique} unique pairs x{reps} distinct HBM copies per GPU",
            "config": {"workload": WORKLOADS[args.workload][0],
                       "pairs_per_gpu_per_step": n, "read_len": L, "Mpairs_per_s": round(value / 2, 3),
                       "kernel": {0: "auto", 1: "generic+hist", 2: "tiled", 3: "generic"}[args.kernel],
                       "parallelism": f"shard{world}" if world > 1 else "single",
                       "clean_pairs_per_step_per_gpu": kept // (args.steps * world)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel_ms": round(k_ms, 4), "bytes_per_launch": bytes_launch},
        }
        if args.child_headline:
            print(json.dumps(out), flush=True)
            return
        if world == 1:
            del batch, rec, dev          # (our own device memory goes back first: the children bring their own batches)
            ctx.close()
            torch.cuda.empty_cache()
            finish_single(args, out, data, t_start)
            if use_dist:
                dist.destroy_process_group()
            return
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
