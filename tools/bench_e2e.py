"""End to end, like for like: this repo's `SOAPnuke filter` and the compiled reference binary on the SAME
FASTQ files in /dev/shm, whole-process wall clock, BASELINE configs[1] parameters (`-f/-r README adapters -J -l 10 -q 0.1`).

    python tools/bench_e2e.py [pairs] [threads] [modes] [--keep]

modes: comma list of plain,gz (input and output of the same kind; default both).  Prints one JSON object (also written to
gpurun_out/e2e_<pairs>.json): Mreads/s of both tools per mode, byte-identity of the decompressed clean FASTQ, and the
host configuration.  The reference is given `.gz` input in gz mode only; in plain mode with more than one merge cycle of
reads it loses a patch -- its wall clock is still what it is, but its output is then not compared.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from soapnuke_amd import synth  # noqa: E402

OURS = os.path.join(ROOT, "soapnuke_amd", "SOAPnuke")
REF = os.path.join(ROOT, "oracle", "_ref", "SOAPnuke")


def md5_of(path):
    """md5 of the (decompressed) bytes"""
    h = hashlib.md5()
    if path.endswith(".gz"):
        p = subprocess.Popen(["gzip", "-dc", path], stdout=subprocess.PIPE)
        src = p.stdout
    else:
        p, src = None, open(path, "rb")
    while True:
        b = src.read(1 << 24)
        if not b:
            break
        h.update(b)
    if p:
        p.wait()
    return h.hexdigest()


def make_inputs(tmp, n, modes, L=150, dup_frac=0.0):
    """n pairs of PE<L> FASTQ: the first million synthetic pairs repeated with new read names; dup_frac > 0: that fraction of
    the pairs of the million carries the SEQUENCE of another pair (what rmdup keys on; the qualities differ)"""
    d = synth.make_batch(min(n, 1_000_000), L, paired=True)
    if dup_frac > 0:
        import numpy as np
        rng = np.random.default_rng(11)
        dst = rng.choice(d["n"], int(d["n"] * dup_frac), replace=False)
        src = rng.integers(0, d["n"], len(dst))
        for m in range(2):
            d["seq"][m][dst] = d["seq"][m][src]
    f = [os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")]
    u = d["n"]
    t0 = time.time()
    for k in range((n + u - 1) // u):
        cnt = min(u, n - k * u)
        for m in range(2):
            part = f[m] + ".part"
            synth.write_fastq(part, d["seq"][m][:cnt], d["qual"][m][:cnt], L, m + 1, first_index=k * u)
            with open(f[m], "ab") as out, open(part, "rb") as src:
                while True:
                    b = src.read(1 << 26)
                    if not b:
                        break
                    out.write(b)
            os.unlink(part)
    t1 = time.time()
    if "gz" in modes:      # ONE gzip member per file (what `gzip` makes of a FASTQ), both files at once
        ps = [subprocess.Popen(["gzip", "-1", "-k", x]) for x in f]
        for p in ps:
            assert p.wait() == 0
    return f, t1 - t0, time.time() - t1


C3 = False      # --c3: BASELINE configs[2] parameters (full trim + filter) instead of configs[1]'s
EXTRA_CFG = []  # more config-file lines (tools/bench_e2e_big.py: rmdup)


REPORTS = ["Statistics_of_Filtered_Reads.txt", "Basic_Statistics_of_Sequencing_Quality.txt"] + [
    f"{n}_{m}.txt" for n in ("Base_distributions_by_read_position", "Base_quality_value_distribution_by_read_position",
                             "Distribution_of_Q20_Q30_bases_by_read_position", "Statistics_of_Trimming_Position_of_Reads") for m in (1, 2)]

C3_ARGS = ["-n", "0.01", "-m", "20", "-g", "10", "-X", "50", "-p", "0.8"]


def command(exe, inputs, out_dir, ext, threads, c3=None, extra_cfg=None):
    """the command line of one run (both tools take the same); c3 / extra_cfg default to the module switches"""
    c3 = C3 if c3 is None else c3
    extra_cfg = EXTRA_CFG if extra_cfg is None else extra_cfg
    args = ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-l", "10", "-q", "0.1", "-T", str(threads)]
    if c3 or extra_cfg:
        cfg = os.path.join(os.path.dirname(inputs[0]), "c3.cfg" if c3 else "extra.cfg")
        with open(cfg, "w") as fh:
            fh.write("".join(x + "\n" for x in (["trimBadTail=20,30"] if c3 else []) + list(extra_cfg)))
        args += (C3_ARGS if c3 else []) + ["-c", cfg]
    return [exe, "filter", "-1", inputs[0], "-2", inputs[1], "-C", "c1" + ext, "-D", "c2" + ext, "-o", out_dir] + args, None


def run(exe, inputs, out_dir, ext, threads, env=None, c3=None, extra_cfg=None):
    cmd, _ = command(exe, inputs, out_dir, ext, threads, c3, extra_cfg)
    t0 = time.time()
    try:      # (a run that hangs -- a first contact of new device code with the hardware -- must not take the caller's bench line with it)
        r = subprocess.run(cmd, capture_output=True, env=env, timeout=float(os.environ.get("SNK_E2E_TIMEOUT", "900")))
    except subprocess.TimeoutExpired as ex:
        r = subprocess.CompletedProcess(cmd, -9, ex.stdout or b"", (ex.stderr or b"") + b"\n[bench_e2e: timed out, killed]")
    return time.time() - t0, r


def params_text(c3, extra_cfg):
    return ("-f/-r README adapters -J -l 10 -q 0.1" + (" -n 0.01 -m 20 -g 10 -X 50 -p 0.8 + trimBadTail=20,30 (configs[2])" if c3 else "") +
            ("".join(" + config key " + x for x in extra_cfg)))


def compare(tmp, mode, ext, entry, side_prefix=None):
    """clean FASTQ (decompressed bytes), the ten reports and -- side_prefix -- the side files of both tools' output directories"""
    a, b = os.path.join(tmp, f"ours_{mode}"), os.path.join(tmp, f"reference_{mode}")
    entry["clean_fastq_identical"] = all(md5_of(os.path.join(a, c + ext)) == md5_of(os.path.join(b, c + ext)) for c in ("c1", "c2"))
    differing = [rep for rep in REPORTS if open(os.path.join(a, rep), "rb").read() != open(os.path.join(b, rep), "rb").read()]
    entry["report_identical"] = not differing          # all ten report files, byte for byte
    entry["reports_compared"] = len(REPORTS)
    if differing:
        entry["reports_differing"] = differing
    if side_prefix:
        names = sorted(x for x in os.listdir(b) if x.startswith(side_prefix))
        entry["side_files"] = len(names)
        entry["side_files_identical"] = all(os.path.exists(os.path.join(a, x)) and md5_of(os.path.join(a, x)) == md5_of(os.path.join(b, x)) for x in names)


def measure(tmp, n, T, modes, c3=None, extra_cfg=None, L=150, dup_frac=0.0, deadline=None):
    """deadline (time.time() value): legs that would start after it are recorded as skipped"""
    c3 = C3 if c3 is None else c3
    extra_cfg = EXTRA_CFG if extra_cfg is None else extra_cfg
    res = {"pairs": n, "read_len": L, "threads_T": T, "host_cores": os.cpu_count(), "params": params_text(c3, extra_cfg),
           "where": "/dev/shm", "modes": {}}
    if dup_frac > 0:
        res["duplicate_pairs"] = dup_frac
    if True:
        f, t_gen, t_gz = make_inputs(tmp, n, ["gz"] if any(m in ("gz", "gz2plain", "gz_c3") for m in modes) else [], L, dup_frac)
        res["generate_s"] = round(t_gen, 1)
        res["gzip_inputs_s"] = round(t_gz, 1)
        for mode in modes:
            if deadline is not None and time.time() > deadline:
                res["modes"][mode] = {"skipped": "the caller's time budget was spent before this leg"}
                continue
            # plain: plain -> plain; gz: .gz -> .gz; gz2plain: .gz -> plain (the reference's plain-INPUT path stalls 60 s in
            # remove_tmpDir past one merge cycle and loses a patch, SURVEY Q10: this leg is its plain-output time without that);
            # gz_c3: .gz -> .gz with BASELINE configs[2]'s parameters on the same files; plain_ours: plain -> plain, this CLI only
            # (to be held against the reference's gz2plain time: its plain-input time is the Q10 stall)
            ext = ".fq.gz" if mode in ("gz", "gz_c3") else ".fq"
            inputs = [x + ".gz" for x in f] if mode in ("gz", "gz2plain", "gz_c3") else f
            leg_c3 = True if mode == "gz_c3" else c3
            entry = {"params": params_text(leg_c3, extra_cfg)} if mode == "gz_c3" else {}
            for name, exe in (("ours", OURS), ("reference", REF)):
                if not os.path.exists(exe) or (mode == "plain_ours" and name == "reference"):
                    continue
                o = os.path.join(tmp, f"{name}_{mode}")
                w, r = run(exe, inputs, o, ext, T, c3=leg_c3, extra_cfg=extra_cfg)
                entry[name] = {"wall_s": round(w, 2), "Mreads_per_s": round(2 * n / w / 1e6, 3), "rc": r.returncode}
                if r.returncode != 0:
                    entry[name]["stderr"] = r.stderr[-300:].decode(errors="replace")
                print(f"{name:10s} {mode:5s} {n} pairs: {w:7.2f} s  {2 * n / w / 1e6:8.3f} Mreads/s  rc {r.returncode}", file=sys.stderr, flush=True)
            if "ours" in entry and "reference" in entry and entry["ours"]["rc"] == 0 and entry["reference"]["rc"] == 0:
                entry["speedup"] = round(entry["reference"]["wall_s"] / entry["ours"]["wall_s"], 2)
                # the reference's plain-input runs drop a patch past one merge cycle (Q10): compare only what is comparable
                comparable = mode != "plain" or n <= 6_000_000
                if comparable:
                    compare(tmp, mode, ext, entry, "dupReads." if "rmdup" in extra_cfg else None)
            for name in ("ours", "reference"):
                subprocess.call(["rm", "-rf", os.path.join(tmp, f"{name}_{mode}")])
            res["modes"][mode] = entry
    return res


def main():
    global C3
    C3 = "--c3" in sys.argv
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(argv[0]) if len(argv) > 0 else 16_000_000
    T = int(argv[1]) if len(argv) > 1 else 16
    modes = (argv[2] if len(argv) > 2 else "plain,gz").split(",")
    tmp = tempfile.mkdtemp(prefix="snke2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        res = measure(tmp, n, T, modes)
    finally:
        if "--keep" not in sys.argv:
            subprocess.call(["rm", "-rf", tmp])
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"e2e_{n}{'_c3' if C3 else ''}.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
