"""BASELINE configs[2]-shaped end-to-end run at sizes up to the README's 628 M pairs: `.gz -> .gz`, full trim + filter
parameters, this repo's CLI and the compiled reference binary on the same files in /dev/shm, ALL ten report files and the
md5 of the decompressed clean FASTQ compared.

    python tools/bench_e2e_big.py [pairs=256000000] [threads=16] [--len 250] [--rmdup] [--dup 0.05] [--bounded] [--stored] [--ours=PATH]

--ours=PATH: another build of this repo's CLI (the emulated one, tests/simt/_build/SOAPnuke_simt, for a rehearsal without a GPU).
SNK_BIG_MEM_LIMIT_MB / SNK_BIG_HEADROOM_MB: hold the run against a limit of its own (see Watchdog) -- the rehearsal of the 300 GiB
box at a small size: `--bounded` has to stay under a limit that the stored mode breaks.

--len 250 --rmdup: BASELINE configs[4]'s shape (PE250, configs[1] parameters + config key `rmdup`; --dup: fraction of pairs that
repeat an earlier pair of the same million), where the dupReads.<thread>.<mate>.gz side files are compared as well.

Inputs are written as multi-member gzip (one member per million pairs, compressed by a pool of `gzip -1` processes: a
single `gzip` stream of 85 GB takes 15 minutes on its own); nothing plain is kept.

**Memory.**  Everything lives in tmpfs, which is charged to the container's memory cgroup (300 GiB on the GPU box): 628 M
pairs are 205 GB of `.gz` input and ~150 GB of `.gz` output PER TOOL, which does not fit (round 3's attempt was OOM-killed two
minutes into this CLI's run).  `--bounded` (the default above 300 M pairs) therefore never keeps a clean file whole:
  * timed run of this CLI: regular output files, a thread punches holes (fallocate PUNCH_HOLE) behind the writer -- the bytes
    are gone, the wall clock and the ten reports are what this run gives;
  * verification run of this CLI (not timed for the headline): the clean files are named pipes read by md5 consumers
    (zlib -> md5 of the decompressed bytes); the CLI writes pipes in order through write();
  * the reference (timed): regular output files that a tail reader decompresses into an md5 as they grow, punching holes
    behind itself (its `cat tmp >> clean` appends keep working; the readers cost about one of the 16 CPUs, the reference
    is bound by its one gzgets thread).
A watchdog aborts the run cleanly when the cgroup comes within 16 GiB of its limit.  Prints one JSON object, also written
to gpurun_out/e2e_big_<pairs>.json."""
import concurrent.futures as cf
import ctypes
import hashlib
import json
import os
import signal
import subprocess
import sys
import tempfile
import threading
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from soapnuke_amd import synth  # noqa: E402
import bench_e2e  # noqa: E402

_libc = ctypes.CDLL("libc.so.6", use_errno=True)
_libc.fallocate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong]


def punch(fd, lo, hi):
    """frees the pages of [lo, hi) of a tmpfs file (FALLOC_FL_KEEP_SIZE | FALLOC_FL_PUNCH_HOLE); the size stays"""
    lo = (lo + 4095) & ~4095
    hi &= ~4095
    if hi > lo:
        _libc.fallocate(fd, 3, lo, hi - lo)
    return max(lo, hi)


def md5_gz(path):
    h = hashlib.md5()
    p = subprocess.Popen(["gzip", "-dc", path], stdout=subprocess.PIPE)
    n = 0
    while True:
        b = p.stdout.read(1 << 24)
        if not b:
            break
        h.update(b)
        n += len(b)
    p.wait()
    return h.hexdigest(), n


class GzMd5:
    """md5 + length of the decompressed bytes of a (multi-member) gzip stream fed piecewise"""

    def __init__(self):
        self.h, self.n, self.z, self.members = hashlib.md5(), 0, zlib.decompressobj(31), 0

    def feed(self, data):
        while data:
            out = self.z.decompress(data)
            if out:
                self.h.update(out)
                self.n += len(out)
            if self.z.eof:                                  # next member
                data = self.z.unused_data
                self.z = zlib.decompressobj(31)
                self.members += 1
            else:
                data = b""

    def result(self):
        return self.h.hexdigest(), self.n


class Puncher(threading.Thread):
    """keeps a growing output file from occupying memory: everything but the last `margin` bytes is punched out"""

    def __init__(self, path, margin=None):
        super().__init__(daemon=True)
        if margin is None:                                   # (SNK_BIG_PUNCH_MARGIN_MB: the rehearsal at a small size)
            margin = int(os.environ.get("SNK_BIG_PUNCH_MARGIN_MB", "1024")) << 20
        self.path, self.margin, self.stop, self.size = path, margin, threading.Event(), 0

    def run(self):
        fd, done = -1, 0
        while True:
            last = self.stop.is_set()
            if fd < 0 and os.path.exists(self.path):
                fd = os.open(self.path, os.O_RDWR)
            if fd >= 0:
                self.size = os.fstat(fd).st_size
                done = max(done, punch(fd, done, self.size - (0 if last else self.margin)))
            if last:
                break
            time.sleep(0.25)
        if fd >= 0:
            os.close(fd)


class TailMd5(threading.Thread):
    """follows a growing regular .gz file (or reads a named pipe to its end): md5 of the decompressed bytes; for a regular file the
    consumed part is punched out.  alive(): the producer is still running."""

    def __init__(self, path, alive, fifo=False):
        super().__init__(daemon=True)
        self.path, self.alive, self.fifo, self.acc, self.compressed = path, alive, fifo, GzMd5(), 0

    def run(self):
        if self.fifo:
            with open(self.path, "rb", buffering=0) as f:        # blocks until the writer opens; EOF when it closes
                while True:
                    b = f.read(1 << 24)
                    if not b:
                        break
                    self.compressed += len(b)
                    self.acc.feed(b)
            return
        while not os.path.exists(self.path):
            if not self.alive():
                return
            time.sleep(0.2)
        fd, done = os.open(self.path, os.O_RDWR), 0
        while True:
            was_alive = self.alive()
            b = os.pread(fd, 1 << 24, self.compressed)
            if b:
                self.compressed += len(b)
                self.acc.feed(b)
                done = max(done, punch(fd, done, self.compressed))
            elif not was_alive:
                break
            else:
                time.sleep(0.2)
        os.close(fd)


class Watchdog(threading.Thread):
    """aborts (kills the running child, sets .tripped) before the memory cgroup's limit is reached"""

    def __init__(self, headroom=16 << 30, where="/dev/shm"):
        super().__init__(daemon=True)
        self.headroom, self.child, self.tripped, self.peak, self.stop = headroom, None, False, 0, threading.Event()
        self.limit, self.where, self.meter = None, where, "cgroup memory.current"
        try:
            v = open("/sys/fs/cgroup/memory.max").read().strip()
            self.limit = None if v == "max" else int(v)
        except OSError:
            pass
        # SNK_BIG_MEM_LIMIT_MB / SNK_BIG_HEADROOM_MB: a limit of the run's own (a rehearsal of the 300 GiB box at a small size, or a box
        # whose cgroup files are not visible): what is held against it is the cgroup's figure where there is one, else the bytes the
        # tmpfs holds (which is what the cgroup charges for this tool: its files ARE its memory)
        if os.environ.get("SNK_BIG_MEM_LIMIT_MB"):
            self.limit = int(os.environ["SNK_BIG_MEM_LIMIT_MB"]) << 20
            self.headroom = int(os.environ.get("SNK_BIG_HEADROOM_MB", "0")) << 20
            self.meter = "bytes held by " + where
        if not os.path.exists("/sys/fs/cgroup/memory.current"):
            self.meter = "bytes held by " + where
        self.base = self.current()

    def current(self):
        if self.meter.startswith("cgroup"):
            try:
                return int(open("/sys/fs/cgroup/memory.current").read())
            except (OSError, ValueError):
                pass
        st = os.statvfs(self.where)
        return (st.f_blocks - st.f_bfree) * st.f_frsize - getattr(self, "base", 0)

    def run(self):
        while not self.stop.is_set():
            cur = self.current()
            self.peak = max(self.peak, cur)
            if self.limit and cur > self.limit - self.headroom and not self.tripped:
                self.tripped = True
                print(f"watchdog: {cur >> 30} GiB of {self.limit >> 30} GiB in use -- aborting", file=sys.stderr, flush=True)
                c = self.child
                if c is not None and c.poll() is None:
                    try:
                        os.killpg(c.pid, signal.SIGKILL)
                    except OSError:
                        pass
            time.sleep(0.5)


def run_tool(exe, inputs, out_dir, T, wd):
    """like bench_e2e.run(), with the child in its own process group and known to the watchdog"""
    cmd, env = bench_e2e.command(exe, inputs, out_dir, ".fq.gz", T)
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, start_new_session=True)
    wd.child = p
    return p, t0


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = sys.argv[1:]
    for a in opt:
        if a.startswith("--ours="):
            bench_e2e.OURS = a[len("--ours="):]
    L = int(opt[opt.index("--len") + 1]) if "--len" in opt else 150
    rmdup = "--rmdup" in opt
    dupf = float(opt[opt.index("--dup") + 1]) if "--dup" in opt else 0.05
    argv = [a for a in argv if a not in (str(L), str(dupf))] if ("--len" in opt or "--dup" in opt) else argv
    n = int(argv[0]) if len(argv) > 0 else 256_000_000
    T = int(argv[1]) if len(argv) > 1 else 16
    bounded = ("--bounded" in opt or n > 300_000_000) and "--stored" not in opt
    if bounded and rmdup:
        sys.exit("--bounded does not cover the dupReads side files")
    tmp = tempfile.mkdtemp(prefix="snkbig_", dir="/dev/shm")
    res = {"pairs": n, "read_len": L, "threads_T": T, "host_cores": os.cpu_count(), "where": "/dev/shm",
           "params": ("-f/-r README adapters -J -l 10 -q 0.1 + config key rmdup (BASELINE configs[4] shape), %.0f %% duplicate pairs" % (100 * dupf)) if rmdup else
                     "-f/-r README adapters -J -l 10 -q 0.1 -n 0.01 -m 20 -g 10 -X 50 -p 0.8 + trimBadTail=20,30 (BASELINE configs[2])",
           "inputs": "multi-member .gz, one member per 1 M pairs",
           "mode": "bounded memory: outputs never stored whole (see the module docstring)" if bounded else "outputs stored in /dev/shm"}
    wd = Watchdog()
    wd.start()
    res["memory_limit_GiB"] = None if wd.limit is None else round(wd.limit / (1 << 30), 3)
    res["memory_meter"] = wd.meter
    try:
        u = int(os.environ.get("SNK_BIG_UNIT", "1000000"))       # pairs per gzip member (tests/test_simt_bench.py shrinks it)
        d = synth.make_batch(u, L, paired=True)
        if rmdup:                                           # duplicates by sequence (qualities differ), as rmdup sees them
            import numpy as np
            rng = np.random.default_rng(11)
            dst = rng.choice(u, int(u * dupf), replace=False)
            src = rng.integers(0, u, len(dst))
            for m in range(2):
                d["seq"][m][dst] = d["seq"][m][src]

        f = [os.path.join(tmp, "r1.fq.gz"), os.path.join(tmp, "r2.fq.gz")]
        t0 = time.time()
        parts = (n + u - 1) // u

        lock, blocks = threading.Lock(), {}

        def block(k):                                       # rmdup: every million pairs its own reads (duplicates only inside it)
            if not rmdup:
                return d
            with lock:
                if k not in blocks:
                    import numpy as np
                    b = synth.make_batch(u, L, paired=True, seed=synth.SEED + 1 + k)
                    rng = np.random.default_rng(11 + k)
                    dst = rng.choice(u, int(u * dupf), replace=False)
                    src = rng.integers(0, u, len(dst))
                    for mm in range(2):
                        b["seq"][mm][dst] = b["seq"][mm][src]
                    blocks[k] = [b, 2]
                e = blocks[k]
                e[1] -= 1
                if e[1] == 0:
                    del blocks[k]
                return e[0]

        def make(k, m):
            part = os.path.join(tmp, f"p{m}.{k}.fq")
            cnt = min(u, n - k * u)
            d = block(k)
            synth.write_fastq(part, d["seq"][m][:cnt], d["qual"][m][:cnt], L, m + 1, first_index=k * u)
            subprocess.check_call(["gzip", "-1", "-f", part])
            return part + ".gz"

        with cf.ThreadPoolExecutor(max_workers=12) as ex:
            for lo in range(0, parts, 12):                    # bounded space: 12 parts per mate in flight, appended in order
                if wd.tripped:
                    raise RuntimeError("memory watchdog tripped during input generation")
                futs = [[ex.submit(make, k, m) for k in range(lo, min(parts, lo + 12))] for m in range(2)]
                for m in range(2):
                    with open(f[m], "ab") as out:
                        for fu in futs[m]:
                            p = fu.result()
                            with open(p, "rb") as src:
                                while True:
                                    b = src.read(1 << 24)
                                    if not b:
                                        break
                                    out.write(b)
                            os.unlink(p)
                if lo % 120 == 0:
                    print(f"generated {min(parts, lo + 12)} / {parts} M pairs, {time.time() - t0:.0f} s, cgroup {wd.current() >> 30} GiB", file=sys.stderr, flush=True)
        res["generate_and_gzip_s"] = round(time.time() - t0, 1)
        res["input_gz_bytes"] = [os.path.getsize(x) for x in f]
        bench_e2e.C3 = not rmdup
        bench_e2e.EXTRA_CFG = ["rmdup"] if rmdup else []
        entry = {}
        md = {}
        clean = lambda who, c: os.path.join(tmp, who, c + ".fq.gz")     # noqa: E731

        def finish(name, p, t0):
            out, err = p.communicate()
            w = time.time() - t0
            e = {"wall_s": round(w, 2), "Mreads_per_s": round(2 * n / w / 1e6, 3), "rc": p.returncode}
            if p.returncode != 0:
                e["stderr"] = err[-300:].decode(errors="replace")
            print(name, e, f"cgroup peak {wd.peak >> 30} GiB", file=sys.stderr, flush=True)
            return e

        if not bounded:
            for name, exe in (("ours", bench_e2e.OURS), ("reference", bench_e2e.REF)):
                p, t0 = run_tool(exe, f, os.path.join(tmp, name), T, wd)
                entry[name] = finish(name, p, t0)
        else:
            # 1. this CLI, timed: regular files, holes punched behind the writer
            os.makedirs(os.path.join(tmp, "ours"), exist_ok=True)
            pun = [Puncher(clean("ours", c)) for c in ("c1", "c2")]
            for x in pun:
                x.start()
            p, t0 = run_tool(bench_e2e.OURS, f, os.path.join(tmp, "ours"), T, wd)
            entry["ours"] = finish("ours", p, t0)
            for x in pun:
                x.stop.set()
            for x in pun:
                x.join()
            entry["ours"]["clean_gz_bytes"] = [x.size for x in pun]
            for c in ("c1", "c2"):
                os.unlink(clean("ours", c))
            # 2. this CLI once more, into named pipes read by md5 consumers (the content check of the run above)
            if entry["ours"]["rc"] == 0 and not wd.tripped:
                os.makedirs(os.path.join(tmp, "ours_verify"), exist_ok=True)
                for c in ("c1", "c2"):
                    os.mkfifo(clean("ours_verify", c))
                cons = {c: TailMd5(clean("ours_verify", c), None, fifo=True) for c in ("c1", "c2")}
                for x in cons.values():
                    x.start()
                p, t0 = run_tool(bench_e2e.OURS, f, os.path.join(tmp, "ours_verify"), T, wd)
                entry["ours_verify"] = finish("ours_verify (pipes into md5 consumers: bound by them)", p, t0)
                if p.returncode != 0:                       # unblock consumers that never saw a writer
                    for c in ("c1", "c2"):
                        try:
                            os.close(os.open(clean("ours_verify", c), os.O_WRONLY | os.O_NONBLOCK))
                        except OSError:
                            pass
                for c, x in cons.items():
                    x.join()
                    md[("ours", c)] = x.acc.result()
                entry["ours_verify"]["reports_same_as_timed_run"] = all(
                    open(os.path.join(tmp, "ours", rep), "rb").read() == open(os.path.join(tmp, "ours_verify", rep), "rb").read() for rep in bench_e2e.REPORTS)
            # 3. the reference, timed: regular files, read (md5) and punched as they grow
            if entry["ours"]["rc"] == 0 and not wd.tripped:
                os.makedirs(os.path.join(tmp, "reference"), exist_ok=True)
                p, t0 = run_tool(bench_e2e.REF, f, os.path.join(tmp, "reference"), T, wd)
                cons = {c: TailMd5(clean("reference", c), lambda: p.poll() is None) for c in ("c1", "c2")}
                for x in cons.values():
                    x.start()
                entry["reference"] = finish("reference", p, t0)
                for c, x in cons.items():
                    x.join()
                    md[("reference", c)] = x.acc.result()
                entry["reference"]["clean_gz_bytes"] = [cons[c].compressed for c in ("c1", "c2")]
        ok = all(entry.get(k, {}).get("rc") == 0 for k in ("ours", "reference"))
        if ok:
            entry["speedup"] = round(entry["reference"]["wall_s"] / entry["ours"]["wall_s"], 2)
            differing = [rep for rep in bench_e2e.REPORTS
                         if open(os.path.join(tmp, "ours", rep), "rb").read() != open(os.path.join(tmp, "reference", rep), "rb").read()]
            entry["reports_compared"] = len(bench_e2e.REPORTS)
            entry["reports_differing"] = differing
            if not bounded:
                with cf.ThreadPoolExecutor(max_workers=4) as ex:
                    jobs = {(who, c): ex.submit(md5_gz, clean(who, c)) for who in ("ours", "reference") for c in ("c1", "c2")}
                    md = {k: v.result() for k, v in jobs.items()}
            entry["clean_fastq_identical"] = all(md.get(("ours", c)) is not None and md.get(("ours", c)) == md.get(("reference", c)) for c in ("c1", "c2"))
            entry["clean_bytes"] = [md[("ours", c)][1] for c in ("c1", "c2") if ("ours", c) in md]
            entry["clean_md5"] = {who + "." + c: md[(who, c)][0] for who in ("ours", "reference") for c in ("c1", "c2") if (who, c) in md}
            if rmdup:
                names = sorted(x for x in os.listdir(os.path.join(tmp, "reference")) if x.startswith("dupReads."))
                with cf.ThreadPoolExecutor(max_workers=8) as ex:
                    dj = {(who, x): ex.submit(md5_gz, os.path.join(tmp, who, x)) for who in ("ours", "reference") for x in names}
                    dm = {k: v.result() for k, v in dj.items()}
                entry["dup_side_files"] = len(names)
                entry["dup_side_files_identical"] = all(dm[("ours", x)] == dm[("reference", x)] for x in names)
                entry["dup_side_bytes"] = sum(dm[("ours", x)][1] for x in names)
            # the rows where the reference's data_num * 9 wrapped (more than 238.6 M reads in a per-position bin set): keep one as evidence
            q = open(os.path.join(tmp, "ours", "Base_quality_value_distribution_by_read_position_1.txt")).read().splitlines()
            entry["quality_row_position_1"] = q[2][-80:] if len(q) > 2 else None
        res["gz"] = entry
    except Exception as ex:      # whatever happened, the JSON (with what was measured) is written
        res["error"] = repr(ex)[:300]
    finally:
        wd.stop.set()
        res["watchdog_tripped"] = wd.tripped
        res["cgroup_peak_GiB"] = round(wd.peak / (1 << 30), 1)
        res["memory_peak_MiB"] = wd.peak >> 20
        subprocess.call(["rm", "-rf", tmp])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"e2e_big_{n}{'_L%d_rmdup' % L if rmdup else ''}.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
