"""End-to-end wall clock of this repo's `SOAPnuke filter` CLI vs the compiled reference on the same FASTQ
(GPU box): python tools/bench_cli.py [pairs] [threads]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from soapnuke_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
only = sys.argv[3] if len(sys.argv) > 3 else ""
tmp = tempfile.mkdtemp(prefix="snkcli_", dir="/dev/shm")
d = synth.make_batch(min(n, 1_000_000), 150, paired=True)
f = [os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")]
u = d["n"]
for k in range((n + u - 1) // u):
    for m in range(2):
        synth.write_fastq(f[m] + ".part", d["seq"][m], d["qual"][m], 150, m + 1, first_index=k * u)
        with open(f[m], "ab") as out, open(f[m] + ".part", "rb") as src:
            out.write(src.read())
        os.unlink(f[m] + ".part")
gzin = len(sys.argv) > 4 and sys.argv[4] == "gz"     # python tools/bench_cli.py 4000000 16 "" gz
if gzin:
    for m in range(2):
        subprocess.check_call(["gzip", "-1", "-k", f[m]])
args = ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-l", "10", "-q", "0.1", "-T", str(T)]
res = {}
for name, exe in (("ours", os.path.join(ROOT, "soapnuke_amd", "SOAPnuke")), ("reference", os.path.join(ROOT, "oracle", "_ref", "SOAPnuke"))):
    if not os.path.exists(exe) or (only and only != name):
        continue
    for out_ext in (".fq", ".fq.gz"):
        o = os.path.join(tmp, name + out_ext.replace(".", "_"))
        t0 = time.time()
        inp = [x + ".gz" for x in f] if gzin else f
        r = subprocess.run([exe, "filter", "-1", inp[0], "-2", inp[1], "-C", "c1" + out_ext, "-D", "c2" + out_ext, "-o", o] + args,
                           capture_output=True)
        w = time.time() - t0
        print(f"{name:10s} {'gz   ' if gzin else 'plain'} -> {out_ext:7s} {n} pairs  wall {w:6.2f} s  {2 * n / w / 1e6:7.3f} Mreads/s  rc {r.returncode}", flush=True)
        subprocess.call(["rm", "-rf", o])
subprocess.call(["rm", "-rf", tmp])
