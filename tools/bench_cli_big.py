"""Larger end-to-end run of this repo's CLI only (sanity at scale + rate): python tools/bench_cli_big.py [pairs] [threads] [rmdup]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from soapnuke_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rmdup = len(sys.argv) > 3 and sys.argv[3] == "rmdup"
tmp = tempfile.mkdtemp(prefix="snkbig_", dir="/dev/shm")
u = 1_000_000
f = [os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")]
for k in range((n + u - 1) // u):
    d = synth.make_batch(u, 150, paired=True, seed=1000 + (k % 8))       # 8 distinct blocks: later ones repeat -> duplicates
    for m in range(2):
        synth.write_fastq(f[m] + ".part", d["seq"][m], d["qual"][m], 150, m + 1, first_index=k * u)
        with open(f[m], "ab") as out, open(f[m] + ".part", "rb") as src:
            out.write(src.read())
        os.unlink(f[m] + ".part")
cfg = os.path.join(tmp, "cfg")
open(cfg, "w").write("rmdup\n" if rmdup else "")
cmd = [os.path.join(ROOT, "soapnuke_amd", "SOAPnuke"), "filter", "-1", f[0], "-2", f[1], "-C", "c1.fq", "-D", "c2.fq", "-o", os.path.join(tmp, "o"),
       "-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-l", "10", "-q", "0.1", "-T", str(T), "-c", cfg]
t0 = time.time()
r = subprocess.run(cmd, capture_output=True)
w = time.time() - t0
print(f"{n} pairs rmdup={rmdup} -T {T}: wall {w:.2f} s  {2 * n / w / 1e6:.2f} Mreads/s  rc {r.returncode} {r.stderr[-200:]}")
print(open(os.path.join(tmp, "o", "Statistics_of_Filtered_Reads.txt")).read()[:400])
print(subprocess.run(["tail", "-3", os.path.join(tmp, "o", "log")], capture_output=True).stdout.decode())
subprocess.call(["rm", "-rf", tmp])
