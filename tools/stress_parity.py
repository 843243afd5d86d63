"""Repeats HIP-vs-oracle comparisons of a few parameter sets many times in one process (hunting intermittent differences):
    python tools/stress_parity.py [reps=30] [n=20000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import snk_testlib as T  # noqa: E402
from cases import PE_CASES  # noqa: E402
from soapnuke_amd import abi, synth  # noqa: E402
from soapnuke_amd.filter import FilterContext, records_to_numpy  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
bad = 0
for name in ("meanq_polyx", "C3_full", "C2_adatrim_lowq", "hard_lq_trim"):
    for var_len in (False, True):
        d = synth.make_batch(n, 150, paired=True, seed=5, var_len=var_len)
        p = abi.default_params(paired=True, max_read_len=150, **PE_CASES[name])
        o = T.run_oracle(p, d)
        for rep in range(reps):
            ctx = FilterContext(p, device=0)
            dev = ctx.upload(d)
            rec = ctx.alloc_records(n)
            ctx.filter_batch(ctx.make_batch(dev), rec, kernel=2)
            s, mx, err = ctx.fetch()
            r = [records_to_numpy(x) for x in rec]
            ctx.close()
            diffs = []
            for m in range(2):
                w = np.nonzero(r[m] != o["rec"][m])[0]
                if len(w):
                    diffs.append(f"mate {m}: {len(w)} records differ, first {w[:4]}: got {r[m][w[:2]]} want {o['rec'][m][w[:2]]}")
            if not np.array_equal(s, o["sum"]):
                diffs.append("stats: " + str(T.describe_stats_diff(p, s, o["sum"]))[:600])
            if not np.array_equal(mx, o["max"]):
                diffs.append("max block differs")
            if diffs:
                bad += 1
                print(f"DIFF {name} var_len={var_len} rep {rep}: " + " | ".join(diffs), flush=True)
        print(f"{name} var_len={var_len}: done", flush=True)
print("differences:", bad)
